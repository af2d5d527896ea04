"""Re-export of the seeded scenario generators (they live in the package because bench.py uses them as well)."""
from graphbasedlocaltrajectoryplanner_amd.scenario_gen import *  # noqa: F401,F403
from graphbasedlocaltrajectoryplanner_amd.scenario_gen import random_scenarios, raceline_state, c2_scenarios  # noqa: F401
