"""CPU: the C restatement (oracle/ltpl_oracle.c) + the host mirror reproduce the recorded reference calls at seam (1)."""
import pytest

from helpers import load_golden, replay_path_call, check_path_output
from graphbasedlocaltrajectoryplanner_amd import _capi
from graphbasedlocaltrajectoryplanner_amd.path_gen import OnlinePathGenerator


@pytest.mark.parametrize("fixture", ["c2_path_calls.npz", "c1_path_calls.npz", "zonewall_path_calls.npz"])
def test_oracle_matches_reference_recordings(monteblanco, oracle_backend, fixture):
    recs = load_golden(fixture)
    gen = OnlinePathGenerator(monteblanco, oracle_backend)
    assert len(recs) > 10
    for rec in recs:
        sc = replay_path_call(gen, rec)
        batch = _capi.PathsBatch([sc], w_last_edges=rec['w_last_edges'])
        res = oracle_backend.plan_paths(batch)
        out6 = res.action_sets(0, rec['start_node'][0], monteblanco.num_layers)
        check_path_output(out6, rec, what="%s tick %d" % (fixture, rec['tick']))
