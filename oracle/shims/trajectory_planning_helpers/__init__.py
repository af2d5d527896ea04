"""
ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Stand-in for ``trajectory_planning_helpers==0.75`` (requirements.txt:5 of the reference): a third-party dependency
whose source is NOT under /root/reference and which is not installed in this image. Each sub-module restates the
published algorithm of the tph function of the same name, limited to the arguments the reference uses
(call sites listed in SURVEY.md §8c / appendix B). Parity at this boundary is UNPINNED by the reference (no tests,
no golden vectors); independent cross-checks (scipy clamped cubic spline, closed-form velocity profiles) live in
tests/test_oracle_shims.py.
"""
from . import normalize_psi          # noqa: F401
from . import calc_splines           # noqa: F401
from . import calc_spline_lengths    # noqa: F401
from . import interp_splines         # noqa: F401
from . import calc_head_curv_an      # noqa: F401
from . import calc_head_curv_num     # noqa: F401
from . import calc_ax_poss           # noqa: F401
from . import calc_vel_profile       # noqa: F401
from . import calc_vel_profile_brake  # noqa: F401
from . import calc_ax_profile        # noqa: F401
from . import conv_filt              # noqa: F401
from . import progressbar            # noqa: F401

__version__ = "0.75-oracle-shim"
