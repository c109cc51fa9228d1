"""irs_lqr/irs_lqr.py of the reference -> irs_mpc_amd.irs_lqr."""
from irs_mpc_amd.irs_lqr import *      # noqa: F401,F403
