"""irs_lqr/tv_lqr.py of the reference -> irs_mpc_amd.tv_lqr."""
from irs_mpc_amd.tv_lqr import *      # noqa: F401,F403
