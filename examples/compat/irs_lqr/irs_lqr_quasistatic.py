"""irs_lqr/irs_lqr_quasistatic.py of the reference -> irs_mpc_amd.irs_lqr_quasistatic."""
from irs_mpc_amd.irs_lqr_quasistatic import *      # noqa: F401,F403
