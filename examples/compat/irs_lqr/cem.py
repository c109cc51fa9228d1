"""irs_lqr/cem.py of the reference -> irs_mpc_amd.cem."""
from irs_mpc_amd.cem import *      # noqa: F401,F403
