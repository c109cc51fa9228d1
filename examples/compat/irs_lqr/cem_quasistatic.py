"""irs_lqr/cem_quasistatic.py of the reference -> irs_mpc_amd.cem_quasistatic."""
from irs_mpc_amd.cem_quasistatic import *      # noqa: F401,F403
