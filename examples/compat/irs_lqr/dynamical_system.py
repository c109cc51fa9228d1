"""irs_lqr/dynamical_system.py of the reference -> irs_mpc_amd.dynamical_system."""
from irs_mpc_amd.dynamical_system import *      # noqa: F401,F403
