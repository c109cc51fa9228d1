"""irs_lqr/irs_lqr_first_order.py of the reference -> irs_mpc_amd.irs_lqr.IrsLqrFirstOrder."""
from irs_mpc_amd.irs_lqr import IrsLqrFirstOrder      # noqa: F401
