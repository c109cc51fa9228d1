"""irs_lqr/irs_lqr_zero_order.py of the reference -> irs_mpc_amd.irs_lqr.IrsLqrZeroOrder."""
from irs_mpc_amd.irs_lqr import IrsLqrZeroOrder      # noqa: F401
