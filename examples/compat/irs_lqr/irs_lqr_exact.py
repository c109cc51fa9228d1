"""irs_lqr/irs_lqr_exact.py of the reference -> irs_mpc_amd.irs_lqr.IrsLqrExact."""
from irs_mpc_amd.irs_lqr import IrsLqrExact      # noqa: F401
