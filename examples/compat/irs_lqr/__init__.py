"""Name shim: `irs_lqr` -> irs_mpc_amd (see examples/compat/README.md).  Sub-modules keep the reference's
file names (irs_lqr/all.py, irs_lqr.py, irs_lqr_zero_order.py, ... cem.py, tv_lqr.py)."""
