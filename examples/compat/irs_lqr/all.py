"""irs_lqr/all.py:5-11 of the reference, served by the device-backed twins (+ the CEM names the reference's
`*_cem.py` scripts import from here)."""
from irs_mpc_amd.all import *                                      # noqa: F401,F403
from irs_mpc_amd.cem import CemParameters, CrossEntropyMethod      # noqa: F401
