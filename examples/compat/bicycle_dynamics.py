"""examples/bicycle/bicycle_dynamics.py of the reference -> the device-backed twin (irs_mpc_amd.systems.BicycleDynamics)."""
from irs_mpc_amd.systems import BicycleDynamics      # noqa: F401
