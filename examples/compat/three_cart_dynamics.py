"""examples/three_cart/three_cart_dynamics.py of the reference -> the device-backed twin (irs_mpc_amd.systems.ThreeCartDynamics)."""
from irs_mpc_amd.systems import ThreeCartDynamics      # noqa: F401
