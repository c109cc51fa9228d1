"""examples/pendulum/pendulum_dynamics.py of the reference -> the device-backed twin (irs_mpc_amd.systems.PendulumDynamics)."""
from irs_mpc_amd.systems import PendulumDynamics      # noqa: F401
