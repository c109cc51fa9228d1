"""examples/quadrotor/quadrotor_dynamics.py of the reference -> the device-backed twin (irs_mpc_amd.systems.QuadrotorDynamics)."""
from irs_mpc_amd.systems import QuadrotorDynamics      # noqa: F401
