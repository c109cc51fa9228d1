"""Device-backed twins of the reference's analytic example plugins."""
from ._lib import MODEL_PENDULUM, MODEL_QUADROTOR
from .dynamical_system import DynamicalSystem


class PendulumDynamics(DynamicalSystem):
    """examples/pendulum/pendulum_dynamics.py:8-127: x=[angle, speed], u=[torque]."""
    device_model = MODEL_PENDULUM

    def __init__(self, h):
        super().__init__()
        self.h = h
        self.dim_x = 2
        self.dim_u = 1


class QuadrotorDynamics(DynamicalSystem):
    """examples/quadrotor/quadrotor_dynamics.py:15-231: x=[xyz,rpy,xyz_d,rpy_d], u=rotors."""
    device_model = MODEL_QUADROTOR

    def __init__(self, h):
        super().__init__()
        self.h = h
        self.dim_x = 12
        self.dim_u = 4
        # quadrotor_dynamics.py:26-37
        self.m = 0.775
        self.L = 0.15
        self.g = 9.81
        self.Ixx, self.Iyy, self.Izz = 0.0015, 0.0025, 0.0035
        self.kF = 1.0
        self.kM = 0.0245

    def device_params(self):
        return [self.h, self.m, self.L, self.g, self.Ixx, self.Iyy, self.Izz, self.kF, self.kM]
