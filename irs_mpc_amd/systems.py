"""Device-backed twins of the reference's analytic example plugins."""
from ._lib import MODEL_BICYCLE, MODEL_PENDULUM, MODEL_QUADROTOR, MODEL_THREE_CART
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


class BicycleDynamics(DynamicalSystem):
    """examples/bicycle/bicycle_dynamics.py:8-132: x=[x,y,heading,speed,steer], u=[accel,steer rate]."""
    device_model = MODEL_BICYCLE

    def __init__(self, h):
        super().__init__()
        self.h = h
        self.dim_x = 5
        self.dim_u = 2


class ThreeCartDynamics(DynamicalSystem):
    """examples/three_cart/three_cart_dynamics.py:8-107: x=[q1,q2,q3,v1,v2,v3], u=[u1,u3].
    Follows the scalar `dynamics` (the reference's `dynamics_batch` resolves penetration
    differently, :175-188).  The reference offers no Jacobian for it; here `jacobian_xu`
    returns the active branch's derivative."""
    device_model = MODEL_THREE_CART

    def __init__(self, dt):
        super().__init__()
        self.h = dt
        self.dim_x = 6
        self.dim_u = 2
        self.d = 0.2

    def device_params(self):
        return [self.h, self.d]
