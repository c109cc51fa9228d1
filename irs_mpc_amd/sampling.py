"""Declarative Gaussian smoothing schedule = the sampling closure every example
script defines (e.g. examples/pendulum/pendulum_zero_order.py:33-43,
examples/quadrotor/quadrotor_first_order.py:42-52):

    dx ~ N(0, std_x / iter**power),  du ~ N(0, std_u / iter**power),  N samples.

Passing one of these instead of a Python callable lets the library draw the samples
on the GPU (Philox4x32-10) -- no host RNG, no host->device sample traffic.
"""
import numpy as np


class GaussianSmoothing:
    def __init__(self, std_x, std_u, num_samples, power=0.5, seed=0):
        self.std_x = np.atleast_1d(np.asarray(std_x, float))
        self.std_u = np.atleast_1d(np.asarray(std_u, float))
        self.num_samples = int(num_samples)
        self.power = float(power)
        self.seed = int(seed)

    def stds(self, it):
        s = float(it) ** self.power
        return self.std_x / s, self.std_u / s

    def __call__(self, xbar, ubar, it):
        """Host draw with the reference's closure semantics (global NumPy RNG)."""
        sx, su = self.stds(it)
        dx = np.random.normal(0.0, sx, size=(self.num_samples, len(sx)))
        du = np.random.normal(0.0, su, size=(self.num_samples, len(su)))
        return dx, du
