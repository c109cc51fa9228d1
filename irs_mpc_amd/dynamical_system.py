"""Host mirror of the reference's plugin surface (irs_lqr/dynamical_system.py:1-66).

Same attributes (`h, dim_x, dim_u`) and the same four methods.  A subclass that is
to run through the HIP path additionally names a device functor: `device_model`
(an irs_model_id) and `device_params()` (its constants, h first).  The four methods
are then served by the device kernels; they take and return NumPy float64 arrays
exactly like the reference's.
"""
import numpy as np

from . import device as dev


class DynamicalSystem:
    device_model = None          # irs_model_id of the device functor, or None

    def __init__(self):
        self.h = 0
        self.dim_x = 0
        self.dim_u = 0
        self._dm = None

    def device_params(self):
        return [self.h]

    def dm(self):
        """The bound device functor (created on first use)."""
        if self.device_model is None:
            raise NotImplementedError(
                "%s has no device model: the HIP path needs a functor registered in "
                "irs_mpc_amd/csrc/models.hpp (there is no CPU fallback)." % type(self).__name__)
        if getattr(self, "_dm", None) is None:
            self._dm = dev.DeviceModel(self.device_model, self.device_params())
        return self._dm

    # ---- the reference's four virtuals, device-backed when a functor exists ----
    def dynamics(self, x, u):
        if self.device_model is None:
            raise NotImplementedError("This class is virtual.")
        return self.dynamics_batch(np.asarray(x, float)[None, :], np.asarray(u, float)[None, :])[0]

    def dynamics_batch(self, x, u):
        if self.device_model is None:
            raise NotImplementedError("This class is virtual.")
        return self.dm().dynamics_batch(dev.to_dev(x), dev.to_dev(u)).cpu().numpy()

    def jacobian_xu(self, x, u):
        if self.device_model is None:
            raise NotImplementedError("This class is virtual.")
        return self.jacobian_xu_batch(np.asarray(x, float)[None, :], np.asarray(u, float)[None, :])[0]

    def jacobian_xu_batch(self, x, u):
        if self.device_model is None:
            raise NotImplementedError("This class is virtual.")
        return self.dm().jacobian_xu_batch(dev.to_dev(x), dev.to_dev(u)).cpu().numpy()
