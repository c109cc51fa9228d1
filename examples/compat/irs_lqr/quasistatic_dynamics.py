"""irs_lqr/quasistatic_dynamics.py of the reference.  `QuasistaticDynamics` wraps the external simulator there;
the device twins (same method names: dynamics, dynamics_batch, jacobian_xu, calc_AB_exact, calc_AB_first_order,
calc_B_zero_order, calc_AB_zero_order, calc_AB_batch, get_*_dict helpers) are model-specific classes here."""
from irs_mpc_amd.systems import (BoxOnBoxDynamics, BoxPivotingDynamics, BoxPushingDynamics,      # noqa: F401
                                 PlanarHandDynamics, QuasistaticDeviceDynamics)
