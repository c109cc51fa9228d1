"""irs_mpc_amd -- MI355X-native iRS-LQR inner loop (drop-in for the smoothing +
TV-LQR hot path of hjsuh94/irs_mpc).  See DESIGN.md."""
from .cem import CemParameters, CrossEntropyMethod                  # noqa: F401
from .cem_quasistatic import (CemQuasistaticParameters,            # noqa: F401
                              CrossEntropyMethodQuasistatic)
from .dynamical_system import DynamicalSystem                       # noqa: F401
from .irs_lqr import (IrsLqr, IrsLqrExact, IrsLqrFirstOrder,        # noqa: F401
                      IrsLqrParameters, IrsLqrZeroOrder)
from .irs_lqr_quasistatic import (IrsLqrQuasistatic,               # noqa: F401
                                   IrsLqrQuasistaticParameters)
from .sampling import GaussianSmoothing                             # noqa: F401
from .systems import (BicycleDynamics, BoxOnBoxDynamics,             # noqa: F401
                      BoxPivotingDynamics, BoxPushingDynamics,
                      PendulumDynamics, PlanarHandDynamics,
                      QuadrotorDynamics, QuasistaticDeviceDynamics,
                      ThreeCartDynamics)
from .tv_lqr import get_solver, solve_tvlqr                         # noqa: F401
