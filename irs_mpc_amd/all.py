"""Convenience module mirroring irs_lqr/all.py:5-11."""
from .dynamical_system import *   # noqa: F401,F403
from .irs_lqr import *            # noqa: F401,F403
from .tv_lqr import *             # noqa: F401,F403
from .systems import *            # noqa: F401,F403
from .sampling import *           # noqa: F401,F403
from .cem import *                # noqa: F401,F403
from .irs_lqr_quasistatic import *   # noqa: F401,F403
from .cem_quasistatic import *       # noqa: F401,F403
