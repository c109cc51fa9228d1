"""Phase stamps of the matrix-core Riccati recursion (csrc/tvlqr.hip, riccati_backward_mfma) on the quadrotor:
build a tuning library with -DIRS_RIC_STAMPS and point IRS_HIP_LIB at it, on the GPU box:

    cd irs_mpc_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DIRS_RIC_STAMPS \
        -c tvlqr.hip -o /tmp/tvlqr_st.o && hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libirs_hip_ricst.so \
        plugin.o smooth.o smooth_ug.o /tmp/tvlqr_st.o cem.o boxqp.o ctrlbox.o ctrlbox_mfma.o collective.o iterate.o -ldl
    IRS_HIP_LIB=$PWD/variants/libirs_hip_ricst.so python tools/riccati_stamps.py

Prints s_memtime cycles per step: head + 14 products, gain solve, the ten products of the Joseph update, and how long a
step waits for its prefetched operands.  The product build contains no stamp."""
import sys, ctypes, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import irs_mpc_amd as amd
from irs_mpc_amd import _lib
from examples.problems import quadrotor
sysd,p,smp,_,_=quadrotor(50); p.xbound=p.ubound=None
sol=amd.IrsLqrExact(sysd,p); sol.verbose=False
sol.iterate(2)
torch.cuda.synchronize()
lib=ctypes.CDLL(_lib.LIB_PATH); lib.irs_debug_riccati_stamps.restype=None; lib.irs_debug_riccati_stamps()
