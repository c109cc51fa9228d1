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
