import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from irs_mpc_amd import device as dev
w = bench.Workload("planar_hand")
dm = w.system.dm(); n,m,T = dm.n, dm.m, w.T
Q,Qd,R = dev.to_dev(w.Q), dev.to_dev(w.Qd), dev.to_dev(w.R)
xd,x0,u_trj = dev.to_dev(w.xd), dev.to_dev(w.x0), dev.to_dev(w.u_trj)
x_trj,_ = dm.rollout_cost(x0,u_trj,Q,R,xd)
g = torch.Generator(device="cuda").manual_seed(1234)
du = w.std_u*torch.randn((T,10000,m),generator=g,device="cuda",dtype=torch.float32)
plan = dev.SmoothPlan(dm, w.mode, x_trj, u_trj, dx=None, du=du, fuse=True)
stream = torch.cuda.current_stream().cuda_stream
fn = lambda: plan.run(stream)
t_end=time.perf_counter()+0.25
while time.perf_counter()<t_end: fn()
for rep in range(6):
    for _ in range(5): fn()
    torch.cuda.synchronize(); torch.cuda.synchronize()
    ev0,ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0=time.perf_counter(); ev0.record(); t1=time.perf_counter()
    for _ in range(20): fn()
    t2=time.perf_counter(); ev1.record(); t3=time.perf_counter()
    while not ev1.query(): pass
    t4=time.perf_counter(); torch.cuda.synchronize(); t5=time.perf_counter()
    print("rep %d: ev0.record %.1f us, 20 launches issued %.1f us, ev1.record %.1f, until done %.1f, sync %.1f | total %.1f us = %.2f us/step; GPU events %.2f us/step" % (rep, 1e6*(t1-t0), 1e6*(t2-t1), 1e6*(t3-t2), 1e6*(t4-t3), 1e6*(t5-t4), 1e6*(t5-t0), 1e6*(t5-t0)/20, ev0.elapsed_time(ev1)*1e3/20))
