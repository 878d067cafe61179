"""Diagnostic: QP rounds / flags of short rollouts (warm and cold) on the config-2 workload."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from linearmpchumanoid_amd.controller import BatchedController, default_config
from helpers import perturbed_velocities, oracle_system
o = oracle_system(1e-3, 0.016)
q0, zc = o.robot()["q"].copy(), o.zcom
B = 256
v = perturbed_velocities(B)
for warm in (0, 1):
    for extra in ({}, {"bpp_rounds": 10}, {"max_qp_iters": 200}):
        ctl = BatchedController(B, default_config(dt=1e-3, time_horizon=0.016, z_com=zc, warm_start=warm, **extra))
        ctl.set_refs_stance(2.0, 2)
        st = ctl.new_state(q0, v, t=0.0)
        for nt in (1, 5, 20):
            out, status, _ = ctl.rollout(st, nt)
            torch.cuda.synchronize()
            s = status.cpu().numpy()
            print("warm", warm, extra, "after +%d ticks: max rounds per robot: mean %.2f max %d; flags or %d; flagged %d" % (nt, s[:, 1].mean(), s[:, 1].max(), int(np.bitwise_or.reduce(s[:, 2])), int((s[:, 2] != 0).sum())), flush=True)
