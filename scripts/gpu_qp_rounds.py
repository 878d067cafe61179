"""Diagnostic: distribution of the worst cone-QP round count per robot and per 10-tick launch of the bench workload
(the slowest robot sets the launch time)."""
import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from linearmpchumanoid_amd.controller import BatchedController, default_config
from helpers import perturbed_velocities
ik = json.load(open('tests/golden/ik_posture.json'))
B = 1024
ctl = BatchedController(B, default_config(dt=1e-3, time_horizon=0.016, z_com=ik['z_com'], warm_start=1))
ctl.set_refs_stance(2.0, 2)
st = ctl.new_state(np.array(ik['q']), perturbed_velocities(B), t=0.0)
out, status = ctl.new_out(), ctl.new_status()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for step in range(23):
    e0.record(); ctl.rollout(st, 10, out, status); e1.record(); torch.cuda.synchronize()
    it = status.cpu().numpy()[:, 1]
    h = np.bincount(np.minimum(it, 12), minlength=13)
    cyc = out.cpu().numpy()[:, 78]
    extra = ("  robot cycles mean %.0f p50 %.0f p99 %.0f max %.0f (max/mean %.3f)" % (cyc.mean(), np.median(cyc), np.percentile(cyc, 99), cyc.max(), cyc.max() / cyc.mean())) if cyc.max() > 0 else ""
    print("step %2d  %.3f ms  max rounds %2d  robots with rounds [1,2,3,4,5-10,>10]: %4d %4d %4d %4d %4d %4d" % (
        step, e0.elapsed_time(e1), it.max(), h[1], h[2], h[3], h[4], h[5:11].sum(), h[11:].sum()) + extra)
