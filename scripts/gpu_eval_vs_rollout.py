"""Diagnostic: cost of one plain evaluation launch vs one evaluation inside the fused rollout."""
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
ctl.rollout(st, 40, out, status)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn, n_eval in (("plain evaluation kernel", lambda: ctl.stand_step(st, out, status), 1), ("rollout, 10 ticks", lambda: ctl.rollout(st, 10, out, status), 40)):
    st2 = st.clone()
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-26s %8.2f us per launch, %6.2f us per evaluation" % (name, e0.elapsed_time(e1) * 100, e0.elapsed_time(e1) * 100 / n_eval))
    st.copy_(st2)
