"""Diagnostic: per-phase cycle stamps (s_memtime) of one controller evaluation, debug kernel."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch, json
from linearmpchumanoid_amd.controller import BatchedController, default_config
from helpers import perturbed_velocities
ik = json.load(open('tests/golden/ik_posture.json'))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctl = BatchedController(B, default_config(dt=1e-3, time_horizon=0.016, z_com=ik['z_com'], warm_start=warm))
ctl.set_refs_stance(2.0, 2)
v = perturbed_velocities(B)
st = ctl.new_state(np.array(ik['q']), v, t=0.0)
pre = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # rollout ticks before the stamped evaluation (realistic active sets)
status = ctl.new_status(); out = ctl.new_out()
if pre:
    ctl.rollout(st, pre, out, status)
for rep in range(3):
    out, status, dbg = ctl.stand_step(st, out=out, status=status, debug=True)   # status carries the active set (warm start)
torch.cuda.synchronize()
d = dbg.cpu().numpy(); s = status.cpu().numpy()
names = ['fk', 'com_x', 'dump', 'newton_euler', 'crba', 'jacobian', 'refs', 'qp', 'outputs']
st_ = d[:, 4000:4010]
dur = np.diff(st_, axis=1)
print("instances", B, "warm", warm, "mean qp iters", s[:, 1].mean(), "max", s[:, 1].max())
tot = (st_[:, 9] - st_[:, 0])
print("total cycles/eval: mean %.0f  p50 %.0f  max %.0f" % (tot.mean(), np.median(tot), tot.max()))
for n, c in zip(names, dur.mean(axis=0)):
    print("  %-14s %8.0f cycles  %5.1f%%" % (n, c, 100 * c / tot.mean()))
q = d[:, 4010:4014]
# qp_setup15 stamps: 4070 fills done, 4074 15x15 solve (+ Z, Mb bp' tiles) done, 4011 [W|h] done, 4012 cone start, 4013 cone done.
# A row is printed only when both of its stamps were written by this build (a stamp the kernel never wrote reads 0).
def row(name, a, b):
    ok = (a != 0) & (b != 0)
    if ok.all():
        c = (b - a).mean()
        print("  %-26s %8.0f cycles  %5.1f%%" % (name, c, 100 * c / tot.mean()))
    else:
        print("  %-26s (stamp not written by this build)" % name)
w = d[:, 4070:4075]
row('qp:fills', st_[:, 7], w[:, 0])
row('qp:Cm,V + 15x15 solve + Z', w[:, 0], w[:, 4])
row('qp:S, S^-1, T1, [W|h]', w[:, 4], q[:, 1])
row('qp:qv, Y join', q[:, 1], q[:, 2])
row('qp:cone solve', q[:, 2], q[:, 3])
row('qp:recover', q[:, 3], st_[:, 8])
# per-solve stamps inside the cone QP loop
for itn in range(1, 7):
    m = s[:, 1] >= itn
    if m.sum() == 0: break
    if not (d[m, 4020 + 2 * itn] != 0).all(): continue     # the all-free fast path does not enter the loop
    dur_i = d[m, 4021 + 2 * itn] - d[m, 4020 + 2 * itn]
    print("  cone loop solve #%d: n=%4d  mean %7.0f cycles  (mean |F| %.1f)" % (itn, m.sum(), dur_i.mean(), d[m, 4050 + itn].mean()))

ss = d[:, 3900:3915]
def seg(a, b): return (b - a).mean()
rows = [("fk:sincos", st_[:,0], ss[:,0]), ("fk:Lc+T0 build", ss[:,0], ss[:,1]), ("fk:chain 8 steps", ss[:,1], st_[:,1]),
        ("comx:CoM", st_[:,1], ss[:,2]), ("comx:E,p", ss[:,2], ss[:,3]), ("comx:B", ss[:,3], ss[:,4]), ("comx:copies,vhat", ss[:,4], st_[:,2]),
        ("ne:vel + acc sweeps", st_[:,3], ss[:,6]), ("ne:forces", ss[:,6], ss[:,7]), ("ne:backward", ss[:,7], ss[:,8]), ("ne:base,C,Jpqp", ss[:,8], st_[:,4]),
        ("crba:init", st_[:,4], ss[:,9]), ("crba:levels", ss[:,9], ss[:,10]), ("crba:roots", ss[:,10], ss[:,11]), ("crba:joint cols", ss[:,11], st_[:,5]),
        ("refs:AG", st_[:,6], ss[:,12]), ("refs:h,vfoot,pdj", ss[:,12], ss[:,13]), ("refs:mpc", ss[:,13], ss[:,14]), ("refs:pd mom/feet", ss[:,14], st_[:,7])]
for n, a, b in rows:
    print("  %-20s %8.0f cycles" % (n, seg(a, b)))

# distribution of the free-set size and what it costs (the launch time of the rollout is set by the slowest robot)
nF = np.array([bin((~int(x)) & 0xFFFFFFFF).count("1") for x in s[:, 3]])
print("free-set size |F| histogram (warm):")
for lo, hi in ((0, 8), (9, 16), (17, 24), (25, 31), (32, 32)):
    m = (nF >= lo) & (nF <= hi)
    if m.sum():
        cone = (q[m, 3] - q[m, 2]).mean(); s1 = (d[m, 4023] - d[m, 4022]).mean()
        print("  |F| in [%2d,%2d]: %4d robots  mean eval cycles %7.0f  qp %7.0f  max %7.0f  | cone phase %6.0f  free-set solve #1 %6.0f  iters %.2f" % (lo, hi, m.sum(), tot[m].mean(), dur[m, 7].mean(), tot[m].max(), cone, s1, s[m, 1].mean()))

if os.environ.get("LMH_DIAG_NW2"):
    w1 = d[:, 3950:3960]
    print("two-wave schedule, cycles from the start of the evaluation (wave 0 | wave 1) at the phase boundaries:")
    for i, n in enumerate(['start', 'fk done', 'com_x done', '(dump)', 'NE done (w0)', 'own tree share done', 'tree phases joined', 'refs done', 'qp done', 'outputs done']):
        a0 = (st_[:, i] - st_[:, 0]).mean(); a1 = (w1[:, i] - st_[:, 0]).mean() if w1[:, i].any() else float('nan')
        print("  %-18s %8.0f | %8.0f" % (n, a0, a1))
