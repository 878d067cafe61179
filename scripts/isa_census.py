#!/usr/bin/env python3
"""Static instruction census of lmh_rollout_kernel<double>: compiles the kernels with -DLMH_PMARK (the per-wave stamps become
'; PMARK n' comments), then counts VALU / SALU / LDS / VMEM instructions between consecutive marks in listing order.  The main path of
one evaluation is straight-line between marks (exec-masked bodies, no counted loops except the cone iteration), so the counts
approximate what a wave issues per phase.  Usage: python scripts/isa_census.py"""
import os, re, subprocess, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "linearmpchumanoid_amd", "csrc", "lmh_kernels.hip")
out = "/tmp/lmh_census.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-DLMH_PMARK",
                       "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-mllvm", "-disable-machine-licm", src, "-o", out],
                      stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z18lmh_rollout_kernelIdLb0EE"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
names = {0: "eval start", 1: "fk / prep done", 2: "joined", 3: "com_x share", 4: "joined", 5: "tree share", 6: "(no join)", 7: "refs share (+prefill)", 8: "joined",
         10: "qp fills", 11: "joined", 12: "Cm | V tile", 13: "joined", 14: "rows loaded", 15: "15x15 solve", 16: "joined", 17: "Y tiles", 18: "set-up done",
         19: "S tile", 20: "S^-1", 21: "T1", 22: "[W|h]", 23: "qv", 24: "cone start", 25: "cone done", 26: "recovery", 27: "joined", 28: "outputs",
         30: "cone entry", 31: "qmax", 32: "12x12 rows", 33: "12x12 solve", 34: "c = Gpinv u", 35: "feasible?", 40: "ldl start", 41: "ldl fwd", 42: "ldl park", 43: "ldl back"}
cnt = collections.Counter(); prev = "kernel entry"; rows = []
def kind(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return None
for l in lines[start:end]:
    m = re.search(r";\s*PMARK (\d+)", l)
    if m:
        rows.append((prev, dict(cnt))); cnt = collections.Counter(); prev = names.get(int(m.group(1)), m.group(1)) + f" [{m.group(1)}]"
        continue
    t = l.strip().split()
    if t and not t[0].startswith((";", ".", "//")) and not t[0].endswith(":"):
        k = kind(t[0])
        if k: cnt[k] += 1
rows.append((prev, dict(cnt)))
print("%-34s %6s %6s %6s %6s" % ("listing segment AFTER this mark", "VALU", "SALU", "LDS", "VMEM"))
for n, c in rows:
    print("%-34s %6d %6d %6d %6d" % (n, c.get("valu", 0), c.get("salu", 0), c.get("lds", 0), c.get("vmem", 0)))
