#!/usr/bin/env python3
"""Static instruction census of lmh_rollout_kernel<double>: compiles the kernels with -DLMH_PMARK (the per-wave stamps become
'; PMARK n' comments), then counts instructions between consecutive marks in listing order, by class.  The main path of one evaluation
is straight-line between marks (exec-masked bodies, no counted loops except the cone iteration), so the counts approximate what a wave
issues per phase.  Classes of VALU instructions: f64 = fp64 arithmetic (add / mul / fma / fmac incl. DPP forms, rcp, rndne, min / max ...),
mfma, sel = v_cndmask, lane = v_readlane / v_readfirstlane / v_writelane, mov = v_mov (incl. DPP moves) / v_accvgpr, int = integer
add / shift / mul / mad / logic, cmp = v_cmp*, cvt, oth.
Usage: python scripts/isa_census.py [--reuse] [--out FILE] [extra -D flags]      (--reuse: take /tmp/lmh_census.s as it is)"""
import collections
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "linearmpchumanoid_amd", "csrc", "lmh_kernels.hip")
out = "/tmp/lmh_census.s"
argv = sys.argv[1:]
reuse = "--reuse" in argv
dest = argv[argv.index("--out") + 1] if "--out" in argv else None
extra = [a for a in argv if a.startswith("-D")]
if not reuse:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-DLMH_PMARK", "-DLMH_ROLLOUT_ONLY",
                           "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-mllvm", "-disable-machine-licm", *extra, src, "-o", out],
                          stderr=subprocess.DEVNULL if "--verbose" not in argv else None)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z18lmh_rollout_kernelIdLb0EE"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
names = {0: "eval start", 1: "fk / prep done", 2: "joined", 3: "com_x share", 4: "joined", 5: "tree share", 6: "(no join)", 7: "refs share (+prefill)", 8: "joined",
         10: "qp fills", 11: "joined", 12: "Cm | V tile", 13: "joined", 14: "rows loaded", 15: "15x15 solve", 16: "joined", 17: "Y tiles", 18: "set-up done",
         19: "S tile", 20: "S^-1", 21: "T1", 22: "[W|h]", 23: "qv", 24: "cone start", 25: "cone done", 26: "recovery", 27: "joined", 28: "outputs",
         30: "cone entry", 31: "qmax", 32: "12x12 rows", 33: "12x12 solve", 34: "c = Gpinv u", 35: "feasible?", 40: "ldl start", 41: "ldl fwd", 42: "ldl park", 43: "ldl back",
         44: "fk: sincos", 45: "fk: local T", 46: "com", 47: "E, p", 48: "B", 49: "NE fwd sweep", 50: "NE body forces", 51: "NE bwd sweep", 52: "crba levels",
         60: "w1 rk4 stage", 61: "ahead: refs", 62: "ahead: fk", 63: "kinv prework", 64: "jacobian", 66: "refs A (w1)", 67: "refs B (w0)", 68: "prefill15",
         69: "w1 rk4 done", 70: "ahead done", 71: "eval done", 72: "stage end", 80: "cone: thin solve", 81: "cone: thin done", 82: "cone: general solve", 83: "cone: general done"}
CLS = ["f64", "mfma", "sel", "lane", "mov", "int", "cmp", "cvt", "oth"]


def vclass(op):
    if "mfma" in op:
        return "mfma"
    if op.startswith("v_cndmask"):
        return "sel"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if op.startswith(("v_mov_", "v_accvgpr", "v_swap")):
        return "mov"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_cvt"):
        return "cvt"
    if "f64" in op:
        return "f64"
    if re.match(r"v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|and|or|xor|not|bfe|bfi|min|max|add3|lshl_add|lshl_or|and_or|or3|xad|perm|alignbit|mbcnt|bcnt|ffb|med3)", op) and "f32" not in op and "f16" not in op:
        return "int"
    return "oth"


def kind(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return None


cnt = collections.Counter(); prev = "kernel entry"; rows = []
tot = collections.Counter()
for l in lines[start:end]:
    m = re.search(r";\s*PMARK (\d+)", l)
    if m:
        rows.append((prev, dict(cnt))); cnt = collections.Counter(); prev = names.get(int(m.group(1)), m.group(1)) + f" [{m.group(1)}]"
        continue
    t = l.strip().split()
    if t and not t[0].startswith((";", ".", "//")) and not t[0].endswith(":"):
        k = kind(t[0])
        if k:
            cnt[k] += 1; tot[k] += 1
            if k == "valu":
                c = vclass(t[0]); cnt[c] += 1; tot[c] += 1
            if k == "salu" and t[0].startswith(("s_nop", "s_waitcnt")):
                cnt["nopwait"] += 1; tot["nopwait"] += 1
rows.append((prev, dict(cnt)))
hdr = "%-30s %6s %5s %5s %5s | " % ("listing segment AFTER this mark", "VALU", "SALU", "LDS", "VMEM") + " ".join("%5s" % c for c in CLS) + " | nop/wait"
txt = [hdr]
for n, c in rows:
    if not c:
        continue
    txt.append("%-30s %6d %5d %5d %5d | " % (n, c.get("valu", 0), c.get("salu", 0), c.get("lds", 0), c.get("vmem", 0)) + " ".join("%5d" % c.get(k, 0) for k in CLS) + " | %5d" % c.get("nopwait", 0))
txt.append("%-30s %6d %5d %5d %5d | " % ("TOTAL (static)", tot["valu"], tot["salu"], tot["lds"], tot["vmem"]) + " ".join("%5d" % tot[k] for k in CLS) + " | %5d" % tot["nopwait"])
meta = "\n".join(lines).split(".name:           _Z18lmh_rollout_kernelIdLb0EE")[1][:900] if ".name:           _Z18lmh_rollout_kernelIdLb0EE" in "\n".join(lines) else ""
for key in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
    mm = re.search(r"\." + key + r":\s*(\d+)", meta)
    if mm:
        txt.append(f"{key}: {mm.group(1)}")
print("\n".join(txt))
if dest:
    open(dest, "w").write("\n".join(txt) + "\n")
