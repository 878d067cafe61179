#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (scripts/profile_rollout.sh) into profiles/<tag>_kernel_stats.csv and
profiles/<tag>_rollout_summary.json.  Usage: python scripts/summarise_profile.py r02_c3 [instances ticks]
(defaults: the workload of the bench line found in the profile directory)"""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
_bl = os.path.join(src, "bench_line.json")
_cfg = json.loads(open(_bl).read().strip().splitlines()[-1])["config"] if os.path.exists(_bl) and os.path.getsize(_bl) else {}
B = int(sys.argv[2]) if len(sys.argv) > 2 else int(_cfg.get("instances_per_gpu", 1024))
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else int(_cfg.get("ticks_per_step", 10))
KERNEL = "lmh_rollout_kernel"


def find(sub, suffix):
    hits = glob.glob(os.path.join(src, sub, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit(f"missing {sub}/*{suffix}")
    return hits[0]


stats = find("kt", "kernel_stats.csv")
shutil.copy(stats, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
avg_ns = calls = None
for row in csv.DictReader(open(stats)):
    if KERNEL in row["Name"]:                                  # "void lmh_rollout_kernel<double>(...)"
        avg_ns, calls = float(row["AverageNs"]), int(row["Calls"])
pmc = {}
meta = {}
for sub in ("fetch", "write", "sq", "sq2", "f64", "ins", "grbm"):
    try:
        f = find(sub, "counter_collection.csv")
    except SystemExit:
        continue
    per = {}
    for row in csv.DictReader(open(f)):
        if KERNEL not in row["Kernel_Name"]:
            continue
        meta = {k: row[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
        per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, d in per.items():
        vals = sorted(d.items(), key=lambda kv: int(kv[0]))[1:]      # drop the first (warm-up) launch
        pmc[name] = sum(v for _, v in vals) / max(1, len(vals))
evals = B * ticks * 4
g = lambda k: pmc.get(k, float("nan"))
# counted fp64 work: the SQ counts wave-level instructions per class; one instruction occupies 64 lane slots whatever the EXEC mask, so
# x 64 is the lane CAPACITY the kernel spent on fp64 arithmetic -- an upper bound of the useful flop (lanes switched off by EXEC, and
# lanes computing padding / safe-address dummies, are included).  FMA = 2 flop; one MFMA_MOPS_F64 unit = 512 flop.
_f64 = [pmc.get(k) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")]
counted = None
if all(v is not None for v in _f64) and "SQ_INSTS_VALU_MFMA_MOPS_F64" in pmc:
    counted = (64.0 * (_f64[0] + _f64[1] + 2.0 * _f64[2] + _f64[3]) + 512.0 * pmc["SQ_INSTS_VALU_MFMA_MOPS_F64"]) / evals
derived = {
    "counted_fp64_flop_per_eval": counted,
    "counted_fp64_flop_note": None if counted is None else
        "64 lanes x (SQ_INSTS_VALU_ADD_F64 + MUL_F64 + 2 FMA_F64 + TRANS_F64) + 512 x SQ_INSTS_VALU_MFMA_MOPS_F64 per evaluation: lane capacity of the fp64 "
        "instructions issued (EXEC-masked and padding lanes included), an upper bound of the useful arithmetic",
    "f64_valu_insts_per_eval": None if counted is None else {"add": _f64[0] / evals, "mul": _f64[1] / evals, "fma": _f64[2] / evals, "trans": _f64[3] / evals},
    "other_valu_insts_per_eval": {k[len("SQ_INSTS_VALU_"):].lower(): pmc[k] / evals for k in ("SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT",
                                  "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32") if k in pmc},
    "smem_insts_per_eval": pmc["SQ_INSTS_SMEM"] / evals if "SQ_INSTS_SMEM" in pmc else None,
    "branch_insts_per_eval": pmc["SQ_INSTS_BRANCH"] / evals if "SQ_INSTS_BRANCH" in pmc else None,
    "valu_insts_per_eval": g("SQ_INSTS_VALU") / evals, "salu_insts_per_eval": g("SQ_INSTS_SALU") / evals,
    "lds_insts_per_eval": g("SQ_INSTS_LDS") / evals, "mfma_f64_mops_per_eval": g("SQ_INSTS_VALU_MFMA_MOPS_F64") / evals,
    "wave_cycles_per_eval": g("SQ_WAVE_CYCLES") / evals,
    "wait_any_frac": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), "wait_inst_frac": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
    "active_inst_frac": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
    "lds_bank_conflict_frac": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"),
    "hbm_write_bytes_per_launch": g("WRITE_SIZE") * 1024.0,
    "hbm_fetch_bytes_per_launch_raw": g("FETCH_SIZE") * 1024.0,
    "hbm_fetch_bytes_per_launch_x2_gfx950_correction": g("FETCH_SIZE") * 2048.0,
    "effective_clock_ghz_from_grbm": (g("GRBM_GUI_ACTIVE") / 8.0) / avg_ns if avg_ns else None,
    "note": "FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section); "
            "access width here is 8 B/lane (uncalibrated), so the read side is bracketed by [raw, 2x raw]",
}
out = {
    "command": "scripts/profile_rollout.sh: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline [config args] (the default bench command); PMC: separate --pmc passes, --steps 3 --warmup 1",
    "kernel": KERNEL, "dispatch": meta, "workload": f"{B} instances x {ticks} RK4 ticks per launch",
    "avg_launch_ns": avg_ns, "calls": calls, "pmc_per_launch": pmc, "derived": derived,
}
bl = os.path.join(src, "bench_line.json")
if os.path.exists(bl) and os.path.getsize(bl):
    out["bench_line_under_profiler"] = json.loads(open(bl).read().strip().splitlines()[-1])
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_rollout_summary.json"), "w"), indent=1)
print(json.dumps({"avg_launch_ns": avg_ns, **derived}, indent=1))
