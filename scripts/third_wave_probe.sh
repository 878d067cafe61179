#!/bin/bash
# Runs ON THE GPU BOX.  What would a third wave per SIMD (a fifth / sixth robot per CU) buy?  (measure, do not estimate.)
# Probe builds whose RESULTS ARE WRONG on purpose -- the LDS array is cut to 32 / 27 KB, so everything the map places above that is out of
# range (LDS reads beyond a workgroup's allocation return zero, writes are dropped: no fault) -- but whose instruction stream is the shipped
# one; same method as the look-ahead upper bound of round 3.  Controls separate the three effects:
#   lds32 @ 4 robots / CU            garbage data alone (same occupancy, same registers)
#   v168 @ 4 / CU                    the 168-register cap alone (results right)
#   lds32v168 @ 4 / CU               both, still two waves per SIMD
#   lds32v168 @ 5 / CU               + the fifth robot (2.5 waves per SIMD)
#   lds27v168 @ 6 / CU               three waves per SIMD
# The variants are built on the CPU box (python scripts/third_wave_probe.py build) and travel with the snapshot.
# usage: bash scripts/third_wave_probe.sh OUTDIR [bench args]
O=$1; shift
EXTRA=("$@")
mkdir -p $O
run() {   # name variant groups_per_cu
  LMH_VARIANT=$2 LMH_ROLLOUT_GROUPS_PER_CU=$3 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "${EXTRA[@]}" > $O/$1.json 2> $O/$1.err
  python - "$O/$1.json" "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:>22}: {d['value']/1e6:8.3f} M ticks/s  kernel {d['roofline']['kernel_ms']:9.2f} ms  flagged {d['instances_flagged']}")
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
}
run shipped_4 "" 4
run lds32_4 wg_lds32 4
run v168_4 wg_v168 4
run lds32v168_4 wg_lds32v168 4
run lds32v168_5 wg_lds32v168 5
run lds27v168_6 wg_lds27v168 6
run shipped_4_again "" 4
python - <<'PY'
import ctypes as C, os
for name in ("", "wg_lds32", "wg_v168", "wg_lds32v168", "wg_lds27v168"):
    p = os.path.join("linearmpchumanoid_amd", "liblmh_hip%s.so" % ("_var_" + name if name else ""))
    lib = C.CDLL(os.path.abspath(p)); g, r, l = C.c_int(), C.c_int(), C.c_int()
    rc = lib.lmh_debug_rollout_occupancy(C.byref(g), C.byref(r), C.byref(l))
    print(f"{name or 'shipped':>14}: runtime occupancy {g.value} workgroups / CU, {r.value} registers, {l.value} B of static LDS (rc {rc})")
PY
