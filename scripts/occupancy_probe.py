#!/usr/bin/env python3
"""What would it take to hold a fifth robot per CU?  (VERDICT r02 item 1 iii.)

Builds variants of the kernel library that are only QUERIED, never launched -- the LDS map is cut by a macro, so running them would be
wrong -- and asks the HIP runtime's occupancy calculator (hipOccupancyMaxActiveBlocksPerMultiprocessor) and the code object
(hipFuncGetAttributes) about lmh_rollout_kernel<double>:

    shipped                         LDS 40.9 KB, ~246 VGPRs
    LDS cut to 32 KB                same registers
    LDS cut to 32 KB + 168 VGPRs    (-DLMH_NUM_VGPR=84: the cap that lets a SIMD hold three waves)
    LDS cut to 27 KB + 168 VGPRs    six robots per CU

Run on the GPU box: python scripts/occupancy_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
from linearmpchumanoid_amd import build


def query(path):
    lib = C.CDLL(path)
    g, r, l = C.c_int(), C.c_int(), C.c_int()
    rc = lib.lmh_debug_rollout_occupancy(C.byref(g), C.byref(r), C.byref(l))
    assert rc == 0, rc
    return g.value, r.value, l.value


rows = [("shipped", None, []),
        ("LDS 32 KB", "probe32", ["-DLMH_LDS_PROBE_DOUBLES=4090"]),
        ("LDS 32 KB, 168 VGPRs", "probe32v", ["-DLMH_LDS_PROBE_DOUBLES=4090", "-DLMH_NUM_VGPR=84"]),
        ("LDS 27 KB, 168 VGPRs", "probe27v", ["-DLMH_LDS_PROBE_DOUBLES=3400", "-DLMH_NUM_VGPR=84"])]
print("%-24s %14s %10s %12s" % ("build", "groups per CU", "registers", "static LDS B"))
for name, var, flags in rows:
    path = build.build() if var is None else build.build_variant(var, flags)
    path = path[0] if isinstance(path, (tuple, list)) else path
    g, r, l = query(path)
    print("%-24s %14d %10d %12d" % (name, g, r, l))
