"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/lmh.h declares.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lmh.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lmh_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(hip_lib):
    from linearmpchumanoid_amd import capi
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/lmh.h but not exported"
    assert sorted(capi.EXPORTS) == names


def test_no_torch_types_in_abi():
    src = open(os.path.join(ROOT, "include", "lmh.h")).read()
    assert "torch" not in src and "at::" not in src and "std::" not in src


def test_config_defaults_are_reference_literals(hip_lib):
    from linearmpchumanoid_amd.capi import LmhConfig
    cfg = LmhConfig()
    hip_lib.lmh_config_default(C.byref(cfg))
    assert (cfg.dt, cfg.time_horizon, cfg.gravity, cfg.alpha, cfg.beta) == (0.01, 0.5, 9.81, 1e-3, 1.0)
    assert (cfg.mu, cfg.kp_joints, cfg.kd_joints, cfg.kp_mom, cfg.kd_mom, cfg.kp_feet, cfg.kd_feet) == (0.7, 300, 34, 10, 6.32, 500, 44)
    assert (cfg.w_com_lin, cfg.w_com_ang, cfg.w_base_pos, cfg.w_base_ang, cfg.w_joints, cfg.w_force, cfg.w_foot) == (4000, 0, 10, 10, 1, 1, 100000)
    assert cfg.eps_coeff == 1e-8


def test_nominal_links_match_oracle_table(hip_lib):
    """Product table == oracle table == createNaoParameters values (typos included)."""
    import numpy as np
    from oracle.pyoracle import nao_raw_links
    raw = np.zeros((28, 13))
    hip_lib.lmh_nominal_links(raw.ctypes.data_as(C.c_void_p))
    assert np.array_equal(raw, nao_raw_links())
    assert raw[13, 4 + 5] == 1.8740920495e-55 and raw[19, 4 + 1] == 5.71599e-5 and raw[19, 4 + 3] == 5.71599e-6


def test_fails_loudly_without_gpu(hip_lib):
    """No CPU fallback: on a machine without a HIP device create() must fail, not compute."""
    import torch
    if torch.cuda.is_available():
        return
    from linearmpchumanoid_amd.capi import LmhConfig
    cfg = LmhConfig()
    hip_lib.lmh_config_default(C.byref(cfg))
    h = C.c_void_p()
    rc = hip_lib.lmh_create(C.byref(cfg), 4, 0, C.byref(h))
    assert rc == -1 and not h.value
    assert b"no HIP device" in hip_lib.lmh_last_error()


def test_product_does_not_touch_oracle():
    """The shipped package must not import / link the oracle."""
    pkg = os.path.join(ROOT, "linearmpchumanoid_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("# oracle-free", ""), f"{f} mentions the oracle"


def test_header_is_plain_c_and_a_c_caller_links(tmp_path):
    """include/lmh.h must be consumable by a C compiler (the boundary is a C ABI, not a C++ one): a C99 translation
    unit that takes the address of every entry point compiles with -pedantic and links against liblmh_hip.so; it also
    checks that sizeof(lmh_config) agrees with the ctypes mirror (capi.LmhConfig)."""
    import subprocess
    from linearmpchumanoid_amd import capi
    names = declared_symbols()
    src = tmp_path / "c_caller.c"
    refs = "\n".join(f"    p[{i}] = (void (*)(void))&{n};" for i, n in enumerate(names))
    src.write_text(f"""#include <stdio.h>
#include "lmh.h"
int main(void) {{
    void (*p[{len(names)}])(void);
    lmh_config cfg;
{refs}
    lmh_config_default(&cfg);
    printf("%zu %d %g %g\\n", sizeof(lmh_config), cfg.precision, cfg.dt, cfg.eps_coeff);
    return p[0] == 0;
}}
""")
    exe = tmp_path / "c_caller"
    libdir = os.path.dirname(capi.SO_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-Wno-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe), "-L", libdir, "-llmh_hip", f"-Wl,-rpath,{libdir}"])
    out = subprocess.check_output([str(exe)]).decode().split()       # lmh_config_default touches no device
    assert int(out[0]) == C.sizeof(capi.LmhConfig)
    assert int(out[1]) == capi.PRECISION_FP64 and float(out[2]) == 0.01 and float(out[3]) == 1e-8


def test_config_carries_the_mpc_sample_time(hip_lib):
    """lmh_config.mpc_dt (0 = dt) is the last field of the record; N = int(time_horizon / mpc_dt) is validated at create time even
    without a device (bad values are refused before the device is looked for)."""
    import ctypes as C
    from linearmpchumanoid_amd.capi import LmhConfig
    from linearmpchumanoid_amd.controller import default_config
    assert LmhConfig._fields_[-1][0] == "mpc_dt" and C.sizeof(LmhConfig) == 8 * 21 + 4 * 6 + 8 * 5
    cfg = default_config()
    assert cfg.mpc_dt == 0.0
    h = C.c_void_p()
    bad = default_config(dt=1e-3, time_horizon=0.32, mpc_dt=-1.0)
    assert hip_lib.lmh_create(C.byref(bad), 1, 0, C.byref(h)) == -2 and b"mpc_dt" in hip_lib.lmh_last_error()
    bad = default_config(dt=1e-3, time_horizon=0.7, mpc_dt=1e-2)         # N = 70 > 64
    assert hip_lib.lmh_create(C.byref(bad), 1, 0, C.byref(h)) == -2 and b"mpc_dt" in hip_lib.lmh_last_error()
    ok = default_config(dt=1e-3, time_horizon=0.7)                        # mpc_dt = dt: N = 700 is refused the same way
    assert hip_lib.lmh_create(C.byref(ok), 1, 0, C.byref(h)) == -2
