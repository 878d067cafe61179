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
