"""Builds linearmpchumanoid_amd/liblmh_hip.so (gfx950) in-tree with hipcc."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
DIAG = os.environ.get("LMH_DIAG") == "1"          # diagnostic build (in-kernel sub-phase stamps): its own file, never the shipped library
VARIANT = os.environ.get("LMH_VARIANT", "")       # experiment builds: LMH_VARIANT=name[:-DFLAG...] -> liblmh_hip_var_<name>.so (never shipped)
SO = os.path.join(_HERE, "liblmh_hip_diag.so" if DIAG else ("liblmh_hip_var_%s.so" % VARIANT.split(":")[0] if VARIANT else "liblmh_hip.so"))
SOURCES = ["lmh_kernels.hip", "lmh_capi.hip"]
HEADERS = ["lmh_device.h", "lmh_nao_model.h", os.path.join("..", "..", "include", "lmh.h")]


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    diag = ["-DLMH_SUBSTAMPS"] if DIAG else []                    # in-kernel sub-phase stamps (diagnostic build)
    diag += VARIANT.split(":")[1:]
    # iterative-ilp machine scheduler: the kernels run one wave per SIMD, so latency (not register pressure /
    # occupancy) is what the scheduler should optimise; measured +19 % ticks/s over the default strategy.
    sched = ["-mllvm", "-amdgpu-sched-strategy=" + os.environ.get("LMH_SCHED", "iterative-ilp")]     # LMH_SCHED: experiments only
    # machine LICM off: in the fused rollout loop it hoists ~100 literal constants (libm polynomial coefficients, LDS offsets) into VGPRs
    # that stay live for the whole launch -> 256 VGPRs + scratch spills; without it the kernel needs 178 VGPRs and no scratch.
    if os.environ.get("LMH_KEEP_MACHINE_LICM") != "1":
        sched += ["-mllvm", "-disable-machine-licm"]
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *sched, *diag,
           *[os.path.join(CSRC, f) for f in SOURCES], "-o", SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


def build_variant(name, flags, force=False, verbose=False):
    """Experiment / checker builds of the same sources into liblmh_hip_var_<name>.so (never the shipped library; loaded with
    LMH_VARIANT=<name>).  `poison` (-DLMH_POISON: every robot starts from an LDS image full of NaNs) is built by
    __graft_entry__.build() and used by tests/test_gpu_round3.py to prove that no result depends on LDS nobody wrote."""
    so = os.path.join(_HERE, "liblmh_hip_var_%s.so" % name)
    srcs = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    if not force and os.path.exists(so) and all(os.path.getmtime(f) <= os.path.getmtime(so) for f in srcs):
        return so
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp",
           "-mllvm", "-disable-machine-licm", *flags, *[os.path.join(CSRC, f) for f in SOURCES], "-o", so]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return so


SHIM_DIR = os.path.join(CSRC, "shim")
SHIM_SO = os.path.join(_HERE, "liblmh_shim.so")
ROOT = os.path.dirname(_HERE)
OFFLINE_BIN = os.path.join(ROOT, "apps", "offline_stand")


def build_shim(force=False, verbose=False):
    """C++ class surface of the reference (Robot, Controller, ...) over the C ABI + the offline app."""
    build(force=False)
    src = os.path.join(SHIM_DIR, "lmh_shim.cpp")
    app = os.path.join(ROOT, "apps", "offline_stand.cpp")
    inc = ["-I" + SHIM_DIR, "-I" + os.path.join(ROOT, "include")]
    newest = max(os.path.getmtime(os.path.join(dp, f)) for dp, _, fs in os.walk(SHIM_DIR) for f in fs)
    if force or not os.path.exists(SHIM_SO) or os.path.getmtime(SHIM_SO) < max(newest, os.path.getmtime(SO)):
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", *inc, src, "-o", SHIM_SO,
               "-L" + _HERE, "-llmh_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if force or not os.path.exists(OFFLINE_BIN) or os.path.getmtime(OFFLINE_BIN) < max(os.path.getmtime(app), os.path.getmtime(SHIM_SO)):
        cmd = ["g++", "-std=c++17", "-O2", *inc, app, "-o", OFFLINE_BIN, "-L" + _HERE, "-llmh_shim", "-llmh_hip",
               "-Wl,-rpath," + _HERE]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SHIM_SO, OFFLINE_BIN


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_shim(force=True, verbose=True))
