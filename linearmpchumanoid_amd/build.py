"""Builds linearmpchumanoid_amd/liblmh_hip.so (gfx950) in-tree with hipcc."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO = os.path.join(_HERE, "liblmh_hip.so")
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
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           *[os.path.join(CSRC, f) for f in SOURCES], "-o", SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
