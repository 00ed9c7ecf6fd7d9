"""Builds the HIP library (csrc/pinn_kernels.hip -> libpinn_hip.so) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this also runs in the build container.
The .so is git-ignored but travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libpinn_hip.so")
SOURCES = ["pinn_kernels.hip"]
HEADERS = ["pinn_math.h", os.path.join("..", "..", "include", "pinn.h")]
# -fno-slp-vectorize: the SLP pass packs scalar FMAs into v_pk_* and then has to copy
# the SGPR-resident weights into VGPR pairs first (+40 % VALU instructions).
FLAGS = ["-O3", "-fno-slp-vectorize", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
         "-Wno-unused-value"]


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    """Compile if the .so is missing or older than its sources. Returns the .so path."""
    if not force and not _stale():
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build %s" % SO)
    cmd = [hipcc] + FLAGS + ["-o", SO] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return SO


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
