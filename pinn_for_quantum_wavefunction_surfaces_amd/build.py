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
# translation units: (object name, source, extra flags).  pinn_kernels.hip: -fno-slp-vectorize, because the SLP
# pass packs scalar FMAs into v_pk_* and then has to copy the SGPR-resident weights into VGPR pairs first (+40 %
# VALU instructions).  The matrix-core trunk kernels are one translation unit per kernel family (they compile in
# parallel); pinn_trunk_general.hip and pinn_trunk_deep.hip are compiled twice: part 0 (everything but the 4x128
# kernels) keeps SLP (2-4 % faster narrow kernels), part 1 (the 4x128 kernels) drops it (fewer spills at that
# register pressure: 4-19 % faster).
NO_SLP_128X4 = ["-DPINN_TRUNK_PART=1", "-fno-slp-vectorize"]
UNITS = [("pinn_kernels.o", "pinn_kernels.hip", ["-fno-slp-vectorize"]),
         ("pinn_trunk_mfma.o", "pinn_trunk_mfma.hip", []),
         ("pinn_trunk_merged.o", "pinn_trunk_merged.hip", []),
         ("pinn_trunk_deep.o", "pinn_trunk_deep.hip", []),
         ("pinn_trunk_deep_p1.o", "pinn_trunk_deep.hip", NO_SLP_128X4),
         ("pinn_trunk_general.o", "pinn_trunk_general.hip", []),
         ("pinn_trunk_general_p1.o", "pinn_trunk_general.hip", NO_SLP_128X4),
         ("pinn_rnet_mfma.o", "pinn_rnet_mfma.hip", []),
         ("pinn_xchg.o", "pinn_xchg.hip", [])]
SOURCES = sorted({u[1] for u in UNITS})
HEADERS = ["pinn_math.h", "pinn_common.h", "pinn_mfma_common.h", os.path.join("..", "..", "include", "pinn.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in list(SOURCES) + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    """Compile if the .so is missing or older than its sources. Returns the .so path."""
    if not force and not _stale():
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build %s" % SO)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for objname, src, extra in UNITS:
        obj = os.path.join(objdir, objname)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(
                os.path.getmtime(os.path.join(CSRC, d)) for d in [src] + HEADERS):
            continue
        cmd = [hipcc] + FLAGS + extra + ["-c", "-o", obj, os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return SO


def build_example(force=False, verbose=False):
    """examples/train_capi.cpp -> examples/train_capi: the C ABI driven from plain C++/HIP."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "examples", "train_capi.cpp")
    exe = os.path.join(root, "examples", "train_capi")
    if not force and os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(src), os.path.getmtime(SO)):
        return exe
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-O2", "--offload-arch=gfx950", "-I" + os.path.join(root, "include"), src,
           "-L" + HERE, "-lpinn_hip", "-Wl,-rpath,$ORIGIN/../pinn_for_quantum_wavefunction_surfaces_amd", "-o", exe]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return exe


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
    print(build_example(force=True, verbose=True))
