#!/usr/bin/env python3
"""Phase-by-phase cycle breakdown of k_trunk2 from a DIAGNOSTIC build with in-kernel s_memtime stamps.

    python tools/stamps.py --build        # here (no GPU): libpinn_hip_stamps.so with -DPINN_STAMPS
    python tools/stamps.py [--arch wide64] [--points 1000000]     # on the GPU box

The stamped kernel is ~10 % slower (the stamps themselves); what is read from it is the SPLIT of a trip
between the four phases and the three barrier waits, per wave, averaged over the waves of the grid."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pinn_for_quantum_wavefunction_surfaces_amd")
SO = os.path.join(PKG, "libpinn_hip_stamps.so")


def build(flags=("-DPINN_STAMPS",), out=SO, tag="stamps"):
    """every translation unit with extra flags into another .so (the diagnostic build; tools/build_flags.py: experiments)"""
    sys.path.insert(0, ROOT)
    from pinn_for_quantum_wavefunction_surfaces_amd.build import UNITS
    objs, procs = [], []
    os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
    for objname, src, extra in UNITS:
        o = os.path.join(PKG, "build", objname.replace(".o", ".%s.o" % tag))
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC",
                                       "-Wno-unused-value"] + list(flags) + extra + ["-c", "-o", o, src],
                                      cwd=os.path.join(PKG, "csrc")))
        objs.append(o)
    for pr in procs:
        if pr.wait() != 0:
            raise RuntimeError("hipcc failed")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--arch", default="wide64")
    ap.add_argument("--points", type=int, default=1000000)
    args = ap.parse_args()
    if args.build:
        return build()
    os.environ["PINN_HIP_LIB"] = SO
    sys.path.insert(0, ROOT)
    import importlib.util
    import numpy as np
    import torch
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    sys.argv = ["bench.py"]
    spec.loader.exec_module(bench)
    import pinn_for_quantum_wavefunction_surfaces_amd as pkg
    a = bench.ARCHS[args.arch]
    dev = torch.device("cuda", 0)
    params = pkg.set_params()
    torch.manual_seed(12345)
    m = pkg.NN_ion(params, a["H"], a["HE"], a["HG"], n_hidden=a["L"], act=a["act_name"], precision=a["precision"]).to(dev)
    tr = pkg.Trainer(m, params, n_points=args.points, lr=8e-3, seed=12345, device=dev)
    waves = a["H"] // 16
    deep = a["L"] == 4                     # k_trunkD: 24 slots per wave, k_trunk2: 8
    NS = 24 if deep else 8
    dbg = torch.zeros(4096 * 8 * 32 + 4096 * 8 * 6, device=dev, dtype=torch.int64)   # phase stamps, then the life records (tools/lifecycle.py)
    for _ in range(50):
        tr.step()
    torch.cuda.synchronize()
    os.environ["PINN_STAMP_PTR"] = "%x" % dbg.data_ptr()
    tr.h.profile_enable(4)
    for _ in range(4):
        tr.step()
    torch.cuda.synchronize()
    ms = tr.h.profile_read(4)
    d = dbg.cpu().numpy()[:4096 * waves * NS].reshape(-1, NS)
    d = d[d.sum(1) > 0].astype(np.float64)
    if deep and a["precision"] != "bf16":
        return report_deep_x3(args, a, d, ms, waves)
    if deep:
        return report_deep(args, a, d, ms, waves)
    trips = (args.points + 15) // 16 / (d.shape[0] / waves)
    names = ["P1 inputs+layer0", "barrier 1", "P2 gemm+act", "barrier 2", "P3 tail+last hidden", "barrier 3",
             "P4 wgrad+adjoint+layer0", "loop top"]
    tot = d.sum(1).mean()
    print("%s: stamped kernel %.4f ms, %d waves, %.1f trips per wave, %.0f cycles per trip (%.2f GHz implied)" % (
        args.arch, ms, d.shape[0], trips, tot / trips, tot / (ms * 1e6)))
    for i in (7, 0, 1, 2, 3, 4, 5, 6):
        print("  %-26s %7.0f cycles per trip  %5.1f %%   (min wave %.0f, max wave %.0f)" % (
            names[i], d[:, i].mean() / trips, 100 * d[:, i].mean() / tot, d[:, i].min() / trips, d[:, i].max() / trips))


def report_deep(args, a, d, ms, waves):
    import numpy as np
    trips = (args.points + 15) // 16 / (d.shape[0] / waves)
    names = {22: "loop top / ring refill", 0: "layer 0", 1: "  barrier", 2: "forward layer 1 (G E G E)", 3: "  barrier",
             4: "forward layer 2", 5: "  barrier", 6: "forward layer 3 + output sums", 7: "  barrier", 8: "tail + last-hidden adjoints",
             9: "  barrier", 10: "reverse 3: wgrad + 2 x (G, E)", 11: "  barrier", 12: "  adjoint stores", 13: "  barrier",
             14: "reverse 2", 15: "  barrier", 16: "  adjoint stores", 17: "  barrier", 18: "reverse 1 (+ layer 0 adjoint)", 19: "  barrier"}
    order = [22, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19]
    tot = d.sum(1).mean()
    print("%s variant %s: stamped kernel %.4f ms, %d waves, %.1f trips per wave, %.0f cycles per trip (%.2f GHz implied)" % (
        args.arch, os.environ.get("PINN_MFMA_VARIANT", "0"), ms, d.shape[0], trips, tot / trips, tot / (ms * 1e6)))
    w = np.arange(d.shape[0]) % waves
    lo, hi = d[w < waves // 2], d[w >= waves // 2]
    bar = sum(d[:, i].mean() for i in order if names[i].startswith("  barrier"))
    print("  %-32s %8s %6s   %8s %8s" % ("phase", "cycles", "%", "waves 0-3", "waves 4-7"))
    for i in order:
        print("  %-32s %8.0f %5.1f %%   %8.0f %8.0f" % (names[i], d[:, i].mean() / trips, 100 * d[:, i].mean() / tot,
                                                      lo[:, i].mean() / trips, hi[:, i].mean() / trips))
    print("  barrier waits in all: %.0f cycles per trip (%.1f %%)" % (bar / trips, 100 * bar / tot))


def report_deep_x3(args, a, d, ms, waves):
    """k_trunkDx: cycles by KIND of phase, summed over the three stages of a tile (forward A | forward B, tail, reverse B |
    forward A, reverse A)"""
    import numpy as np
    trips = (args.points + 15) // 16 / (d.shape[0] / waves)
    names = {23: "loop top / ring refill", 0: "layer 0 (x3)", 1: "  barrier (x3)",
             2: "forward GEMM 1 (x3)", 5: "  algebra + stores 1", 3: "forward GEMM 2", 6: "  algebra + stores 2",
             4: "forward GEMM 3", 7: "  algebra / output sums 3", 8: "  barriers of the forward layers (x9)",
             9: "tail + last-hidden adjoints (x2)", 10: "  barrier (x2)",
             11: "weight gradient 3 (x2)", 14: "reverse GEMM 3", 17: "  adjoint algebra 3",
             12: "weight gradient 2", 15: "reverse GEMM 2", 18: "  adjoint algebra 2",
             13: "weight gradient 1", 16: "reverse GEMM 1", 19: "  layer-0 adjoints",
             20: "  barrier before the adjoint stores (x4)", 21: "  adjoint stores (x4)", 22: "  barrier after them / end of pass (x6)"}
    order = [23, 0, 1, 2, 5, 3, 6, 4, 7, 8, 9, 10, 11, 14, 17, 12, 15, 18, 13, 16, 19, 20, 21, 22]
    tot = d.sum(1).mean()
    print("%s: stamped kernel %.4f ms, %d waves, %.1f trips per wave, %.0f cycles per trip (%.2f GHz implied)" % (
        args.arch, ms, d.shape[0], trips, tot / trips, tot / (ms * 1e6)))
    w = np.arange(d.shape[0]) % waves
    lo, hi = d[w < waves // 2], d[w >= waves // 2]
    print("  %-44s %8s %6s   %8s %8s" % ("phase", "cycles", "%", "waves 0-3", "waves 4-7"))
    for i in order:
        print("  %-44s %8.0f %5.1f %%   %8.0f %8.0f" % (names[i], d[:, i].mean() / trips, 100 * d[:, i].mean() / tot,
                                                      lo[:, i].mean() / trips, hi[:, i].mean() / trips))
    bar = sum(d[:, i].mean() for i in (1, 8, 10, 20, 22))
    print("  barrier waits in all: %.0f cycles per trip (%.1f %%)" % (bar / trips, 100 * bar / tot))


if __name__ == "__main__":
    main()
