#!/usr/bin/env python3
"""Life cycle of the waves of one k_trunk2 / k_trunk16 launch (--arch wide64 | wide128 | ref16) on the chip-wide 100 MHz clock (diagnostic -DPINN_STAMPS build,
`python tools/stamps.py --build` first): when each wave entered the kernel, reached its first tile, finished its last
tile and left, relative to the first wave's entry.

    python tools/lifecycle.py [--points 4000]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pinn_for_quantum_wavefunction_surfaces_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=4000)
    ap.add_argument("--arch", default="wide64")
    ap.add_argument("--save", default="", help="write per-block records (block, xcc, se, sh, cu, entry, done) to this .npy")
    args = ap.parse_args()
    os.environ["PINN_HIP_LIB"] = os.path.join(PKG, "libpinn_hip_stamps.so")
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
    if a.get("fine_tune"):
        m.freezeDecayUnit()
        m.freezeBase()
    tr = pkg.Trainer(m, params, n_points=args.points, lr=8e-3, seed=12345, device=dev)
    waves = max(a["H"] // 16, 4)                 # waves per block of the trunk kernel (k_trunk16: 4)
    OFF = 4096 * 8 * 32                           # life records sit behind the phase-stamp region (csrc/pinn_mfma_common.h)
    dbg = torch.zeros(OFF + 4096 * 8 * 6, device=dev, dtype=torch.int64)
    for _ in range(50):
        tr.step()
    torch.cuda.synchronize()
    os.environ["PINN_STAMP_PTR"] = "%x" % dbg.data_ptr()
    tr.h.profile_enable(4)
    for _ in range(4):
        tr.step()
    torch.cuda.synchronize()
    ms = tr.h.profile_read(4)
    lf = dbg.cpu().numpy()[OFF:].reshape(-1, 6)
    lf = lf[lf[:, 0] > 0].astype(np.float64)
    hw, xcc = lf[:, 4].astype(np.int64), lf[:, 5].astype(np.int64)
    lf = lf[:, :4]
    t0 = lf[:, 0].min()
    us = (lf - t0) / 100.0
    print("%s, %d points: kernel %.1f us by HIP events (with the weight pack); %d waves wrote a record" % (args.arch, args.points, 1e3 * ms, len(us)))
    for k, name in enumerate(["kernel entry", "first tile starts", "last tile done", "exit"]):
        print("  %-18s min %7.2f  median %7.2f  max %7.2f us after the first wave's entry" % (name, us[:, k].min(), np.median(us[:, k]), us[:, k].max()))
    d = us[:, 1:] - us[:, :-1]
    for k, name in enumerate(["prologue", "tile loop", "epilogue"]):
        print("  %-18s min %7.2f  median %7.2f  max %7.2f us per wave" % (name, d[:, k].min(), np.median(d[:, k]), d[:, k].max()))


    # who shares a CU: HW_ID bits (gfx9 layout) wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13; XCC_ID 3:0
    cu = ((xcc & 15) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    simd = (hw >> 4) & 3
    blocks = np.arange(len(us)) // waves
    print("  distinct CUs %d, blocks %d; SIMD ids seen %s" % (len(set(cu.tolist())), blocks.max() + 1, sorted(set(simd.tolist()))))
    done = {}
    for b in range(blocks.max() + 1):
        m = blocks == b
        done.setdefault(int(cu[m][0]), []).append((float(us[m, 2].max()), float(us[m, 0].min()), b))
    sizes = sorted(set(len(v) for v in done.values()))
    print("  blocks per CU: %s" % sizes)
    if sizes and sizes[-1] > 2:
        k = sizes[-1]
        rows = np.array([[t[0] for t in sorted(v, key=lambda t: t[1])] for v in done.values() if len(v) == k])
        print("  tiles done after (us), blocks of a CU in the order they entered: " + "  ".join("%.1f" % x for x in np.median(rows, axis=0)))
    pairs = [sorted(v) for v in done.values() if len(v) == 2]
    if pairs:
        first = np.array([p[0][0] for p in pairs]); second = np.array([p[1][0] for p in pairs])
        print("  CUs with two blocks: %d; the block that finishes its tiles first: median %.1f us (min %.1f, max %.1f); the other: median %.1f us (min %.1f, max %.1f)" % (
            len(pairs), np.median(first), first.min(), first.max(), np.median(second), second.min(), second.max()))
        early = np.array([p[0][1] < p[1][1] for p in pairs])
        print("  the first finisher is the block that ENTERED first on %d of %d CUs; block-index distance of the partners: %s" % (
            early.sum(), len(pairs), sorted(set(abs(p[0][2] - p[1][2]) for p in pairs))[:8]))
    # per XCD (XCC_ID) and per shader engine: is a part of the chip consistently slower?
    xid = (xcc & 15)
    bdone = np.array([us[blocks == b, 2].max() for b in range(blocks.max() + 1)])
    bx = np.array([int(xid[blocks == b][0]) for b in range(blocks.max() + 1)])
    bse = np.array([int(((hw >> 13) & 7)[blocks == b][0]) for b in range(blocks.max() + 1)])
    bcu = np.array([int(((hw >> 8) & 15)[blocks == b][0]) for b in range(blocks.max() + 1)])
    print("  last tile done by XCD (median us; blockIdx %% 8 of its blocks): " + "  ".join(
        "%d: %.0f %s" % (x, np.median(bdone[bx == x]), sorted(set((np.nonzero(bx == x)[0] % 8).tolist()))) for x in sorted(set(bx.tolist()))))
    print("  by shader engine: " + "  ".join("%d: %.0f" % (e, np.median(bdone[bse == e])) for e in sorted(set(bse.tolist()))))
    print("  by CU id within its engine: " + "  ".join("%d: %.0f" % (c, np.median(bdone[bcu == c])) for c in sorted(set(bcu.tolist()))))
    if args.save:
        ben = np.array([us[blocks == b, 0].min() for b in range(blocks.max() + 1)])
        bsh = np.array([int(((hw >> 12) & 1)[blocks == b][0]) for b in range(blocks.max() + 1)])
        np.save(args.save, np.stack([np.arange(len(bdone)), bx, bse, bsh, bcu, ben, bdone], 1))
    h, edges = np.histogram(us[:, 2], bins=12)
    print("  histogram of 'last tile done' (us): " + "  ".join("%.0f-%.0f:%d" % (edges[i], edges[i + 1], h[i]) for i in range(len(h))))


if __name__ == "__main__":
    main()
