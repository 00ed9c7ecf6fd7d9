#!/usr/bin/env python3
"""Diagnostics for kernel measurement variants (PINN_MFMA_VARIANT): one fused loss+gradient call per variant on the same
points and weights, compared entry by entry with variant 0 (GPU box; uses the oracle only for the printed reference error).
    python tests/diag/var_check.py --arch 128,4 --variants 0,11,12,14 [--n 100000]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("PINN_MFMA_VARIANT_LIVE", "1")   # variants are switched inside this process
import numpy as np  # noqa: E402
import torch  # noqa: E402
import pinn_for_quantum_wavefunction_surfaces_amd as pkg  # noqa: E402
from pinn_for_quantum_wavefunction_surfaces_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="128,4")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    H, L = [int(v) for v in a.arch.split(",")]
    f = pkg.model.FORMULATION_A
    prec = {"bf16": capi.PINN_PREC_BF16, "bf16x3": capi.PINN_PREC_BF16X3}[a.prec]
    arch = capi.Arch(H, L, 32, 10, 1, 2, 1.0, prec)
    rng = np.random.default_rng(3)
    n = a.n
    x, y, z = rng.uniform(-18, 18, (3, n)).astype(np.float32)
    R = rng.uniform(0.2, 4, n).astype(np.float32)
    r1 = np.sqrt((x - R) ** 2 + y * y + z * z)
    r2 = np.sqrt((x + R) ** 2 + y * y + z * z)
    mask = ((r1 >= 17.5).astype(np.uint8) + 2 * (r2 >= 17.5).astype(np.uint8))
    h = capi.Handle(arch, 0, n)
    theta = (rng.normal(0, 1, h.P) * 0.08).astype(np.float32)
    dev = "cuda:0"
    t = lambda v: torch.tensor(v, device=dev)
    args = [t(theta), t(x), t(y), t(z), t(R), t(mask)]
    n1, n2 = int((mask & 1).sum()), int(((mask >> 1) & 1).sum())
    ham = capi.Ham(f["a"], f["b"], f["c"], 1.0, 0.0, 0.0, 1.0 / n, 1.0 / max(n1, 1), 1.0 / max(n2, 1))
    base = None
    for v in [int(s) for s in a.variants.split(",")]:
        os.environ["PINN_MFMA_VARIANT"] = str(v)
        outs = []
        for _ in range(a.reps):
            out = torch.empty(h.P + 3, device=dev)
            h.residual_fwd_bwd(*args, n, ham, None, (1 << 63) - 1, out)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy())
        o = outs[0]
        same_runs = all(np.array_equal(o, q, equal_nan=True) for q in outs[1:])
        nan = np.isnan(o)
        msg = "variant %2d: loss %.6e  nan entries %d of %d  run-to-run identical %s" % (v, o[h.P:].sum(), nan.sum(), o.size, same_runs)
        if base is None:
            base = o
        else:
            d = np.abs(o.astype(np.float64) - base)
            msg += "  vs first: bit-identical %s, max |diff| %.3e (rel to max|grad| %.2e)" % (
                np.array_equal(o, base), np.nanmax(d), np.nanmax(d) / np.abs(base[:h.P]).max())
        print(msg)
        if nan.any():
            off, sz = capi.tensor_layout(arch)
            for k, (o_, s_) in enumerate(zip(off, sz)):
                c = int(nan[o_:o_ + s_].sum())
                if c:
                    print("     tensor %d (offset %d, size %d): %d nan" % (k, o_, s_, c))
            print("     loss terms", o[h.P:])


if __name__ == "__main__":
    main()
