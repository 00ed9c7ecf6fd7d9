#!/usr/bin/env python3
"""diag_deep (test infrastructure: uses the oracle, hence under tests/) -- where does the gradient error of the 4x128 split-bf16 path on the reference-generated deep128
fixture come from?  Compares default kernel (k_trunkDx) and the pass-by-pass kernel (PINN_MFMA_VARIANT=3), with and
without the fixture's hand-placed points next to the nuclei, against the fixture / the fp64 oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("PINN_MFMA_VARIANT_LIVE", "1")   # variants are switched inside this process
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pinn_for_quantum_wavefunction_surfaces_amd as pkg  # noqa: E402
from pinn_for_quantum_wavefunction_surfaces_amd import capi  # noqa: E402
from oracle import oracle as O  # noqa: E402

DEV = "cuda:0"
rel = lambda a, b: np.max(np.abs(np.asarray(a, np.float64) - b)) / max(np.max(np.abs(b)), 1e-300)
dev32 = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=DEV)


def loss_grad(h, theta, x, y, z, R, mask):
    f = pkg.model.FORMULATION_A
    n = len(x)
    n1, n2 = int((mask & 1).sum()), int(((mask >> 1) & 1).sum())
    out = torch.empty(h.P + 3, device=DEV)
    ham = capi.Ham(f["a"], f["b"], f["c"], 1.0, 0.0, 0.0, 1.0 / n, 1.0 / n1 if n1 else 0.0, 1.0 / n2 if n2 else 0.0)
    h.residual_fwd_bwd(dev32(theta), dev32(x), dev32(y), dev32(z), dev32(R), torch.tensor(mask, device=DEV), n, ham,
                       None, (1 << 63) - 1, out)
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.float64)


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "g2_formA_deep128.npz"))
    A = O.arch(H=128, L=4, act=1)
    off, sz = O.tensor_layout(A)
    names = [k for l in range(4) for k in ("W%d" % l, "b%d" % l)] + ["wout", "bout", "WE1", "bE1", "WE2", "bE2", "WE3",
                                                                      "bE3", "WG1", "bG1", "WG2", "bG2"]
    for prec, pr in (("bf16x3", capi.PINN_PREC_BF16X3),):
        h = capi.Handle(capi.Arch(128, 4, 32, 10, 1, 2, 1.0, pr), 0, 4096)
        for label, sl in (("all 768 points", slice(0, None)), ("without the 12 hand-placed points", slice(12, None)),
                          ("only points 12..267 (one tile row)", slice(12, 268))):
            x, y, z, R, mask = [g[k][sl] for k in ("x", "y", "z", "R", "mask")]
            ref = O.loss_grad(A, O.ham_A(), g["theta"], x, y, z, R, mask)
            for var in ("0", "3"):
                os.environ["PINN_MFMA_VARIANT"] = var
                out = loss_grad(h, g["theta"], x, y, z, R, mask)
                te = {nm: np.max(np.abs(out[off[t]:off[t] + sz[t]] - ref["grad"][off[t]:off[t] + sz[t]]))
                      / max(np.max(np.abs(ref["grad"][off[t]:off[t] + sz[t]])), 1e-3 * np.max(np.abs(ref["grad"])))
                      for t, nm in enumerate(names)}
                print("%s %s variant %s: loss %.2e grad %.2e |grad|max %.3e  worst tensors: %s" % (
                    prec, label, var, abs(out[h.P:].sum() / ref["Ltot"] - 1), rel(out[:h.P], ref["grad"]),
                    np.abs(ref["grad"]).max(),
                    ", ".join("%s %.1e" % kv for kv in sorted(te.items(), key=lambda kv: -kv[1])[:5])))
            os.environ.pop("PINN_MFMA_VARIANT", None)
    # the same fixture through the oracle with theta rounded the way the split-bf16 kernels see it (hi + lo bf16 = 16
    # mantissa bits): how much of the error is the weights' representation alone?
    def split16(a):
        t = torch.tensor(a, dtype=torch.float32)
        hi = t.to(torch.bfloat16).to(torch.float32)
        lo = (t - hi).to(torch.bfloat16).to(torch.float32)
        return (hi + lo).double().numpy()
    th2 = g["theta"].copy()
    for t in (2, 4, 6):          # W1..W3 (the dense layers are the only rounded operands among the parameters)
        th2[off[t]:off[t] + sz[t]] = split16(g["theta"][off[t]:off[t] + sz[t]])
    ref = O.loss_grad(A, O.ham_A(), g["theta"], g["x"], g["y"], g["z"], g["R"], g["mask"])
    r2 = O.loss_grad(A, O.ham_A(), th2, g["x"], g["y"], g["z"], g["R"], g["mask"])
    print("oracle with W1..W3 rounded to 16 mantissa bits vs exact: loss %.2e grad %.2e" % (
        abs(r2["Ltot"] / ref["Ltot"] - 1), rel(r2["grad"], ref["grad"])))
    print("per-point |res| of the fixture: max %.3e, median %.3e; 12 hand-placed: %s" % (
        np.abs(g["res"]).max(), np.median(np.abs(g["res"])), np.array2string(np.abs(g["res"][:12]), precision=2)))


if __name__ == "__main__":
    main()
