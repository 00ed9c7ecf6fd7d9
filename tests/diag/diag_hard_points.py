#!/usr/bin/env python3
"""diag_hard_points (GPU box; numbers only) -- forward errors of the 4x128 split-bf16 path at the deep128 fixture's 12
hand-placed points next to the nuclei, where the batch's gradient is conditioned worst: per point |res| and the error of
res, lap, psi of (a) the HIP forward, (b) the operand-rounding emulation of tests/diag/diag_operand_bits.py (fp64 elementwise,
2-piece GEMM operands), (c) the reference run in fp32 torch (fixture: psi, lap).
    python tests/diag/diag_hard_points.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import pinn_for_quantum_wavefunction_surfaces_amd as pkg  # noqa: E402
from pinn_for_quantum_wavefunction_surfaces_amd import capi  # noqa: E402
import diag_operand_bits as D  # noqa: E402


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "g2_formA_deep128.npz"))
    f = pkg.model.FORMULATION_A
    dev = "cuda:0"
    t32 = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    n = len(g["x"])
    out = {}
    for prec, pr in (("bf16x3", capi.PINN_PREC_BF16X3),):
        h = capi.Handle(capi.Arch(128, 4, 32, 10, 1, 2, 1.0, pr), 0, 4096)
        o = [torch.empty(n, device=dev) for _ in range(4)]
        h.forward(t32(g["theta"]), t32(g["x"]), t32(g["y"]), t32(g["z"]), t32(g["R"]), n,
                  capi.Ham(f["a"], f["b"], f["c"], 1.0, 0.0, 0.0, 0.0, 0.0, 0.0), *o)
        torch.cuda.synchronize()
        out[prec] = dict(zip(("psi", "E", "lap", "res"), [t.cpu().numpy().astype(np.float64) for t in o]))
    _, _, emu = D.loss_grad(g, (2,) * 6, want_fields=True)
    _, _, ex = D.loss_grad(g, (D.INF,) * 6, want_fields=True)
    print("max|res| %.3f  max|lap| %.3f  max|psi| %.3f" % (np.abs(g["res"]).max(), np.abs(g["lap"]).max(), np.abs(g["psi"]).max()))
    print("%3s %9s %9s | %-29s | %-29s | %-19s" % ("pt", "res", "lap", "HIP split-bf16: dres dlap dpsi", "emulation 2-piece: dres dlap dpsi", "fp32 torch: dlap dpsi"))
    for i in list(range(12)) + [100, 300]:
        k = out["bf16x3"]
        print("%3d %9.3e %9.3e | %9.2e %9.2e %9.2e | %9.2e %9.2e %9.2e | %9.2e %9.2e" % (
            i, g["res"][i], g["lap"][i], k["res"][i] - g["res"][i], k["lap"][i] - g["lap"][i], k["psi"][i] - g["psi"][i],
            emu["res"][i] - ex["res"][i], emu["lap"][i] - ex["lap"][i], emu["psi"][i] - ex["psi"][i],
            g["fp32_lap"][i] - g["lap"][i], g["fp32_psi"][i] - g["psi"][i]))
    for name, d in (("HIP", out["bf16x3"]["res"] - g["res"]), ("emulation", emu["res"] - ex["res"])):
        print("%s: sum over points of res * dres (what the PDE loss and its gradient see): %.3e   max|dres| %.2e" % (
            name, float(np.sum(g["res"] * d)), np.abs(d).max()))


if __name__ == "__main__":
    main()
