#!/usr/bin/env python3
"""callers_bench -- the callers next to the hot path on the GPU path, timed:
energy_from_psi at n_test = 80 (the reference's setting) and 250 (the size poc/main.py:1461 says needs "more than 32GB"
with the autograd Laplacian), calculate_E_R (39 R values), dEdR_int, force_curves, psi_cut, with the published
checkpoint's weights (tests/golden/g1_weights.npz).  (The CPU restatement of the reference beside it:
tests/diag/callers_cpu_ref.py -- anything that touches oracle/ lives under tests/.)
    python tools/callers_bench.py"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pinn_for_quantum_wavefunction_surfaces_amd as pkg  # noqa: E402
from pinn_for_quantum_wavefunction_surfaces_amd import analysis  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out


def main():
    argparse.ArgumentParser().parse_args()
    dev = "cuda:0"
    params = pkg.set_params()
    model = pkg.NN_ion(params).to(dev)
    model.load_flat_params(torch.tensor(np.load(os.path.join(ROOT, "tests", "golden", "g1_weights.npz"))["fineTune"]))
    for n in (80, 250):
        p = dict(params, n_test=n)
        torch.cuda.reset_peak_memory_stats()
        dt, (Eint, Enet) = timed(lambda: analysis.energy_from_psi(1.0, p, model))
        print("energy_from_psi(R = 1.0), n_test = %3d (%.2e grid points): %.1f ms, E_int = %.6f, E_net = %.6f, peak device "
              "memory %.2f GB" % (n, n ** 3, dt * 1e3, Eint, Enet, torch.cuda.max_memory_allocated() / 2 ** 30))
    p = dict(params, n_test=80)
    dt, tab = timed(lambda: analysis.calculate_E_R(p, model), reps=1)
    print("calculate_E_R (%d values of R, n_test = 80, LCAO curve included): %.2f s; max |E_int - E_net| for R >= 1: %.4f" % (
        len(tab["R"]), dt, float(np.abs(tab["E_int"] - tab["E_net"])[tab["R"] >= 1.0].max())))
    dt, v = timed(lambda: analysis.dEdR_int(1.0, p, model))
    print("dEdR_int(R = 1.0), n_test = 80: %.1f ms, %.6f" % (dt * 1e3, v))
    R = torch.linspace(0.2, 4.0, 191, dtype=torch.float64, device=dev)
    dt, (F, F2) = timed(lambda: analysis.force_curves(model, R))
    print("force_curves on 191 values of R (E twice differentiable w.r.t. R through pinn_energy_R): %.2f ms, F(R=2) = %.6f" % (
        dt * 1e3, float(F[90])))
    dt, _ = timed(lambda: analysis.psi_cut(model, 2.0))
    print("psi_cut (1000 points on the x axis): %.2f ms" % (dt * 1e3))


if __name__ == "__main__":
    main()
