#!/usr/bin/env python3
"""callers_bench -- the callers next to the hot path on the GPU path, timed:
energy_from_psi at n_test = 80 (the reference's setting) and 250 (the size poc/main.py:1461 says needs "more than 32GB"
with the autograd Laplacian), calculate_E_R (39 R values), dEdR_int, force_curves, psi_cut, with the published
checkpoint's weights (tests/golden/g1_weights.npz).  `--cpu-ref N` also times the oracle's torch restatement of the
reference formulation (nested autograd Laplacian, fp64, CPU) on an N^3 grid for comparison.
    python tools/callers_bench.py [--cpu-ref 40]"""
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu-ref", type=int, default=0)
    args = ap.parse_args()
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
    if args.cpu_ref:
        from oracle import torch_baseline as TB
        n = args.cpu_ref
        th = torch.tensor(np.load(os.path.join(ROOT, "tests", "golden", "g1_weights.npz"))["fineTune"], dtype=torch.float64)
        ax = torch.linspace(-18, 18, n, dtype=torch.float64)
        xg, yg, zg = torch.meshgrid(ax, ax, ax, indexing="ij")
        col = lambda t: t.reshape(-1, 1).clone().requires_grad_(True)
        x, y, z = col(xg), col(yg), col(zg)
        Rr = torch.full_like(x, 1.0)
        arch = dict(H=16, L=2, HE=32, HG=10, act=0, n_pass=2, out_scale=1.0)
        ham = dict(a=-0.5, b=-1.0, c=-1.0, P=1.0)
        t0 = time.perf_counter()
        empty = torch.zeros(0, dtype=torch.long)
        TB.loss_terms(th, arch, ham, x, y, z, Rr, empty, empty)
        dt = time.perf_counter() - t0
        print("reference formulation (PyTorch CPU, fp64, nested autograd Laplacian, %d threads) psi + H psi on a %d^3 grid: "
              "%.2f s = %.2e points/s" % (torch.get_num_threads(), n, dt, n ** 3 / dt))


if __name__ == "__main__":
    main()
