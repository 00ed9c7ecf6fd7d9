#!/usr/bin/env python3
"""callers_cpu_ref (test infrastructure) -- the reference formulation (oracle/torch_baseline.py: PyTorch CPU, fp64, nested
autograd Laplacian) evaluating psi and H psi on an N^3 grid, timed: the number quoted beside tools/callers_bench.py in
profiles/r03_callers.md.    python tests/diag/callers_cpu_ref.py [N = 60]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import torch_baseline as TB  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
th = torch.tensor(np.load(os.path.join(ROOT, "tests", "golden", "g1_weights.npz"))["fineTune"], dtype=torch.float64)
ax = torch.linspace(-18, 18, n, dtype=torch.float64)
xg, yg, zg = torch.meshgrid(ax, ax, ax, indexing="ij")
col = lambda t: t.reshape(-1, 1).clone().requires_grad_(True)
x, y, z = col(xg), col(yg), col(zg)
Rr = torch.full_like(x, 1.0)
arch = dict(H=16, L=2, HE=32, HG=10, act=0, n_pass=2, out_scale=1.0)
ham = dict(a=-0.5, b=-1.0, c=-1.0, P=1.0)
empty = torch.zeros(0, dtype=torch.long)
t0 = time.perf_counter()
TB.loss_terms(th, arch, ham, x, y, z, Rr, empty, empty)
dt = time.perf_counter() - t0
print("reference formulation (PyTorch CPU, fp64, nested autograd Laplacian, %d threads) psi + H psi on a %d^3 grid: "
      "%.2f s = %.2e points/s" % (torch.get_num_threads(), n, dt, n ** 3 / dt))
