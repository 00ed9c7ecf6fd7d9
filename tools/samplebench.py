#!/usr/bin/env python3
"""samplebench -- time pinn_sample alone (torch events around K back-to-back calls) at a few sizes:
    [PINN_HIP_LIB=build_ab/libpinn_hip_X.so] python tools/samplebench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import pinn_for_quantum_wavefunction_surfaces_amd as pkg  # noqa: E402
from pinn_for_quantum_wavefunction_surfaces_amd import capi  # noqa: E402


def main():
    dev = "cuda:0"
    h = capi.Handle(capi.Arch(16, 2, 32, 10, 0, 2, 1.0, 0), 0, 1 << 20)
    for n in (1000000, 100000, 4000):
        x, y, z, R = [torch.empty(n, device=dev) for _ in range(4)]
        mask = torch.empty(n, device=dev, dtype=torch.uint8)
        cnt = torch.zeros(2, device=dev, dtype=torch.int64)
        for _ in range(50):
            h.sample(1, 0, 0, n, 18.0, 0.2, 4.0, 0.005, 17.5, x, y, z, R, mask, cnt)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(200):
                h.sample(1, k, 0, n, 18.0, 0.2, 4.0, 0.005, 17.5, x, y, z, R, mask, cnt)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
        print("pinn_sample n = %7d: %.2f us per call (counts %s)" % (n, best, cnt.tolist()))


if __name__ == "__main__":
    main()
