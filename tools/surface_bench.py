#!/usr/bin/env python3
"""surface_bench -- the reference's own training-loop statements (poc/main.py:389-404: sampling, radial, torch.where,
LossFunctions, Ltot.backward(), optimizer.step() with torch.optim.Adam) run against the drop-in module surface, per
step, next to the fused Trainer at the same size.  What a user pays who keeps their loop and only swaps the import."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import pinn_for_quantum_wavefunction_surfaces_amd as pkg  # noqa: E402


def main():
    dev = "cuda"
    params = pkg.set_params()
    for n in (100000, 1000000):
        model = pkg.NN_ion(params).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=8e-3)
        def one(tt):
            x, y, z, R = pkg.sampling(params, n, linearSampling=False, device=dev, step=tt, model=model)
            r1, r2 = pkg.radial(x, y, z, R, params)
            b1 = torch.where(r1 >= params["BCcutoff"])
            b2 = torch.where(r2 >= params["BCcutoff"])
            Ltot, Lpde, Lbc, E = model.LossFunctions(x, y, z, R, params, b1, b2)
            opt.zero_grad()
            Ltot.backward()
            opt.step()
            return Ltot
        for tt in range(5):
            one(tt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 30
        for tt in range(K):
            L = one(5 + tt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        model2 = pkg.NN_ion(params).to(dev)
        tr = pkg.Trainer(model2, params, n_points=n, lr=8e-3, device=dev)
        for _ in range(20):
            tr.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            tr.step()
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / 200
        print("n = %7d: module surface + torch.optim.Adam %.3f ms/step (%.3g points/s), fused Trainer %.3f ms/step (%.3g points/s)"
              % (n, dt * 1e3, n / dt, dt2 * 1e3, n / dt2))


if __name__ == "__main__":
    main()
