#!/usr/bin/env python3
"""recipe_e2e -- the reference's two-stage recipe end to end on the fused path, from a fresh nn.Linear-style init:
main run (poc/main.py:36-43,919-922: 5 000 epochs x 100 000 points, Adam lr 8e-3, resampling every step during the
first 90 %) then fine-tune (poc/main.py:932-942: 2 000 epochs, lr 5e-4, trunk and gate frozen, R in [0.5, 4]), and
the E_net(R) curve of the result against Wind's exact table and against the curve the reference's authors published
(poc/energy_R_ion.pkl, tests/golden/g5_energy.npz).  The reference's CPU mt19937 sampling stream is not reproduced
(SURVEY G7), so this compares outcomes, not trajectories."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pinn_for_quantum_wavefunction_surfaces_amd as pkg  # noqa: E402


def bands(R, err):
    return "all R %.4f, R >= 0.5 %.4f, R >= 1.0 %.4f, mean %.4f" % (
        err.max(), err[R >= 0.5].max(), err[R >= 1.0].max(), err.mean())


def main():
    g5 = np.load(os.path.join(ROOT, "tests", "golden", "g5_energy.npz"))
    R, E_pub, E_exact = g5["R"], g5["E_net"], g5["E_exact"]
    print("published curve (poc/energy_R_ion.pkl E_net) vs exact: max |err| " + bands(R, np.abs(E_pub - E_exact)))
    print("published main run (poc/loss_ionH.pkl, opcode-parsed): Ltot at epochs 0: 9.60e-02, 10: 2.50e-03, 100: 3.31e-05, 500: "
          "1.05e-05, 1000: 4.57e-06, 2000: 5.12e-05, 3000: 1.65e-05, 4000: 9.86e-07, 4999: 7.42e-07; median of [0.8n, 0.9n) "
          "1.29e-06, of the last 5 % 4.90e-07, min 4.08e-07 at 4867")
    for seed in (int(a) for a in (sys.argv[1:] or ["1"])):
        torch.manual_seed(seed)
        params = pkg.set_params()
        params.update(saveModelPath="", lossPath="", loadModelPath="")
        t0 = time.perf_counter()
        model, hist = pkg.train(params, verbose=False, save=False, seed=12345 + seed)
        p2 = dict(params, epochs=2000, lr=5e-4, RxL=0.5)                  # poc/main.py:932-942
        model, hist2 = pkg.train(p2, freezeUnits=True, model=model, verbose=False, save=False, seed=777 + seed)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        L = hist["Ltot"][:, 0]
        n = len(L)
        print("   main run Ltot at epochs " + ", ".join("%d: %.2e" % (e, L[e]) for e in (0, 10, 100, 500, 1000, 2000, 3000, 4000, n - 1))
              + "; median of [0.8n, 0.9n) %.2e, of the last 5 %% %.2e, min %.2e at %d" % (
                  np.median(L[int(0.8 * n):int(0.9 * n)]), np.median(L[int(0.95 * n):]), L.min(), L.argmin()))
        Rt = torch.tensor(R, device="cuda", dtype=torch.float32)
        E = model.energy_R(Rt)[0].double().cpu().numpy()
        print("seed %d: %.2f s for both stages; best Ltot main %.3e, fine-tune %.3e; E_net vs exact: max |err| %s; "
              "vs the published curve: max |diff| R >= 1.0 %.4f" % (
                  seed, dt, hist["Ltot"].min(), hist2["Ltot"].min(), bands(R, np.abs(E - E_exact)),
                  np.abs(E - E_pub)[R >= 1.0].max()))


if __name__ == "__main__":
    main()
