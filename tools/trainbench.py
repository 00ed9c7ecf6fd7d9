import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import pinn_for_quantum_wavefunction_surfaces_amd as pkg
for n, ep in ((100000, 400), (1000000, 200)):
    params = pkg.set_params()
    params.update(epochs=ep, n_train=n, saveModelPath="", lossPath="", loadModelPath="")
    torch.manual_seed(0)
    t0 = time.perf_counter()
    model, hist = pkg.train(params, verbose=False, save=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # second run: steady state (handles built, library warm)
    t0 = time.perf_counter()
    model, hist = pkg.train(params, verbose=False, save=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("train(): n_train = %d, %d epochs: %.3f ms/epoch (%.3g points/s), Ltot %.3e -> %.3e" % (
        n, ep, dt / ep * 1e3, n * ep / dt, hist["Ltot"][0, 0], hist["Ltot"][-1, 0]))
