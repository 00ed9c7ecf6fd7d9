#!/usr/bin/env python3
"""kbench -- A/B timing of trunk-kernel variants in ONE process, interleaved rounds (HIP events inside the
library, pinn_profile_*): `python tools/kbench.py --arch wide64 --variants 0,1,2 [--points 1000000]`.
A variant is an alternative instantiation selected by PINN_MFMA_VARIANT (csrc/pinn_trunk_mfma.hip picks the kernel).
Caveat (round 4): with the variants alternating every `--steps` steps the chip's clock control sees a mix, and a variant that
stalls less is flattered (k_trunk2's swizzle: -4.4 % here, -2.4 % as separate steady processes).  Quote gains from
tools/ab_steady.sh; use this tool to rank variants and to check that they agree bit for bit (theta checksum)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PINN_MFMA_VARIANT_LIVE", "1")   # variants are switched inside this process
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="wide64")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--points", type=int, default=1000000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    sys.argv = ["bench.py"]
    spec.loader.exec_module(bench)
    import pinn_for_quantum_wavefunction_surfaces_amd as pkg
    a = bench.ARCHS[args.arch]
    dev = torch.device("cuda", 0)
    params = pkg.set_params()
    variants = [int(v) for v in args.variants.split(",")]
    trainers = {}
    for v in variants:
        os.environ["PINN_MFMA_VARIANT"] = str(v)
        torch.manual_seed(12345)
        m = pkg.NN_ion(params, a["H"], a["HE"], a["HG"], n_hidden=a["L"], act=a["act_name"],
                       formulation=a.get("formulation", "A"), precision=a["precision"]).to(dev)
        tr = pkg.Trainer(m, params, n_points=args.points, lr=8e-3, seed=12345, device=dev)
        tr.h.profile_enable(args.steps)
        for _ in range(30):
            tr.step()
        trainers[v] = tr
    torch.cuda.synchronize()
    res = {v: [] for v in variants}
    for r in range(args.rounds):
        for v in variants:
            os.environ["PINN_MFMA_VARIANT"] = str(v)
            tr = trainers[v]
            for _ in range(args.steps):
                tr.step()
            torch.cuda.synchronize()
            res[v].append(tr.h.profile_read(args.steps))
    for v in variants:
        # theta checksum: variants that only re-order or re-place work must agree bit for bit (same seed, same step count)
        print("arch %s variant %d: trunk kernel ms median %.4f min %.4f  (rounds: %s)  loss %.4e  theta bits %d" % (
            args.arch, v, float(np.median(res[v])), min(res[v]), " ".join("%.4f" % t for t in res[v]),
            trainers[v].losses()[0], int(trainers[v].theta.view(torch.int32).to(torch.int64).sum())))


if __name__ == "__main__":
    main()
