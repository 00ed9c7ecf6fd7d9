"""Stress of the sampler's cross-block count hand-off: many calls with varying batch sizes, the two set sizes against the
sums of the mask bits computed on the device.  python tests/diag/diag_sample_counts.py [calls]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pinn_for_quantum_wavefunction_surfaces_amd as pkg  # noqa: E402
from pinn_for_quantum_wavefunction_surfaces_amd import capi  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
nmax = 1 << 20
h = capi.Handle(capi.Arch(16, 2, 32, 10, 0, 2, 1.0, capi.PINN_PREC_F32), 0, nmax)
x, y, z, R = [torch.empty(nmax, device=dev) for _ in range(4)]
mask = torch.empty(nmax, device=dev, dtype=torch.uint8)
counts = torch.zeros(2, device=dev, dtype=torch.int64)
cf = torch.zeros(2, device=dev)
g = torch.Generator().manual_seed(1)
bad = 0
for it in range(calls):
    n = int(torch.randint(1, nmax + 1, (1,), generator=g)) if it % 3 else nmax
    # a large BC cutoff keeps the sets non-trivial: bc_cutoff 12 of a box of 18
    h.sample(12345, it, 0, n, 18.0, 0.2, 4.0, 0.1, 12.0, x, y, z, R, mask, counts, counts_f32=cf)
    ref = torch.stack([(mask[:n] & 1).sum(), ((mask[:n] >> 1) & 1).sum()]).to(torch.int64)
    if not torch.equal(counts, ref) or not torch.equal(cf.to(torch.int64), ref):
        bad += 1
        print("call %d n %d: counts %s float %s expected %s" % (it, n, counts.tolist(), cf.tolist(), ref.tolist()))
print("%d calls, %d mismatches" % (calls, bad))
sys.exit(1 if bad else 0)
