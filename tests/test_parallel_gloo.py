"""world_size-2 test of the data-parallel plan on CPU (gloo).

The per-rank compute is stood in for by the CPU oracle (no GPU here); what is under test is
the product's sharding logic (pinn_..._amd/parallel.py): rank offsets into the global Philox
stream, global 1/N and BC counts, and that SUM all-reduce of [grad | loss terms] over shards
equals the single-process result."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_local, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from pinn_for_quantum_wavefunction_surfaces_amd.parallel import DataParallelPlan
    plan = DataParallelPlan(n_local)
    assert (plan.rank, plan.world, plan.first, plan.n_global) == (rank, world, rank * n_local,
                                                                  world * n_local)
    w = np.load(os.path.join(ROOT, "tests", "golden", "g1_weights.npz"))["ionHsym"]
    x, y, z, R, mask, cnt = O.sample(12345, 3, plan.first, n_local)
    counts = plan.reduce_counts(torch.tensor(cnt, dtype=torch.int64))
    A, Hm = O.arch(), O.ham_A()
    lg = O.loss_grad(A, Hm, w, x, y, z, R, mask, invN=plan.invN, invN1=1.0 / int(counts[0]),
                     invN2=1.0 / int(counts[1]))
    s = lg["sums"]
    buf = torch.tensor(np.concatenate([lg["grad"], [s[0] * plan.invN, s[1] / int(counts[0]),
                                                    s[2] / int(counts[1])]]))
    plan.reduce_grad(buf)
    # the Trainer's one-message form: the next batch's local counts ride behind the gradient
    lg32 = np.concatenate([lg["grad"], [s[0] * plan.invN, s[1] / int(counts[0]),
                                        s[2] / int(counts[1])], [0, 0]]).astype(np.float32)
    buf1 = torch.tensor(lg32)
    nxt = O.sample(12345, 4, plan.first, n_local)[5]
    nxt_global = torch.zeros(2, dtype=torch.int64)
    plan.reduce_grad_and_counts(buf1, torch.tensor(nxt, dtype=torch.int64), nxt_global)
    tmax = plan.max_over_ranks(float(rank + 1), torch.device("cpu"))
    assert tmax == float(world)
    if rank == 0:
        np.save(os.path.join(out_dir, "dp.npy"), buf.numpy())
        np.save(os.path.join(out_dir, "counts.npy"), counts.numpy())
        np.save(os.path.join(out_dir, "dp1.npy"), buf1.numpy())
        np.save(os.path.join(out_dir, "counts_next.npy"), nxt_global.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_shards_equal_one(tmp_path):
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    world, n_local = 2, 1500
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, n_local, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "dp.npy")
    counts = np.load(tmp_path / "counts.npy")
    w = np.load(os.path.join(ROOT, "tests", "golden", "g1_weights.npz"))["ionHsym"]
    x, y, z, R, mask, cnt = O.sample(12345, 3, 0, world * n_local)
    assert tuple(counts) == cnt
    ref = O.loss_grad(O.arch(), O.ham_A(), w, x, y, z, R, mask)
    P = ref["grad"].size
    assert np.abs(got[:P] - ref["grad"]).max() / np.abs(ref["grad"]).max() < 1e-12
    assert abs(got[P:].sum() / ref["Ltot"] - 1) < 1e-12
    # one-message form: same sums (fp32 buffer), and the next batch's global set sizes exactly
    got1 = np.load(tmp_path / "dp1.npy")
    assert np.abs(got1[:P] - ref["grad"]).max() / np.abs(ref["grad"]).max() < 1e-6
    assert tuple(np.load(tmp_path / "counts_next.npy")) == O.sample(12345, 4, 0, world * n_local)[5]
