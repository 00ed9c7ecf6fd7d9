#!/usr/bin/env python3
"""bench.py -- collocation points/s through one full training step on N MI355X.

A step = sample (Philox, on device) -> fused forward + Laplacian + residual + loss +
backward -> SUM all-reduce of [grad | loss terms] when N > 1 -> Adam.  Inputs are
generated on the device, so they are resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--points 1000000]

For N > 1 there is one process per GPU; rank 0 prints ONE JSON line.  Started under a launcher
(WORLD_SIZE set: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) this
process is one rank.  Started bare (`python bench.py --gpus N`, WORLD_SIZE unset) it becomes the
launcher itself: before torch is imported or any GPU call is made it starts that same
torch.distributed.run command as a CHILD process, relays the child's output and exits with its status
(a process that has touched the GPU is never re-exec'd).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it: be the launcher."""
    if "WORLD_SIZE" in os.environ:
        return
    n = 1
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


if __name__ == "__main__":
    _self_launch()

import numpy as np  # noqa: E402
import torch  # noqa: E402


def trunk_flops_per_point(H, L, n_pass, S=5, fine_tune=False):
    """fwd+bwd MACs per pass 4SH + 3S(L-1)H^2 + 6H, 2 flop per MAC; frozen trunk
    (fine-tune): the forward sweep only, S(2H + (L-1)H^2) + 2H MACs per pass."""
    if fine_tune:
        return 2.0 * n_pass * (S * (2 * H + (L - 1) * H * H) + 2 * H)
    return 2.0 * n_pass * (4 * S * H + 3 * S * (L - 1) * H * H + 6 * H)


def total_flops_per_point(H, L, HE, HG, n_pass):
    return trunk_flops_per_point(H, L, n_pass) + 2.0 * (3 * (HE * HE + 2 * HE) + 6 * HG)


# workloads.  "ref16": the reference network (poc/main.py:225-245) on the fp32 parity path with
# the published ionHsym weights.  "wide64": the 2x64 tanh trunk named in BASELINE.json's configs,
# NN_ion formulation (symmetric double pass), bf16 MFMA trunk, seeded nn.Linear-style init.
ARCHS = {
    "ref16": dict(H=16, L=2, HE=32, HG=10, act=0, n_pass=2, out_scale=1.0, precision="f32",
                  act_name="sigmoid", peak=157.3, dtype="f32",
                  label="reference arch 2x16 trunk + 2x32 E-net + 1x10 gate, sigmoid, fp32-grade: fp32 "
                        "elementwise algebra, 16x16 / 32x32 layers as split-bf16 MFMA (3 pieces forward, 2 "
                        "reverse, fp32 accumulate), ionHsym weights"),
    "wide64": dict(H=64, L=2, HE=32, HG=10, act=1, n_pass=2, out_scale=1.0, precision="bf16",
                   act_name="tanh", peak=2500.0, dtype="bf16",
                   label="2x64 tanh trunk + 2x32 E-net + 1x10 gate, bf16 MFMA trunk with fp32 "
                         "accumulate, seeded random init"),
}
# BASELINE config 4 (fine-tune sweep, poc/main.py:384-387 freezeUnits): warm start from the published
# weights, trunk and gate frozen (forward-only trunk sweep, MODE_HEAD), R ~ U(0.5, 4), lr 5e-4
ARCHS["ref16ft"] = dict(ARCHS["ref16"], fine_tune=True,
                        label="reference arch, fine-tune mode: trunk + gate frozen, only the E-net trains "
                              "(forward-only trunk sweep), R in [0.5, 4], ionHsym weights, fp32")
# train.py's own formulation (single pass over (f1, f2), output 2h, coefficients (1, 1, 1)) at 2x64 tanh
ARCHS["wide64B"] = dict(ARCHS["wide64"], n_pass=1, out_scale=2.0, formulation="B",
                        label="train.py's formulation (single pass, 2h): 2x64 tanh trunk + 2x32 E-net + 1x10 gate, "
                              "bf16 MFMA trunk with fp32 accumulate, seeded random init")
ARCHS["wide128"] = dict(H=128, L=2, HE=32, HG=10, act=1, n_pass=2, out_scale=1.0, precision="bf16",
                        act_name="tanh", peak=2500.0, dtype="bf16",
                        label="2x128 tanh trunk + 2x32 E-net + 1x10 gate, bf16 MFMA trunk with fp32 "
                              "accumulate, seeded random init")
ARCHS["wide4x64"] = dict(H=64, L=4, HE=32, HG=10, act=1, n_pass=2, out_scale=1.0, precision="bf16",
                         act_name="tanh", peak=2500.0, dtype="bf16",
                         label="4x64 tanh trunk + 2x32 E-net + 1x10 gate, bf16 MFMA trunk with fp32 "
                               "accumulate, seeded random init")
ARCHS["wide4x128"] = dict(H=128, L=4, HE=32, HG=10, act=1, n_pass=2, out_scale=1.0, precision="bf16",
                          act_name="tanh", peak=2500.0, dtype="bf16",
                          label="4x128 tanh trunk + 2x32 E-net + 1x10 gate, bf16 MFMA trunk with fp32 "
                                "accumulate, seeded random init")
ARCHS["wide64x3"] = dict(ARCHS["wide64"], precision="bf16x3", dtype="bf16x3",
                         label="2x64 tanh trunk + 2x32 E-net + 1x10 gate, split-bf16 MFMA trunk (hi+lo "
                               "operands, 3 MFMAs per product, fp32 accumulate), seeded random init")
ARCHS["wide128x3"] = dict(ARCHS["wide128"], precision="bf16x3", dtype="bf16x3",
                          label="2x128 tanh trunk + 2x32 E-net + 1x10 gate, split-bf16 MFMA trunk (hi+lo "
                                "operands, 3 MFMAs per product, fp32 accumulate), seeded random init")
ARCHS["mid32"] = dict(H=32, L=2, HE=16, HG=8, act=0, n_pass=2, out_scale=1.0, precision="f32",
                      act_name="sigmoid", peak=157.3, dtype="f32",
                      label="2x32 trunk + 2x16 E-net + 1x8 gate, sigmoid, fp32 (the 32-wide instantiation of "
                            "the packed-VALU trunk: functional, not tuned), seeded random init")
ARCHS["wide4x64x3"] = dict(ARCHS["wide4x64"], precision="bf16x3", dtype="bf16x3",
                           label="4x64 tanh trunk + 2x32 E-net + 1x10 gate, split-bf16 MFMA trunk (hi+lo "
                                 "operands, 3 MFMAs per product, fp32 accumulate), seeded random init")
# BASELINE config 1: "100k points/step, 4x128 MLP, fp32": the split-bf16 4x128 kernel (16 operand bits, fp32 accumulation
# and elementwise; psi, lap, E at <= 1e-5; gradient vs genuine fp32: DESIGN.md section 6) at 100 000 points per step
ARCHS["wide4x128x3"] = dict(ARCHS["wide4x128"], precision="bf16x3", dtype="bf16x3", points=100000,
                            label="BASELINE config 1: 4x128 tanh trunk + 2x32 E-net + 1x10 gate, split-bf16 MFMA trunk "
                                  "(hi+lo operands = 16 mantissa bits, 3 MFMAs per product, fp32 accumulate and "
                                  "elementwise; psi/lap/E <= 1e-5), 100 000 points/step, seeded random init")
# HBM bytes per launch of each workload's trunk kernel: read from profiles/traffic.json, which tools/profile_round.sh
# (tools/summarize_prof.py) writes from the rocprofv3 PMC passes of the same bench command -- FETCH_SIZE x 2 (the gfx950
# correction for wide coalesced reads, MI355X_MICROARCH.md) + WRITE_SIZE, mean per launch, keyed by kernel.  No file, or
# no entry for a kernel: `traffic` is null.
TRAFFIC_KERNEL_KEY = {"wide64": "trunk2_64", "wide64B": "trunk2s_64", "wide64x3": "trunk2x", "wide128": "trunk2_128",
                      "wide4x64": "trunkD_64", "wide4x128": "trunkD_128", "wide4x128x3": "trunkDx",
                      "ref16": "trunk16_bwd", "ref16ft": "trunk16_head"}


def load_traffic():
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
    except (OSError, ValueError):
        return {}, "profiles/traffic.json not found"
    return ({a: t["bytes_per_launch"].get(k) for a, k in TRAFFIC_KERNEL_KEY.items()},
            "%s (%s)" % (t.get("formula", "FETCH_SIZE*2+WRITE_SIZE"), t.get("source", "profiles/traffic.json")))


TRAFFIC, TRAFFIC_SOURCE = load_traffic()
DEFAULT_ARCH = "wide64"


def _mfma_kernel_name(arch):
    """the trunk kernel launch_trunk_mfma dispatches this workload to (csrc/pinn_trunk_mfma.hip)"""
    if arch.get("n_pass", 2) == 1 and arch["L"] == 2 and arch.get("precision") == "bf16" and not arch.get("fine_tune", False):
        return "k_trunk2s"
    if arch.get("n_pass", 2) == 2 and not arch.get("fine_tune", False):
        if arch["L"] == 2:
            if arch.get("precision") == "bf16x3":
                return "k_trunk2x" if arch["H"] == 64 else "k_trunk_mfma2"
            return "k_trunk2"
        if arch["L"] == 4 and arch.get("precision") != "bf16x3":
            return "k_trunkD"
        if arch["L"] == 4 and arch["H"] == 128:
            return "k_trunkDx"
    return "k_trunk_mfma2"
SECONDARY = {"wide64": ["wide64x3", "wide4x128x3", "ref16", "ref16ft", "wide128", "wide4x64", "wide4x128", "wide64B"], "ref16": ["wide64"], "wide64x3": ["wide64"],
             "wide128": ["wide128x3"], "wide128x3": ["wide128"]}
# tolerances the GPU parity tests hold each path to, against the fp64 oracle / reference vectors
# (tests/test_gpu_parity.py).  What THIS run observed is measured in the cpu_baseline leg ("parity_observed").
PARITY_BAR = {
    "f32": "psi, E, lap 1e-5 rel; residual 2e-5; loss 1e-4; gradient 1e-4",
    "bf16": "psi 1e-2, lap 1e-2, residual 2e-2, loss 3e-2, gradient 3e-2 rel (bf16 operands cannot meet 1e-5)",
    "bf16x3": "psi, lap 1e-5 rel; residual 5e-5; loss, gradient 2e-4 on sampled batches (split-bf16: 16 operand bits, fp32 "
              "accumulate; on a batch dominated by points 0.01-0.2 a.u. from a nucleus the gradient is within 16x of what "
              "the reference gives in fp32 torch, observed 9.5x: tests/test_gpu_parity.py, DESIGN.md section 6)",
}
REF_ARCH = dict(H=16, L=2, HE=32, HG=10, act=0, n_pass=2, out_scale=1.0)
HAM_A = dict(a=-0.5, b=-1.0, c=-1.0, P=1.0)


def _median_step_time(step, batch, warm=2, reps=5, budget_s=8.0):
    for _ in range(warm):
        step(*batch())
    ts, t0 = [], time.time()
    while len(ts) < reps and (len(ts) < 2 or time.time() - t0 < budget_s):
        b = batch()
        t1 = time.time()
        step(*b)
        ts.append(time.time() - t1)
    return float(np.median(ts)), len(ts)


def cpu_baseline(theta_np, arch, gpu_check=None):
    """The reference formulation (nested autograd.grad Laplacian, loss, backward, torch.optim.Adam;
    oracle/torch_baseline.py) in PyTorch on the host's CPU cores, on a BOUNDED sample:
      * the reference's own architecture (2x16 sigmoid trunk, the only one it has) at its own batch size,
        n = 100 000 (poc/main.py:39), median of 5 steps after 2 warm-ups: fp64 (the reference's dtype) on all
        host threads torch uses by default and on 16 threads (`value` = the faster), fp32 on 16 threads;
      * `primary_arch`: the benchmark's primary architecture (nested autograd through its wider layers),
        fp64, 20 000 points.
    The same leg also CHECKS the GPU path (gpu_check): a few thousand of the benchmark's own points and
    its own weights through the fp64 restatement here, against what the HIP kernels returned for them."""
    from oracle import torch_baseline as TB

    def make_batch(n, dtype):
        def batch():
            x, y, z = [(torch.rand(n, dtype=dtype) * 36 - 18) for _ in range(3)]
            return x, y, z, torch.rand(n, dtype=dtype) * 3.8 + 0.2
        return batch

    torch.manual_seed(12345)
    nthreads_all = torch.get_num_threads()
    ncpu = os.cpu_count() or 1
    w = np.load(os.path.join(ROOT, "tests", "golden", "g1_weights.npz"))["ionHsym"]
    legs, n = {}, 100000
    for tag, dtype, nt in (("fp64", torch.float64, nthreads_all), ("fp64_16thr", torch.float64, min(16, ncpu)),
                           ("fp32_16thr", torch.float32, min(16, ncpu))):
        torch.set_num_threads(nt)
        _, step = TB.make_step(torch.tensor(w), REF_ARCH, HAM_A, lr=8e-3, dtype=dtype)
        dt, k = _median_step_time(step, make_batch(n, dtype), warm=1 if tag == "fp64" else 2,
                                  reps=3 if tag == "fp64" else 5)
        legs[tag] = {"value": n / dt, "unit": "points/s", "s_per_step": dt, "steps": k, "threads": nt}
        print("[cpu_baseline] reference arch n=%d %s %d threads: %.3f s/step" % (n, tag, nt, dt),
              file=sys.stderr, flush=True)
    torch.set_num_threads(legs["fp64_16thr"]["threads"])
    # SURVEY 8(d)'s other two sizes, fp64, 16 threads: BASELINE config 0's batch (4 000 points; the reference architecture
    # and config 0's own "2x64 tanh MLP" under train.py's formulation, train.py:41-57) and config 2's batch (1 000 000
    # points as ten chunks of 100 000 accumulated into one gradient, one Adam step: ~40 GB in one autograd graph)
    n4k = 4000
    _, step = TB.make_step(torch.tensor(w), REF_ARCH, HAM_A, lr=8e-3, dtype=torch.float64)
    dt, k = _median_step_time(step, make_batch(n4k, torch.float64), warm=2, reps=7, budget_s=5.0)
    legs["n4000"] = {"value": n4k / dt, "unit": "points/s", "s_per_step": dt, "steps": k, "threads": torch.get_num_threads(),
                     "sample": "reference architecture, 4 000 points/step, fp64 (BASELINE config 0's batch)"}
    torch.manual_seed(12345)
    th64 = torch.cat([(torch.rand(o * i + o, dtype=torch.float64) * 2 - 1) / max(i, 1) ** 0.5 for o, i in
                      ((64, 2), (64, 64), (1, 64), (32, 1), (32, 32), (1, 32), (10, 1), (1, 10))])
    a64B = dict(H=64, L=2, HE=32, HG=10, act=1, n_pass=1, out_scale=2.0)
    _, step = TB.make_step(th64, a64B, dict(a=1.0, b=1.0, c=1.0, P=1.0), lr=8e-3, dtype=torch.float64)
    dt, k = _median_step_time(step, make_batch(n4k, torch.float64), warm=2, reps=7, budget_s=5.0)
    legs["n4000_2x64tanh_trainpy"] = {"value": n4k / dt, "unit": "points/s", "s_per_step": dt, "steps": k,
                                      "threads": torch.get_num_threads(),
                                      "sample": "BASELINE config 0: 2x64 tanh trunk, train.py's formulation (single pass, 2h, "
                                                "coefficients (1,1,1)), 4 000 points/step, fp64"}
    n1m = 1000000
    _, step = TB.make_step_chunked(torch.tensor(w), REF_ARCH, HAM_A, lr=8e-3, dtype=torch.float64, chunk=100000)
    dt, k = _median_step_time(step, make_batch(n1m, torch.float64), warm=0, reps=2, budget_s=6.0)
    legs["n1000000_chunked"] = {"value": n1m / dt, "unit": "points/s", "s_per_step": dt, "steps": k,
                                "threads": torch.get_num_threads(),
                                "sample": "reference architecture, 1 000 000 points/step as 10 chunks of 100 000 accumulated "
                                          "into one gradient, one Adam step, fp64 (BASELINE config 2's batch)"}
    for tag in ("n4000", "n4000_2x64tanh_trainpy", "n1000000_chunked"):
        print("[cpu_baseline] %s: %.3f s/step" % (tag, legs[tag]["s_per_step"]), file=sys.stderr, flush=True)
    # `value` = the FASTER of the two fp64 runs: the ~700 small ops per step of this formulation do not
    # scale to a whole socket (all cores can be several times slower than 16 threads), and a baseline
    # slowed by oversubscription would flatter the GPU number
    best = max(("fp64", "fp64_16thr"), key=lambda k: legs[k]["value"])
    out = {"value": legs[best]["value"], "unit": "points/s", "cores": legs[best]["threads"], "kind": "port",
           "sample": "reference architecture (2x16 sigmoid trunk + 2x32 E-net + 1x10 gate, ionHsym weights), "
                     "n = 100 000 points/step, fp64, median of %d steps after 2 warm-ups, PyTorch-CPU restatement "
                     "of the reference formulation (nested autograd.grad Laplacian + backward + Adam); run on %d "
                     "threads (torch's default = all physical cores of the %d host CPUs) and on 16 threads, the "
                     "faster of the two (%d threads) is `value`"
                     % (legs[best]["steps"], nthreads_all, ncpu, legs[best]["threads"]),
           "fp64_all_cores": legs["fp64"], "fp64_16_threads": legs["fp64_16thr"], "fp32_16_threads": legs["fp32_16thr"],
           "n4000": legs["n4000"], "n4000_2x64tanh_trainpy": legs["n4000_2x64tanh_trainpy"],
           "n1000000_chunked": legs["n1000000_chunked"]}
    if arch["H"] != 16:
        n2 = 20000
        a2 = {k: arch[k] for k in ("H", "L", "HE", "HG", "act", "n_pass", "out_scale")}
        _, step = TB.make_step(torch.tensor(theta_np), a2, HAM_A, lr=8e-3, dtype=torch.float64)
        dt, k = _median_step_time(step, make_batch(n2, torch.float64), warm=1, reps=3, budget_s=10.0)
        out["primary_arch"] = {"value": n2 / dt, "unit": "points/s", "s_per_step": dt, "steps": k,
                               "threads": torch.get_num_threads(), "sample": "%dx%d %s trunk, %d points, fp64"
                               % (arch["L"], arch["H"], arch["act_name"], n2)}
    if gpu_check is not None:
        out["parity_observed"] = gpu_check(TB)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=1000000, help="points per GPU per step")
    ap.add_argument("--arch", choices=sorted(ARCHS), default=DEFAULT_ARCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="primary workload only (profiling)")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the untimed clock pre-warm")
    ap.add_argument("--one-shot", action="store_true",
                    help="N > 1: exchange the gradient message with the library's one-shot peer-memory "
                         "all-reduce instead of RCCL (default: RCCL)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one captured HIP graph (helps below ~10^5 points)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:                       # before any GPU or process-group call
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (start it bare, or with --nproc-per-node %d)"
                 % (args.gpus, world, args.gpus))
    use_dist = world > 1 or os.environ.get("PINN_FORCE_DIST") == "1"   # the latter: RCCL sanity at N=1
    # rehearsal of the N > 1 code path on a one-GPU box: all ranks on cuda:0, exchange over gloo
    # (RCCL refuses two ranks on one device).  Never used by the driver's runs.
    rehearsal = os.environ.get("PINN_REHEARSE_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():      # (counting devices does not initialise the GPU)
        sys.exit("bench.py: rank %d needs GPU %d but this node shows %d device(s); --gpus must not exceed the GPUs of "
                 "ONE node" % (rank, local_rank, torch.cuda.device_count()))
    # stdout carries ONE line, the JSON record.  Library banners (RCCL prints its version to stdout when the
    # first communicator is built, gloo its connection notes) would precede it, so in distributed runs file
    # descriptor 1 points at stderr until the record is written to the saved descriptor.
    json_fd = None
    if use_dist:
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import pinn_for_quantum_wavefunction_surfaces_amd as pkg
    params = pkg.set_params()
    arch = ARCHS[args.arch]

    def build_model(name):
        a = ARCHS[name]
        torch.manual_seed(12345)
        m = pkg.NN_ion(params, a["H"], a["HE"], a["HG"], n_hidden=a["L"], act=a["act_name"],
                       formulation=a.get("formulation", "A"), precision=a["precision"]).to(dev)
        if name.startswith("ref16"):
            m.load_flat_params(torch.tensor(np.load(os.path.join(ROOT, "tests", "golden", "g1_weights.npz"))["ionHsym"]))
        return m

    model = build_model(args.arch)
    theta0 = model.flat_params(dtype=torch.float64).cpu().numpy()

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    xstat = {}

    def run(model, arch_name):
        """W untimed + K timed steps of one workload; returns (seconds, trunk kernel ms, losses, points/GPU)"""
        a = ARCHS[arch_name]
        npts = min(args.points, a["points"]) if "points" in a else args.points
        plan = pkg.DataParallelPlan(npts, rank=rank, world=world, force_collectives=use_dist,
                                    one_shot=True if args.one_shot else None)
        p, lr = params, 8e-3
        if a.get("fine_tune"):
            model.freezeDecayUnit()
            model.freezeBase()
            p, lr = dict(params, RxL=0.5), 5e-4
        tr = pkg.Trainer(model, p, n_points=npts, lr=lr, seed=12345, device=dev, plan=plan)
        if args.graph:
            step = tr.capture_graph(warmup=max(args.warmup, 1))
            tr.h.profile_enable(1)          # events recorded at capture time only
        else:
            step = tr.step
            if not args.no_prewarm:
                # untimed: keep the GPU busy for >= 0.3 s so that a short driver run (--steps 20) measures
                # steady clocks instead of the ramp from idle; the timed region below is unchanged.  The
                # step count is agreed over the ranks (every step is a collective when N > 1).
                t_w = time.perf_counter()
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                per_step = plan.max_over_ranks((time.perf_counter() - t_w) / 10, dev)
                for _ in range(min(int(0.3 / max(per_step, 1e-5)) + 1, 3000)):
                    step()
                torch.cuda.synchronize()
            tr.h.profile_enable(min(max(args.steps, 1), 1024))
            for _ in range(args.warmup):
                step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = plan.max_over_ranks(time.perf_counter() - t0, dev)
        kms = float("nan") if args.graph else tr.h.profile_read(min(args.steps, 1024))
        # losses() checks the one-shot exchange's status first: a peer that missed a call's timeout
        # raises here (and the message of that call was NaN), so no number is printed for such a run
        losses = tr.losses()
        xstat["name"] = plan.exchange_name
        xstat["status"] = "ok" if plan._xchg is not None else "n/a"
        return dt, kms, losses, npts

    def gpu_check(TB):
        """parity numbers observed in THIS run: 16 384 of the benchmark's points and its own initial
        weights through the fp64 CPU restatement vs the HIP path (forward outputs, loss, gradient)"""
        from pinn_for_quantum_wavefunction_surfaces_amd import capi
        n = 16384
        m = build_model(args.arch)
        h = m._handle_for(dev, n)
        th = m.flat_params(dev).contiguous()
        x, y, z, R = [torch.empty(n, device=dev) for _ in range(4)]
        mask = torch.empty(n, device=dev, dtype=torch.uint8)
        cnt = torch.zeros(2, device=dev, dtype=torch.int64)
        h.sample(12345, 0, 0, n, 18.0, float(params["RxL"]), float(params["RxR"]), float(params["cutOff"]),
                 float(params["BCcutoff"]), x, y, z, R, mask, cnt)
        r1 = ((x - R) ** 2 + y * y + z * z).sqrt()
        r2 = ((x + R) ** 2 + y * y + z * z).sqrt()
        far = torch.minimum(r1, r2) > 0.25
        a2 = {k: arch[k] for k in ("H", "L", "HE", "HG", "act", "n_pass", "out_scale")}
        rel = lambda got, ref: float((got.double().cpu().reshape(-1) - ref.detach().reshape(-1)).abs().max()
                                     / ref.detach().abs().max())

        def compare(sel):
            xs, ys, zs, Rs, ms = [t[sel].contiguous() for t in (x, y, z, R, mask)]
            k = int(xs.numel())
            c2 = torch.stack([(ms & 1).sum(), ((ms >> 1) & 1).sum()]).to(torch.int64)
            outs = [torch.empty(k, device=dev) for _ in range(4)]
            h.forward(th, xs, ys, zs, Rs, k, m.ham(), *outs)
            g = torch.empty(h.P + 3, device=dev)
            h.residual_fwd_bwd(th, xs, ys, zs, Rs, ms, k, m.ham(1.0 / k, 0.0, 0.0), c2, (1 << 63) - 1, g)
            torch.cuda.synchronize()
            t64 = torch.tensor(theta0, dtype=torch.float64, requires_grad=True)
            col = lambda t, rg: t.double().cpu().reshape(-1, 1).requires_grad_(rg)
            xc, yc, zc, Rc = col(xs, True), col(ys, True), col(zs, True), col(Rs, False)
            mk = ms.cpu()
            i1, i2 = torch.where(mk & 1)[0], torch.where((mk >> 1) & 1)[0]
            Ltot, Lpde, Lbc, psi, E, lap, res = TB.loss_terms(t64, a2, HAM_A, xc, yc, zc, Rc, i1, i2)
            Ltot.backward()
            return {"points": k, "psi": rel(outs[0], psi), "E": rel(outs[1], E), "lap": rel(outs[2], lap),
                    "res": rel(outs[3], res), "loss": abs(float(g[h.P:].sum()) / float(Ltot.detach()) - 1),
                    "grad": rel(g[:h.P], t64.grad)}

        # reported twice: on ALL sampled points, and without the few closer than 0.25 a.u. to a nucleus -- there the
        # Coulomb and Laplacian terms of a random-init net are so large that those points alone set every max-norm (and
        # the loss), so the second set is the one that speaks about the network part
        out = {"all_points": compare(torch.ones_like(far)), "farther_than_0.25_au": compare(far),
               "against": "fp64 PyTorch-CPU restatement of the reference formulation, same points, same weights "
                          "(max-norm relative errors)"}
        return out

    dt, trunk_ms, (Ltot, Lpde, Lbc), npts = run(model, args.arch)
    xprimary = dict(xstat)
    # other workloads ride along as secondary records on ONE GPU only (same K, W): the split-bf16 variant
    # of the primary (fp32-grade parity on the MFMA pipe), BASELINE config 1 (4x128 fp32-grade at 100 000
    # points), the reference architecture on the fp32 parity path, its fine-tune mode, and the 2x128 trunk
    others = [] if (args.no_secondary or world > 1) else [a for a in SECONDARY.get(args.arch, []) if a != args.arch]
    sec = []
    for other in others:
        omodel = build_model(other)
        odt, otrunk_ms, olosses, onpts = run(omodel, other)
        sec.append((other, ARCHS[other], odt, otrunk_ms, olosses, onpts))
        del omodel
    if rank == 0:
        n_total = npts * world
        fl_trunk = trunk_flops_per_point(arch["H"], arch["L"], arch["n_pass"], fine_tune=arch.get("fine_tune", False))
        achieved = fl_trunk * npts / (trunk_ms * 1e-3) / 1e12
        peak = arch["peak"]
        out = {
            "metric": "PDE-residual collocation points/sec (H2+ ion, batch 1M)",
            "value": n_total * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": arch["dtype"],
            "data": "synthetic",
            "config": {"workload": "H2+ ion symmetric (NN_ion), %d points/step/GPU, %s, fused "
                                   "Laplacian kernel" % (npts, arch["label"]),
                       "arch": args.arch,
                       "points_per_gpu": npts, "global_batch": n_total,
                       "parallelism": "dp%d" % world,
                       "exchange": ("none" if not use_dist else "one-shot peer-memory all-reduce"
                                    if (args.one_shot or os.environ.get("PINN_ONE_SHOT") == "1") else
                                    ("gloo (one-GPU rehearsal)" if rehearsal else "RCCL all-reduce")),
                       "exchange_status": xprimary.get("status", "n/a")},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": TRAFFIC.get(args.arch),
                         "traffic_source": TRAFFIC_SOURCE,
                         "kernel": ("k_trunk16" if arch["H"] == 16 else "k_trunk") if arch["dtype"] == "f32" else _mfma_kernel_name(arch),
                         "kernel_ms": trunk_ms, "flops_per_point": fl_trunk,
                         "note": ("fp32-grade path: algorithmic flops against the fp32 vector peak (157.3 TFLOP/s); the layer GEMMs "
                                  "run as split-bf16 MFMAs, the kernel is bound by its fp32 elementwise algebra"
                                  if arch["dtype"] == "f32" else
                                  "bf16 dense MFMA peak; algorithmic trunk flops only (SURVEY 8d formula, 5 streams)")},
            "loss": {"Ltot": Ltot, "Lpde": Lpde, "Lbc": Lbc},
        }
        out["parity_bar"] = PARITY_BAR[arch["precision"]]
        out["secondary"] = []
        for other, oarch, odt, otrunk_ms, olosses, onpts in sec:
            ofl = trunk_flops_per_point(oarch["H"], oarch["L"], oarch["n_pass"], fine_tune=oarch.get("fine_tune", False))
            oach = ofl * onpts / (otrunk_ms * 1e-3) / 1e12
            out["secondary"].append({
                "workload": oarch["label"], "arch": other, "dtype": oarch["dtype"], "points_per_step": onpts,
                "value": onpts * args.steps / odt, "unit": "points/s",
                "ms_per_step": odt / args.steps * 1e3,
                "roofline": {"bound": "mfma", "achieved": oach, "peak": oarch["peak"],
                             "unit": "TFLOP/s", "frac": oach / oarch["peak"], "traffic": TRAFFIC.get(other),
                             "kernel_ms": otrunk_ms, "flops_per_point": ofl},
                "parity_bar": PARITY_BAR[oarch["precision"]], "loss": {"Ltot": olosses[0]}})
        # the parity-grade throughput as a first-class number: the fastest record of this run whose path is held to
        # north_star's "psi / E(R) within 1e-5" (split-bf16 or fp32 paths; plain bf16 operands cannot meet it)
        grade = [r for r in [dict(arch=args.arch, dtype=arch["dtype"], value=out["value"], ms_per_step=out["ms_per_step"],
                                  roofline=out["roofline"], workload=arch["label"], points_per_step=npts)] + out["secondary"]
                 if r["dtype"] in ("bf16x3", "f32") and ARCHS[r["arch"]]["H"] >= 64 and not ARCHS[r["arch"]].get("fine_tune")]
        if grade:
            best = max(grade, key=lambda r: r["value"])
            out["north_star_grade"] = {"arch": best["arch"], "workload": best["workload"], "dtype": best["dtype"],
                                       "points_per_step": best["points_per_step"], "value": best["value"], "unit": "points/s",
                                       "ms_per_step": best["ms_per_step"], "roofline_frac": best["roofline"]["frac"],
                                       "kernel_ms": best["roofline"]["kernel_ms"], "parity_bar": PARITY_BAR[best["dtype"]],
                                       "note": "fastest MFMA-width (H >= 64) record of this run on a path held to psi, lap <= 1e-5 "
                                               "against the reference (tests/test_gpu_parity.py)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(theta0, arch, gpu_check)
            out["parity_observed"] = out["cpu_baseline"].pop("parity_observed")
        line = json.dumps(out) + "\n"
        if json_fd is not None:
            os.write(json_fd, line.encode())
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
