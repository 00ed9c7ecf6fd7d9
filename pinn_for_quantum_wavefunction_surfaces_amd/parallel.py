"""Data-parallel plan for the collocation batch.

Points are i.i.d., the loss is a mean over points, so the path shards naturally:
rank r draws points [r*n_local, (r+1)*n_local) of the global Philox stream and
the only exchanged data are SUMs, of
  * the BC set sizes (2 integers per batch), so every rank weights its boundary
    points with the GLOBAL 1/N_bc1, 1/N_bc2;
  * one flat fp32 buffer [grad(P) | Lpde, Lbc1, Lbc2] after the fused kernel
    (already scaled with global 1/N), after which every rank runs the same Adam.
The Trainer sends both in ONE all-reduce per step: the next batch is sampled while
the current step runs and its set sizes ride behind the gradient as two floats -- written there by the
sampler itself (pinn_sample_dp) and turned back into int64 by the Adam kernel that consumes the message
(pinn_adam_dp), so the step contains no conversion kernels (exact below 2^24 points; above that, or for a
one-off batch, reduce_counts is used).
The backend is whatever torch.distributed was initialised with ("nccl" = RCCL over
xGMI on MI355X; "gloo" in the CPU tests).  With world_size 1 nothing is called.

one_shot=True (or PINN_ONE_SHOT=1) replaces the all-reduce of the gradient message by the
library's own exchange over peer-mapped memory (include/pinn.h pinn_xchg_*: every rank writes its
6-205 KB vector into a slot in every peer's memory over the xGMI mesh and sums the N slots in rank
order -- one hop, two small kernels on the compute stream, bitwise identical sums on all ranks);
torch.distributed is then only used at setup (to trade the 64-byte memory handles, and one barrier
before the first call); the boundary-set sizes of reduce_counts travel through the same exchange as
16-bit pieces in fp32 (exact), so a captured step contains no torch.distributed call at all.  A peer
that misses the exchange timeout turns the message into NaN and sets a status that check() -- called
wherever the Trainer synchronises anyway -- raises as PinnError.  RCCL stays the default until the
one-shot path has been measured on an 8-GPU node.
"""
import os

import torch
import torch.distributed as dist


class DataParallelPlan:
    def __init__(self, n_local, rank=None, world=None, group=None, force_collectives=False, one_shot=None):
        if world is None:
            world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if world > 1 else 0
        self.n_local, self.rank, self.world, self.group = int(n_local), int(rank), int(world), group
        # world == 1 normally skips the collectives; forcing them exercises the RCCL path on one GPU
        self.collectives = self.world > 1 or bool(force_collectives)
        self.one_shot = (os.environ.get("PINN_ONE_SHOT") == "1") if one_shot is None else bool(one_shot)
        self._xchg = None

    def attach_exchange(self, device, max_floats):
        """one_shot: build the peer-memory exchange for messages of up to max_floats fp32 on `device`
        (called by the Trainer once it knows the message length); no-op otherwise"""
        if not (self.one_shot and self.collectives) or self._xchg is not None:
            return
        from . import capi

        def gather(mine):
            if self.world == 1:
                return [mine]
            out = [None] * self.world
            dist.all_gather_object(out, mine, group=self.group)
            return out
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self._xchg = capi.Exchange(idx, self.rank, self.world, max(int(max_floats), 8), gather)
        self._cnt_f = torch.zeros(8, device=device, dtype=torch.float32)    # 2 counts x 4 pieces of 16 bits
        self._shifts = torch.tensor([0, 16, 32, 48], device=device, dtype=torch.int64)
        if self.world > 1:
            # every rank has mapped every peer before anybody's first push; also absorbs start-up skew
            # (library load, first-touch) that would otherwise eat into the first call's timeout
            dist.barrier(group=self.group)

    def check(self):
        """Raise PinnError if a one-shot exchange call has timed out on a peer (synchronises the
        current stream); a no-op on the RCCL/gloo path, where a missing peer blocks instead."""
        if self._xchg is not None:
            self._xchg.status()

    @property
    def exchange_name(self):
        if not self.collectives:
            return "none"
        return "one-shot peer-memory all-reduce" if self._xchg is not None else "all-reduce"

    def _sum(self, buf):
        if self._xchg is not None:
            self._xchg.allreduce(buf)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)

    @property
    def first(self):
        """offset of this rank's shard in the global point stream"""
        return self.rank * self.n_local

    @property
    def n_global(self):
        return self.n_local * self.world

    @property
    def invN(self):
        return 1.0 / self.n_global

    def counts_f32(self):
        """one-shot path below 2^24 global points: the float[2] buffer pinn_sample_dp writes the local set sizes
        into, so that reduce_counts needs no conversion kernels before the exchange (None otherwise)"""
        if self._xchg is not None and self.n_global < (1 << 24):
            return self._cnt_f[:2]
        return None

    def reduce_counts(self, counts, counts_f32=None):
        """int64[2] BC set sizes: local -> global (in place).  counts_f32: the same local sizes as floats, already
        written by the sampler (see counts_f32())."""
        if not self.collectives:
            return counts
        if self._xchg is not None and counts_f32 is not None:
            self._xchg.allreduce(counts_f32)           # integers below 2^24: exact in fp32, on every rank alike
            counts.copy_(counts_f32)
        elif self._xchg is not None:
            # 2^24 points and more: 16-bit pieces as floats, sums over <= 256 ranks stay below 2^24, i.e. exact in
            # fp32 (the exchange itself takes at most kMaxWorld = 16 ranks)
            self._cnt_f.copy_(torch.bitwise_and(counts[:, None] >> self._shifts, 0xFFFF).reshape(8))
            self._xchg.allreduce(self._cnt_f)
            counts.copy_((self._cnt_f.reshape(2, 4).to(torch.int64) << self._shifts).sum(1))
        else:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)
        return counts

    def reduce_grad(self, buf):
        """fp32[P+3] = [grad | three loss terms]: local partial -> global (in place)."""
        if self.collectives:
            self._sum(buf)
        return buf

    def reduce_message(self, buf):
        """One message for both exchanges: buf = fp32[P+5] = [grad | three loss terms | the NEXT batch's two local
        BC set sizes, written there as floats by pinn_sample_dp]; comes back summed.  pinn_adam_dp stores the two
        summed sizes as int64 for the next pinn_residual_fwd_bwd.  Needs n_global < 2^24 (exact in fp32)."""
        assert self.n_global < (1 << 24)
        if self.collectives:
            self._sum(buf)
        return buf

    def reduce_grad_and_counts(self, buf, counts_local, counts_global):
        """reduce_message with the conversions done by torch copies (round 3's form; kept for callers that hold the
        local sizes as int64): counts_local travel in the two spare floats, come back summed in counts_global."""
        buf[-2:].copy_(counts_local)            # int64 -> fp32 inside the copy kernel (exact < 2^24)
        self.reduce_message(buf)
        counts_global.copy_(buf[-2:])           # sums of integers below 2^24 are exact in fp32
        return buf

    def max_over_ranks(self, seconds, device):
        t = torch.tensor([seconds], dtype=torch.float64, device=device)
        if self.collectives:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())
