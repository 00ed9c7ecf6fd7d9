"""Fused training loop: sample -> residual fwd+bwd -> (all-reduce) -> Adam, all on-stream.

Replaces the reference's hot loops (train.py:21-72; poc/main.py:359-430).  The
loop body issues only C-ABI launches (plus two small collectives when
data-parallel); there is no host synchronisation per step, unlike the
reference's `.numpy()` / `.cpu()` reads (train.py:58; poc/main.py:408-411).
"""
import os
import pickle
import time
from os import path

import numpy as np
import torch

from . import capi
from .model import NN_ion
from .parallel import DataParallelPlan


def check_cubic_box(params):
    """pinn_sample draws from the symmetric cube [-xR, xR]^3 -- the only box the reference ever uses
    (poc/main.py:18-23, train.py:80) -- so anything else is refused instead of silently ignored."""
    box = [float(params[k]) for k in ("xL", "xR", "yL", "yR", "zL", "zR")]
    if not (box[0] == -box[1] and box[0:2] == box[2:4] == box[4:6] and box[1] > 0):
        raise capi.PinnError("the sampler draws from the symmetric cube [-xR, xR]^3; got "
                             "(xL, xR, yL, yR, zL, zR) = %r" % (box,))


class Trainer:
    """Owns the flat fp32 state (theta, Adam moments, point buffers) of one rank and runs steps.

    Data parallel (plan.collectives): ONE small all-reduce per step.  When step t says that the next
    resampling step is t + k (`next_resample=k`, default 1: the reference resamples every step), the
    points of step t + k are drawn into the second point buffer while step t is in flight, and their
    local boundary-set sizes travel as two extra floats behind [grad | loss terms] of step t, so the
    global 1/N_bc of that batch arrives with this step's gradient sum (exact: counts < 2^24).  The
    buffers swap at the START of step t + k.  So, exactly as on one rank: a resampling step t trains on
    the points of Philox step t, a non-resampling step trains on the batch of the step before it, and
    after step() the properties x, y, z, R, mask, counts are the batch that step trained on.
    Without the hint, above 2^24 global points, or with pipelined=False, a resampling step samples
    and all-reduces the set sizes (int64[2]) before the fused kernel: two collectives.

    One handle = one stream: the handle's scratch (per-point E/g and adjoints, per-block partials,
    packed weights) is shared by all calls on it, so calls on one handle must be ordered on ONE
    stream (include/pinn.h); the Trainer issues everything on the current stream.
    """

    def __init__(self, model, params, n_points=None, lr=None, seed=12345, device=None,
                 plan=None, betas=(0.9, 0.999), eps=1e-8, pipelined=None, handle=None):
        self.model, self.params = model, params
        self.device = torch.device(device if device is not None else "cuda")
        # `handle`: test seam only (tests/test_parallel_gloo.py drives the step sequencing below over
        # gloo with a stand-in for the kernels); the product always builds the HIP handle
        if self.device.type != "cuda" and handle is None:
            raise capi.PinnError("Trainer runs on the GPU only; there is no CPU fallback")
        check_cubic_box(params)
        self.n = int(n_points if n_points is not None else params["n_train"])
        self.plan = plan if plan is not None else DataParallelPlan(self.n)
        self.lr = float(lr if lr is not None else params["lr"])
        self.b1, self.b2, self.eps = float(betas[0]), float(betas[1]), float(eps)
        self.seed = int(seed)
        self.h = handle if handle is not None else model._handle_for(self.device, self.n)
        P, dev, f32 = self.h.P, self.device, torch.float32
        self.P = P
        self.theta = model.flat_params(dev).contiguous()
        self.m = torch.zeros(P, device=dev, dtype=f32)
        self.v = torch.zeros(P, device=dev, dtype=f32)
        self.gbuf = torch.zeros(P + 5, device=dev, dtype=f32)   # [grad | Lpde Lbc1 Lbc2 | next counts]
        self.grad = self.gbuf[:P + 3]
        if dev.type == "cuda":
            self.plan.attach_exchange(dev, P + 5)               # one-shot exchange, if the plan asks for it
        self.step_count = torch.zeros(1, device=dev, dtype=torch.int32)
        self.best_theta = self.theta.clone()
        self.best_loss = torch.full((1,), float("inf"), device=dev, dtype=f32)
        self._pts = [[torch.empty(self.n, device=dev, dtype=f32) for _ in range(4)] +
                     [torch.empty(self.n, device=dev, dtype=torch.uint8)] for _ in range(2)]
        self._cnt = [torch.zeros(2, device=dev, dtype=torch.int64) for _ in range(2)]  # GLOBAL sizes
        self._cur = 0
        self.tmask = model.trainable_mask()
        self.t = 0
        self._sampled = False
        self._ahead_for = -1        # step index at which the pre-sampled second buffer becomes current
        # best-model rule: False = train.py:58-60 (the parameters the loss was evaluated at, fused in
        # pinn_adam); True = poc/main.py:414-417 (the parameters AFTER that step's update, with the
        # optimizer state of that moment, as saveModel stores them)
        self.best_post_step = False
        # reduction + Adam as one launch (pinn_adam_fused) where no exchange sits between them; PINN_FUSE_ADAM=0: two calls
        self.fuse_adam = os.environ.get("PINN_FUSE_ADAM", "1") != "0"
        self.best_m = self.best_v = None
        self.best_step = torch.zeros(1, device=dev, dtype=torch.int32)
        # one-message pipelining needs exact float counts
        auto = self.plan.collectives and self.plan.n_global < (1 << 24)
        self.pipelined = auto if pipelined is None else (bool(pipelined) and auto)

    # the current batch (views change when the buffers swap)
    x = property(lambda self: self._pts[self._cur][0])
    y = property(lambda self: self._pts[self._cur][1])
    z = property(lambda self: self._pts[self._cur][2])
    R = property(lambda self: self._pts[self._cur][3])
    mask = property(lambda self: self._pts[self._cur][4])
    counts = property(lambda self: self._cnt[self._cur])

    def _sample_into(self, buf, counts_out, step, ahead, counts_f32=None):
        p = self.params
        x, y, z, R, mask = self._pts[buf]
        self.h.sample(self.seed, ahead if step is None else step, self.plan.first, self.n,
                      float(p["xR"]), float(p["RxL"]), float(p["RxR"]), float(p["cutOff"]),
                      float(p["BCcutoff"]), x, y, z, R, mask, counts_out,
                      step_dev=self.step_count if step is None else None, counts_f32=counts_f32)

    def resample(self, step=None):
        """Draw this rank's points of Philox step `step` into the current buffer and make their
        global boundary-set sizes available; by default the step is the device-side Adam step
        counter (so a captured graph of step() draws fresh points on every replay)."""
        cf = self.plan.counts_f32()                 # one-shot exchange: the sampler writes the floats it sends
        self._sample_into(self._cur, self._cnt[self._cur], step, 0, counts_f32=cf)
        self.plan.reduce_counts(self._cnt[self._cur], counts_f32=cf)
        self._sampled = True

    def capture_graph(self, resample=True, track_best=True, warmup=3):
        """Capture one whole step (sample -> fused fwd+bwd -> all-reduce -> Adam) into a HIP graph
        and return a callable that replays it (bit-identical to eager steps).  Measured: no gain,
        even at 10^4 points the step is bound by per-tile kernel latency, not by launches."""
        self.pipelined = False                      # a graph is bound to one set of buffers
        # (the one-shot exchange counts its calls in device memory, so it replays correctly; data
        # parallel, every rank must capture and replay the same step)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step(resample=resample, track_best=track_best)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.step(resample=resample, track_best=track_best)

        def replay():
            graph.replay()
            self.t += 1
        self._graph = graph
        return replay

    def step(self, resample=True, track_best=True, next_resample=1):
        """One training step; returns nothing (read .losses() when a host value is wanted).
        resample: draw fresh points for this step (Philox step = this step's index) or keep the batch.
        next_resample: in how many steps the next resampling step comes (1 = the next one; 0/None =
        unknown or never): lets the data-parallel path fold that batch's set sizes into this step's
        one all-reduce."""
        if resample and self.pipelined and self._ahead_for == self.t:
            self._cur = 1 - self._cur               # the batch drawn ahead for exactly this step
            self._ahead_for = -1
        elif resample or not self._sampled:
            self.resample()                         # sample, then all-reduce the set sizes
        cur = self._cur
        x, y, z, R, mask = self._pts[cur]
        ham = self.model.ham(self.plan.invN, 0.0, 0.0)
        # single device, best model tracked by the Adam kernel itself (or not at all): reduction and Adam are one launch
        fused = self.fuse_adam and not self.plan.collectives and not (track_best and self.best_post_step)
        self.h.residual_fwd_bwd(self.theta, x, y, z, R, mask, self.n, ham, self._cnt[cur],
                                self.tmask, None if fused else self.grad)
        if fused:
            self.h.adam_fused(self.theta, self.grad, self.m, self.v, self.step_count, self.lr, self.b1, self.b2, self.eps,
                              self.best_theta if track_best else None, self.best_loss if track_best else None)
            self.t += 1
            return
        k = int(next_resample or 0)
        # points of step t + k and their local counts -- unless an earlier step already drew them (with
        # sc_sampling > 1 the driver hints at the same step twice: at the resampling step and at t+k-1)
        msg = {}
        if self.pipelined and k >= 1 and self._ahead_for != self.t + k:
            # ONE message: [grad | losses | that batch's local set sizes], the sizes written as floats behind the
            # loss terms by the sampler; the Adam kernel below stores their sums as int64 for the step that
            # trains on that batch -- no conversion kernels in between
            self._sample_into(1 - cur, None, None, k, counts_f32=self.gbuf[self.P + 3:])
            self.plan.reduce_message(self.gbuf)
            msg = dict(msg_counts=self.gbuf[self.P + 3:], counts_out=self._cnt[1 - cur])
            self._ahead_for = self.t + k
        else:
            self.plan.reduce_grad(self.grad)
        if track_best and self.best_post_step:
            # poc/main.py:414-417: `if Ltot < Llim` is tested on this step's loss, the model saved is
            # the one optimizer.step() has just produced; all on the device, no host sync
            if self.best_m is None:
                self.best_m, self.best_v = torch.zeros_like(self.m), torch.zeros_like(self.v)
            ltot = self.grad[self.P:].sum()
            better = ltot < self.best_loss
            self.h.adam(self.theta, self.grad, self.m, self.v, self.step_count, self.lr, self.b1,
                        self.b2, self.eps, self.tmask, None, None, **msg)
            for dst, src in ((self.best_theta, self.theta), (self.best_m, self.m), (self.best_v, self.v),
                             (self.best_step, self.step_count), (self.best_loss, ltot.reshape(1))):
                torch.where(better, src, dst, out=dst)
        else:
            self.h.adam(self.theta, self.grad, self.m, self.v, self.step_count, self.lr, self.b1,
                        self.b2, self.eps, self.tmask,
                        self.best_theta if track_best else None,
                        self.best_loss if track_best else None, **msg)
        self.t += 1

    def evaluate(self, resample=True, track_best=True):
        """One loss evaluation WITHOUT a parameter update -- train.py's loop runs epochs + 1 of them
        for epochs optimizer steps (train.py:24-69: the last pass only evaluates, updates the best
        snapshot and restores it).  Leaves the loss terms in .grad[P:]."""
        if resample or not self._sampled:
            self.resample()
        x, y, z, R, mask = self._pts[self._cur]
        self.h.residual_fwd_bwd(self.theta, x, y, z, R, mask, self.n, self.model.ham(self.plan.invN, 0.0, 0.0),
                                self._cnt[self._cur], self.tmask, self.grad)
        self.plan.reduce_grad(self.grad)
        if track_best:
            ltot = self.grad[self.P:].sum()
            better = ltot < self.best_loss
            torch.where(better, self.theta, self.best_theta, out=self.best_theta)
            torch.where(better, ltot.reshape(1), self.best_loss, out=self.best_loss)

    def optimizer_state_dict(self, best=False):
        """An object whose state_dict() has the schema torch.optim.Adam writes into the reference's
        checkpoints (per trainable parameter id `step` as an fp64 scalar tensor,
        exp_avg, exp_avg_sq in the parameter's shape; one param_group) -- what NN_ion.saveModel
        (poc/main.py:331-339) stores next to model_state_dict.  best=True: the moments saved with the
        best model (best_post_step rule)."""
        m, v, step = (self.best_m, self.best_v, self.best_step) if best and self.best_m is not None \
            else (self.m, self.v, self.step_count)
        off, sz = capi.tensor_layout(self.model.arch)
        state = {}
        plist = list(self.model.parameters())
        for t, p in enumerate(plist):
            if not (self.tmask >> t) & 1:
                continue
            sl = slice(off[t], off[t] + sz[t])
            state[t] = {"step": step[0].to(torch.float64).cpu(),
                        "exp_avg": m[sl].reshape(p.shape).to(torch.float64).cpu().clone(),
                        "exp_avg_sq": v[sl].reshape(p.shape).to(torch.float64).cpu().clone()}
        group = {"lr": self.lr, "betas": (self.b1, self.b2), "eps": self.eps, "weight_decay": 0,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                 "initial_lr": self.lr, "params": list(range(len(plist)))}
        sd = {"state": state, "param_groups": [group]}

        class _State:
            def state_dict(self_inner):
                return sd
        return _State()

    def losses(self):
        """(Ltot, Lpde, Lbc) of the last step as floats (this one does synchronise; on the one-shot
        exchange it is also where a peer that missed the timeout is reported)."""
        self.plan.check()
        l = self.grad[self.P:].tolist()
        return l[0] + l[1] + l[2], l[0], l[1] + l[2]

    def sync_to_model(self, best=False):
        self.plan.check()
        self.model.load_flat_params(self.best_theta if best else self.theta)


def train(params, loadWeights=False, freezeUnits=False, model=None, device="cuda", seed=12345,
          verbose=True, save=True):
    """Reference driver poc/main.py:359-430 on the fused path.  Returns (model, history).

    Kept from the reference: resample every sc_sampling steps during the first 90 % of the epochs
    (:396); after 50 % of the epochs a step whose total loss is below the best so far (initially
    Llim = 10, :372,414-417) makes the weights AFTER that step's update the saved model; the loss
    history is written with saveLoss (:421-427) and the best model with saveModel's schema
    (model_state_dict + optimizer_state_dict, :331-339) to params['saveModelPath'].  Different on
    purpose: the best weights are tracked on the device (no host sync per step) and written once at
    the end instead of torch.save inside the loop; the history's Energy column is E of the last
    sample point of the batch each step trained on (:411)."""
    model = model if model is not None else NN_ion(params)
    if path.exists(params["loadModelPath"]) and loadWeights:
        model.loadModel(params)
    if freezeUnits:                                     # poc/main.py:384-387
        model.freezeDecayUnit()
        model.freezeBase()
    model.to(device)
    epochs = int(params["epochs"])
    sc = int(params["sc_sampling"])
    tr = Trainer(model, params, params["n_train"], params["lr"], seed=seed, device=device)
    tr.best_loss.fill_(10.0)                            # Llim, poc/main.py:372
    tr.best_post_step = True                            # poc: saveModel runs after optimizer.step()
    hist = torch.zeros(epochs, 3, device=tr.device, dtype=torch.float32)
    E_h = torch.zeros(epochs, device=tr.device, dtype=torch.float32)
    t0 = time.time()

    def resamples(tt):
        return tt % sc == 0 and tt < 0.9 * epochs       # poc/main.py:396

    for tt in range(epochs):
        nxt = next((k for k in range(1, sc + 1) if tt + k < epochs and resamples(tt + k)), 0)
        tr.step(resample=resamples(tt), track_best=tt > 0.5 * epochs,     # poc/main.py:414
                next_resample=nxt if resamples(tt) or nxt == 1 else 0)
        hist[tt].copy_(tr.grad[tr.P:])
        tr.h.forward(tr.theta, tr.x[-1:], tr.y[-1:], tr.z[-1:], tr.R[-1:], 1, model.ham(),
                     E=E_h[tt:tt + 1])                                     # poc/main.py:411
    torch.cuda.synchronize()
    tr.plan.check()
    runtime = time.time() - t0
    hist = hist.cpu().numpy().astype(np.float64)
    history = {"Ltot": hist.sum(1, keepdims=True), "Lbc": hist[:, 1:].sum(1, keepdims=True),
               "Lpde": hist[:, :1].copy(), "Energy": E_h.cpu().numpy().astype(np.float64)[:, None]}
    found_best = bool((tr.best_loss < 10.0).item())
    tr.sync_to_model(best=found_best)
    if save:
        for key in ("saveModelPath", "lossPath"):       # the reference expects models/ and data/ to exist
            d = path.dirname(params.get(key) or "")
            if d:
                os.makedirs(d, exist_ok=True)
        if found_best and params.get("saveModelPath"):
            model.saveModel(params, tr.optimizer_state_dict(best=True))
        if params.get("lossPath"):
            saveLoss(params, history)
    if verbose:
        print("Runtime (min): ", runtime / 60)
    return model, history


def saveLoss(params, lossDictionary):
    """poc/main.py:159-161 (same pickle schema: dict of (epochs,1) float64 arrays)."""
    with open(params["lossPath"], "wb") as f:
        pickle.dump(lossDictionary, f)
