"""Fused training loop: sample -> residual fwd+bwd -> (all-reduce) -> Adam, all on-stream.

Replaces the reference's hot loops (train.py:21-72; poc/main.py:359-430).  The
loop body issues only C-ABI launches (plus two small collectives when
data-parallel); there is no host synchronisation per step, unlike the
reference's `.numpy()` / `.cpu()` reads (train.py:58; poc/main.py:408-411).
"""
import pickle
import time
from os import path

import numpy as np
import torch

from . import capi
from .model import NN_ion
from .parallel import DataParallelPlan


class Trainer:
    """Owns the flat fp32 state (theta, Adam moments, point buffers) of one rank and runs steps.

    Data parallel (plan.collectives): ONE small all-reduce per step.  The points of step t+1 are
    sampled while step t is in flight (two point buffers), and their local boundary-set sizes
    travel as two extra floats behind [grad | loss terms] of step t, so the global 1/N_bc of the
    next step arrives with this step's gradient sum (exact: counts < 2^24).  Philox step t still
    draws the points of training step t, so the point sequence does not depend on this pipelining.
    """

    def __init__(self, model, params, n_points=None, lr=None, seed=12345, device=None,
                 plan=None, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.params = model, params
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise capi.PinnError("Trainer runs on the GPU only; there is no CPU fallback")
        self.n = int(n_points if n_points is not None else params["n_train"])
        self.plan = plan if plan is not None else DataParallelPlan(self.n)
        self.lr = float(lr if lr is not None else params["lr"])
        self.b1, self.b2, self.eps = float(betas[0]), float(betas[1]), float(eps)
        self.seed = int(seed)
        self.h = model._handle_for(self.device, self.n)
        P, dev, f32 = self.h.P, self.device, torch.float32
        self.P = P
        self.theta = model.flat_params(dev).contiguous()
        self.m = torch.zeros(P, device=dev, dtype=f32)
        self.v = torch.zeros(P, device=dev, dtype=f32)
        self.gbuf = torch.zeros(P + 5, device=dev, dtype=f32)   # [grad | Lpde Lbc1 Lbc2 | next counts]
        self.grad = self.gbuf[:P + 3]
        self.step_count = torch.zeros(1, device=dev, dtype=torch.int32)
        self.best_theta = self.theta.clone()
        self.best_loss = torch.full((1,), float("inf"), device=dev, dtype=f32)
        self._pts = [[torch.empty(self.n, device=dev, dtype=f32) for _ in range(4)] +
                     [torch.empty(self.n, device=dev, dtype=torch.uint8)] for _ in range(2)]
        self._cnt = [torch.zeros(2, device=dev, dtype=torch.int64) for _ in range(2)]  # GLOBAL sizes
        self._cnt_local = torch.zeros(2, device=dev, dtype=torch.int64)
        self._cur = 0
        self.tmask = model.trainable_mask()
        self.t = 0
        self._sampled = False
        # one-message pipelining needs exact float counts
        self.pipelined = self.plan.collectives and self.plan.n_global < (1 << 24)

    # the current batch (views change when the buffers swap)
    x = property(lambda self: self._pts[self._cur][0])
    y = property(lambda self: self._pts[self._cur][1])
    z = property(lambda self: self._pts[self._cur][2])
    R = property(lambda self: self._pts[self._cur][3])
    mask = property(lambda self: self._pts[self._cur][4])
    counts = property(lambda self: self._cnt[self._cur])

    def _sample_into(self, buf, counts_out, step, ahead):
        p = self.params
        x, y, z, R, mask = self._pts[buf]
        self.h.sample(self.seed, ahead if step is None else step, self.plan.first, self.n,
                      float(p["xR"]), float(p["RxL"]), float(p["RxR"]), float(p["cutOff"]),
                      float(p["BCcutoff"]), x, y, z, R, mask, counts_out,
                      step_dev=self.step_count if step is None else None)

    def resample(self, step=None):
        """Draw this rank's points of Philox step `step` into the current buffer and make their
        global boundary-set sizes available; by default the step is the device-side Adam step
        counter (so a captured graph of step() draws fresh points on every replay)."""
        self._sample_into(self._cur, self._cnt[self._cur], step, 0)
        self.plan.reduce_counts(self._cnt[self._cur])
        self._sampled = True

    def capture_graph(self, resample=True, track_best=True, warmup=3):
        """Capture one whole step (sample -> fused fwd+bwd -> all-reduce -> Adam) into a HIP graph
        and return a callable that replays it (bit-identical to eager steps).  Measured: no gain,
        even at 10^4 points the step is bound by per-tile kernel latency, not by launches."""
        self.pipelined = False                      # a graph is bound to one set of buffers
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

    def step(self, resample=True, track_best=True):
        """One training step; returns nothing (read .losses() when a host value is wanted)."""
        pipe = self.pipelined and resample
        if not self._sampled or (resample and not self.pipelined):
            self.resample()                         # non-pipelined: sample, then reduce the counts
        cur = self._cur
        x, y, z, R, mask = self._pts[cur]
        ham = self.model.ham(self.plan.invN, 0.0, 0.0)
        self.h.residual_fwd_bwd(self.theta, x, y, z, R, mask, self.n, ham, self._cnt[cur],
                                self.tmask, self.grad)
        if pipe:                                    # next step's points + their local counts
            self._sample_into(1 - cur, self._cnt_local, None, 1)
            # ONE message: [grad | losses | next counts]
            self.plan.reduce_grad_and_counts(self.gbuf, self._cnt_local, self._cnt[1 - cur])
            self._cur = 1 - cur
        else:
            self.plan.reduce_grad(self.grad)
        self.h.adam(self.theta, self.grad, self.m, self.v, self.step_count, self.lr, self.b1,
                    self.b2, self.eps, self.tmask,
                    self.best_theta if track_best else None,
                    self.best_loss if track_best else None)
        self.t += 1

    def losses(self):
        """(Ltot, Lpde, Lbc) of the last step as floats (this one does synchronise)."""
        l = self.grad[self.P:].tolist()
        return l[0] + l[1] + l[2], l[0], l[1] + l[2]

    def sync_to_model(self, best=False):
        self.model.load_flat_params(self.best_theta if best else self.theta)


def train(params, loadWeights=False, freezeUnits=False, model=None, device="cuda", seed=12345,
          verbose=True):
    """Reference driver poc/main.py:359-430 on the fused path.  Returns (model, history)."""
    model = model if model is not None else NN_ion(params)
    if path.exists(params["loadModelPath"]) and loadWeights:
        model.loadModel(params)
    if freezeUnits:                                     # poc/main.py:384-387
        model.freezeDecayUnit()
        model.freezeBase()
    model.to(device)
    epochs = int(params["epochs"])
    tr = Trainer(model, params, params["n_train"], params["lr"], seed=seed, device=device)
    hist = torch.zeros(epochs, 3, device=tr.device, dtype=torch.float32)
    E_h = torch.zeros(epochs, device=tr.device, dtype=torch.float32)
    t0 = time.time()
    for tt in range(epochs):
        resample = tt % params["sc_sampling"] == 0 and tt < 0.9 * epochs   # poc/main.py:396
        tr.step(resample=resample, track_best=tt > 0.5 * epochs)           # poc/main.py:414
        hist[tt].copy_(tr.grad[tr.P:])
        tr.h.forward(tr.theta, tr.x[-1:], tr.y[-1:], tr.z[-1:], tr.R[-1:], 1, model.ham(),
                     E=E_h[tt:tt + 1])                                     # poc/main.py:411
    torch.cuda.synchronize()
    runtime = time.time() - t0
    hist = hist.cpu().numpy().astype(np.float64)
    history = {"Ltot": hist.sum(1, keepdims=True), "Lbc": hist[:, 1:].sum(1, keepdims=True),
               "Lpde": hist[:, :1].copy(), "Energy": E_h.cpu().numpy().astype(np.float64)[:, None]}
    tr.sync_to_model(best=bool(torch.isfinite(tr.best_loss).item()))
    if verbose:
        print("Runtime (min): ", runtime / 60)
    return model, history


def saveLoss(params, lossDictionary):
    """poc/main.py:159-161 (same pickle schema: dict of (epochs,1) float64 arrays)."""
    with open(params["lossPath"], "wb") as f:
        pickle.dump(lossDictionary, f)
