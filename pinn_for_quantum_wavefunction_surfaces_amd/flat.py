"""train.py's flat-script surface on the fused path.

train.py keeps the network as a tuple of 16 leaf tensors
    (H1a, H1b, H2a, H2b, H3a, H3b, L1a, L1b, L2a, L2b, E1a, E1b, E2a, E2b, E3a, E3b)
with weights stored (in, out) (`x @ A + b`, train.py:4-5), initialised by `ini` (train.py:13-18),
trained by `train(params, lr, epochs)` (train.py:21-73) and written to `model.bin`
(train.py:112-119).  This module offers the same three entry points; the loop body -- resample,
loss (train.py:41-57: one trunk pass, `2*h`, res = lap(psi) + (e + 1/r1 + 1/r2) psi), backward, Adam,
best-so-far tracking from step 0 with the best parameters restored at the end (train.py:58-60,
65-69) -- is the HIP path: pinn_sample -> pinn_residual_fwd_bwd -> pinn_adam.
"""
import numpy as np
import torch

from . import formats
from .model import NN_ion, set_params
from .trainer import Trainer

NAMES = ("H1a", "H1b", "H2a", "H2b", "H3a", "H3b", "L1a", "L1b", "L2a", "L2b", "E1a", "E1b", "E2a",
         "E2b", "E3a", "E3b")
# train.py's module-level constants (train.py:78-87)
DEFAULTS = dict(bcutoff=17.5, cutoff=0.005, L=18, n=10000, Rlo=0.2, Rhi=3, sc_sampling=1, nh=16,
                ne=32, nl=10)


def ini(*shape, dtype=torch.double, generator=None):
    """train.py:13-18: U(-1/sqrt(shape[0]), 1/sqrt(shape[0])) -- for a bias, shape[0] is its length."""
    x = torch.empty(shape, dtype=dtype, requires_grad=True)
    lim = 1 / shape[0] ** 0.5
    with torch.no_grad():
        x.uniform_(-lim, lim, generator=generator)
    return x


def init_params(nh=16, ne=32, nl=10, dtype=torch.double):
    """The 16 tensors in train.py's creation order (train.py:88-103), so that under the same
    torch.manual_seed the same numbers come out."""
    shapes = [(2, nh), (nh,), (nh, nh), (nh,), (nh, 1), (1,), (1, nl), (nl,), (nl, 1), (1,),
              (1, ne), (ne,), (ne, ne), (ne,), (ne, 1), (1,)]
    return tuple(ini(*s, dtype=dtype) for s in shapes)


def _to_model(params, device):
    if len(params) != 16:
        raise ValueError("expected train.py's 16 tensors %s" % (NAMES,))
    nh, ne, nl = params[0].shape[1], params[10].shape[1], params[6].shape[1]
    p = set_params()
    model = NN_ion(p, nh, ne, nl, formulation="B")
    sd = formats.bin_arrays_to_state_dict([t.detach().cpu().numpy() for t in params])
    model.load_state_dict({k: torch.tensor(v, dtype=torch.float32) for k, v in sd.items()})
    return model.to(device)


def train(params, lr, epochs, n=None, Rlo=None, Rhi=None, L=None, cutoff=None, bcutoff=None,
          device="cuda", seed=12345, verbose=False):
    """train.py:21-73.  `params`: the 16 tensors (updated IN PLACE with the best parameters, like
    train.py:65-69).  Returns {"Ltot", "Lpde", "Lbc", "Lbest"} with one row per optimizer step;
    "Lfinal" is the loss of the closing evaluation (train.py's pass tt == epochs)."""
    d = DEFAULTS
    n = d["n"] if n is None else n
    box = d["L"] if L is None else L
    model = _to_model(params, device)
    p = set_params()
    p.update(xL=-box, xR=box, yL=-box, yR=box, zL=-box, zR=box,
             RxL=d["Rlo"] if Rlo is None else Rlo, RxR=d["Rhi"] if Rhi is None else Rhi,
             cutOff=d["cutoff"] if cutoff is None else cutoff,
             BCcutoff=d["bcutoff"] if bcutoff is None else bcutoff)
    tr = Trainer(model, p, n_points=n, lr=lr, seed=seed, device=device)
    hist = torch.zeros(epochs, 3, device=tr.device, dtype=torch.float32)
    best = torch.zeros(epochs, device=tr.device, dtype=torch.float32)
    e_buf = torch.empty(n, device=tr.device, dtype=torch.float32) if verbose else None
    for tt in range(epochs):
        show = verbose and tt % 10 == 0
        theta_eval = tr.theta.clone() if show else None  # the parameters this pass's loss is evaluated at
        tr.step(resample=True, track_best=True)          # sc_sampling = 1; best tracked from step 0
        hist[tt].copy_(tr.grad[tr.P:])
        best[tt:tt + 1].copy_(tr.best_loss)
        if show:                                         # train.py:61-64: Ltot Lpde Lbc (mean(e)) [Lbest]; synchronises
            tr.h.forward(theta_eval, tr.x, tr.y, tr.z, tr.R, n, model.ham(), E=e_buf)
            l = hist[tt].tolist()
            print("%8d: %.2e %.2e %.2e (%.2e) [%.5e]" % (tt, sum(l), l[0], l[1] + l[2], float(e_buf.mean()),
                                                        float(best[tt])))
    tr.evaluate(resample=True, track_best=True)          # train.py:24-69: pass epochs + 1 only evaluates,
    torch.cuda.synchronize()                             # so the final parameters can become the best
    tr.sync_to_model(best=True)                          # train.py:65-69
    arrs = formats.state_dict_to_bin_arrays(model.state_dict())
    with torch.no_grad():
        for t, a in zip(params, arrs):
            t.copy_(torch.tensor(a, dtype=t.dtype).reshape(t.shape))
    h = hist.cpu().numpy().astype(np.float64)
    return {"Ltot": h.sum(1), "Lpde": h[:, 0].copy(), "Lbc": h[:, 1:].sum(1),
            "Lbest": best.cpu().numpy().astype(np.float64), "Lfinal": float(tr.grad[tr.P:].sum())}


def write_model_bin(path, params):
    """train.py:112-119 for the 16 tensors (same bytes as formats.write_model_bin of the
    equivalent state_dict)."""
    with open(path, "wb") as f:
        for t in params:
            a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float64)
            f.write(int(a.ndim).to_bytes(4, "little"))
            for dim in a.shape:
                f.write(int(dim).to_bytes(4, "little"))
            f.write(a.tobytes())
