"""NN_ion-compatible module whose evaluation runs on the fused HIP kernels.

Mirrors the object surface of the reference (poc/main.py): constructor
arguments, sub-module names (= state_dict keys, so models/ionHsym.pt loads),
forward / parametricPsi / LossFunctions / loadModel / saveModel / freezeBase /
freezeDecayUnit, plus the free functions set_params, exactE, radial, V, sampling.
Nothing here computes psi with torch ops: forward and LossFunctions call the
C ABI (include/pinn.h) and raise if the HIP library or a GPU is missing.
"""

import numpy as np
import torch
import torch.nn as nn

from . import capi


def set_params():
    """Same keys and defaults as poc/main.py:16-45."""
    params = dict()
    boundaries = 18
    for ax in "xyz":
        params[ax + "L"] = -boundaries
        params[ax + "R"] = boundaries
    params["BCcutoff"] = 17.5
    params["RxL"] = 0.2
    params["RxR"] = 4
    params["Ry"] = 0
    params["Rz"] = 0
    params["cutOff"] = 0.005
    params["lossPath"] = "data/loss_ionH.pkl"
    params["EnergyPath"] = "data/energy_ionH.pkl"
    params["saveModelPath"] = "models/ionHsym.pt"
    params["loadModelPath"] = "models/ionHsym.pt"
    params["EnrR_path"] = "data/energy_R_ion.pkl"
    params["sc_step"] = 3000
    params["sc_decay"] = .7
    params["sc_sampling"] = 1
    params["n_train"] = 100000
    params["n_test"] = 80
    params["epochs"] = int(5e3)
    params["lr"] = 8e-3
    params["inversion_symmetry"] = 1
    return params


def radial(x, y, z, R, params):
    """poc/main.py:101-108 (plain tensor expression; not on the fused path)."""
    r1 = torch.sqrt((x - R).pow(2) + (y - params["Ry"]).pow(2) + (z - params["Rz"]).pow(2))
    r2 = torch.sqrt((x + R).pow(2) + (y + params["Ry"]).pow(2) + (z + params["Rz"]).pow(2))
    return r1, r2


def V(x, y, z, R, params):
    """poc/main.py:111-115."""
    r1, r2 = radial(x, y, z, R, params)
    return -1 / r1 - 1 / r2


class toR(nn.Module):
    """Parameter-free child kept so that state_dict()._metadata matches (poc/main.py:64-71)."""

    @staticmethod
    def forward(input):
        return torch.sqrt(input.pow(2).sum(1)).reshape(-1, 1)


class atomicAct_s(nn.Module):
    """poc/main.py:74-78."""

    @staticmethod
    def forward(input):
        return torch.exp(-input)


FORMULATION_A = dict(a=-0.5, b=-1.0, c=-1.0, n_pass=2, out_scale=1.0)   # poc/main.py:118-120,345
FORMULATION_B = dict(a=1.0, b=1.0, c=1.0, n_pass=1, out_scale=2.0)      # train.py:47,54


class _FusedLoss(torch.autograd.Function):
    """Ltot with a backward that hands out the gradients the fused kernel already computed."""

    @staticmethod
    def forward(ctx, model, x, y, z, R, mask, *plist):
        out = model._fused_loss_grad(x, y, z, R, mask)
        P = model._handle_for(x.device, x.numel()).P
        ctx.model = model
        ctx.grad_flat = out[:P]
        ctx.meta = [(p.shape, p.dtype, p.requires_grad) for p in plist]
        Lpde = out[P]
        Lbc = out[P + 1] + out[P + 2]
        return (Lpde + Lbc).double(), Lpde.double(), Lbc.double()

    @staticmethod
    def backward(ctx, gtot, gpde, gbc):
        scaled = ctx.grad_flat * gtot.to(ctx.grad_flat.dtype)       # one multiply, then views per parameter
        grads, k = [], 0
        for shape, dtype, req in ctx.meta:
            n = int(np.prod(shape))
            grads.append(scaled[k:k + n].to(dtype).reshape(shape) if req else None)
            k += n
        return (None,) * 6 + tuple(grads)


class _EnergyOfR(torch.autograd.Function):
    """E(R) of the E-net, differentiable twice w.r.t. R the way the reference's callers use it:
    grad([E + 1/(2R)], [R], create_graph=True) for the force and once more for its slope
    (poc/main.py:1316-1328).  Every level is one pinn_energy_R launch (value, d/dR, d2/dR2 as
    three forward streams); nothing is differentiated numerically or on the host."""

    @staticmethod
    def forward(ctx, model, R):
        ctx.model = model
        ctx.save_for_backward(R)
        return model.energy_R(R)[0]

    @staticmethod
    def backward(ctx, gE):
        (R,) = ctx.saved_tensors
        return None, gE * _dEdR.apply(ctx.model, R)


class _dEdR(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, R):
        _, dE, d2E = model.energy_R(R)
        ctx.save_for_backward(d2E)
        return dE

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (d2E,) = ctx.saved_tensors
        return None, g * d2E


class NN_ion(nn.Module):
    """Drop-in for poc/main.py:223-355.  `n_hidden`, `act`, `formulation` and `precision`
    are build extensions with the reference's values as defaults."""

    def __init__(self, params, dense_neurons=16, dense_neurons_E=32, netDecay_neurons=10,
                 n_hidden=2, act="sigmoid", formulation="A", precision="f32"):
        super(NN_ion, self).__init__()
        self.sig = nn.Sigmoid()
        self.toR = toR()
        self.actAO_s = atomicAct_s()
        self.Lin_H1 = nn.Linear(2, dense_neurons)
        self.Lin_H2 = nn.Linear(dense_neurons, dense_neurons, bias=True)
        for k in range(3, n_hidden + 1):
            setattr(self, "Lin_H%d" % k, nn.Linear(dense_neurons, dense_neurons, bias=True))
        self.Lin_out = nn.Linear(dense_neurons, 1)
        self.Lin_E1 = nn.Linear(1, dense_neurons_E)
        self.Lin_E2 = nn.Linear(dense_neurons_E, dense_neurons_E)
        self.Lin_Eout = nn.Linear(dense_neurons_E, 1)
        nn.init.constant_(self.Lin_Eout.bias[0], -1)          # poc/main.py:240
        self.Ry = params["Ry"]
        self.Rz = params["Rz"]
        self.P = params["inversion_symmetry"]
        self.netDecayL = nn.Linear(1, netDecay_neurons, bias=True)
        self.netDecay = nn.Linear(netDecay_neurons, 1, bias=True)
        form = FORMULATION_A if formulation == "A" else FORMULATION_B
        self._form = dict(form)
        self._arch = capi.Arch(dense_neurons, n_hidden, dense_neurons_E, netDecay_neurons,
                               {"sigmoid": 0, "tanh": 1}[act], form["n_pass"], form["out_scale"],
                               {"f32": 0, "bf16": 1, "bf16x3": 2}[precision])
        self._handles = {}

    # ---- plumbing -------------------------------------------------------------
    @property
    def arch(self):
        return self._arch

    def trainable_mask(self):
        m = 0
        for t, p in enumerate(self.parameters()):
            if p.requires_grad:
                m |= 1 << t
        return m

    def flat_params(self, device=None, dtype=torch.float32):
        """theta in state_dict order (include/pinn.h), as one contiguous tensor."""
        ps = [p.detach().reshape(-1).to(dtype) for p in self.parameters()]
        th = torch.cat(ps)
        return th.to(device) if device is not None else th

    def load_flat_params(self, theta):
        k = 0
        with torch.no_grad():
            for p in self.parameters():
                n = p.numel()
                p.copy_(theta[k:k + n].reshape(p.shape).to(p.dtype))
                k += n

    def _handle_for(self, device, n):
        if device.type != "cuda":
            raise capi.PinnError("NN_ion evaluates on the GPU only (got %s tensors): the fused "
                                 "HIP path has no CPU fallback" % device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        h = self._handles.get(idx)
        if h is None or h.max_points < n:
            # a larger workspace is a NEW handle; the old one is only let go of here, never closed: a Trainer (or any
            # other caller) may still hold it, and it is destroyed when its last reference goes (Handle.__del__)
            h = capi.Handle(self._arch, idx, max(int(n), 4096))
            self._handles[idx] = h
        return h

    def ham(self, invN=0.0, invN1=0.0, invN2=0.0):
        f = self._form
        return capi.Ham(f["a"], f["b"], f["c"], float(self.P), float(self.Ry), float(self.Rz),
                        invN, invN1, invN2)

    @staticmethod
    def _f32(t):
        return t.detach().reshape(-1).to(torch.float32).contiguous()

    def evaluate(self, x, y, z, R, want=("psi", "E")):
        """Fused forward: any of psi, E, lap, res as (n,1) tensors in x's dtype."""
        n = x.numel()
        h = self._handle_for(x.device, n)
        theta = self.flat_params(x.device)
        xs = [self._f32(t) for t in (x, y, z, R)]
        outs = {k: torch.empty(n, device=x.device, dtype=torch.float32) for k in want}
        h.forward(theta, *xs, n, self.ham(), psi=outs.get("psi"), E=outs.get("E"),
                  lap=outs.get("lap"), res=outs.get("res"))
        return {k: v.reshape(-1, 1).to(x.dtype) for k, v in outs.items()}

    def _fused_loss_grad(self, x, y, z, R, mask):
        n = x.numel()
        h = self._handle_for(x.device, n)
        theta = self.flat_params(x.device)
        xs = [self._f32(t) for t in (x, y, z, R)]
        # boundary-set sizes stay on the device (bc_counts of pinn_residual_fwd_bwd): no host read per call
        counts = torch.stack([(mask & 1).sum(), ((mask >> 1) & 1).sum()]).to(torch.int64)
        out = torch.empty(h.P + 3, device=x.device, dtype=torch.float32)
        h.residual_fwd_bwd(theta, *xs, mask, n, self.ham(1.0 / n, 0.0, 0.0), counts, self.trainable_mask(), out)
        return out

    def energy_R(self, R):
        """(E, dE/dR, d2E/dR2), each shaped and typed like R, from one pinn_energy_R launch."""
        n = R.numel()
        h = self._handle_for(R.device, n)
        out = [torch.empty(n, device=R.device, dtype=torch.float32) for _ in range(3)]
        h.energy_R(self.flat_params(R.device), self._f32(R), n, *out)
        return tuple(o.reshape(R.shape).to(R.dtype) for o in out)

    # ---- reference surface ------------------------------------------------------
    def forward(self, x, y, z, R):
        """-> (Nout, E), each (n,1): poc/main.py:247-267.  When R requires grad, E carries a
        graph to R (twice differentiable: the force curves of poc/main.py:1316-1328); psi is
        a plain value (the reference never differentiates it outside the loss)."""
        if torch.is_grad_enabled() and R.requires_grad:
            psi = self.evaluate(x, y, z, R, want=("psi",))["psi"]
            return psi, _EnergyOfR.apply(self, R.reshape(-1, 1))
        o = self.evaluate(x, y, z, R, want=("psi", "E"))
        return o["psi"], o["E"]

    def parametricPsi(self, x, y, z, R):
        N, E = self.forward(x, y, z, R)
        return N, E

    def hamiltonian(self, x, y, z, R):
        """H psi = -1/2 lap psi + V psi (poc/main.py:118-120), from the fused forward."""
        o = self.evaluate(x, y, z, R, want=("psi", "E", "res"))
        return o["res"] + o["E"] * o["psi"]

    def LossFunctions(self, x, y, z, R, params, bIndex1, bIndex2):
        """-> (Ltot, LossPDE, Lbc, E): poc/main.py:341-355.  Ltot.backward() fills .grad of the
        trainable parameters with what the fused kernel computed in the same pass."""
        n = x.numel()
        mask = torch.zeros(n, dtype=torch.uint8, device=x.device)
        i1 = bIndex1[0] if isinstance(bIndex1, (tuple, list)) else bIndex1
        i2 = bIndex2[0] if isinstance(bIndex2, (tuple, list)) else bIndex2
        mask[i1] |= 1
        mask[i2] |= 2
        plist = list(self.parameters())
        Ltot, Lpde, Lbc = _FusedLoss.apply(self, x, y, z, R, mask, *plist)
        E = self.evaluate(x, y, z, R, want=("E",))["E"]
        return Ltot, Lpde, Lbc, E

    def freezeBase(self):
        """poc/main.py:305-313."""
        for k in [n for n, _ in self.named_children() if n.startswith("Lin_H")] + ["Lin_out"]:
            getattr(self, k).weight.requires_grad = False
            getattr(self, k).bias.requires_grad = False

    def freezeDecayUnit(self):
        """poc/main.py:315-319."""
        for k in ("netDecayL", "netDecay"):
            getattr(self, k).weight.requires_grad = False
            getattr(self, k).bias.requires_grad = False

    def loadModel(self, params):
        """poc/main.py:325-329; weights_only=True: nothing in the file is executed."""
        checkpoint = torch.load(params["loadModelPath"], map_location=torch.device("cpu"),
                                weights_only=True)
        self.load_state_dict(checkpoint["model_state_dict"])
        self.eval()

    def saveModel(self, params, optimizer):
        """poc/main.py:331-339."""
        torch.save({"model_state_dict": self.state_dict(),
                    "optimizer_state_dict": optimizer.state_dict()}, params["saveModelPath"])


def sampling(params, n_points, linearSampling=False, device="cuda", seed=12345, step=0, model=None):
    """poc/main.py:124-156.  Random points come from the on-device Philox sampler
    (pinn_sample), not torch's CPU mt19937 stream; same distribution, clamp and
    shapes: four (n,1) tensors."""
    dev = torch.device(device)
    if linearSampling:
        x = torch.linspace(params["xL"], params["xR"], n_points, device=dev)
        y = torch.linspace(params["yL"], params["yR"], n_points, device=dev)
        z = torch.linspace(params["zL"], params["zR"], n_points, device=dev)
        R = torch.linspace(params["RxL"], params["RxR"], n_points, device=dev)
        r1, r2 = radial(x, y, z, R, params)
        x[r1 < params["cutOff"]] = params["cutOff"]
        x[r2 < params["cutOff"]] = params["cutOff"]
    else:
        if dev.type != "cuda":
            raise capi.PinnError("sampling() draws on the GPU (pinn_sample); no CPU fallback")
        from .trainer import check_cubic_box
        check_cubic_box(params)
        m = model if model is not None else NN_ion(params)
        h = m._handle_for(dev, n_points)
        x, y, z, R = [torch.empty(n_points, device=dev, dtype=torch.float32) for _ in range(4)]
        h.sample(seed, step, 0, n_points, float(params["xR"]), float(params["RxL"]),
                 float(params["RxR"]), float(params["cutOff"]), float(params["BCcutoff"]),
                 x, y, z, R, None, None)
    return tuple(t.reshape(-1, 1) for t in (x, y, z, R))


def exactE():
    """Wind, J. Chem. Phys. 42, 2371 (1965) table, as tabulated in poc/main.py:48-61."""
    R_exact = np.round(np.arange(0.2, 4.1, .1), 2)
    e_exact = [-1.8008, -1.6715, -1.5545, -1.4518, -1.3623, -1.2843, -1.2159, -1.1558, -1.1026,
               -1.0554, -1.0132, -0.9754, -0.9415, -0.9109, -0.8832, -0.8582, -0.8355, -0.8149,
               -0.7961, -0.7790, -0.7634, -0.7492, -0.7363, -0.7244, -0.7136, -0.7037, -0.6946,
               -0.6863, -0.6786, -0.6716, -0.6651, -0.6591, -0.6536, -0.6485, -0.6437, -0.6392,
               -0.6351, -0.6312, -0.6276]
    return R_exact, e_exact
