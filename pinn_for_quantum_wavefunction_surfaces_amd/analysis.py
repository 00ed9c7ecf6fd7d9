"""Energy of the trained wavefunction by quadrature -- the callers adjacent to the hot path.

Mirrors poc/main.py:179-186 (integra3d), 438-464 (energy_from_psi), 467-492
(energy_from_psi_LCAO) and 495-517 (calculate_E_R): E(R) = <psi|H|psi> / <psi|psi> on an
n_test^3 grid with nested composite-Simpson quadrature.  psi and H psi come from ONE fused
forward launch (pinn_forward: no autograd graph, so the reference's ">32 GB for 250^3 points"
limit, poc/main.py:1461, does not apply); the quadrature is a weighted sum on the device.

Simpson weights are taken from scipy itself: integrating the unit vectors gives the exact
weight vector of whatever rule the installed scipy.integrate.simpson applies (it changed its
even-point-count handling in SciPy 1.11; the reference's `simps` is gone since 1.14).
"""
import numpy as np
import torch

from .model import NN_ion


def simpson_weights(x):
    """w with simpson(f, x=x) == w @ f for every f (the rule is linear in f): one call on the identity."""
    from scipy.integrate import simpson
    x = np.asarray(x, dtype=np.float64)
    return np.asarray(simpson(np.eye(x.size), x=x, axis=1), dtype=np.float64)


_GRID_CACHE = {}


def clear_grid_cache():
    _GRID_CACHE.clear()


def _grid(params, n_test, device, points=True):
    """linear grid of sampling(..., linearSampling=True) (poc/main.py:133-140,442-449), built on `device`.
    Cached per (box, n_test, device): only what is O(n_test) -- the three axes (fp64, host) and the three Simpson
    weight vectors (fp64, device; scipy's rule costs a host round trip, and calculate_E_R asks for the same grid 39
    times).  The flattened fp32 coordinates (3 x 62 MB at n_test = 250) are rebuilt on the device per call: a
    meshgrid is no work for it, the cache pins no bulk memory, and no caller can corrupt a later call's points."""
    key = (tuple(float(params[k + s]) for k in "xyz" for s in "LR"), int(n_test), str(device))
    hit = _GRID_CACHE.get(key)
    if hit is None:
        ax = [torch.linspace(params[k + "L"], params[k + "R"], n_test, dtype=torch.float64) for k in "xyz"]
        w = [torch.tensor(simpson_weights(a.numpy()), device=device, dtype=torch.float64) for a in ax]
        if len(_GRID_CACHE) >= 8:
            _GRID_CACHE.clear()
        hit = _GRID_CACHE[key] = (ax, w)
    ax, w = hit
    if not points:
        return ax, w, None
    xg, yg, zg = torch.meshgrid(*[a.to(device) for a in ax], indexing="ij")
    pts = [t.reshape(-1).to(torch.float32) for t in (xg, yg, zg)]
    return ax, w, pts


def _integrate(model, w, u, v=None):
    """nested Simpson over z, y, x of u * v on the grid (integra3d, poc/main.py:179-186): one pinn_quadrature3d
    launch pair on the device (fp64 accumulation, fixed order); u, v: fp32 per-point tensors (v None: u^2)."""
    dev = u.device
    h = model._handle_for(dev, u.numel())
    scratch = torch.empty(w[0].numel(), device=dev, dtype=torch.float64)
    out = torch.empty(1, device=dev, dtype=torch.float64)
    h.quadrature3d(u.contiguous(), None if v is None else v.contiguous(), w[0], w[1], w[2], scratch, out)
    return out


def energy_from_psi(Ri, params, model):
    """-> (E_integral, E_net) at nuclear half-distance Ri (poc/main.py:438-464).  Everything stays on the device:
    grid, fused forward (psi, E, residual in one launch), H psi = res + E psi, the two quadratures; one host read
    at the end."""
    n = int(params["n_test"])
    dev = next(model.parameters()).device
    _, w, (x, y, z) = _grid(params, n, dev)
    R = torch.full_like(x, float(Ri))
    o = model.evaluate(x, y, z, R, want=("psi", "E", "res"))
    psi = o["psi"].reshape(-1)
    hpsi = (o["res"] + o["E"] * o["psi"]).reshape(-1)                 # H psi = res + E psi
    num, den = _integrate(model, w, psi, hpsi), _integrate(model, w, psi)
    return float(num / den), float(o["E"][-1])


def _integrate64(w, f):
    """the same nested rule on an fp64 grid function f[n, n, n] that no kernel of the library produced (the LCAO
    closed form, the Hellmann-Feynman integrand with its fp64 geometry factor): a weighted sum on the device"""
    return torch.dot(torch.mv(torch.matmul(f, w[2]), w[1]), w[0])     # innermost axis first, as integra3d nests them


def energy_from_psi_LCAO(Ri, params, device=None):
    """LCAO reference curve psi = f1 + f2 (poc/main.py:467-492), closed form Laplacian; evaluated
    and integrated on `device` (default: the GPU when there is one -- a 250^3 grid in fp64 is
    no work for it -- else the CPU: this curve involves no kernel of the library)."""
    n = int(params["n_test"])
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    ax, w, _ = _grid(params, n, device, points=False)
    xg, yg, zg = torch.meshgrid(*[a.to(device) for a in ax], indexing="ij")   # fp64: this curve is a closed form
    r1 = torch.sqrt((xg - Ri) ** 2 + yg ** 2 + zg ** 2)
    r2 = torch.sqrt((xg + Ri) ** 2 + yg ** 2 + zg ** 2)
    f1, f2 = torch.exp(-r1), torch.exp(-r2)
    psi = f1 + f2
    lap = f1 * (1 - 2 / r1) + f2 * (1 - 2 / r2)
    hpsi = -0.5 * lap - (1 / r1 + 1 / r2) * psi
    return float(_integrate64(w, psi * hpsi) / _integrate64(w, psi * psi))


def dEdR_int(Ri, params, model):
    """Hellmann-Feynman dE/dR at half-distance Ri (poc/main.py:639-676): <psi| dV/dR |psi> with
    the normalised network psi on the n_test^3 grid, minus 1/(2 Ri^2); psi from the fused
    forward kernel, dV/dR in closed form as in hamiltonian_R (poc/main.py:639-644)."""
    n = int(params["n_test"])
    dev = next(model.parameters()).device
    ax, w, (x, y, z) = _grid(params, n, dev)
    R = torch.full_like(x, float(Ri))
    psi = model.evaluate(x, y, z, R, want=("psi",))["psi"].reshape(-1)
    xg, yg, zg = [t.reshape(-1) for t in torch.meshgrid(*[a.to(dev) for a in ax], indexing="ij")]   # fp64 geometry
    r1 = torch.sqrt((xg - Ri) ** 2 + yg ** 2 + zg ** 2)
    r2 = torch.sqrt((xg + Ri) ** 2 + yg ** 2 + zg ** 2)
    VR = -(xg - Ri) / r1.pow(3) + (xg + Ri) / r2.pow(3)
    shp = (n, n, n)
    psid = psi.double()
    norm = _integrate64(w, (psid * psid).reshape(shp))
    return float(_integrate64(w, (psid * VR * psid).reshape(shp)) / norm) - 1 / (2 * Ri ** 2)


def force_curves(model, R):
    """-> (F, dF/dR) with F = -d(E_net + 1/(2R))/dR, the auto-diff force of
    poc/main.py:1316-1328, through NN_ion.forward's R-differentiable E."""
    R = R.detach().reshape(-1, 1).clone().requires_grad_(True)
    zero = torch.zeros_like(R)
    _, E = model(zero + 0.3, zero, zero, R)
    ET = E + 1 / (2 * R)
    ones = torch.ones_like(R)
    F = -torch.autograd.grad([ET], [R], grad_outputs=ones, create_graph=True)[0]
    F2 = torch.autograd.grad([F], [R], grad_outputs=ones)[0]
    return F.detach(), F2.detach()


def psi_cut(model, R0, n=1000, x_max=10.0):
    """psi and the LCAO guess f1 + f2 along the x axis (y = z = 0) at half-distance R0: the
    numbers behind plot.py:34-58 (its model.bin evaluation) and psiX_norm (poc/main.py:628-635)."""
    dev = next(model.parameters()).device
    x = torch.linspace(-x_max, x_max, n, dtype=torch.float64, device=dev)
    zero = torch.zeros_like(x)
    R = torch.full_like(x, float(R0))
    psi = model.evaluate(x, zero, zero, R, want=("psi",))["psi"].reshape(-1)
    lcao = torch.exp(-(x - R0).abs()) + torch.exp(-(x + R0).abs())
    return x, psi, lcao


def calculate_E_R(params, model=None):
    """E(R) table on R = RxL .. RxR step 0.1 (poc/main.py:495-517); returns the same dict."""
    if model is None:
        model = NN_ion(params)
        model.loadModel(params)
        model.to("cuda")
    Rs = np.round(np.arange(params["RxL"], params["RxR"] + .1, .1), 2)
    out = {"R": Rs, "E_int": np.zeros(len(Rs)), "Elcao": np.zeros(len(Rs)), "E_net": np.zeros(len(Rs))}
    for i, Ri in enumerate(Rs):
        out["E_int"][i], out["E_net"][i] = energy_from_psi(float(Ri), params, model)
        out["Elcao"][i] = energy_from_psi_LCAO(float(Ri), params, device=next(model.parameters()).device)
    return out
