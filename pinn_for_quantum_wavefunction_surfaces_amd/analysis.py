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
    """w with simpson(f, x=x) == w @ f for every f (the rule is linear in f)."""
    from scipy.integrate import simpson
    x = np.asarray(x, dtype=np.float64)
    eye = np.eye(x.size)
    return np.array([simpson(eye[i], x=x) for i in range(x.size)])


def _grid(params, n_test, device):
    """linear grid of sampling(..., linearSampling=True) (poc/main.py:133-140,442-449)."""
    ax = [torch.linspace(params[k + "L"], params[k + "R"], n_test, dtype=torch.float64) for k in "xyz"]
    xg, yg, zg = torch.meshgrid(*ax, indexing="ij")
    return ax, [t.reshape(-1).to(device=device, dtype=torch.float32) for t in (xg, yg, zg)]


def _integrate(ax, f):
    """nested Simpson over z, y, x of f[n,n,n] (integra3d, poc/main.py:179-186)."""
    w = [torch.tensor(simpson_weights(a.numpy()), device=f.device, dtype=torch.float64) for a in ax]
    return float(torch.einsum("i,j,k,ijk->", w[0], w[1], w[2], f.double()))


def energy_from_psi(Ri, params, model):
    """-> (E_integral, E_net) at nuclear half-distance Ri (poc/main.py:438-464)."""
    n = int(params["n_test"])
    dev = next(model.parameters()).device
    ax, (x, y, z) = _grid(params, n, dev)
    R = torch.full_like(x, float(Ri))
    o = model.evaluate(x, y, z, R, want=("psi", "E", "res"))
    psi = o["psi"].reshape(n, n, n).double()
    hpsi = (o["res"] + o["E"] * o["psi"]).reshape(n, n, n).double()   # H psi = res + E psi
    E_int = _integrate(ax, psi * hpsi) / _integrate(ax, psi * psi)
    return E_int, float(o["E"][-1])


def energy_from_psi_LCAO(Ri, params, device=None):
    """LCAO reference curve psi = f1 + f2 (poc/main.py:467-492), closed form Laplacian; evaluated
    and integrated on `device` (default: the GPU when there is one -- a 250^3 grid in fp64 is
    no work for it -- else the CPU: this curve involves no kernel of the library)."""
    n = int(params["n_test"])
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    ax, (x, y, z) = _grid(params, n, device)
    x, y, z = x.double(), y.double(), z.double()
    r1 = torch.sqrt((x - Ri) ** 2 + y ** 2 + z ** 2)
    r2 = torch.sqrt((x + Ri) ** 2 + y ** 2 + z ** 2)
    f1, f2 = torch.exp(-r1), torch.exp(-r2)
    psi = f1 + f2
    lap = f1 * (1 - 2 / r1) + f2 * (1 - 2 / r2)
    hpsi = -0.5 * lap - (1 / r1 + 1 / r2) * psi
    shp = (n, n, n)
    return _integrate(ax, (psi * hpsi).reshape(shp)) / _integrate(ax, (psi * psi).reshape(shp))


def dEdR_int(Ri, params, model):
    """Hellmann-Feynman dE/dR at half-distance Ri (poc/main.py:639-676): <psi| dV/dR |psi> with
    the normalised network psi on the n_test^3 grid, minus 1/(2 Ri^2); psi from the fused
    forward kernel, dV/dR in closed form as in hamiltonian_R (poc/main.py:639-644)."""
    n = int(params["n_test"])
    dev = next(model.parameters()).device
    ax, (x, y, z) = _grid(params, n, dev)
    R = torch.full_like(x, float(Ri))
    psi = model.evaluate(x, y, z, R, want=("psi",))["psi"].reshape(-1).double()
    x, y, z = x.double(), y.double(), z.double()
    r1 = torch.sqrt((x - Ri) ** 2 + y ** 2 + z ** 2)
    r2 = torch.sqrt((x + Ri) ** 2 + y ** 2 + z ** 2)
    VR = -(x - Ri) / r1.pow(3) + (x + Ri) / r2.pow(3)
    shp = (n, n, n)
    norm = _integrate(ax, (psi * psi).reshape(shp))
    return _integrate(ax, (psi * VR * psi).reshape(shp)) / norm - 1 / (2 * Ri ** 2)


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
