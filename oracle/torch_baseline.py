"""PyTorch-CPU restatement of the reference's training step -- TEST/BENCH INFRASTRUCTURE ONLY.

The reference's own files cannot travel to the GPU box, so bench.py's
`cpu_baseline` leg times this instead: the same formulation the reference runs
(second derivatives by two nested torch.autograd.grad sweeps per coordinate with
create_graph=True, loss, backward, torch.optim.Adam; train.py:8-10,41-57,71-72 and
poc/main.py:82-97,341-355,403-404), written over a flat parameter vector.  It is
also cross-checked against the C oracle in tests/test_oracle_golden.py.
"""
import torch


def unpack(theta, arch):
    """split the flat state_dict-order vector into (W, b) pairs"""
    H, L, HE, HG = arch["H"], arch["L"], arch["HE"], arch["HG"]
    shapes = [(H, 2)] + [(H, H)] * (L - 1) + [(1, H), (HE, 1), (HE, HE), (1, HE), (HG, 1), (1, HG)]
    out, k = [], 0
    for (o, i) in shapes:
        W = theta[k:k + o * i].reshape(o, i); k += o * i
        b = theta[k:k + o]; k += o
        out.append((W, b))
    return out


def second_derivative(f, x):
    df, = torch.autograd.grad(f.sum(), x, create_graph=True)
    return torch.autograd.grad(df.sum(), x, create_graph=True)[0]


def loss_terms(theta, arch, ham, x, y, z, R, i1, i2):
    act = torch.sigmoid if arch.get("act", 0) == 0 else torch.tanh
    layers = unpack(theta, arch)
    L = arch["L"]
    trunk, (Wo, bo), (WE1, bE1), (WE2, bE2), (WE3, bE3), (WG1, bG1), (WG2, bG2) = \
        layers[:L], layers[L], layers[L + 1], layers[L + 2], layers[L + 3], layers[L + 4], layers[L + 5]
    r1 = torch.sqrt((x - R) ** 2 + y ** 2 + z ** 2)
    r2 = torch.sqrt((x + R) ** 2 + y ** 2 + z ** 2)
    f1, f2 = torch.exp(-r1), torch.exp(-r2)

    def base(u, v):
        h = torch.cat((u, v), 1)
        for W, b in trunk:
            h = act(h @ W.T + b)
        return h

    B = base(f1, f2)
    if arch["n_pass"] == 2:
        B = B + ham["P"] * base(f2, f1)
    N = arch["out_scale"] * (B @ Wo.T) + bo
    g = act(R @ WG1.T + bG1) @ WG2.T + bG2
    E = act(act(R @ WE1.T + bE1) @ WE2.T + bE2) @ WE3.T + bE3
    psi = N * g + f1 + ham["P"] * f2
    lap = second_derivative(psi, x) + second_derivative(psi, y) + second_derivative(psi, z)
    res = ham["a"] * lap + (ham["b"] * (1 / r1 + 1 / r2) + ham["c"] * E) * psi
    Lpde = (res ** 2).mean()
    Lbc = (psi[i1, 0] ** 2).mean() + (psi[i2, 0] ** 2).mean()
    return Lpde + Lbc, Lpde, Lbc, psi, E, lap, res


def make_step(theta0, arch, ham, lr=8e-3, dtype=torch.float64):
    """returns (theta_param, step_fn); step_fn(x,y,z,R) does loss+backward+Adam on CPU"""
    theta = theta0.clone().to(dtype).requires_grad_(True)
    opt = torch.optim.Adam([theta], lr=lr)

    def step(x, y, z, R):
        x = x.reshape(-1, 1).to(dtype).requires_grad_(True)
        y = y.reshape(-1, 1).to(dtype).requires_grad_(True)
        z = z.reshape(-1, 1).to(dtype).requires_grad_(True)
        R = R.reshape(-1, 1).to(dtype)
        with torch.no_grad():
            r1sq = (x - R) ** 2 + y ** 2 + z ** 2
            r2sq = (x + R) ** 2 + y ** 2 + z ** 2
            i1 = torch.where(r1sq[:, 0] >= 17.5 ** 2)[0]
            i2 = torch.where(r2sq[:, 0] >= 17.5 ** 2)[0]
        opt.zero_grad()
        Ltot = loss_terms(theta, arch, ham, x, y, z, R, i1, i2)[0]
        Ltot.backward()
        opt.step()
        return float(Ltot.detach())

    return theta, step


def make_step_chunked(theta0, arch, ham, lr=8e-3, dtype=torch.float64, chunk=100000):
    """the same step on a batch too large for the host's memory in one autograd graph (~40 GB at
    1 000 000 points in fp64): the boundary sets are found on the whole batch first, then the loss is accumulated
    chunk by chunk -- sum(res^2)/N + sum(psi^2[i1])/N1 + sum(psi^2[i2])/N2, one backward per chunk into the same
    .grad -- and Adam steps once.  Identical to the one-graph step in exact arithmetic."""
    theta = theta0.clone().to(dtype).requires_grad_(True)
    opt = torch.optim.Adam([theta], lr=lr)

    def step(x, y, z, R):
        x, y, z, R = [t.reshape(-1, 1).to(dtype) for t in (x, y, z, R)]
        n = x.shape[0]
        with torch.no_grad():
            m1 = ((x - R) ** 2 + y ** 2 + z ** 2)[:, 0] >= 17.5 ** 2
            m2 = ((x + R) ** 2 + y ** 2 + z ** 2)[:, 0] >= 17.5 ** 2
            n1, n2 = int(m1.sum()), int(m2.sum())
        opt.zero_grad()
        total = 0.0
        for lo in range(0, n, chunk):
            sl = slice(lo, min(lo + chunk, n))
            xc, yc, zc = [t[sl].clone().requires_grad_(True) for t in (x, y, z)]
            i1, i2 = torch.where(m1[sl])[0], torch.where(m2[sl])[0]
            _, _, _, psi, _, _, res = loss_terms(theta, arch, ham, xc, yc, zc, R[sl], i1, i2)
            part = (res ** 2).sum() / n
            if n1:
                part = part + (psi[i1, 0] ** 2).sum() / n1
            if n2:
                part = part + (psi[i2, 0] ** 2).sum() / n2
            part.backward()
            total += float(part.detach())
        opt.step()
        return total

    return theta, step
