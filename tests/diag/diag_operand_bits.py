#!/usr/bin/env python3
"""diag_operand_bits (CPU, numbers only) -- which GEMM operand's precision sets the gradient error of a split-bf16 trunk
on the reference-generated deep128 fixture (4x128 tanh, 12 hand-placed points next to the nuclei)?

The stream formulation of the kernels (value, three Cartesian derivatives, Laplacian: SURVEY A.3) in torch fp64, with every
operand of the dense H x H products rounded to a chosen number of bf16 pieces (1 piece = 8 mantissa bits, 2 = 16, 3 = 24)
-- separately for the forward product (W, X), the input-adjoint product (W^T, dY) and the weight-gradient product (dY, X^T).
Everything else (orbitals, activations, tail, the 2-column first layer, sums) stays fp64, so what is printed is the effect of
the operand rounding alone, against the fixture's fp64 gradient (the reference's own autograd).
    python tests/diag/diag_operand_bits.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
INF = 99


def pieces(t, k):
    if k >= INF:
        return t
    acc, r = torch.zeros_like(t), t.clone()
    for _ in range(k):
        p = r.float().to(torch.bfloat16).double()
        acc = acc + p
        r = r - p
    return acc


class RoundedLinear(torch.autograd.Function):
    """Y = X W^T with rounded operands; cfg = (fw, fx, bw, by, gy, gx) pieces"""
    @staticmethod
    def forward(ctx, X, W, cfg):
        ctx.cfg = cfg
        ctx.save_for_backward(X, W)
        return pieces(X, cfg[1]) @ pieces(W, cfg[0]).t()

    @staticmethod
    def backward(ctx, dY):
        X, W = ctx.saved_tensors
        c = ctx.cfg
        dX = pieces(dY, c[3]) @ pieces(W, c[2])
        dW = pieces(dY, c[4]).t() @ pieces(X, c[5])
        return dX, dW, None


def loss_grad(g, cfg, sl=slice(None), want_fields=False):
    L, H = 4, 128
    th = torch.tensor(g["theta"], dtype=torch.float64)
    shapes = [(H, 2), (H,)] + [(H, H), (H,)] * (L - 1) + [(1, H), (1,), (32, 1), (32,), (32, 32), (32,), (1, 32), (1,),
                                                          (10, 1), (10,), (1, 10), (1,)]
    ps, o = [], 0
    for s in shapes:
        n = int(np.prod(s))
        ps.append(th[o:o + n].reshape(s).clone().requires_grad_(True))
        o += n
    assert o == th.numel()
    x, y, z, R = [torch.tensor(g[k][sl], dtype=torch.float64).reshape(-1, 1) for k in ("x", "y", "z", "R")]
    mask = torch.tensor(g["mask"][sl].astype(np.int64)).reshape(-1, 1)
    n = x.shape[0]
    act, d1 = torch.tanh, lambda h: 1 - h * h

    def orbital(dx):
        r = torch.sqrt(dx * dx + y * y + z * z)
        f = torch.exp(-r)
        return [f, -f * dx / r, -f * y / r, -f * z / r, f * (1 - 2 / r)], r

    o1, r1 = orbital(x - R)
    o2, r2 = orbital(x + R)

    def base(ia, ib):   # streams of the two inputs -> last hidden layer's (value, lap) streams
        W0, b0 = ps[0], ps[1]
        a = [ia[s] * W0[:, 0] + ib[s] * W0[:, 1] for s in range(5)]   # (n, H) each; K = 2: not a matrix-pipe product
        a[0] = a[0] + b0
        for l in range(L):
            h = act(a[0])
            s1 = d1(h)
            s2 = -2 * h * s1
            q = a[1] ** 2 + a[2] ** 2 + a[3] ** 2
            hs = [h, s1 * a[1], s1 * a[2], s1 * a[3], s1 * a[4] + s2 * q]
            if l == L - 1:
                return hs
            W, b = ps[2 + 2 * l], ps[3 + 2 * l]
            out = RoundedLinear.apply(torch.cat(hs, 0), W, cfg)
            a = list(out.split(n, 0))
            a[0] = a[0] + b

    hA, hB = base(o1, o2), base(o2, o1)
    wo, bo = ps[2 * L], ps[2 * L + 1]
    N = (hA[0] + hB[0]) @ wo.t() + bo
    lapN = (hA[4] + hB[4]) @ wo.t()
    sig = torch.tanh
    E = sig(sig(R @ ps[2 * L + 2].t() + ps[2 * L + 3]) @ ps[2 * L + 4].t() + ps[2 * L + 5]) @ ps[2 * L + 6].t() + ps[2 * L + 7]
    gate = sig(R @ ps[2 * L + 8].t() + ps[2 * L + 9]) @ ps[2 * L + 10].t() + ps[2 * L + 11]
    psi = gate * N + o1[0] + o2[0]
    lap = gate * lapN + o1[4] + o2[4]
    res = -0.5 * lap - (1 / r1 + 1 / r2) * psi - E * psi
    m1, m2 = (mask & 1).double(), ((mask >> 1) & 1).double()
    Ltot = (res ** 2).mean() + (psi ** 2 * m1).sum() / m1.sum().clamp(min=1) + (psi ** 2 * m2).sum() / m2.sum().clamp(min=1)
    Ltot.backward()
    grad = np.concatenate([p.grad.numpy().ravel() for p in ps])
    if want_fields:
        f = lambda t: t.detach().numpy().ravel()
        return float(Ltot.detach()), grad, dict(psi=f(psi), lap=f(lap), res=f(res))
    return float(Ltot.detach()), grad


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "g2_formA_deep128.npz"))
    ref = g["grad"]
    rel = lambda a: np.abs(a - ref).max() / np.abs(ref).max()
    L0, g0 = loss_grad(g, (INF,) * 6)
    print("fp64 stream formulation vs the fixture: loss %.1e, grad %.1e (the formulation itself)" % (abs(L0 / g["Ltot"] - 1), rel(g0)))
    print("fp32 torch run of the reference on this batch (fixture): grad %.3e" % rel(g["fp32_grad"]))
    rows = [("all operands 2 pieces (what the split-bf16 kernels carry)", (2,) * 6),
            ("all operands 3 pieces", (3,) * 6),
            ("only forward W at 2", (2, INF, INF, INF, INF, INF)),
            ("only forward X at 2", (INF, 2, INF, INF, INF, INF)),
            ("only input-adjoint W^T at 2", (INF, INF, 2, INF, INF, INF)),
            ("only input-adjoint dY at 2", (INF, INF, INF, 2, INF, INF)),
            ("only weight-gradient dY at 2", (INF, INF, INF, INF, 2, INF)),
            ("only weight-gradient X^T at 2", (INF, INF, INF, INF, INF, 2)),
            ("forward at 3, reverse at 2", (3, 3, 2, 2, 2, 2)),
            ("forward at 2, reverse at 3", (2, 2, 3, 3, 3, 3)),
            ("weights at 3, activations/adjoints at 2", (3, 2, 3, 2, 2, 2)),
            ("weights at 2, activations/adjoints at 3", (2, 3, 2, 3, 3, 3)),
            ("all operands 1 piece (plain bf16)", (1,) * 6)]
    for name, cfg in rows:
        Lc, gc = loss_grad(g, cfg)
        print("%-58s grad %.3e of max|grad|   loss %.1e" % (name, rel(gc), abs(Lc / g["Ltot"] - 1)))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
