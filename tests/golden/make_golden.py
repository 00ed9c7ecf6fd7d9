#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py

Nothing from the reference is copied into this repository: the script loads the
reference's own sources at run time (poc/main.py definitions, train.py's loss
statements and serializer, energy.py), feeds them fixed inputs and stores only
numbers (inputs + outputs) as .npz files.

What is produced:
  g1_weights.npz   both checkpoints as flat fp64 vectors (state_dict order)
  g2_formA_*.npz   NN_ion (poc/main.py) on a fixed batch: psi,E,lap,Hpsi,res,losses,grads
                   (tanh64 / tanh16: the same module with its `sig` handle set to nn.Tanh())
                   (tanh64, deep128 also carry fp32_*: the same module run in fp32 torch on the same batch)
                   (deep128 / deep16: NN_ion with Lin_H3(, Lin_H4) added the way the reference's own
                   commented-out line poc/main.py:302 does -- only `base` is overridden; forward,
                   LossFunctions, lapl and the third-order autograd are the reference's)
  g3_formB.npz     train.py formulation on a fixed batch (its own statements, its `d`)
  g3_formB_tanh64.npz  the same statements with nh = 64 and `torch.sigmoid` resolving to tanh (BASELINE config 0)
  g4_adam_*.npz    10 torch.optim.Adam steps on stored batches (full and frozen trunk)
  g5_energy.npz    poc/energy_R_ion.pkl table (parsed opcode-by-opcode, never unpickled)
                   + Wind-1965 table from poc/main.py:exactE
  g6_modelbin.npz  model.bin bytes written by train.py's serializer for the fine-tune
                   checkpoint + energy.py's e(R) evaluated from that file
  g7_force.npz     callers next to the path: E(R), dE/dR, d2E/dR2 and the force curves by the
                   reference's autograd statements (poc/main.py:1313-1332), the Hellmann-Feynman
                   integral dEdR_int (poc/main.py:639-676), plot.py's psi cut from model.bin

    python tests/golden/make_golden.py g7      # only the fixtures whose name starts with g7
"""
import ast
import hashlib
import os
import pickletools
import sys
import tempfile

import numpy as np
import torch

REF = os.environ.get("PINN_REFERENCE", "/root/reference")
OUT = os.environ.get("PINN_GOLDEN_OUT", os.path.dirname(os.path.abspath(__file__)))

STATE_KEYS = ["Lin_H1", "Lin_H2", "Lin_out", "Lin_E1", "Lin_E2", "Lin_Eout", "netDecayL",
              "netDecay"]


# ----------------------------------------------------------------- loaders
def load_main_defs():
    """exec the definitions (imports, defs, classes) of poc/main.py; skip its script body."""
    src = open(os.path.join(REF, "poc", "main.py")).read()
    tree = ast.parse(src)
    keep = []
    seen = set()
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            if isinstance(node, ast.ImportFrom) and node.module == "scipy.integrate":
                continue  # `simps` was removed from SciPy >= 1.14 (poc/main.py:12)
            keep.append(node)
        elif isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            if node.name in seen:
                continue  # later duplicates are stale
            seen.add(node.name)
            keep.append(node)
    os.environ.setdefault("MPLBACKEND", "Agg")
    ns = {"__name__": "ref_main"}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "poc/main.py", "exec"), ns)
    import scipy.integrate
    ns["simps"] = lambda yv, xv: scipy.integrate.simpson(yv, x=xv)
    ns["dtype"] = torch.double            # poc/main.py:914
    ns["params"] = ns["set_params"]()     # poc/main.py:918
    return ns


def load_train_parts():
    """train.py: helper defs, the loss statements of train()'s loop, and the serializer."""
    src = open(os.path.join(REF, "train.py")).read()
    tree = ast.parse(src)
    defs, writer, loss_stmts = [], None, []
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            defs.append(node)
        elif isinstance(node, ast.FunctionDef):
            if node.name == "train":
                loop = [n for n in node.body if isinstance(n, ast.While)][0]
                started = False
                for st in loop.body:
                    if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
                        name = st.targets[0].id
                        if name == "r1":
                            started = True
                        if started:
                            loss_stmts.append(st)
                        if name == "Ltot":
                            break
            else:
                defs.append(node)
        elif isinstance(node, ast.With):
            writer = node
    ns = {"__name__": "ref_train"}
    exec(compile(ast.Module(body=defs, type_ignores=[]), "train.py", "exec"), ns)
    loss_code = compile(ast.Module(body=loss_stmts, type_ignores=[]), "train.py:41-57", "exec")
    writer_code = compile(ast.Module(body=[writer], type_ignores=[]), "train.py:112-119", "exec")
    return ns, loss_code, writer_code


def flat_state(sd):
    return np.concatenate([sd[k + s].detach().cpu().numpy().ravel()
                           for k in STATE_KEYS for s in (".weight", ".bias")])


def sha256(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def fixed_batch(n, seed, Rhi):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-18, 18, n)
    y = rng.uniform(-18, 18, n)
    z = rng.uniform(-18, 18, n)
    R = rng.uniform(0.2, Rhi, n)
    # hand-placed hard points: close to a nucleus, the clamp value, far corners, axes
    k = 0
    for dr in (0.012, 0.03, 0.05, 0.2):
        x[k], y[k], z[k] = R[k] + dr * 0.6, dr * 0.64, dr * 0.48
        k += 1
        x[k], y[k], z[k] = -R[k] - dr * 0.6, -dr * 0.64, dr * 0.48
        k += 1
    x[k], y[k], z[k] = 0.005, 0.0, 0.0        # what the clamp writes (poc/main.py:148)
    k += 1
    x[k], y[k], z[k] = 0.0, 0.0, 0.0
    k += 1
    x[k], y[k], z[k] = 17.9, -17.9, 17.9
    k += 1
    x[k], y[k], z[k] = -17.99, 0.3, -0.2
    k += 1
    return x, y, z, R


def as_col(a, grad=False):
    t = torch.tensor(np.asarray(a, dtype=np.float64).reshape(-1, 1), dtype=torch.float64)
    t.requires_grad_(grad)
    return t


# ------------------------------------------------------------ formulation A
def eval_form_A_fp32(ns, model, x, y, z, R, grad_order=None):
    """The SAME module and statements run in fp32 torch (weights and points rounded to fp32): what genuine fp32
    arithmetic of the reference formulation gives on this batch -- the yardstick for an "fp32" path (round 4:
    the GPU tests hold the split-bf16 kernels to a multiple of THIS error instead of to a measured sensitivity)."""
    import copy
    params = ns["params"]
    m32 = copy.deepcopy(model).float()
    col = lambda a: torch.tensor(np.asarray(a, dtype=np.float64).reshape(-1, 1)).float().requires_grad_(True)
    xt, yt, zt, Rt = col(x), col(y), col(z), col(R)
    r1, r2 = ns["radial"](xt, yt, zt, Rt, params)
    b1 = torch.where(r1 >= params["BCcutoff"])
    b2 = torch.where(r2 >= params["BCcutoff"])
    psi, E = m32.parametricPsi(xt, yt, zt, Rt)
    lap = ns["lapl"](xt, yt, zt, psi)
    m32.zero_grad()
    Ltot, _, _, _ = m32.LossFunctions(xt, yt, zt, Rt, params, b1, b2)
    Ltot.backward()
    named = dict(m32.named_parameters())
    keys = grad_order or [k + s_ for k in STATE_KEYS for s_ in (".weight", ".bias")]
    grad = np.concatenate([named[k].grad.detach().double().numpy().ravel() for k in keys])
    d = lambda t: t.detach().double().numpy().ravel().copy()
    return dict(fp32_psi=d(psi), fp32_lap=d(lap), fp32_Ltot=float(Ltot.detach()), fp32_grad=grad)


def eval_form_A(ns, model, x, y, z, R):
    params = ns["params"]
    xt, yt, zt, Rt = as_col(x, True), as_col(y, True), as_col(z, True), as_col(R, True)
    r1, r2 = ns["radial"](xt, yt, zt, Rt, params)
    b1 = torch.where(r1 >= params["BCcutoff"])      # poc/main.py:392-393
    b2 = torch.where(r2 >= params["BCcutoff"])
    psi, E = model.parametricPsi(xt, yt, zt, Rt)
    lap = ns["lapl"](xt, yt, zt, psi)
    hpsi = ns["hamiltonian"](xt, yt, zt, Rt, psi, params)
    res = hpsi - E * psi
    model.zero_grad()
    Ltot, Lpde, Lbc, _ = model.LossFunctions(xt, yt, zt, Rt, params, b1, b2)
    Ltot.backward()
    grads = np.concatenate([p.grad.detach().numpy().ravel() for p in model.parameters()])
    mask = ((r1 >= params["BCcutoff"]).to(torch.uint8)
            + 2 * (r2 >= params["BCcutoff"]).to(torch.uint8)).numpy().ravel().astype(np.uint8)
    d = lambda t: t.detach().numpy().ravel().copy()
    return dict(x=x, y=y, z=z, R=R, mask=mask, theta=flat_state(model.state_dict()),
                psi=d(psi), E=d(E), lap=d(lap), hpsi=d(hpsi), res=d(res),
                Ltot=float(Ltot), Lpde=float(Lpde), Lbc=float(Lbc), grad=grads)


def gen_form_A(ns):
    NN_ion, params = ns["NN_ion"], ns["params"]
    x, y, z, R = fixed_batch(2048, 20250221, 4.0)
    out = {}
    for tag, ck in (("ionHsym", "ionHsym.pt"), ("fineTune", "ionHsym_fineTune.pt")):
        model = NN_ion(params)
        sd = torch.load(os.path.join(REF, "models", ck), map_location="cpu",
                        weights_only=True)["model_state_dict"]
        model.load_state_dict(sd)
        out[tag] = eval_form_A(ns, model, x, y, z, R)
    torch.manual_seed(7)
    out["rand16"] = eval_form_A(ns, NN_ion(params), x, y, z, R)
    torch.manual_seed(11)
    xs, ys, zs, Rs = fixed_batch(1024, 99, 4.0)
    out["rand64"] = eval_form_A(ns, NN_ion(params, 64, 32, 10), xs, ys, zs, Rs)
    torch.manual_seed(13)
    out["rand32"] = eval_form_A(ns, NN_ion(params, 32, 16, 8), xs[:512], ys[:512], zs[:512],
                                Rs[:512])
    # tanh: the reference's only activation handle is NN_ion.sig (poc/main.py:231), used by the
    # trunk, the E-net and the gate alike; swapping that module is all it takes to run the
    # reference's unchanged forward / LossFunctions / autograd with tanh (BASELINE "2x64 tanh")
    torch.manual_seed(17)
    m64 = NN_ion(params, 64, 32, 10)
    m64.sig = torch.nn.Tanh()
    out["tanh64"] = eval_form_A(ns, m64, xs, ys, zs, Rs)
    out["tanh64"].update(eval_form_A_fp32(ns, m64, xs, ys, zs, Rs))
    torch.manual_seed(19)
    m16 = NN_ion(params)
    m16.sig = torch.nn.Tanh()
    out["tanh16"] = eval_form_A(ns, m16, xs, ys, zs, Rs)
    # more than two hidden layers (BASELINE config 1 is "4x128"): the reference's trunk has two, with a
    # third commented out (`# fi_r = self.Lin_H3(fi_r); fi_r = self.sig(fi_r)`, poc/main.py:302).  The
    # subclass below re-enables exactly that pattern for L layers and changes nothing else, so psi, the
    # Laplacian, the residual, the losses and every gradient still come from the reference's own
    # forward / lapl / hamiltonian / LossFunctions and torch's autograd.
    class DeepIon(NN_ion):
        def __init__(self, params, H, HE, HG, L):
            super().__init__(params, H, HE, HG)
            self.n_layers = L
            for k in range(3, L + 1):
                setattr(self, "Lin_H%d" % k, torch.nn.Linear(H, H, bias=True))

        def base(self, fi_r1, fi_r2):
            fi_r = torch.cat((fi_r1, fi_r2), 1)
            for k in range(1, self.n_layers + 1):
                fi_r = self.sig(getattr(self, "Lin_H%d" % k)(fi_r))
            return fi_r

    def deep_order(model, vec_of):
        keys = ["Lin_H%d" % k for k in range(1, model.n_layers + 1)] + STATE_KEYS[2:]
        return np.concatenate([vec_of(k + s) for k in keys for s in (".weight", ".bias")])

    for tag, seed, H, L, act, n in (("deep128", 23, 128, 4, torch.nn.Tanh(), 768),
                                    ("deep16", 29, 16, 3, None, 1024)):
        torch.manual_seed(seed)
        m = DeepIon(params, H, 32, 10, L)
        if act is not None:
            m.sig = act
        d = eval_form_A(ns, m, xs[:n], ys[:n], zs[:n], Rs[:n])
        sd, named = m.state_dict(), dict(m.named_parameters())
        d["theta"] = deep_order(m, lambda k: sd[k].detach().numpy().ravel())
        d["grad"] = deep_order(m, lambda k: named[k].grad.detach().numpy().ravel())
        d["L"] = L
        if tag == "deep128":   # BASELINE config 1's shape: the fp32-torch run of the same module on the same batch
            keys = ["Lin_H%d" % k for k in range(1, L + 1)] + STATE_KEYS[2:]
            d.update(eval_form_A_fp32(ns, m, xs[:n], ys[:n], zs[:n], Rs[:n],
                                      [k + s_ for k in keys for s_ in (".weight", ".bias")]))
        out[tag] = d
    for tag, dct in out.items():
        if len(sys.argv) > 2 and tag not in sys.argv[2:]:
            continue
        np.savez_compressed(os.path.join(OUT, "g2_formA_%s.npz" % tag), **dct)
        print("g2_formA_%s: Ltot=%.6e |grad|=%.3e" % (tag, dct["Ltot"],
                                                      np.linalg.norm(dct["grad"])))
        if "fp32_grad" in dct:
            print("    fp32 torch on the same batch: grad %.3e of max|grad|, psi %.2e, lap %.2e, Ltot %.2e" % (
                np.abs(dct["fp32_grad"] - dct["grad"]).max() / np.abs(dct["grad"]).max(),
                np.abs(dct["fp32_psi"] - dct["psi"]).max() / np.abs(dct["psi"]).max(),
                np.abs(dct["fp32_lap"] - dct["lap"]).max() / np.abs(dct["lap"]).max(),
                abs(dct["fp32_Ltot"] / dct["Ltot"] - 1)))


# ------------------------------------------------------------ formulation B
FLAT_NAMES = ["H1a", "H1b", "H2a", "H2b", "H3a", "H3b", "L1a", "L1b", "L2a", "L2b", "E1a", "E1b",
              "E2a", "E2b", "E3a", "E3b"]


def flat_to_state_order(tensors):
    """train.py order/layout (in,out; H,L,E) -> state_dict order/layout ((out,in); H,E,L)."""
    t = dict(zip(FLAT_NAMES, tensors))
    order = ["H1", "H2", "H3", "E1", "E2", "E3", "L1", "L2"]
    parts = []
    for k in order:
        parts.append(t[k + "a"].detach().numpy().T.ravel())
        parts.append(t[k + "b"].detach().numpy().ravel())
    return np.concatenate(parts)


class _TanhTorch:
    """stands in for the name `torch` inside train.py's loss statements: `torch.sigmoid` (the only activation
    train.py:45-51 names) resolves to torch.tanh, every other attribute is torch's own"""
    sigmoid = staticmethod(torch.tanh)

    def __getattr__(self, name):
        return getattr(torch, name)


def gen_form_B(tanh64=False):
    """tanh64: BASELINE config 0's "2x64 tanh MLP, train.py reference" -- train.py's own statements
    (train.py:41-57, verbatim, at run time) with nh = 64 and `torch.sigmoid` resolving to tanh"""
    ns, loss_code, _ = load_train_parts()
    torch.manual_seed(4321 if tanh64 else 12345)  # train.py:75
    ns["dtype"] = torch.double                    # train.py:76
    nh, ne, nl = (64 if tanh64 else 16), 32, 10   # train.py:85-87
    shapes = [(2, nh), (nh,), (nh, nh), (nh,), (nh, 1), (1,), (1, nl), (nl,), (nl, 1), (1,),
              (1, ne), (ne,), (ne, ne), (ne,), (ne, 1), (1,)]
    tensors = [ns["ini"](*s) for s in shapes]     # train.py:88-103 (same call order)
    x, y, z, R = fixed_batch(1024 if tanh64 else 2048, 4243 if tanh64 else 4242, 3.0)
    env = dict(ns)
    if tanh64:
        env["torch"] = _TanhTorch()
    env.update(dict(zip(FLAT_NAMES, tensors)))
    env["x"], env["y"], env["z"] = as_col(x, True), as_col(y, True), as_col(z, True)
    env["R"] = as_col(R)
    r1sq = (x - R) ** 2 + y ** 2 + z ** 2
    r2sq = (x + R) ** 2 + y ** 2 + z ** 2
    env["i1"] = torch.tensor(np.where(r1sq >= 17.5 ** 2)[0])   # train.py:38-39
    env["i2"] = torch.tensor(np.where(r2sq >= 17.5 ** 2)[0])
    exec(loss_code, env)                          # train.py:41-57 verbatim, at run time
    grads = torch.autograd.grad(env["Ltot"], tensors)
    mask = ((r1sq >= 17.5 ** 2).astype(np.uint8) + 2 * (r2sq >= 17.5 ** 2).astype(np.uint8))
    d = lambda t: t.detach().numpy().ravel().copy()
    np.savez_compressed(
        os.path.join(OUT, "g3_formB_tanh64.npz" if tanh64 else "g3_formB.npz"), x=x, y=y, z=z, R=R, mask=mask,
        theta=flat_to_state_order(tensors), psi=d(env["psi"]), E=d(env["e"]),
        res=d(env["res"]), Ltot=float(env["Ltot"]), Lpde=float(env["Lpde"]),
        Lbc=float(env["Lbc"]), grad=flat_to_state_order(grads))
    print("g3_formB%s: Ltot=%.6e" % ("_tanh64" if tanh64 else "", float(env["Ltot"])))


# --------------------------------------------------------------------- Adam
def gen_adam(ns):
    NN_ion, params = ns["NN_ion"], ns["params"]
    for tag, frozen, lr in (("full", False, 8e-3), ("frozen", True, 5e-4)):
        model = NN_ion(params)
        sd = torch.load(os.path.join(REF, "models", "ionHsym.pt"), map_location="cpu",
                        weights_only=True)["model_state_dict"]
        model.load_state_dict(sd)
        if frozen:                                 # poc/main.py:384-387
            model.freezeDecayUnit()
            model.freezeBase()
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0)  # poc/main.py:363
        theta0 = flat_state(model.state_dict())
        n, steps = 512, 10
        xs, ys, zs, Rs, Ls, thetas = [], [], [], [], [], []
        for t in range(steps):
            x, y, z, R = fixed_batch(n, 1000 + t, 4.0)
            xt, yt, zt, Rt = as_col(x, True), as_col(y, True), as_col(z, True), as_col(R, True)
            r1, r2 = ns["radial"](xt, yt, zt, Rt, params)
            b1 = torch.where(r1 >= params["BCcutoff"])
            b2 = torch.where(r2 >= params["BCcutoff"])
            opt.zero_grad()
            Ltot, Lpde, Lbc, _ = model.LossFunctions(xt, yt, zt, Rt, params, b1, b2)
            Ltot.backward()
            opt.step()                             # poc/main.py:401-404
            xs.append(x); ys.append(y); zs.append(z); Rs.append(R)
            Ls.append([float(Ltot), float(Lpde), float(Lbc)])
            thetas.append(flat_state(model.state_dict()))
        st = opt.state_dict()["state"]
        plist = list(model.parameters())
        m = np.concatenate([(st[i]["exp_avg"].numpy().ravel() if i in st
                             else np.zeros(plist[i].numel())) for i in range(len(plist))])
        v = np.concatenate([(st[i]["exp_avg_sq"].numpy().ravel() if i in st
                             else np.zeros(plist[i].numel())) for i in range(len(plist))])
        np.savez_compressed(os.path.join(OUT, "g4_adam_%s.npz" % tag), theta0=theta0, lr=lr,
                            x=np.array(xs), y=np.array(ys), z=np.array(zs), R=np.array(Rs),
                            losses=np.array(Ls), thetas=np.array(thetas), m=m, v=v,
                            frozen=int(frozen))
        print("g4_adam_%s: L0=%.4e L9=%.4e" % (tag, Ls[0][0], Ls[-1][0]))


# ------------------------------------------------------------ energy tables
def parse_energy_pickle(path):
    """Walk the pickle opcodes (no unpickling): a key string, later a 312-byte <f8 blob."""
    out, key = {}, None
    for op, arg, _ in pickletools.genops(open(path, "rb").read()):
        if op.name == "SHORT_BINUNICODE" and arg in ("R", "E_int", "Elcao", "E_net"):
            key = arg
        elif op.name == "BINBYTES" and key is not None and len(arg) == 39 * 8:
            out[key] = np.frombuffer(arg, dtype="<f8").copy()
            key = None
    return out


def gen_energy(ns):
    tab = parse_energy_pickle(os.path.join(REF, "poc", "energy_R_ion.pkl"))
    R_exact, e_exact = ns["exactE"]()
    np.savez_compressed(os.path.join(OUT, "g5_energy.npz"), R=tab["R"], E_int=tab["E_int"],
                        Elcao=tab["Elcao"], E_net=tab["E_net"], R_exact=np.asarray(R_exact),
                        E_exact=np.asarray(e_exact, dtype=np.float64))
    print("g5_energy: E_net[0..2] =", tab["E_net"][:3])


def gen_modelbin():
    ns, _, writer_code = load_train_parts()
    sd = torch.load(os.path.join(REF, "models", "ionHsym_fineTune.pt"), map_location="cpu",
                    weights_only=True)["model_state_dict"]
    g = lambda k: sd[k].clone()
    # H1a = Lin_H1.weight.T, ... ; bin order H, L(gate), E
    pairs = [("Lin_H1", "H1"), ("Lin_H2", "H2"), ("Lin_out", "H3"), ("netDecayL", "L1"),
             ("netDecay", "L2"), ("Lin_E1", "E1"), ("Lin_E2", "E2"), ("Lin_Eout", "E3")]
    params = []
    for key, _ in pairs:
        params.append(g(key + ".weight").T.contiguous())
        params.append(g(key + ".bias"))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            env = dict(ns)
            env["params"] = tuple(params)
            exec(writer_code, env)                 # train.py:112-119
            blob = open("model.bin", "rb").read()
            os.environ["MPLBACKEND"] = "Agg"
            np.product = np.prod                   # energy.py:18 predates NumPy 2
            esrc = open(os.path.join(REF, "energy.py")).read()
            eenv = {"__name__": "ref_energy"}
            exec(compile(esrc, "energy.py", "exec"), eenv)
            eR, Rg = eenv["e"][:, 0].copy(), eenv["R"][:, 0].copy()
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "g6_modelbin.npz"),
                        blob=np.frombuffer(blob, dtype=np.uint8), R=Rg, e=eR)
    print("g6_modelbin: %d bytes, e(R=0.2)=%.6f" % (len(blob), eR[0]))


def gen_force(ns):
    from torch.autograd import grad
    NN_ion, params = ns["NN_ion"], ns["params"]
    R = np.linspace(0.2, 4.0, 191)
    out = dict(R=R)
    for tag, ck in (("ionHsym", "ionHsym.pt"), ("fineTune", "ionHsym_fineTune.pt")):
        model = NN_ion(params)
        sd = torch.load(os.path.join(REF, "models", ck), map_location="cpu",
                        weights_only=True)["model_state_dict"]
        model.load_state_dict(sd)
        Rt = as_col(R, True)
        xt, yt, zt = as_col(0.3 + 0 * R, True), as_col(0 * R, True), as_col(0 * R, True)
        _, E = model.parametricPsi(xt, yt, zt, Rt)
        ones = torch.ones(Rt.shape, dtype=torch.float64)
        ET = E + 1 / (2 * Rt)                                        # poc/main.py:1316
        F = -grad([ET], [Rt], grad_outputs=ones, create_graph=True)[0]   # :1322-1324
        F2 = grad([F], [Rt], grad_outputs=ones, create_graph=True)[0]    # :1326-1328
        dE = grad([E], [Rt], grad_outputs=ones, create_graph=True)[0]
        d2E = grad([dE], [Rt], grad_outputs=ones)[0]
        d = lambda t: t.detach().numpy().ravel().copy()
        out.update({tag + "_E": d(E), tag + "_dE": d(dE), tag + "_d2E": d(d2E),
                    tag + "_F": d(F), tag + "_F2": d(F2)})
    # Hellmann-Feynman integral on a 40^3 grid, the reference's function as is
    p = dict(params)
    p["n_test"] = 40
    p["loadModelPath"] = os.path.join(REF, "models", "ionHsym.pt")
    ns["params"] = p                       # poc/main.py functions read the module-level dict too
    Ri = np.array([0.7, 1.0, 2.0])
    out["hf_Ri"] = Ri
    out["hf_n_test"] = 40
    out["hf_dEdR"] = np.array([float(ns["dEdR_int"](float(r), p)) for r in Ri])
    ns["params"] = params
    # plot.py: psi along the x axis from model.bin (its statements, run in a scratch cwd)
    blob = np.load(os.path.join(OUT, "g6_modelbin.npz"))["blob"].tobytes()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            open("model.bin", "wb").write(blob)
            os.environ["MPLBACKEND"] = "Agg"
            np.product = np.prod
            penv = {"__name__": "ref_plot"}
            exec(compile(open(os.path.join(REF, "plot.py")).read(), "plot.py", "exec"), penv)
            out["cut_R0"] = float(penv["R0"])           # the loop's last pass (R0 = 2)
            out["cut_x"] = penv["x"][:, 0].copy()
            out["cut_psi"] = penv["psi"][:, 0].copy()
            out["cut_psi_lcao"] = penv["psi_lcao"][:, 0].copy()
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "g7_force.npz"), **out)
    print("g7_force: F(R=2)=%.6f  HF dE/dR=%s  psi_cut(0)=%.6f" % (
        out["ionHsym_F"][90], out["hf_dEdR"], out["cut_psi"][500]))


def gen_weights():
    out = {}
    for tag, ck in (("ionHsym", "ionHsym.pt"), ("fineTune", "ionHsym_fineTune.pt")):
        path = os.path.join(REF, "models", ck)
        c = torch.load(path, map_location="cpu", weights_only=True)
        out[tag] = flat_state(c["model_state_dict"])
        out[tag + "_sha256"] = np.frombuffer(bytes.fromhex(sha256(path)), dtype=np.uint8)
        out[tag + "_lr"] = c["optimizer_state_dict"]["param_groups"][0]["lr"]
        st = c["optimizer_state_dict"]["state"]
        out[tag + "_step"] = float(st[sorted(st)[0]]["step"])
    np.savez_compressed(os.path.join(OUT, "g1_weights.npz"), **out)
    print("g1_weights:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


def main():
    torch.set_default_dtype(torch.float64)         # poc/main.py:915, train.py:77
    torch.set_num_threads(4)
    ns = load_main_defs()
    only = sys.argv[1] if len(sys.argv) > 1 else "g"
    jobs = [("g1", gen_weights, ()), ("g2", gen_form_A, (ns,)), ("g3", gen_form_B, ()), ("g3", gen_form_B, (True,)),
            ("g4", gen_adam, (ns,)), ("g5", gen_energy, (ns,)), ("g6", gen_modelbin, ()),
            ("g7", gen_force, (ns,))]
    for name, fn, args in jobs:
        if name.startswith(only):
            fn(*args)


if __name__ == "__main__":
    sys.exit(main())
