"""On-disk formats either side of the hot path.

* `model.bin` (train.py:112-119; readers energy.py:8-19, plot.py:6-17): for each
  of 16 arrays in train.py order (H1a,H1b,H2a,H2b,H3a,H3b,L1a,L1b,L2a,L2b,E1a,...,
  E3b): int32 LE ndim, ndim int32 LE dims, raw C-order float64; weights (in,out).
* NN_ion checkpoints (poc/main.py:331-339): torch.save({'model_state_dict',
  'optimizer_state_dict'}), keys in registration order, weights (out,in).

Order change between the two: bin = trunk, gate, E-net; state_dict = trunk, E-net, gate.
"""
import numpy as np

# (state_dict module name, train.py prefix), in model.bin order
BIN_ORDER = [("Lin_H1", "H1"), ("Lin_H2", "H2"), ("Lin_out", "H3"), ("netDecayL", "L1"),
             ("netDecay", "L2"), ("Lin_E1", "E1"), ("Lin_E2", "E2"), ("Lin_Eout", "E3")]


# NN_ion registration order = state_dict key order (poc/main.py:234-245)
STATE_ORDER = ["Lin_H1", "Lin_H2", "Lin_out", "Lin_E1", "Lin_E2", "Lin_Eout", "netDecayL", "netDecay"]


def state_dict_to_bin_arrays(sd):
    """state_dict (tensors or arrays, L = 2) -> list of 16 float64 arrays in model.bin order."""
    out = []
    for key, _ in BIN_ORDER:
        w = np.asarray(_np(sd[key + ".weight"]), dtype=np.float64)
        b = np.asarray(_np(sd[key + ".bias"]), dtype=np.float64)
        out.append(np.ascontiguousarray(w.T))   # (out,in) -> (in,out): x @ A + b (train.py:4-5)
        out.append(np.ascontiguousarray(b))
    return out


def bin_arrays_to_state_dict(arrs):
    """inverse of state_dict_to_bin_arrays; returns {key: float64 ndarray}."""
    sd = {}
    for k, (key, _) in enumerate(BIN_ORDER):
        sd[key + ".weight"] = np.ascontiguousarray(np.asarray(arrs[2 * k]).T)
        sd[key + ".bias"] = np.asarray(arrs[2 * k + 1]).copy()
    return sd


def write_model_bin(path, sd):
    with open(path, "wb") as f:
        for a in state_dict_to_bin_arrays(sd):
            f.write(int(a.ndim).to_bytes(4, "little"))
            for d in a.shape:
                f.write(int(d).to_bytes(4, "little"))
            f.write(a.tobytes())


def read_model_bin(path):
    """Same stop rule as energy.py:8-19: stop when the 4-byte ndim read comes back empty."""
    arrs = []
    with open(path, "rb") as f:
        while True:
            ndim = int.from_bytes(f.read(4), "little")
            if ndim == 0:
                break
            shape = [int.from_bytes(f.read(4), "little") for _ in range(ndim)]
            n = int(np.prod(shape))
            arrs.append(np.frombuffer(f.read(n * 8), dtype="<f8").reshape(shape).copy())
    return arrs


def pt_to_model_bin(pt_path, bin_path):
    """NN_ion checkpoint (poc/main.py:331-339) -> model.bin for energy.py / plot.py."""
    import torch
    ck = torch.load(pt_path, map_location="cpu", weights_only=True)   # executes nothing
    write_model_bin(bin_path, ck["model_state_dict"])


def model_bin_to_pt(bin_path, pt_path, optimizer_state=None):
    """model.bin -> checkpoint that NN_ion.loadModel reads (keys in registration order,
    float64 like the reference's files)."""
    import collections
    import torch
    sd = bin_arrays_to_state_dict(read_model_bin(bin_path))
    osd = collections.OrderedDict()
    for key in STATE_ORDER:
        for suffix in (".weight", ".bias"):
            osd[key + suffix] = torch.tensor(sd[key + suffix], dtype=torch.float64)
    torch.save({"model_state_dict": osd, "optimizer_state_dict": optimizer_state or {}}, pt_path)


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
