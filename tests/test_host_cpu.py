"""CPU-side checks of the product: the C-ABI library loads and exports every symbol of
include/pinn.h, its layout helpers agree with the oracle, the file formats are byte-compatible
with what the reference wrote, and the host module keeps the reference's state_dict surface.
No compute call is made here (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import pinn_for_quantum_wavefunction_surfaces_amd as pkg
from pinn_for_quantum_wavefunction_surfaces_amd import capi, formats
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pinn.h")).read()
    declared = set(re.findall(r"\b(pinn_[A-Za-z_0-9]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    lib = C.CDLL(capi.LIB_PATH)
    for name in declared:
        assert getattr(lib, name) is not None


@pytest.mark.parametrize("H,L,HE,HG", [(16, 2, 32, 10), (64, 2, 32, 10), (128, 4, 32, 10), (8, 3, 6, 4)])
def test_layout_helpers_match_oracle(H, L, HE, HG):
    a = capi.Arch(H, L, HE, HG, 0, 2, 1.0, 0)
    assert capi.param_count(a) == O.param_count(O.arch(H=H, L=L, HE=HE, HG=HG))
    assert capi.tensor_layout(a) == O.tensor_layout(O.arch(H=H, L=L, HE=HE, HG=HG))


def test_bad_arguments_are_reported_not_fatal():
    lib = capi.load_library()
    bad = capi.Arch(16, 1, 32, 10, 0, 2, 1.0, 0)        # L < 2
    assert lib.pinn_param_count(C.byref(bad)) == -1
    assert lib.pinn_strerror(-2).decode().startswith("unsupported")
    h = C.c_void_p()
    assert lib.pinn_create(C.byref(bad), 0, 100, C.byref(h)) == -1 and not h
    assert lib.pinn_destroy(None) == -1


def test_state_dict_surface_matches_reference_checkpoints(golden_dir):
    m = pkg.NN_ion(pkg.set_params())
    keys = list(m.state_dict().keys())
    want = [k + s for k in ("Lin_H1", "Lin_H2", "Lin_out", "Lin_E1", "Lin_E2", "Lin_Eout",
                            "netDecayL", "netDecay") for s in (".weight", ".bias")]
    assert keys == want                                   # registration order
    shapes = [tuple(v.shape) for v in m.state_dict().values()]
    assert shapes == [(16, 2), (16,), (16, 16), (16,), (1, 16), (1,), (32, 1), (32,), (32, 32),
                      (32,), (1, 32), (1,), (10, 1), (10,), (1, 10), (1,)]
    assert float(m.Lin_Eout.bias.detach()[0]) == -1.0              # poc/main.py:240
    # a reference-format fp64 checkpoint loads and round-trips through the flat vector
    w = np.load(os.path.join(golden_dir, "g1_weights.npz"))
    m.load_flat_params(torch.tensor(w["fineTune"]))
    assert np.allclose(m.flat_params(dtype=torch.float64).numpy(), w["fineTune"], rtol=1e-7)
    m.freezeBase()
    m.freezeDecayUnit()
    assert m.trainable_mask() == sum(1 << t for t in range(6, 12))   # E-net tensors only


def test_checkpoint_file_roundtrip(tmp_path, golden_dir):
    torch.set_default_dtype(torch.float64)
    try:
        params = pkg.set_params()
        m = pkg.NN_ion(params)
        w = np.load(os.path.join(golden_dir, "g1_weights.npz"))
        m.load_flat_params(torch.tensor(w["ionHsym"]))
        params["saveModelPath"] = params["loadModelPath"] = str(tmp_path / "m.pt")
        m.saveModel(params, torch.optim.Adam(m.parameters(), lr=8e-3))
        ck = torch.load(params["saveModelPath"], weights_only=True)
        assert set(ck) == {"model_state_dict", "optimizer_state_dict"}
        m2 = pkg.NN_ion(params)
        m2.loadModel(params)
        assert np.array_equal(m2.flat_params(dtype=torch.float64).numpy(), w["ionHsym"])
    finally:
        torch.set_default_dtype(torch.float32)


def test_model_bin_is_byte_identical_to_reference_writer(tmp_path, golden_dir):
    """g6: the blob train.py's serializer wrote for ionHsym_fineTune.pt (12 328 bytes)."""
    w = np.load(os.path.join(golden_dir, "g1_weights.npz"))
    g = np.load(os.path.join(golden_dir, "g6_modelbin.npz"))
    torch.set_default_dtype(torch.float64)
    try:
        m = pkg.NN_ion(pkg.set_params())
        m.load_flat_params(torch.tensor(w["fineTune"]))
    finally:
        torch.set_default_dtype(torch.float32)
    path = str(tmp_path / "model.bin")
    formats.write_model_bin(path, m.state_dict())
    blob = open(path, "rb").read()
    assert len(blob) == 12328 and blob == g["blob"].tobytes()
    arrs = formats.read_model_bin(path)
    assert [a.shape for a in arrs][:6] == [(2, 16), (16,), (16, 16), (16,), (16, 1), (1,)]
    sd = formats.bin_arrays_to_state_dict(arrs)
    assert np.array_equal(sd["Lin_E2.weight"], m.state_dict()["Lin_E2.weight"].numpy())
    # energy.py's evaluation of that file (its e(R)) from the read-back arrays
    E1a, E1b, E2a, E2b, E3a, E3b = arrs[10:16]
    sig = lambda t: 1 / (1 + np.exp(-t))
    R = g["R"][:, None]
    e = sig(sig(R @ E1a + E1b) @ E2a + E2b) @ E3a + E3b
    assert np.max(np.abs(e[:, 0] - g["e"])) < 1e-13


def test_pt_and_model_bin_convert_both_ways(tmp_path, golden_dir):
    """.pt -> model.bin gives the reference writer's bytes; model.bin -> .pt loads back into
    NN_ion with the reference's key order and values"""
    import collections
    g = np.load(os.path.join(golden_dir, "g6_modelbin.npz"))
    w = np.load(os.path.join(golden_dir, "g1_weights.npz"))["fineTune"]
    torch.set_default_dtype(torch.float64)          # the reference's files are fp64
    try:
        m = pkg.NN_ion(pkg.set_params())
        m.load_flat_params(torch.tensor(w))
        pt, binp, pt2 = [str(tmp_path / n) for n in ("a.pt", "model.bin", "b.pt")]
        torch.save({"model_state_dict": collections.OrderedDict(m.state_dict()),
                    "optimizer_state_dict": {}}, pt)
        pkg.formats.pt_to_model_bin(pt, binp)
        assert open(binp, "rb").read() == g["blob"].tobytes()
        pkg.formats.model_bin_to_pt(binp, pt2)
        sd = torch.load(pt2, map_location="cpu", weights_only=True)["model_state_dict"]
        assert list(sd) == list(m.state_dict())
        m2 = pkg.NN_ion(pkg.set_params())
        m2.load_state_dict(sd)
        assert torch.equal(m2.flat_params(dtype=torch.float64), torch.tensor(w))
    finally:
        torch.set_default_dtype(torch.float32)


def test_flat_script_surface_matches_train_py_init_and_writer(tmp_path, golden_dir):
    """flat.init_params under train.py's seed reproduces the tensors train.py's own `ini` calls
    produced (g3 was generated by executing them); flat.write_model_bin writes what
    formats.write_model_bin writes for the equivalent state_dict"""
    from pinn_for_quantum_wavefunction_surfaces_amd import flat
    g = np.load(os.path.join(golden_dir, "g3_formB.npz"))
    torch.manual_seed(12345)                              # train.py:75
    params = flat.init_params()
    assert [tuple(t.shape) for t in params][:6] == [(2, 16), (16,), (16, 16), (16,), (16, 1), (1,)]
    sd = formats.bin_arrays_to_state_dict([t.detach().numpy() for t in params])
    theta = np.concatenate([sd[k + s].ravel() for k in formats.STATE_ORDER for s in (".weight", ".bias")])
    assert np.array_equal(theta, g["theta"])
    a, b = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    flat.write_model_bin(a, params)
    formats.write_model_bin(b, sd)
    assert open(a, "rb").read() == open(b, "rb").read()
    with pytest.raises(capi.PinnError):                   # the loop itself is GPU only
        flat.train(params, lr=8e-3, epochs=1, device="cpu")


def test_no_cpu_fallback():
    m = pkg.NN_ion(pkg.set_params())
    x = torch.zeros(8, 1)
    with pytest.raises(pkg.PinnError):
        m(x, x, x, x + 1)
    with pytest.raises(pkg.PinnError):
        pkg.Trainer(m, pkg.set_params(), n_points=128, device="cpu")


def test_torch_baseline_matches_oracle(golden_dir):
    """bench.py's cpu_baseline leg (oracle/torch_baseline.py) computes the reference numbers."""
    from oracle import torch_baseline as TB
    g = np.load(os.path.join(golden_dir, "g2_formA_rand16.npz"))
    arch = dict(H=16, L=2, HE=32, HG=10, act=0, n_pass=2, out_scale=1.0)
    ham = dict(a=-0.5, b=-1.0, c=-1.0, P=1.0)
    th = torch.tensor(g["theta"], requires_grad=True)
    col = lambda a, rg=False: torch.tensor(a).reshape(-1, 1).requires_grad_(rg)
    i1 = torch.tensor(np.where(g["mask"] & 1)[0])
    i2 = torch.tensor(np.where(g["mask"] & 2)[0])
    Ltot = TB.loss_terms(th, arch, ham, col(g["x"], True), col(g["y"], True), col(g["z"], True),
                         col(g["R"]), i1, i2)[0]
    Ltot.backward()
    assert abs(float(Ltot) / g["Ltot"] - 1) < 1e-12
    assert np.abs(th.grad.numpy() - g["grad"]).max() / np.abs(g["grad"]).max() < 1e-11


def test_no_packed_opsel_forms_in_any_trunk_kernel():
    """hipcc's v_pk_{mul,fma,add}_f32 ... op_sel:[..] (the LOW result lane reading the HIGH dword of an operand
    pair) intermittently computed with 0 in place of that dword on MI355X (profiles/r02_packed_opsel.md,
    tools/opselbench; cause open: compiler or hardware, profiles/r03_packed_opsel.md).  The matrix-core trunk
    kernels (csrc/pinn_trunk_*.hip, csrc/pinn_rnet_mfma.hip) are written so that none is emitted (lin2, fmac_bcast,
    ring_splats in csrc/pinn_mfma_common.h), and since round 4 so are the exact-fp32 vector kernels of
    csrc/pinn_kernels.hip (bcast_elem in pinn_math.h); this compiles every arithmetic translation unit to assembly
    (no GPU needed) and looks."""
    import shutil
    import sys
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import opsel_check
    hits, tot = opsel_check.compile_and_scan()
    assert sum(tot.values()) > 10000          # the scan saw the kernels
    assert any("7k_trunkI" in k for k in tot), "pinn_kernels.hip was not scanned"
    bad = {k: v for k, v in hits.items() if v}
    assert not bad, "packed op_sel forms in: %s" % sorted(bad)[:5]
    # and, while the assembly is at hand: the kernels of the benchmark's primary workload, of train.py's formulation
    # and of the reference architecture do not spill (a silent performance regression otherwise)
    for sym in ("8k_trunk2ILi64E", "9k_trunk2sILi64E", "9k_trunk16I"):
        ks = {k: v for k, v in opsel_check.SCRATCH.items() if sym in k}
        assert ks and all(v == 0 for v in ks.values()), {k[-40:]: v for k, v in ks.items() if v}


@pytest.mark.parametrize("n", [3, 7, 8, 80, 81, 250])
def test_simpson_weight_vector_is_scipys_rule(n):
    """analysis.simpson_weights(x) @ f == scipy.integrate.simpson(f, x=x) for any f, odd and even point counts
    (the reference's integra3d nests scipy's rule over z, y, x: poc/main.py:179-186; pinn_quadrature3d applies the
    three weight vectors as one weighted sum)"""
    from scipy.integrate import simpson
    from pinn_for_quantum_wavefunction_surfaces_amd.analysis import simpson_weights
    rng = np.random.default_rng(n)
    x = np.linspace(-18.0, 18.0, n)
    w = simpson_weights(x)
    for _ in range(3):
        f = rng.normal(size=n)
        assert abs(w @ f - simpson(f, x=x)) < 1e-12 * np.abs(w).sum() * np.abs(f).max()
    # and nested over three axes, as integra3d does it
    if n <= 8:
        f3 = rng.normal(size=(n, n, n))
        nested = simpson([simpson([simpson(fx, x=x) for fx in fy], x=x) for fy in f3], x=x)
        assert abs(np.einsum("i,j,k,ijk->", w, w, w, f3) - nested) < 1e-10 * np.abs(f3).sum()


def test_profile_summary_keys_and_the_traffic_file_bench_reads():
    """tools/summarize_prof.py names a trunk kernel the same whether rocprofv3 printed it demangled or mangled, and
    bench.py takes `roofline.traffic` of every workload from the committed profiles/traffic.json (round 4; round 3 had the
    numbers pasted into bench.py)"""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "tools", "summarize_prof.py")).read()
    ns = {}
    exec(src[:src.index("root = sys.argv[1]")], ns)
    key = ns["kernel_key"]
    assert key("void (anonymous namespace)::k_trunkD<128, 4, 1, 0>(float const*, bool") == "trunkD_128"
    assert key("_ZN12_GLOBAL__N_18k_trunk2ILi64ELi1ELb1EEEvPKfPKDF16b9TrunkArgs") == "trunk2_64"
    assert key("space)::k_trunk16<0, 2, 1>(float const*, TrunkArgs)") == "trunk16_bwd"
    assert key("space)::k_trunk16<0, 2, 2>(float const*, TrunkArgs)") == "trunk16_head"
    assert key("k_trunk2sILi64ELi1EEEvPKfPKDF16b9TrunkArgs") == "trunk2s_64" and key("k_trunkDx<1>(float") == "trunkDx"
    assert key("k_sample(unsigned long") is None
    t = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys_mod = list(__import__("sys").argv), __import__("sys")
    sys_mod.argv = ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys_mod.argv = argv
    for arch, k in bench.TRAFFIC_KERNEL_KEY.items():
        assert arch in bench.ARCHS and k in t["bytes_per_launch"], (arch, k)
        assert bench.TRAFFIC[arch] == t["bytes_per_launch"][k] > 1e6
    assert "FETCH_SIZE" in bench.TRAFFIC_SOURCE


def test_every_tile_is_taken_exactly_once_whatever_the_shares():
    """The tile -> block map of the trunk kernels (csrc/pinn_common.h block_tiles_of, driven on the host through
    pinn_debug_block_tiles): for any grid, tile count, tile unit and TileShares -- equal, two or four classes, bounds beyond the
    end, out of order, grids smaller than the class count -- the blocks' tile sets partition [0, ntile)."""
    import ctypes as C
    lib = capi.load_library()
    lib.pinn_debug_block_tiles.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.c_int,
                                           C.POINTER(C.c_int64)]
    lib.pinn_debug_block_tiles.restype = C.c_int
    rng = np.random.default_rng(0)
    cases = [(512, 62500, 2, [41563], 1), (512, 62500, 0, [], 1), (1024, 15625, 4, [9000, 12500, 14000], 4),
             (1, 7, 2, [3], 1), (2, 0, 2, [0], 1), (3, 10, 4, [2, 4, 6], 1), (5, 100, 4, [50, 75, 90], 1),
             (512, 8193, 2, [10 ** 9], 1), (512, 8193, 2, [0], 1), (8, 1000, 4, [900, 100, 500], 2), (7, 3, 2, [2], 1)]
    for _ in range(200):
        G = int(rng.integers(1, 1100))
        nt = int(rng.integers(0, 5000))
        K = int(rng.choice([0, 2, 4]))
        unit = int(rng.choice([1, 2, 4]))
        b = sorted(int(v) for v in rng.integers(0, max(nt * unit, 1) + 5, max(K - 1, 0)))
        cases.append((G, nt, K, b, unit))
    out = (C.c_int64 * 3)()
    for G, nt, K, bounds, unit in cases:
        arr = (C.c_int64 * 3)(*(list(bounds) + [0, 0, 0])[:3])
        seen = np.zeros(nt, np.int32)
        for blk in range(G):
            assert lib.pinn_debug_block_tiles(G, blk, nt, K, arr, unit, out) == 0
            first, end, stride = out[0], out[1], out[2]
            assert stride >= 1 and 0 <= end <= nt and first >= 0
            if end > first:
                seen[first:end:stride] += 1
        assert (seen == 1).all(), (G, nt, K, bounds, unit, int((seen != 1).sum()))

