"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module (see oracle/pinn_oracle.h).  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpinn_oracle.so")


class Arch(C.Structure):
    _fields_ = [("H", C.c_int32), ("L", C.c_int32), ("HE", C.c_int32), ("HG", C.c_int32),
                ("act", C.c_int32), ("n_pass", C.c_int32), ("out_scale", C.c_double)]


class Ham(C.Structure):
    _fields_ = [("a", C.c_double), ("b", C.c_double), ("c", C.c_double), ("P", C.c_double),
                ("Ry", C.c_double), ("Rz", C.c_double)]


def build(force=False):
    src = os.path.join(_HERE, "pinn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpinn_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        _lib.oracle_param_count.restype = C.c_int64
        _lib.oracle_param_count.argtypes = [C.POINTER(Arch)]
        _lib.oracle_tensor_count.restype = C.c_int32
        _lib.oracle_tensor_count.argtypes = [C.POINTER(Arch)]
        _lib.oracle_tensor_layout.argtypes = [C.POINTER(Arch), C.POINTER(C.c_int64),
                                              C.POINTER(C.c_int64)]
        _lib.oracle_forward.argtypes = [C.POINTER(Arch), C.POINTER(Ham)] + [dp] * 5 + \
            [C.c_int64] + [dp] * 4
        _lib.oracle_energy_R.argtypes = [C.POINTER(Arch), dp, dp, C.c_int64, dp, dp, dp]
        _lib.oracle_loss_grad.argtypes = [C.POINTER(Arch), C.POINTER(Ham)] + [dp] * 5 + \
            [C.POINTER(C.c_uint8), C.c_int64, C.c_double, C.c_double, C.c_double,
             C.c_uint64, dp, dp]
        _lib.oracle_adam.argtypes = [C.POINTER(Arch), dp, dp, dp, dp, C.c_int64] + \
            [C.c_double] * 4 + [C.c_uint64]
        _lib.oracle_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        fp = C.POINTER(C.c_float)
        _lib.oracle_sample.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int64] + \
            [C.c_float] * 5 + [fp] * 4 + [C.POINTER(C.c_uint8), C.POINTER(C.c_int64)]
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def arch(H=16, L=2, HE=32, HG=10, act=0, n_pass=2, out_scale=1.0):
    return Arch(H, L, HE, HG, act, n_pass, out_scale)


def ham_A(P=1.0, Ry=0.0, Rz=0.0):
    """NN_ion: res = -1/2 lap + V psi - E psi (poc/main.py:118-120,345)."""
    return Ham(-0.5, -1.0, -1.0, P, Ry, Rz)


def ham_B():
    """train.py:54: res = lap + (e + 1/r1 + 1/r2) psi."""
    return Ham(1.0, 1.0, 1.0, 1.0, 0.0, 0.0)


def param_count(A):
    return int(lib().oracle_param_count(C.byref(A)))


def tensor_layout(A):
    nt = int(lib().oracle_tensor_count(C.byref(A)))
    off = (C.c_int64 * nt)()
    sz = (C.c_int64 * nt)()
    lib().oracle_tensor_layout(C.byref(A), off, sz)
    return list(off), list(sz)


def _pts(*arrs):
    return [np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel()) for a in arrs]


def forward(A, Hm, theta, x, y, z, R):
    theta, x, y, z, R = _pts(theta, x, y, z, R)
    n = x.size
    out = [np.empty(n) for _ in range(4)]
    lib().oracle_forward(C.byref(A), C.byref(Hm), _d(theta), _d(x), _d(y), _d(z), _d(R), n,
                         *[_d(o) for o in out])
    return dict(psi=out[0], E=out[1], lap=out[2], res=out[3])


def energy_R(A, theta, R):
    """E(R), dE/dR, d2E/dR2 of the E-net"""
    theta, R = _pts(theta, R)
    out = [np.empty(R.size) for _ in range(3)]
    lib().oracle_energy_R(C.byref(A), _d(theta), _d(R), R.size, *[_d(o) for o in out])
    return dict(E=out[0], dE=out[1], d2E=out[2])


def loss_grad(A, Hm, theta, x, y, z, R, mask, invN=None, invN1=None, invN2=None,
              trainable_mask=None, want_grad=True):
    theta, x, y, z, R = _pts(theta, x, y, z, R)
    n = x.size
    mask = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8).ravel())
    n1 = int((mask & 1).sum())
    n2 = int(((mask >> 1) & 1).sum())
    invN = 1.0 / n if invN is None else invN
    invN1 = (1.0 / n1 if n1 else 0.0) if invN1 is None else invN1
    invN2 = (1.0 / n2 if n2 else 0.0) if invN2 is None else invN2
    if trainable_mask is None:
        trainable_mask = (1 << 63) - 1
    grad = np.zeros(param_count(A)) if want_grad else None
    sums = np.zeros(3)
    lib().oracle_loss_grad(C.byref(A), C.byref(Hm), _d(theta), _d(x), _d(y), _d(z), _d(R),
                           mask.ctypes.data_as(C.POINTER(C.c_uint8)), n, invN, invN1, invN2,
                           C.c_uint64(trainable_mask), _d(grad), _d(sums))
    Lpde = sums[0] * invN
    Lbc = sums[1] * invN1 + sums[2] * invN2
    return dict(grad=grad, sums=sums, Lpde=Lpde, Lbc=Lbc, Ltot=Lpde + Lbc)


def adam(A, theta, grad, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, trainable_mask=None):
    if trainable_mask is None:
        trainable_mask = (1 << 63) - 1
    for a in (theta, grad, m, v):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    lib().oracle_adam(C.byref(A), _d(theta), _d(grad), _d(m), _d(v), step, lr, b1, b2, eps,
                      C.c_uint64(trainable_mask))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().oracle_philox4x32_10(c, k, o)
    return list(o)


def sample(seed, step, first, n, box=18.0, Rlo=0.2, Rhi=4.0, cutoff=0.005, bc_cutoff=17.5):
    x, y, z, R = [np.empty(n, dtype=np.float32) for _ in range(4)]
    mask = np.empty(n, dtype=np.uint8)
    counts = (C.c_int64 * 2)()
    fp = C.POINTER(C.c_float)
    lib().oracle_sample(C.c_uint64(seed), C.c_uint64(step), first, n, box, Rlo, Rhi, cutoff,
                        bc_cutoff, *[a.ctypes.data_as(fp) for a in (x, y, z, R)],
                        mask.ctypes.data_as(C.POINTER(C.c_uint8)), counts)
    return x, y, z, R, mask, (int(counts[0]), int(counts[1]))
