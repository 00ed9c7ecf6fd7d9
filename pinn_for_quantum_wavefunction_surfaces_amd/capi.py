"""ctypes binding of the C ABI in include/pinn.h (libpinn_hip.so).

There is no CPU fallback: if the HIP library is missing, or a call returns an
error, this module raises.  torch is used only for device memory and streams.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PINN_HIP_LIB", os.path.join(HERE, "libpinn_hip.so"))  # override: A/B builds

PINN_ACT_SIGMOID, PINN_ACT_TANH = 0, 1
PINN_PREC_F32, PINN_PREC_BF16, PINN_PREC_BF16X3 = 0, 1, 2

EXPORTS = ["pinn_param_count", "pinn_tensor_count", "pinn_tensor_layout", "pinn_strerror",
           "pinn_create", "pinn_destroy", "pinn_last_error", "pinn_sample", "pinn_forward",
           "pinn_residual_fwd_bwd", "pinn_adam", "pinn_sample_dp", "pinn_adam_dp", "pinn_profile_enable", "pinn_profile_read",
           "pinn_debug_block_tiles", "pinn_adam_fused",
           "pinn_energy_R", "pinn_quadrature3d", "pinn_xchg_create", "pinn_xchg_local_handle", "pinn_xchg_connect",
           "pinn_xchg_allreduce", "pinn_xchg_set_timeout", "pinn_xchg_status", "pinn_xchg_destroy", "pinn_xchg_last_error"]


class PinnError(RuntimeError):
    pass


class Arch(C.Structure):
    _fields_ = [("H", C.c_int32), ("L", C.c_int32), ("HE", C.c_int32), ("HG", C.c_int32),
                ("act", C.c_int32), ("n_pass", C.c_int32), ("out_scale", C.c_float),
                ("precision", C.c_int32)]

    def key(self):
        return (self.H, self.L, self.HE, self.HG, self.act, self.n_pass, float(self.out_scale),
                self.precision)


class Ham(C.Structure):
    _fields_ = [("a", C.c_float), ("b", C.c_float), ("c", C.c_float), ("P", C.c_float),
                ("Ry", C.c_float), ("Rz", C.c_float), ("invN", C.c_float), ("invN1", C.c_float),
                ("invN2", C.c_float)]


_lib = None


def load_library():
    """Load libpinn_hip.so; raises PinnError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PinnError("HIP library %s is missing: run `python __graft_entry__.py build` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, fp, i64p = C.c_void_p, C.c_void_p, C.c_void_p  # device pointers travel as integers
    lib.pinn_param_count.restype = C.c_int64
    lib.pinn_param_count.argtypes = [C.POINTER(Arch)]
    lib.pinn_tensor_count.restype = C.c_int32
    lib.pinn_tensor_count.argtypes = [C.POINTER(Arch)]
    lib.pinn_tensor_layout.argtypes = [C.POINTER(Arch), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.pinn_strerror.restype = C.c_char_p
    lib.pinn_strerror.argtypes = [C.c_int]
    lib.pinn_create.argtypes = [C.POINTER(Arch), C.c_int, C.c_int64, C.POINTER(vp)]
    lib.pinn_destroy.argtypes = [vp]
    lib.pinn_last_error.restype = C.c_char_p
    lib.pinn_last_error.argtypes = [vp]
    lib.pinn_sample.argtypes = [vp, C.c_uint64, C.c_uint64, vp, C.c_int64, C.c_int64] + \
        [C.c_float] * 5 + [fp] * 4 + [vp, i64p, vp]
    lib.pinn_sample_dp.argtypes = [vp, C.c_uint64, C.c_uint64, vp, C.c_int64, C.c_int64] + \
        [C.c_float] * 5 + [fp] * 4 + [vp, i64p, fp, vp]
    lib.pinn_forward.argtypes = [vp] + [fp] * 5 + [C.c_int64, C.POINTER(Ham)] + [fp] * 4 + [vp]
    lib.pinn_energy_R.argtypes = [vp, fp, fp, C.c_int64, fp, fp, fp, vp]
    lib.pinn_quadrature3d.argtypes = [vp, fp, fp, fp, fp, fp, C.c_int64, C.c_int64, C.c_int64, fp, fp, vp]
    lib.pinn_residual_fwd_bwd.argtypes = [vp] + [fp] * 5 + [vp, C.c_int64, C.POINTER(Ham), i64p,
                                                             C.c_uint64, fp, vp]
    lib.pinn_adam.argtypes = [vp, fp, fp, fp, fp, vp] + [C.c_double] * 4 + [C.c_uint64, fp, fp, vp]
    lib.pinn_adam_fused.argtypes = [vp, fp, fp, fp, fp, vp] + [C.c_double] * 4 + [fp, fp, vp]
    lib.pinn_adam_dp.argtypes = [vp, fp, fp, fp, fp, vp] + [C.c_double] * 4 + [C.c_uint64, fp, fp, fp, i64p, vp]
    lib.pinn_profile_enable.argtypes = [vp, C.c_int]
    lib.pinn_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    lib.pinn_xchg_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(vp)]
    lib.pinn_xchg_local_handle.argtypes = [vp, C.c_char_p]
    lib.pinn_xchg_connect.argtypes = [vp, C.c_char_p]
    lib.pinn_xchg_allreduce.argtypes = [vp, fp, C.c_int64, vp]
    lib.pinn_xchg_set_timeout.argtypes = [vp, C.c_double]
    lib.pinn_xchg_status.argtypes = [vp, vp]
    lib.pinn_xchg_destroy.argtypes = [vp]
    lib.pinn_xchg_last_error.restype = C.c_char_p
    lib.pinn_xchg_last_error.argtypes = [vp]
    _lib = lib
    return lib


def param_count(arch):
    n = load_library().pinn_param_count(C.byref(arch))
    if n < 0:
        raise PinnError("pinn_param_count: %s" % load_library().pinn_strerror(int(n)).decode())
    return int(n)


def tensor_layout(arch):
    lib = load_library()
    nt = lib.pinn_tensor_count(C.byref(arch))
    if nt < 0:
        raise PinnError("pinn_tensor_count: %s" % lib.pinn_strerror(int(nt)).decode())
    off = (C.c_int64 * nt)()
    sz = (C.c_int64 * nt)()
    rc = lib.pinn_tensor_layout(C.byref(arch), off, sz)
    if rc:
        raise PinnError("pinn_tensor_layout: %s" % lib.pinn_strerror(rc).decode())
    return list(off), list(sz)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Handle:
    """Owns one pinn_handle (workspace for up to max_points points on one device)."""

    def __init__(self, arch, device_index, max_points):
        self.lib = load_library()
        self.arch = arch
        self.device_index = int(device_index)
        self.max_points = int(max_points)
        self.P = param_count(arch)
        h = C.c_void_p()
        rc = self.lib.pinn_create(C.byref(arch), self.device_index, self.max_points, C.byref(h))
        if rc:
            msg = self.lib.pinn_last_error(h).decode() if h else ""
            if h:
                self.lib.pinn_destroy(h)
            raise PinnError("pinn_create failed: %s %s" % (self.lib.pinn_strerror(rc).decode(), msg))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.pinn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc:
            raise PinnError("%s: %s (%s)" % (what, self.lib.pinn_strerror(rc).decode(),
                                             self.lib.pinn_last_error(self.h).decode()))

    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def sample(self, seed, step, first, n, box, Rlo, Rhi, cutoff, bc_cutoff, x, y, z, R, mask,
               counts, step_dev=None, counts_f32=None):
        """counts_f32 (float[2], optional): the set sizes as floats as well (pinn_sample_dp)"""
        self._check(self.lib.pinn_sample_dp(self.h, seed, step, _ptr(step_dev), first, n, box, Rlo, Rhi, cutoff,
                                            bc_cutoff, _ptr(x), _ptr(y), _ptr(z), _ptr(R), _ptr(mask),
                                            _ptr(counts), _ptr(counts_f32), self._stream()), "pinn_sample_dp")

    def forward(self, theta, x, y, z, R, n, ham, psi=None, E=None, lap=None, res=None):
        self._check(self.lib.pinn_forward(self.h, _ptr(theta), _ptr(x), _ptr(y), _ptr(z), _ptr(R),
                                          n, C.byref(ham), _ptr(psi), _ptr(E), _ptr(lap),
                                          _ptr(res), self._stream()), "pinn_forward")

    def energy_R(self, theta, R, n, E=None, dE=None, d2E=None):
        self._check(self.lib.pinn_energy_R(self.h, _ptr(theta), _ptr(R), n, _ptr(E), _ptr(dE),
                                           _ptr(d2E), self._stream()), "pinn_energy_R")

    def quadrature3d(self, u, v, wx, wy, wz, scratch, out):
        """out[0] = sum_ijk wx_i wy_j wz_k u_ijk v_ijk (v None: u squared); u, v fp32 device tensors of
        nx*ny*nz values, weights and scratch (nx) and out (1) fp64 device tensors"""
        self._check(self.lib.pinn_quadrature3d(self.h, _ptr(u), _ptr(v), _ptr(wx), _ptr(wy), _ptr(wz),
                                               wx.numel(), wy.numel(), wz.numel(), _ptr(scratch), _ptr(out),
                                               self._stream()), "pinn_quadrature3d")

    def residual_fwd_bwd(self, theta, x, y, z, R, mask, n, ham, bc_counts, trainable_mask, out):
        self._check(self.lib.pinn_residual_fwd_bwd(self.h, _ptr(theta), _ptr(x), _ptr(y), _ptr(z),
                                                   _ptr(R), _ptr(mask), n, C.byref(ham),
                                                   _ptr(bc_counts), C.c_uint64(trainable_mask),
                                                   _ptr(out), self._stream()),
                    "pinn_residual_fwd_bwd")

    def profile_enable(self, nslots):
        self._check(self.lib.pinn_profile_enable(self.h, int(nslots)), "pinn_profile_enable")

    def profile_read(self, ncalls):
        """mean duration (ms) of the dominant kernel over the last ncalls fwd_bwd calls"""
        ms = C.c_float()
        self._check(self.lib.pinn_profile_read(self.h, int(ncalls), C.byref(ms)), "pinn_profile_read")
        return float(ms.value)

    def adam_fused(self, theta, grad_out, m, v, step_count, lr, b1, b2, eps, best_theta=None, best_loss=None):
        """reduction + Adam in one launch, after residual_fwd_bwd(..., out=None) (include/pinn.h pinn_adam_fused)"""
        self._check(self.lib.pinn_adam_fused(self.h, _ptr(theta), _ptr(grad_out), _ptr(m), _ptr(v), _ptr(step_count),
                                             lr, b1, b2, eps, _ptr(best_theta), _ptr(best_loss), self._stream()),
                    "pinn_adam_fused")

    def adam(self, theta, grad, m, v, step_count, lr, b1, b2, eps, trainable_mask,
             best_theta=None, best_loss=None, msg_counts=None, counts_out=None):
        """msg_counts (float[2]) / counts_out (int64[2]), optional: pinn_adam_dp's conversion of the summed
        boundary-set sizes that travelled behind the gradient message"""
        self._check(self.lib.pinn_adam_dp(self.h, _ptr(theta), _ptr(grad), _ptr(m), _ptr(v),
                                          _ptr(step_count), lr, b1, b2, eps,
                                          C.c_uint64(trainable_mask), _ptr(best_theta),
                                          _ptr(best_loss), _ptr(msg_counts), _ptr(counts_out), self._stream()),
                    "pinn_adam_dp")


XCHG_HANDLE_BYTES = 64


class Exchange:
    """One-shot all-reduce over peer-mapped memory (include/pinn.h, pinn_xchg_*).  `gather` is a
    callable that takes this rank's 64-byte handle and returns all ranks' handles in rank order
    (torch.distributed.all_gather_object in DataParallelPlan)."""

    def __init__(self, device_index, rank, world, max_floats, gather):
        self.lib = load_library()
        x = C.c_void_p()
        rc = self.lib.pinn_xchg_create(int(device_index), int(rank), int(world), int(max_floats), C.byref(x))
        self.x = x if x else None
        self._check(rc, "pinn_xchg_create")
        mine = C.create_string_buffer(XCHG_HANDLE_BYTES)
        self._check(self.lib.pinn_xchg_local_handle(self.x, mine), "pinn_xchg_local_handle")
        handles = gather(bytes(mine.raw))
        if len(handles) != world or any(len(h) != XCHG_HANDLE_BYTES for h in handles):
            raise PinnError("exchange setup: expected %d handles of %d bytes" % (world, XCHG_HANDLE_BYTES))
        self._check(self.lib.pinn_xchg_connect(self.x, b"".join(handles)), "pinn_xchg_connect")
        self.max_floats = int(max_floats)

    def _check(self, rc, what):
        if rc:
            msg = self.lib.pinn_xchg_last_error(self.x).decode() if self.x else ""
            raise PinnError("%s: %s (%s)" % (what, self.lib.pinn_strerror(rc).decode(), msg))

    def allreduce(self, buf):
        """in-place SUM over ranks of a contiguous fp32 device tensor, on the current stream"""
        self._check(self.lib.pinn_xchg_allreduce(self.x, _ptr(buf), buf.numel(), Handle._stream()),
                    "pinn_xchg_allreduce")

    def set_timeout(self, seconds):
        """how long a call waits for the slowest peer before it gives up (NaN result + status)"""
        self._check(self.lib.pinn_xchg_set_timeout(self.x, float(seconds)), "pinn_xchg_set_timeout")

    def status(self):
        """synchronises the current stream; raises PinnError if any call so far timed out on a peer"""
        self._check(self.lib.pinn_xchg_status(self.x, Handle._stream()), "pinn_xchg_status")

    def close(self):
        if getattr(self, "x", None):
            self.lib.pinn_xchg_destroy(self.x)
            self.x = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
