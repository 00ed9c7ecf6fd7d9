/*
 * include/pinn.h -- C ABI of the MI355X-native PINN Schroedinger-residual hot path.
 *
 * The reference (slitvinov/PINN_for_quantum_wavefunction_surfaces) has no FFI or
 * operator interface: its hot path is inline Python over torch tensors.  This
 * header is therefore build-defined; every entry point
 * names the reference code it replaces.  All buffers are caller-owned DEVICE
 * pointers (fp32 unless noted); the library owns only the opaque handle and the
 * workspace sized at pinn_create.  All work is enqueued on the caller's HIP
 * stream (passed as void*, NULL = default stream); nothing here synchronises.
 * Every function returns 0 on success or a negative pinn_status; nothing
 * throws or exits.  A handle is not thread-safe; distinct handles are independent.
 * One handle = one stream at a time: the handle's scratch (per-point E, g and their adjoints,
 * per-block gradient partials, packed weights) is shared by every call on it, so two calls on the
 * same handle must be ordered on ONE stream (or by events); use one handle per concurrent stream.
 *
 * Parameter vector `theta` (P floats): tensors in NN_ion registration order
 * (poc/main.py:234-245) -- Lin_H1 .. Lin_H{L}, Lin_out, Lin_E1, Lin_E2, Lin_Eout,
 * netDecayL, netDecay -- each weight (out,in) row-major followed by its bias.
 * Tensor index t (for trainable_mask bit t) counts weight and bias separately,
 * i.e. optimizer param ids 0..15 of the reference checkpoints for L = 2.
 */
#ifndef PINN_H
#define PINN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pinn_handle pinn_handle;

typedef enum pinn_status {
    PINN_OK = 0,
    PINN_ERR_ARG = -1,          /* null pointer, bad size, bad enum */
    PINN_ERR_UNSUPPORTED = -2,  /* architecture / option has no kernel */
    PINN_ERR_HIP = -3,          /* a HIP runtime call failed (see pinn_last_error) */
    PINN_ERR_CAPACITY = -4      /* n exceeds the max_points given at create */
} pinn_status;

enum { PINN_ACT_SIGMOID = 0, PINN_ACT_TANH = 1 };
enum {
    PINN_PREC_F32 = 0,    /* fp32-grade, the parity path: fp32 elementwise algebra; the 16x16 trunk layer
                             and (on every setting) the 32x32 E-net layer run on the bf16 pipe with split
                             operands, three pieces = 24 bits in the forward sweep, fp32 accumulation.
                             PINN_TRUNK16_MFMA=0 / PINN_RNET_MFMA=0 select the exact-fp32 kernels */
    PINN_PREC_BF16 = 1,   /* bf16 MFMA operands, fp32 accumulate and elementwise */
    PINN_PREC_BF16X3 = 2  /* split-bf16 (hi/lo, 3 products), fp32 accumulate */
};

/* Network shape.  Reference: NN_ion.__init__ poc/main.py:225-245 (H=16, L=2,
 * HE=32, HG=10, sigmoid, n_pass=2, out_scale=1) and train.py:85-103 (same
 * widths, n_pass=1, out_scale=2 for `2*h`, train.py:47). */
typedef struct pinn_arch {
    int32_t H;         /* trunk width */
    int32_t L;         /* trunk hidden layers (>= 2) */
    int32_t HE;        /* E-net width, two hidden layers */
    int32_t HG;        /* gate width, one hidden layer */
    int32_t act;       /* PINN_ACT_* (all three sub-nets) */
    int32_t n_pass;    /* 2: base(f1,f2)+P*base(f2,f1) (poc/main.py:260); 1: train.py */
    float out_scale;   /* multiplies w_out.(sum of passes) */
    int32_t precision; /* PINN_PREC_* */
} pinn_arch;

/* Hamiltonian / loss constants.  res = a*lap(psi) + (b*(1/r1+1/r2) + c*E)*psi.
 * NN_ion (poc/main.py:118-120,345): (a,b,c) = (-0.5,-1,-1); train.py:54: (1,1,1).
 * invN, invN1, invN2 are the GLOBAL 1/N, 1/N_bc1, 1/N_bc2 (whole batch over all
 * data-parallel ranks), so per-rank outputs add up under one SUM all-reduce.
 * Ry, Rz (poc/main.py:241-242) must be 0 (as everywhere in the reference). */
typedef struct pinn_ham {
    float a, b, c;
    float P;           /* params['inversion_symmetry'] (poc/main.py:243) */
    float Ry, Rz;
    float invN, invN1, invN2;
} pinn_ham;

/* ---- layout helpers (host only, no device work) ------------------------ */
int64_t pinn_param_count(const pinn_arch *arch);
int32_t pinn_tensor_count(const pinn_arch *arch);            /* 2*(L+1) + 6 + 4 */
int pinn_tensor_layout(const pinn_arch *arch, int64_t *offsets, int64_t *sizes);
const char *pinn_strerror(int status);

/* ---- lifetime ----------------------------------------------------------- */
/* Allocates the workspace for batches of up to max_points points on `device`. */
int pinn_create(const pinn_arch *arch, int device, int64_t max_points, pinn_handle **out);
int pinn_destroy(pinn_handle *h);
const char *pinn_last_error(const pinn_handle *h);

/* ---- the hot path -------------------------------------------------------- */

/* Replaces the resampling block train.py:26-39 / sampling()+BC index lists
 * poc/main.py:124-156,391-393.  Point i of the global stream (first <= i <
 * first+n) is Philox4x32-10(counter = {lo(i),hi(i),lo(step),hi(step)}, key =
 * seed); x,y,z ~ U(-box,box), R ~ U(Rlo,Rhi); x := cutoff where either
 * pre-clamp radius < cutoff; mask bit0: r1 >= bc_cutoff, bit1: r2 >= bc_cutoff
 * (post-clamp).  counts (device int64[2], may be NULL) receives the two BC set
 * sizes of THIS call.  A rank of a data-parallel job passes its own `first`.
 * step_dev (device int32[1], may be NULL): when given, the Philox step is step + *step_dev,
 * read on the device -- pass pinn_adam's step counter so that a captured HIP graph of the whole
 * training step draws fresh points on every replay. */
int pinn_sample(pinn_handle *h, uint64_t seed, uint64_t step, const int32_t *step_dev,
                int64_t first, int64_t n, float box, float Rlo, float Rhi, float cutoff,
                float bc_cutoff, float *x, float *y, float *z, float *R, uint8_t *mask,
                int64_t *counts, void *stream);

/* pinn_sample for a rank of a data-parallel job (the reference is single device: nothing to cite beyond
 * pinn_sample's lines): counts_f32 (device float[2], may be NULL) also receives the two set sizes as floats --
 * exact below 2^24 -- so that the caller can point it at the two spare floats behind its [grad | loss terms]
 * message and all-reduce the sizes of the batch it has drawn ahead together with the gradient, without a
 * conversion kernel of its own (pinn_adam_dp turns the summed floats back into int64). */
int pinn_sample_dp(pinn_handle *h, uint64_t seed, uint64_t step, const int32_t *step_dev,
                   int64_t first, int64_t n, float box, float Rlo, float Rhi, float cutoff,
                   float bc_cutoff, float *x, float *y, float *z, float *R, uint8_t *mask,
                   int64_t *counts, float *counts_f32, void *stream);

/* Replaces NN_ion.forward + lapl/hamiltonian (poc/main.py:247-267,82-120,345)
 * and train.py:41-54 evaluated without a graph.  Outputs (each n floats, any
 * may be NULL): psi, E, lap = Laplacian(psi), res (the PDE residual; the
 * reference's hamiltonian() equals res + E*psi for formulation A). */
int pinn_forward(pinn_handle *h, const float *theta, const float *x, const float *y,
                 const float *z, const float *R, int64_t n, const pinn_ham *ham,
                 float *psi, float *E, float *lap, float *res, void *stream);

/* E(R) with its first and second R-derivative (each n floats, any may be NULL).  Replaces
 * the reference's autograd over the E-net for the force curves: F = -d(E + 1/(2R))/dR and
 * dF/dR by grad(..., create_graph=True) twice (poc/main.py:249-253,1316-1328). */
int pinn_energy_R(pinn_handle *h, const float *theta, const float *R, int64_t n, float *E,
                  float *dE, float *d2E, void *stream);

/* Nested composite-Simpson quadrature of a product of two grid functions (integra3d, poc/main.py:179-186, as
 * energy_from_psi / dEdR_int use it, poc/main.py:438-464,639-676):
 *   *out = sum_ijk wx[i] wy[j] wz[k] u[(i*ny + j)*nz + k] * v[(i*ny + j)*nz + k]
 * with u, v the fp32 per-point outputs of pinn_forward on the grid (v == NULL: u squared), the weights the fp64
 * weight vectors of the 1-D rule along each axis, accumulation in fp64, a fixed summation order (bitwise
 * reproducible).  scratch: nx doubles of device memory (caller-owned, like everything else); out: one device
 * double.  Replaces the reference's round trip of psi and H psi to the host and scipy's simps over Python lists. */
int pinn_quadrature3d(pinn_handle *h, const float *u, const float *v, const double *wx, const double *wy,
                      const double *wz, int64_t nx, int64_t ny, int64_t nz, double *scratch, double *out,
                      void *stream);

/* Replaces LossFunctions + Ltot.backward() (poc/main.py:341-355,403;
 * train.py:41-57,71): one fused pass computing the loss terms and d(Ltot)/d(theta).
 * out (P+3 floats; NULL: leave the per-block partial sums in the handle for pinn_adam_fused): gradient, then invN*sum res^2, invN1*sum_{bc1} psi^2,
 * invN2*sum_{bc2} psi^2 (so Ltot = out[P]+out[P+1]+out[P+2]).  mask as written
 * by pinn_sample (NULL = no boundary points).  Tensors whose trainable_mask bit
 * is 0 get a zero gradient (freezeBase/freezeDecayUnit, poc/main.py:305-319);
 * when no trunk tensor is trainable the trunk reverse pass is skipped.
 * bc_counts (device int64[2], may be NULL): when given, invN1 = 1/bc_counts[0]
 * and invN2 = 1/bc_counts[1] are taken from device memory instead of `ham`
 * (0 for an empty set -- the reference's mean over an empty index list is NaN,
 * poc/main.py:349-350; unreachable for n >= 100).  This is
 * how a training loop stays free of host syncs: pinn_sample's counts, summed
 * over ranks on the stream, feed this call directly. */
int pinn_residual_fwd_bwd(pinn_handle *h, const float *theta, const float *x, const float *y,
                          const float *z, const float *R, const uint8_t *mask, int64_t n,
                          const pinn_ham *ham, const int64_t *bc_counts,
                          uint64_t trainable_mask, float *out, void *stream);

/* Replaces torch.optim.Adam.step (train.py:23,72; poc/main.py:363,404; no
 * amsgrad, weight_decay 0) fused with train.py:58-60 best-so-far tracking:
 * if Ltot = grad[P]+grad[P+1]+grad[P+2] < *best_loss, theta (pre-update) is
 * copied to best_theta and *best_loss updated (both may be NULL).  step_count
 * (device int32[1]) is incremented first and used for the bias correction.
 * The hyper-parameters are doubles: like torch, the scalars 1-beta, lr/(1-beta1^t),
 * sqrt(1-beta2^t) are formed in double and rounded to fp32 once. */
int pinn_adam(pinn_handle *h, float *theta, const float *grad, float *m, float *v,
              int32_t *step_count, double lr, double beta1, double beta2, double eps,
              uint64_t trainable_mask, float *best_theta, float *best_loss, void *stream);

/* Reduction and Adam in ONE launch for the single-device step (no exchange between the two; same reference lines as
 * pinn_adam): call pinn_residual_fwd_bwd with out == NULL -- the per-block partial sums then stay in the handle -- and
 * this instead of pinn_adam.  grad_out (device float[P + 3], may be NULL) receives what pinn_residual_fwd_bwd would have
 * written to out.  Bitwise the same gradient, losses and parameters as the two-call form (same additions, same order);
 * the frozen-tensor mask is the one given to that pinn_residual_fwd_bwd call.  PINN_ERR_ARG if no such call precedes it. */
int pinn_adam_fused(pinn_handle *h, float *theta, float *grad_out, float *m, float *v, int32_t *step_count,
                    double lr, double beta1, double beta2, double eps, float *best_theta, float *best_loss,
                    void *stream);

/* pinn_adam for a rank of a data-parallel job: the step that consumes the all-reduced message also stores its two
 * trailing floats (msg_counts: the global boundary-set sizes of the batch drawn ahead, summed over ranks; exact
 * below 2^24) as int64 in counts_out, where the next pinn_residual_fwd_bwd takes them as bc_counts.  Both may be
 * NULL (then identical to pinn_adam). */
int pinn_adam_dp(pinn_handle *h, float *theta, const float *grad, float *m, float *v,
                 int32_t *step_count, double lr, double beta1, double beta2, double eps,
                 uint64_t trainable_mask, float *best_theta, float *best_loss,
                 const float *msg_counts, int64_t *counts_out, void *stream);

/* ---- measurement --------------------------------------------------------- */
/* With nslots > 0, every pinn_residual_fwd_bwd call brackets its dominant
 * kernel (the fused trunk kernel) with a HIP event pair on the launch stream,
 * cycling through nslots (<= 1024) pairs; nslots = 0 switches it off (default).
 * pinn_profile_read waits for the pairs of the last `ncalls` calls and returns
 * their mean elapsed milliseconds (bench.py: live roofline figure over the
 * timed region; costs two event records per call, no synchronisation). */
int pinn_profile_enable(pinn_handle *h, int nslots);
int pinn_profile_read(pinn_handle *h, int ncalls, float *mean_trunk_ms);

/* ---- data-parallel exchange (the reference is single device) ------------
 * One-shot all-reduce (SUM, fp32) of the per-step message [grad | loss terms | counts] over
 * peer-mapped memory: every rank writes its vector into a slot in every peer's memory (one xGMI hop,
 * all links at once) and sums the N slots locally in rank order, so all ranks obtain bitwise
 * identical sums.  Two kernels on the caller's stream per call, no host synchronisation.
 * Setup (once, host side): every rank creates its exchange, the ranks trade the 64-byte handles of
 * pinn_xchg_local_handle by any means (the Python layer uses torch.distributed.all_gather_object),
 * and each calls pinn_xchg_connect with all N handles in rank order (its own entry is ignored).
 * Every rank must make the same sequence of pinn_xchg_allreduce calls.  buf: device, 16-byte aligned,
 * n <= max_floats, summed in place.  The call counter lives in device memory, so a call captured into a
 * HIP graph advances with every replay like an eager call (all ranks must then replay the same graph).
 * A peer that does not arrive within the timeout (default 20 s; PINN_XCHG_TIMEOUT_S in the environment at
 * create, or pinn_xchg_set_timeout before the calls it should apply to) makes the call fill buf with NaN
 * -- a local-only gradient is never handed back as if it were the sum -- and sets a sticky status word
 * that pinn_xchg_status (which synchronises the stream) reports as PINN_ERR_HIP. */
typedef struct pinn_xchg pinn_xchg;
#define PINN_XCHG_HANDLE_BYTES 64
int pinn_xchg_create(int device, int rank, int world, int64_t max_floats, pinn_xchg **out);
int pinn_xchg_local_handle(pinn_xchg *x, void *handle64);
int pinn_xchg_connect(pinn_xchg *x, const void *handles /* world x 64 bytes, rank order */);
int pinn_xchg_allreduce(pinn_xchg *x, float *buf, int64_t n, void *stream);
int pinn_xchg_set_timeout(pinn_xchg *x, double seconds);
int pinn_xchg_status(pinn_xchg *x, void *stream);
int pinn_xchg_destroy(pinn_xchg *x);
const char *pinn_xchg_last_error(const pinn_xchg *x);

/* ---- diagnostics (host only, no GPU call) -------------------------------------------------
 * The tile -> block map of the trunk kernels (csrc/pinn_common.h block_tiles_of): block b of a grid of `grid` blocks takes
 * the tiles out[0], out[0] + out[2], ... below out[1].  nclass = 0 (equal shares), 2 or 4 with bounds[nclass - 1] upper
 * tile bounds in 16-point tiles; unit = points per tile / 16.  For the CPU test that checks that every tile is taken
 * exactly once whatever the shares. */
int pinn_debug_block_tiles(int grid, int block, int64_t ntile, int nclass, const int64_t *bounds, int unit,
                           int64_t out[3]);

#ifdef __cplusplus
}
#endif
#endif
