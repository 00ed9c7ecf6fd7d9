// pinn_kernels.hip -- gfx950 kernels and the C ABI of include/pinn.h (fp32 parity path).
//
// Kernel map:
//   k_sample      K3  Philox box sampler + nucleus clamp + BC masks/counts (train.py:26-39)
//   k_rnet_fwd        E(R), g(R) per point: the two R-only sub-nets (poc/main.py:249-253,262-264)
//   k_trunk       K1/K2 fused features -> 5-stream trunk MLP -> psi, lap psi, residual, loss
//                     seeds -> full reverse pass; weight gradients contracted over the wave's
//                     64 points on the f32 MFMA pipe (train.py:41-57,71)
//   k_rnet_bwd        reverse pass of the R-only sub-nets from per-point dL/dE, dL/dg
//   k_reduce          fixed-order second-stage reduction of per-block partials (deterministic)
//   k_adam        K4  Adam + best-so-far snapshot (train.py:23,58-60,72)
//   k_energy_R        E, dE/dR, d2E/dR2 of the E-net (the force curves, poc/main.py:1316-1328)
//
// Execution shape: one collocation point per lane (wave64).  Weights are read with wave-uniform
// addresses, so hipcc keeps them in SGPRs (s_load through the scalar cache); every mat-vec is a
// sequence of rank-1 updates on neuron PAIRS, v_pk_fma_f32 v[acc pair], s[weight pair], v[x]
// (the fp32 vector peak is the packed rate), with the weight rows double-buffered in SGPRs.
// The only cross-lane work is the weight-gradient contraction over points.  It is not done on the
// VALU: each wave transposes (adjoint, input) vectors through a private LDS tile and issues
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains), accumulating 16x16 gradient tiles in registers
// for the whole kernel; vector gradients are lane-local running sums reduced once at the end.
// The R-net's 32x32 layer runs on v_mfma_f32_32x32x2_f32 with the matrix held in VGPRs.
// Per-block partials are written once and summed in fixed order: results are bitwise reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "../../include/pinn.h"
#include "pinn_math.h"
#include "pinn_common.h"

using namespace pinn;

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kPitch = 68;              // floats per staged row: 64 points + 4 pad (272 B, 16-B aligned)

// ---------------------------------------------------------------- wave tiles
// D[16x16] += sum over the wave's 64 points of U[j][pt] * V[i][pt].
// Staging image: row r of a vector (one value per lane) at buf[r*kPitch + lane].
// Fragment of lane (g = lane>>4, c = lane&15): the 16 consecutive points
// 16g..16g+15 of row c -> MFMA step t contracts the four points {16g+t}.
__device__ __forceinline__ void stage_row(float* buf, int row, float v, int lane)
{
    buf[row * kPitch + lane] = v;
}

template <int N>
__device__ __forceinline__ void stage_rows(float* buf, int row0, const float (&v)[N], int lane)
{
#pragma unroll
    for (int j = 0; j < N; j++) buf[(row0 + j) * kPitch + lane] = v[j];
}

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS operations of one wave execute in issue order; this only stops the
    // compiler from moving the fragment reads above the staging writes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Frag {
    float v[16];
};

__device__ __forceinline__ void load_frag(const float* buf, int rowblock, int lane, Frag& f)
{
    const int c = lane & 15, g = lane >> 4;
    const float4* p = reinterpret_cast<const float4*>(buf + (rowblock * 16 + c) * kPitch + 16 * g);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float4 t = p[q];
        f.v[4 * q + 0] = t.x;
        f.v[4 * q + 1] = t.y;
        f.v[4 * q + 2] = t.z;
        f.v[4 * q + 3] = t.w;
    }
}

// fragment of a row that is constant over points: value on row `row`, 0 elsewhere
__device__ __forceinline__ void const_frag(int row, float value, int lane, Frag& f)
{
    const float x = (lane & 15) == row ? value : 0.0f;
#pragma unroll
    for (int t = 0; t < 16; t++) f.v[t] = x;
}

__device__ __forceinline__ void mma(f32x4& acc, const Frag& a, const Frag& b)
{
#pragma unroll
    for (int t = 0; t < 16; t++)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[t], b.v[t], acc, 0, 0, 0);
}

// One MFMA step of several independent accumulator chains at a time: a 16x16x4 f32 MFMA can issue
// every 32 cycles but its result is only available after 40, so back-to-back steps of ONE chain
// stall the (in-order) wave; stepping all tiles of a call together keeps the pipe busy.
template <int NA, int NB>
__device__ __forceinline__ void mma_grid(f32x4 (&acc)[NA][NB], const Frag (&a)[NA], const Frag (&b)[NB])
{
#pragma unroll
    for (int t = 0; t < 16; t++)
#pragma unroll
        for (int i = 0; i < NA; i++)
#pragma unroll
            for (int j = 0; j < NB; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].v[t], b[j].v[t], acc[i][j], 0, 0, 0);
}

template <int NA>
__device__ __forceinline__ void mma_col(f32x4 (&acc)[NA], const Frag (&a)[NA], const Frag& b)
{
#pragma unroll
    for (int t = 0; t < 16; t++)
#pragma unroll
        for (int i = 0; i < NA; i++)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].v[t], b.v[t], acc[i], 0, 0, 0);
}

// tile element (row = 4*(lane>>4) + r, col = lane&15) lives in acc[r]
__device__ __forceinline__ void tile_store(const f32x4& acc, float* dst, int row_stride,
                                           int col_stride, int nrow, int ncol, int lane, bool add)
{
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * g + r;
        if (row < nrow && c < ncol) {
            float* p = dst + row * row_stride + c * col_stride;
            *p = add ? *p + acc[r] : acc[r];
        }
    }
}

// Weights are loop-invariant, so LICM would hoist all P scalar loads out of the
// persistent point loop and spill ~P SGPRs.  Passing the pointer through an empty
// asm offset per phase keeps each phase's s_loads next to their use.
__device__ __forceinline__ const float* launder(const float* __restrict__ p)
{
    int z;  // an opaque zero: keeps the pointer's address space / noalias provenance
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return p + z;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// thT = theta with the forward-contracted matrices stored input-major (see pinn_math.h):
// W_0 as [2][H], W_l (l >= 1) and WE2 transposed; everything else copied.
__global__ void k_prep_theta(const float* __restrict__ th, float* __restrict__ thT, int H, int L,
                             int HE, int PT, int P)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    int src = k;
    if (k < 2 * H) {                       // thT[c*H + j] = W0[j][c]
        const int c = k / H, j = k % H;
        src = 2 * j + c;
    } else if (k < PT) {
        const int per = H * H + H;
        const int r = k - 3 * H;           // offset into the (W_l, b_l) blocks, l >= 1
        if (r >= 0 && r < (L - 1) * per) {
            const int l = r / per, o = r % per;
            if (o < H * H) src = 3 * H + l * per + (o % H) * H + o / H;
        }
    } else {
        const int o = k - PT - 2 * HE;     // WE2 block of the R-net
        if (o >= 0 && o < HE * HE) src = PT + 2 * HE + (o % HE) * HE + o / HE;
    }
    thT[k] = th[src];
}

// ------------------------------------------------------------- trunk kernel
template <int H, int L>
struct TrunkSink {
    static constexpr int HT = H / 16;
    float* U;
    float* V;
    int lane;
    f32x4 tW[L - 1][HT][HT];  // W_l-bar (l = 1..L-1): contraction over the wave's points on the MFMA pipe
    // vector gradients: lane-local running sums over this lane's points, reduced once at the end
    float accB[L][H];         // b_l-bar
    float accW0[H][2];        // W_0-bar
    float accOut[H];          // wout-bar

    __device__ __forceinline__ void zero()
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < L - 1; l++)
#pragma unroll
            for (int a = 0; a < HT; a++)
#pragma unroll
                for (int b = 0; b < HT; b++) tW[l][a][b] = z;
#pragma unroll
        for (int j = 0; j < H; j++) {
#pragma unroll
            for (int l = 0; l < L; l++) accB[l][j] = 0.f;
            accW0[j][0] = 0.f;
            accW0[j][1] = 0.f;
            accOut[j] = 0.f;
        }
    }

    __device__ __forceinline__ void out_weight(const float (&t)[H])
    {
#pragma unroll
        for (int j = 0; j < H; j++) accOut[j] += t[j];
    }

    __device__ __forceinline__ void weight(int l, int s, const float (&ab)[H], const float (&oin)[H])
    {
        stage_rows<H>(U, 0, ab, lane);
        stage_rows<H>(V, 0, oin, lane);
        wave_lds_sync();
        Frag fa[HT], fb[HT];
#pragma unroll
        for (int a = 0; a < HT; a++) load_frag(U, a, lane, fa[a]);
#pragma unroll
        for (int b = 0; b < HT; b++) load_frag(V, b, lane, fb[b]);
        mma_grid<HT, HT>(tW[l - 1], fa, fb);
        if (s == 0) {
#pragma unroll
            for (int j = 0; j < H; j++) accB[l][j] += ab[j];
        }
    }

    __device__ __forceinline__ void weight0(int s, const float (&ab)[H], float in0, float in1)
    {
#pragma unroll
        for (int j = 0; j < H; j++) {
            accW0[j][0] += ab[j] * in0;
            accW0[j][1] += ab[j] * in1;
            if (s == 0) accB[0][j] += ab[j];
        }
    }

    // add this wave's contribution into the block gradient image g (theta order)
    __device__ __forceinline__ void dump(float* g, bool add)
    {
        using Lay = TrunkLayout<H, L>;
#pragma unroll
        for (int l = 1; l < L; l++)
#pragma unroll
            for (int a = 0; a < HT; a++)
#pragma unroll
                for (int b = 0; b < HT; b++)
                    tile_store(tW[l - 1][a][b], g + Lay::W(l) + (16 * a) * H + 16 * b, H, 1, 16, 16,
                               lane, add);
#pragma unroll
        for (int j = 0; j < H; j++) {
            float v[L + 3];
#pragma unroll
            for (int l = 0; l < L; l++) v[l] = wave_sum(accB[l][j]);
            v[L] = wave_sum(accW0[j][0]);
            v[L + 1] = wave_sum(accW0[j][1]);
            v[L + 2] = wave_sum(accOut[j]);
            if (lane == 0) {
#pragma unroll
                for (int l = 0; l < L; l++) {
                    float* p = g + Lay::b(l) + j;
                    *p = add ? *p + v[l] : v[l];
                }
                float* p0 = g + Lay::W(0) + 2 * j;
                p0[0] = add ? p0[0] + v[L] : v[L];
                p0[1] = add ? p0[1] + v[L + 1] : v[L + 1];
                float* po = g + Lay::Wout + j;
                *po = add ? *po + v[L + 2] : v[L + 2];
            }
        }
    }
};

// one wave per SIMD: the packed rank-1 formulation keeps ~300 values live; at the 256-register
// cap of two waves it spills 700 B/lane and runs 1.4x slower (VALU throughput does not depend on
// occupancy: a wave64 VALU instruction occupies its SIMD for 4 cycles either way)
#ifndef PINN_TRUNK_MIN_WAVES
#define PINN_TRUNK_MIN_WAVES 1
#endif
template <int H, int L, int ACT, int NPASS, int MODE>
__global__ __launch_bounds__(kBlock, PINN_TRUNK_MIN_WAVES) void k_trunk(const float* __restrict__ th,
                                                      const float* __restrict__ thT, TrunkArgs A)
{
    using Lay = TrunkLayout<H, L>;
    constexpr int PT = Lay::size;
    __shared__ __attribute__((aligned(16))) float smem[kWavesPerBlock * 2 * H * kPitch + PT + 16];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* gimg = smem + kWavesPerBlock * 2 * H * kPitch;  // block gradient image + 4 scalars/wave

    TrunkSink<H, L> sink;
    sink.U = smem + wave * 2 * H * kPitch;
    sink.V = sink.U + H * kPitch;
    sink.lane = lane;
    if (MODE == MODE_BWD) {
        sink.zero();
        for (int k = lane; k < 2 * H * kPitch; k += kWave) sink.U[k] = 0.0f;
    }
    float acc_bout = 0.f, acc_pde = 0.f, acc_bc1 = 0.f, acc_bc2 = 0.f;
    float invN1 = A.invN1, invN2 = A.invN2;
    if (MODE != MODE_FWD && A.bc_counts) {
        const long long c1 = A.bc_counts[0], c2 = A.bc_counts[1];
        invN1 = c1 > 0 ? 1.0f / (float)c1 : 0.0f;
        invN2 = c2 > 0 ? 1.0f / (float)c2 : 0.0f;
    }

    for (int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * kWave; base < A.n;
         base += (int64_t)gridDim.x * kBlock) {
        const int64_t idx = base + lane;
        const bool valid = idx < A.n;
        const int64_t id = valid ? idx : A.n - 1;
        const float xv = A.x[id], yv = A.y[id], zv = A.z[id], Rv = A.R[id];
        const float Ev = A.E[id], gv = A.g[id];
        float m1 = 0.f, m2 = 0.f;
        if (MODE != MODE_FWD && A.mask) {
            const uint8_t mk = A.mask[id];
            m1 = float(mk & 1);
            m2 = float((mk >> 1) & 1);
        }
        float inA[2][5], inB[2][5], ri1, ri2;
        orbital<float>(xv - Rv, yv, zv, inA[0], ri1);
        orbital<float>(xv + Rv, yv, zv, inA[1], ri2);
#pragma unroll
        for (int s = 0; s < 5; s++) { inB[0][s] = inA[1][s]; inB[1][s] = inA[0][s]; }
        const float F = inA[0][0] + A.P * inA[1][0];
        const float lapF = inA[0][4] + A.P * inA[1][4];

        TrunkPass<H, L, ACT, float> tp;
        Tail<float> t;
        float SvA = 0.f, SlA = 0.f;
        // NPASS == 2 stages: 0 = forward(f1,f2); 1 = forward(f2,f1) + tail + reverse of that
        // pass; 2 = forward(f1,f2) again + its reverse.  One copy of the code, three trips:
        // the second pass's state replaces the first's, so the live set is one pass.
        constexpr int nstage = NPASS == 2 ? (MODE == MODE_BWD ? 3 : 2) : 1;
#pragma unroll 1
        for (int st = 0; st < nstage; st++) {
            const bool useB = NPASS == 2 && st == 1;
            float in[2][5];
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int s = 0; s < 5; s++) in[c][s] = useB ? inB[c][s] : inA[c][s];
            float Sv = 0.f, Sl = 0.f;
            tp.forward(launder(th), launder(thT), in, Sv, Sl);
            const bool do_tail = NPASS == 1 || st == 1;
            if (NPASS == 2 && st == 0) { SvA = Sv; SlA = Sl; }
            if (do_tail) {
                if (NPASS == 2) { Sv = SvA + A.P * Sv; Sl = SlA + A.P * Sl; }
                const float N = A.out_scale * Sv + th[Lay::bout];
                const float lapN = A.out_scale * Sl;
                tail<float>(N, lapN, gv, Ev, F, lapF, ri1 + ri2, A.a, A.b, A.c, m1, m2, A.invN,
                            invN1, invN2, valid ? 1.0f : 0.0f, t);
                if (MODE == MODE_FWD) {
                    if (valid) {
                        if (A.psi) A.psi[idx] = t.psi;
                        if (A.lap) A.lap[idx] = t.lap;
                        if (A.res) A.res[idx] = t.res;
                    }
                } else {
                    if (valid) { A.Eb[idx] = t.Eb; A.gb[idx] = t.gb; }
                    acc_pde += t.l_pde;
                    acc_bc1 += t.l_bc1;
                    acc_bc2 += t.l_bc2;
                    acc_bout += t.Nb;
                }
            }
            if (MODE == MODE_BWD && (NPASS == 1 || st >= 1)) {
                const float pw = useB ? A.P : 1.0f;
                tp.backward(launder(th), launder(thT), in, A.out_scale * pw * t.Nb, A.out_scale * pw * t.lapNb, sink);
            }
        }
    }

    if (MODE == MODE_FWD) return;
    // ---- block reduction in fixed wave order, then one partial row per block
    float* sc = gimg + PT;
    const float s0 = wave_sum(acc_pde), s1 = wave_sum(acc_bc1), s2 = wave_sum(acc_bc2),
                s3 = wave_sum(acc_bout);
    for (int w = 0; w < kWavesPerBlock; w++) {
        if (wave == w) {
            if (MODE == MODE_BWD) sink.dump(gimg, w != 0);
            if (lane == 0) {
                if (w == 0) { sc[0] = s0; sc[1] = s1; sc[2] = s2; sc[3] = s3; }
                else { sc[0] += s0; sc[1] += s1; sc[2] += s2; sc[3] += s3; }
            }
        }
        __syncthreads();
    }
    float* out = A.partial + (size_t)blockIdx.x * A.stride;
    if (MODE == MODE_BWD) {
        for (int k = threadIdx.x; k < PT; k += kBlock) out[k] = k == Lay::bout ? sc[3] : gimg[k];
    } else {
        for (int k = threadIdx.x; k < PT; k += kBlock) out[k] = 0.0f;
    }
    if (threadIdx.x < 3) out[PT + threadIdx.x] = sc[threadIdx.x];
}

// -------------------------------------------------------------- R-only nets
// ---------------------------------------------------------- 32x32 mat-vec on the matrix pipe
// y[j] = sum_i M[j][i] x[i] for each lane's own point (x, y: one value per lane per neuron).
// The 32x32 matrix is the A operand of v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains) and lives
// in 16 VGPRs for the whole kernel: lane l holds M[l&31][2t + (l>>5)], t = 0..15.  The B operand
// of step t is (x[2t] of points 0..31 | x[2t+1] of points 0..31): one v_permlane32_swap of the
// registers x[2t], x[2t+1] yields it for points 0..31 AND, in its second result, for points
// 32..63.  The two 32x32 results come back to lane = point order with 16 more swaps.  No scalar
// weight stream, no LDS: the VALU only does the activations.
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Mat32 {
    float w[16];
};

// v_permlane32_swap: lanes 32..63 of a <-> lanes 0..31 of b.  Inline asm on purpose: hipcc
// (ROCm 7.2) lowers the SECOND result of __builtin_amdgcn_permlane32_swap to the same register as
// the first (checked on hardware with a stand-alone integer test).  The s_nops are the
// VALU-write -> permlane-read and permlane-write -> MFMA-read wait states, which the compiler does
// not pad inside an asm statement.
__device__ __forceinline__ void swap32(float& a, float& b)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

__device__ __forceinline__ void load_mat32(const float* __restrict__ W, bool transpose, int lane, Mat32& m)
{
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int r = lane & 31, k = 2 * t + (lane >> 5);
        m.w[t] = transpose ? W[k * 32 + r] : W[r * 32 + k];
    }
}

__device__ __forceinline__ void matvec32(const Mat32& m, const float (&x)[32], float (&y)[32])
{
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; r++) { a0[r] = 0.f; a1[r] = 0.f; }
#pragma unroll
    for (int t = 0; t < 16; t++) {
        float b0 = x[2 * t], b1 = x[2 * t + 1];
        swap32(b0, b1);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(m.w[t], b0, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(m.w[t], b1, a1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float q0 = a0[r], q1 = a1[r];
        swap32(q0, q1);
        const int j = (r & 3) + 8 * (r >> 2);
        y[j] = q0;
        y[j + 4] = q1;
    }
}

// E(R) and g(R) with the 32x32 layer on the matrix pipe (HE = 32).  Same mathematics as
// pinn::Rnet (poc/main.py:249-253,262-264; train.py:48-52).
template <int HG, int ACT>
struct RnetM32 {
    static constexpr int HE = 32;
    float h1[HE], h2[HE], hg[HG];

    __device__ __forceinline__ void forward(const float* __restrict__ th, const Mat32& M, float R, float& E, float& g)
    {
        constexpr RnetLayout Lo{HE, HG};
        float s1, s2;
#pragma unroll
        for (int c0 = 0; c0 < HE; c0 += 16) {
            float w1[16], b1[16];
#pragma unroll
            for (int j = 0; j < 16; j++) { w1[j] = th[Lo.WE1() + c0 + j]; b1[j] = th[Lo.bE1() + c0 + j]; }
            PINN_ROW_FENCE();
#pragma unroll
            for (int j = 0; j < 16; j++) act012<ACT>(w1[j] * R + b1[j], h1[c0 + j], s1, s2);
            PINN_ROW_FENCE();
        }
        {
            float wg1[HG], bg1[HG], wg2[HG];
#pragma unroll
            for (int j = 0; j < HG; j++) { wg1[j] = th[Lo.WG1() + j]; bg1[j] = th[Lo.bG1() + j]; wg2[j] = th[Lo.WG2() + j]; }
            float gg = th[Lo.bG2()];
            PINN_ROW_FENCE();
#pragma unroll
            for (int j = 0; j < HG; j++) {
                act012<ACT>(wg1[j] * R + bg1[j], hg[j], s1, s2);
                gg += wg2[j] * hg[j];
            }
            g = gg;
            PINN_ROW_FENCE();
        }
        float pre[HE];
        matvec32(M, h1, pre);
        float e = th[Lo.bE3()];
#pragma unroll
        for (int c0 = 0; c0 < HE; c0 += 16) {
            float b2[16], w3[16];
#pragma unroll
            for (int j = 0; j < 16; j++) { b2[j] = th[Lo.bE2() + c0 + j]; w3[j] = th[Lo.WE3() + c0 + j]; }
            PINN_ROW_FENCE();
#pragma unroll
            for (int j = 0; j < 16; j++) {
                act012<ACT>(pre[c0 + j] + b2[j], h2[c0 + j], s1, s2);
                e += w3[j] * h2[c0 + j];
            }
            PINN_ROW_FENCE();
        }
        E = e;
    }

    template <class Sink>
    __device__ __forceinline__ void backward(const float* __restrict__ th, const Mat32& MT, float R, float Eb,
                                             float gb, Sink& sink)
    {
        constexpr RnetLayout Lo{HE, HG};
        float a2[HE], t3[HE];
#pragma unroll
        for (int c0 = 0; c0 < HE; c0 += 16) {
            float w3[16];
#pragma unroll
            for (int j = 0; j < 16; j++) w3[j] = th[Lo.WE3() + c0 + j];
            PINN_ROW_FENCE();
#pragma unroll
            for (int j = 0; j < 16; j++) {
                t3[c0 + j] = Eb * h2[c0 + j];
                a2[c0 + j] = Eb * w3[j] * act1_from_value<ACT>(h2[c0 + j]);
            }
            PINN_ROW_FENCE();
        }
        sink.e3(t3, Eb);
        sink.e2(a2, h1);
        float a1[HE];
        matvec32(MT, a2, a1);
#pragma unroll
        for (int i = 0; i < HE; i++) a1[i] *= act1_from_value<ACT>(h1[i]);
        sink.e1(a1, R);
        float tg[HG], ag[HG];
        {
            float wg2[HG];
#pragma unroll
            for (int j = 0; j < HG; j++) wg2[j] = th[Lo.WG2() + j];
            PINN_ROW_FENCE();
#pragma unroll
            for (int j = 0; j < HG; j++) {
                tg[j] = gb * hg[j];
                ag[j] = gb * wg2[j] * act1_from_value<ACT>(hg[j]);
            }
            PINN_ROW_FENCE();
        }
        sink.g2(tg, gb);
        sink.g1(ag, R);
    }
};

template <int HE, int HG, int ACT>
__global__ __launch_bounds__(kBlock) void k_rnet_fwd(const float* __restrict__ th_r,
                                                     const float* __restrict__ thT_r,
                                                     const float* __restrict__ R, int64_t n,
                                                     float* __restrict__ E, float* __restrict__ g,
                                                     float* __restrict__ E_user)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Mat32 M;
    if (HE == 32) {
        constexpr RnetLayout Lo{HE, HG};
        load_mat32(th_r + Lo.WE2(), false, lane, M);
    }
    // whole waves stay in the loop together (the matrix-pipe path needs all 64 lanes)
    for (int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * kWave; base < n;
         base += (int64_t)gridDim.x * kBlock) {
        const int64_t i = base + lane;
        const bool valid = i < n;
        const float Rv = R[valid ? i : n - 1];
        float e, gg;
        if constexpr (HE == 32) {
            RnetM32<HG, ACT> rn;
            rn.forward(launder(th_r), M, Rv, e, gg);
        } else {
            Rnet<HE, HG, ACT, float> rn;
            rn.forward(launder(th_r), launder(thT_r), Rv, e, gg);
        }
        if (valid) {
            E[i] = e;
            g[i] = gg;
            if (E_user) E_user[i] = e;
        }
    }
}

template <int HE, int HG>
struct RnetSink {
    static constexpr int ET = (HE + 15) / 16;
    float* U;
    float* V;
    int lane;
    f32x4 tE2[ET][ET], tbE2[ET], tE3[ET], tE1[ET], tG2, tG1[1];
    float bE3, bG2;

    __device__ __forceinline__ void zero()
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < ET; a++) {
            tbE2[a] = z; tE3[a] = z; tE1[a] = z;
#pragma unroll
            for (int b = 0; b < ET; b++) tE2[a][b] = z;
        }
        tG2 = z; tG1[0] = z;
        bE3 = 0.f; bG2 = 0.f;
    }
    __device__ __forceinline__ void e3(const float (&t3)[HE], float Eb)
    {
        stage_rows<HE>(U, 0, t3, lane);
        wave_lds_sync();
        Frag one, fa[ET];
        const_frag(0, 1.0f, lane, one);
#pragma unroll
        for (int a = 0; a < ET; a++) load_frag(U, a, lane, fa[a]);
        mma_col<ET>(tE3, fa, one);
        bE3 += Eb;
    }
    __device__ __forceinline__ void e2(const float (&a2)[HE], const float (&h1)[HE])
    {
        stage_rows<HE>(U, 0, a2, lane);
        stage_rows<HE>(V, 0, h1, lane);
        wave_lds_sync();
        Frag fa[ET], fb[ET], one;
        const_frag(0, 1.0f, lane, one);
#pragma unroll
        for (int a = 0; a < ET; a++) load_frag(U, a, lane, fa[a]);
#pragma unroll
        for (int b = 0; b < ET; b++) load_frag(V, b, lane, fb[b]);
        mma_grid<ET, ET>(tE2, fa, fb);
        mma_col<ET>(tbE2, fa, one);
    }
    // cols 0,1 of the tile: sum a*R, sum a
    template <int N>
    __device__ __forceinline__ void times_R_and_1(f32x4 (&tiles)[(N + 15) / 16], const float (&a)[N], float Rv)
    {
        stage_rows<N>(U, 0, a, lane);
        stage_row(V, 0, Rv, lane);
        stage_row(V, 1, 1.0f, lane);
        wave_lds_sync();
        Frag fb, fa[(N + 15) / 16];
        load_frag(V, 0, lane, fb);
#pragma unroll
        for (int t = 0; t < (N + 15) / 16; t++) load_frag(U, t, lane, fa[t]);
        mma_col<(N + 15) / 16>(tiles, fa, fb);
    }
    __device__ __forceinline__ void e1(const float (&a1)[HE], float Rv) { times_R_and_1<HE>(tE1, a1, Rv); }
    __device__ __forceinline__ void g1(const float (&ag)[HG], float Rv) { times_R_and_1<HG>(tG1, ag, Rv); }
    __device__ __forceinline__ void g2(const float (&tg)[HG], float gb)
    {
        stage_rows<HG>(U, 0, tg, lane);
        wave_lds_sync();
        Frag one, fa;
        const_frag(0, 1.0f, lane, one);
        load_frag(U, 0, lane, fa);
        mma(tG2, fa, one);
        bG2 += gb;
    }
    __device__ __forceinline__ void dump(float* g, bool add)
    {
        const RnetLayout Lo{HE, HG};
#pragma unroll
        for (int a = 0; a < ET; a++) {
            const int nr = HE - 16 * a < 16 ? HE - 16 * a : 16;
#pragma unroll
            for (int b = 0; b < ET; b++) {
                const int nc = HE - 16 * b < 16 ? HE - 16 * b : 16;
                tile_store(tE2[a][b], g + Lo.WE2() + (16 * a) * HE + 16 * b, HE, 1, nr, nc, lane, add);
            }
            tile_store(tbE2[a], g + Lo.bE2() + 16 * a, 1, 0, nr, 1, lane, add);
            tile_store(tE3[a], g + Lo.WE3() + 16 * a, 1, 0, nr, 1, lane, add);
            // tE1: col 0 -> WE1-bar, col 1 -> bE1-bar (bE1 follows WE1 at distance HE)
            tile_store(tE1[a], g + Lo.WE1() + 16 * a, 1, HE, nr, 2, lane, add);
        }
        tile_store(tG2, g + Lo.WG2(), 1, 0, HG, 1, lane, add);
        tile_store(tG1[0], g + Lo.WG1(), 1, HG, HG, 2, lane, add);
    }
};

// two waves per SIMD.  Measured at 1M points (HE = 32): 2 waves with MFMA-tile vector sums
// 124 us; 1 wave with lane-local vector sums 225 us -- this kernel alternates short VALU, LDS and
// matrix-pipe phases and needs the second wave to fill the gaps.
template <int HE, int HG, int ACT>
__global__ __launch_bounds__(kBlock, HE > 32 ? 1 : 2) void k_rnet_bwd(const float* __restrict__ th_r,
                                                     const float* __restrict__ thT_r,
                                                     const float* __restrict__ R,
                                                     const float* __restrict__ Eb,
                                                     const float* __restrict__ gb, int64_t n,
                                                     float* __restrict__ partial, int stride)
{
    constexpr int HR = ((HE > HG ? HE : HG) + 15) / 16 * 16;  // staged rows
    constexpr RnetLayout Lo{HE, HG};
    constexpr int PR = Lo.size();
    __shared__ __attribute__((aligned(16))) float smem_r[kWavesPerBlock * 2 * HR * kPitch + PR + 16];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* gimg = smem_r + kWavesPerBlock * 2 * HR * kPitch;
    RnetSink<HE, HG> sink;
    sink.U = smem_r + wave * 2 * HR * kPitch;
    sink.V = sink.U + HR * kPitch;
    sink.lane = lane;
    sink.zero();
    for (int k = lane; k < 2 * HR * kPitch; k += kWave) sink.U[k] = 0.0f;
    Mat32 M, MT;
    if (HE == 32) {
        load_mat32(th_r + Lo.WE2(), false, lane, M);
        load_mat32(th_r + Lo.WE2(), true, lane, MT);
    }

    for (int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * kWave; base < n;
         base += (int64_t)gridDim.x * kBlock) {
        const int64_t idx = base + lane;
        const bool valid = idx < n;
        const int64_t id = valid ? idx : n - 1;
        const float Rv = R[id];
        const float eb = valid ? Eb[id] : 0.0f, gbv = valid ? gb[id] : 0.0f;
        float e, gg;
        if constexpr (HE == 32) {
            RnetM32<HG, ACT> rn;
            rn.forward(launder(th_r), M, Rv, e, gg);
            rn.backward(launder(th_r), MT, Rv, eb, gbv, sink);
        } else {
            Rnet<HE, HG, ACT, float> rn;
            rn.forward(launder(th_r), launder(thT_r), Rv, e, gg);
            rn.backward(launder(th_r), Rv, eb, gbv, sink);
        }
    }
    const float sE3 = wave_sum(sink.bE3), sG2 = wave_sum(sink.bG2);
    for (int w = 0; w < kWavesPerBlock; w++) {
        if (wave == w) {
            sink.dump(gimg, w != 0);
            if (lane == 0) {
                float* sc = gimg + PR;
                if (w == 0) { sc[0] = sE3; sc[1] = sG2; }
                else { sc[0] += sE3; sc[1] += sG2; }
            }
        }
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * stride;
    for (int k = threadIdx.x; k < PR; k += kBlock)
        out[k] = k == Lo.bE3() ? gimg[PR] : (k == Lo.bG2() ? gimg[PR + 1] : gimg[k]);
}

// --------------------------------------------------------- reduce / Adam / sample
// out[e] = sum_b partial[b][e]; frozen tensors -> 0.  Block = 64 entries x 16 row groups:
// thread (tx, ty) sums rows ty, ty+16, ... of column e (coalesced over tx, 8 loads in
// flight), then the 16 group sums are added in fixed order: bitwise reproducible.
// rows ty, ty + 16, ... of column e of the partial sums (e < PT: trunk rows; < P: R-net rows; P .. P+2: the loss terms)
__device__ __forceinline__ float column_part(const float* __restrict__ partT, int nbT, int strideT, int PT,
                                             const float* __restrict__ partR, int nbR, int strideR, int P, int e, int ty)
{
    const float* src;
    int nb, stride;
    if (e < PT) { src = partT + e; nb = nbT; stride = strideT; }
    else if (e < P) { src = partR + (e - PT); nb = nbR; stride = strideR; }
    else { src = partT + PT + (e - P); nb = nbT; stride = strideT; }
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = ty;
    for (; b + 16 * 7 < nb; b += 16 * 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] += src[(size_t)(b + 16 * k) * stride];
    }
    for (; b < nb; b += 16) a[0] += src[(size_t)b * stride];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

__global__ __launch_bounds__(1024) void k_reduce(const float* __restrict__ partT, int nbT, int strideT, int PT,
                                                 const float* __restrict__ partR, int nbR, int strideR, int PR,
                                                 const uint8_t* __restrict__ tensor_of, uint64_t trainable_mask,
                                                 float* __restrict__ out)
{
    __shared__ float sh[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + tx;
    const int P = PT + PR;
    float s = 0.f;
    if (e < P + 3) s = column_part(partT, nbT, strideT, PT, partR, nbR, strideR, P, e, ty);
    sh[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && e < P + 3) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) t += sh[k][tx];
        if (e < P && !((trainable_mask >> tensor_of[e]) & 1)) t = 0.f;
        out[e] = t;
    }
}

// Several blocks (one per 1024 parameters, at most kAdamMaxBlocks): every block reads step_count and best_loss before
// it takes a ticket, and the block that takes the LAST ticket writes them back -- so no block can see the
// incremented counter (one block looping over P = 51 233 parameters took 40 us at 4x128).
constexpr int kAdamMaxBlocks = 64;
__global__ __launch_bounds__(1024) void k_adam(float* __restrict__ theta,
                                               const float* __restrict__ grad,
                                               float* __restrict__ m, float* __restrict__ v,
                                               int32_t* __restrict__ step_count, int P, double lr,
                                               double b1d, double b2d, double epsd,
                                               const uint8_t* __restrict__ tensor_of,
                                               uint64_t trainable_mask,
                                               float* __restrict__ best_theta,
                                               float* __restrict__ best_loss,
                                               unsigned* __restrict__ ticket,
                                               const float* __restrict__ msg_counts,
                                               long long* __restrict__ counts_out)
{
    // the bias corrections are formed in double and rounded once, as torch does for fp32 tensors -- by ONE
    // thread per block (two double pow() per thread were most of this kernel's 10 us), handed out through LDS
    __shared__ float sh_step_size, sh_inv_bc2s;
    __shared__ int sh_step, sh_better;
    if (threadIdx.x == 0) {
        const int step0 = *step_count + 1;
        const double bc1 = 1.0 - pow(b1d, (double)step0);
        const double bc2 = 1.0 - pow(b2d, (double)step0);
        sh_step_size = (float)(lr / bc1);
        sh_inv_bc2s = (float)(1.0 / sqrt(bc2));
        sh_step = step0;
        sh_better = best_loss != nullptr && (grad[P] + grad[P + 1] + grad[P + 2]) < *best_loss;
    }
    __syncthreads();  // thread 0 has read step_count / best_loss before this block writes anything or takes its ticket
    const int step = sh_step;
    const float ltot = grad[P] + grad[P + 1] + grad[P + 2];
    const bool better = sh_better != 0;
    const float step_size = sh_step_size, inv_bc2s = sh_inv_bc2s;
    const float b1 = (float)b1d, b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
    const float eps = (float)epsd;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < P; k += gridDim.x * blockDim.x) {
        const float th = theta[k];
        if (better && best_theta) best_theta[k] = th;  // train.py:58-60: params the loss was taken at
        if (!((trainable_mask >> tensor_of[k]) & 1)) continue;
        const float g = grad[k];
        const float mk = b1 * m[k] + omb1 * g;
        const float vk = b2 * v[k] + omb2 * g * g;
        m[k] = mk;
        v[k] = vk;
        const float denom = sqrtf(vk) * inv_bc2s + eps;
        theta[k] = th - step_size * (mk / denom);
    }
    if (threadIdx.x == 0) {
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {   // every block has read step_count / best_loss by now
            *ticket = 0;                                 // for the next call
            *step_count = step;
            if (better) *best_loss = ltot;
            if (msg_counts && counts_out) {              // data parallel: set sizes of the batch drawn ahead
                counts_out[0] = (long long)msg_counts[0];
                counts_out[1] = (long long)msg_counts[1];
            }
        }
    }
}

// k_reduce and k_adam in one launch, for the single-device step (no exchange between the two): block = 64 columns x 16 row
// groups as in k_reduce; every block also sums the three loss columns (it needs the loss for the best-model test: 3 x nbT
// floats, nothing), then row group 0 applies Adam to its 64 parameters.  Same additions in the same order as the two
// kernels: bitwise identical gradient, losses and parameters.  step_count / best_loss: as in k_adam (ticket).
__global__ __launch_bounds__(1024) void k_reduce_adam(const float* __restrict__ partT, int nbT, int strideT, int PT,
                                                      const float* __restrict__ partR, int nbR, int strideR, int PR,
                                                      const uint8_t* __restrict__ tensor_of, uint64_t trainable_mask,
                                                      float* __restrict__ grad_out, float* __restrict__ theta,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      int32_t* __restrict__ step_count, double lr, double b1d, double b2d,
                                                      double epsd, float* __restrict__ best_theta,
                                                      float* __restrict__ best_loss, unsigned* __restrict__ ticket)
{
    __shared__ float sh[16][64], shl[16][4];
    __shared__ float sh_step_size, sh_inv_bc2s, sh_best;
    __shared__ int sh_step;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + tx;
    const int P = PT + PR;
    sh[ty][tx] = e < P + 3 ? column_part(partT, nbT, strideT, PT, partR, nbR, strideR, P, e, ty) : 0.f;
    if (tx < 3) shl[ty][tx] = column_part(partT, nbT, strideT, PT, partR, nbR, strideR, P, P + tx, ty);
    if (threadIdx.x == 0) {
        const int step0 = *step_count + 1;
        const double bc1 = 1.0 - pow(b1d, (double)step0);
        const double bc2 = 1.0 - pow(b2d, (double)step0);
        sh_step_size = (float)(lr / bc1);
        sh_inv_bc2s = (float)(1.0 / sqrt(bc2));
        sh_step = step0;
        sh_best = best_loss ? *best_loss : 0.f;
    }
    __syncthreads();      // thread 0 has read step_count / best_loss before this block writes anything or takes its ticket
    float l3[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) t += shl[k][j];
        l3[j] = t;
    }
    const float ltot = l3[0] + l3[1] + l3[2];
    const bool better = best_loss != nullptr && ltot < sh_best;
    if (ty == 0 && e < P + 3) {
        float g = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) g += sh[k][tx];
        if (e < P && !((trainable_mask >> tensor_of[e]) & 1)) g = 0.f;
        if (grad_out) grad_out[e] = g;
        if (e < P) {
            const float th = theta[e];
            if (better && best_theta) best_theta[e] = th;     // train.py:58-60: params the loss was taken at
            if ((trainable_mask >> tensor_of[e]) & 1) {
                const float b1 = (float)b1d, b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
                const float mk = b1 * m[e] + omb1 * g;
                const float vk = b2 * v[e] + omb2 * g * g;
                m[e] = mk;
                v[e] = vk;
                const float denom = sqrtf(vk) * sh_inv_bc2s + (float)epsd;
                theta[e] = th - sh_step_size * (mk / denom);
            }
        }
    }
    if (threadIdx.x == 0) {
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {   // every block has read step_count / best_loss by now
            *ticket = 0;
            *step_count = sh_step;
            if (better) *best_loss = ltot;
        }
    }
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&o)[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

constexpr int kSampleBlock = 1024;   // few, large blocks
constexpr int kSampleMaxBlocks = 256;
__global__ __launch_bounds__(kSampleBlock) void k_sample(uint64_t seed, uint64_t step0,
                                                   const int32_t* __restrict__ step_dev, int64_t first,
                                                   int64_t n, float box, float Rlo, float Rhi,
                                                   float cutoff, float bc_cutoff,
                                                   float* __restrict__ x, float* __restrict__ y,
                                                   float* __restrict__ z, float* __restrict__ R,
                                                   uint8_t* __restrict__ mask,
                                                   unsigned long long* __restrict__ counts,
                                                   unsigned long long* __restrict__ part,
                                                   float* __restrict__ counts_f)
{
    const uint64_t step = step0 + (step_dev ? (uint64_t)(int64_t)*step_dev : 0);
    const float two_box = box + box, dR = Rhi - Rlo;
    const float cut2 = cutoff * cutoff, bc2 = bc_cutoff * bc_cutoff;
    unsigned c1 = 0, c2 = 0;
    for (int64_t i = (int64_t)blockIdx.x * kSampleBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kSampleBlock) {
        const uint64_t gi = (uint64_t)(first + i);
        uint32_t o[4];
        philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)step, (uint32_t)(step >> 32),
                      (uint32_t)seed, (uint32_t)(seed >> 32), o);
        float xv = __fmaf_rn(two_box, u01(o[0]), -box);
        const float yv = __fmaf_rn(two_box, u01(o[1]), -box);
        const float zv = __fmaf_rn(two_box, u01(o[2]), -box);
        const float Rv = __fmaf_rn(dR, u01(o[3]), Rlo);
        const float yz = __fmaf_rn(yv, yv, __fmul_rn(zv, zv));
        float d1 = __fsub_rn(xv, Rv), d2 = __fadd_rn(xv, Rv);
        float r1sq = __fmaf_rn(d1, d1, yz), r2sq = __fmaf_rn(d2, d2, yz);
        if (r1sq < cut2 || r2sq < cut2) xv = cutoff;  // train.py:34-35
        d1 = __fsub_rn(xv, Rv);
        d2 = __fadd_rn(xv, Rv);
        r1sq = __fmaf_rn(d1, d1, yz);
        r2sq = __fmaf_rn(d2, d2, yz);
        const uint8_t mk = (uint8_t)((r1sq >= bc2 ? 1 : 0) | (r2sq >= bc2 ? 2 : 0));
        x[i] = xv; y[i] = yv; z[i] = zv; R[i] = Rv;
        if (mask) mask[i] = mk;
        c1 += mk & 1;
        c2 += (mk >> 1) & 1;
    }
    if (counts || counts_f) {
        // per-block partial counts, summed by whichever block finishes last (ticket): no memset of the counts
        // before the kernel (a 5 us node of its own in every step), no atomics on them; integer sums, so the
        // result does not depend on the order
        __shared__ unsigned sh[2];
        __shared__ unsigned last;
        if (threadIdx.x < 2) sh[threadIdx.x] = 0;
        __syncthreads();
        unsigned w1 = c1, w2 = c2;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            w1 += (unsigned)__shfl_xor((int)w1, off, kWave);
            w2 += (unsigned)__shfl_xor((int)w2, off, kWave);
        }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&sh[0], w1); atomicAdd(&sh[1], w2); }
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(&part[2 * blockIdx.x], (unsigned long long)sh[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&part[2 * blockIdx.x + 1], (unsigned long long)sh[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // The two stores above are agent-scope atomic stores (write-through, sc1): once the wave's store counter
            // has drained they are visible to every XCD, which is all the ticket below must wait for.  (__threadfence()
            // here was a release fence for EVERYTHING this block wrote: buffer_wbl2 + buffer_inv, i.e. one write-back
            // of the XCD's L2 -- full of the batch just sampled -- per block: most of this kernel's 15 us at 1M points.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            last = atomicAdd(reinterpret_cast<unsigned*>(part + 2 * kSampleMaxBlocks), 1u) == gridDim.x - 1;
        }
        __syncthreads();
        if (last && threadIdx.x < 64) {
            __threadfence();
            unsigned long long s1 = 0, s2 = 0;
            for (unsigned b = threadIdx.x; b < gridDim.x; b += 64) {
                s1 += __hip_atomic_load(&part[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s2 += __hip_atomic_load(&part[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                s1 += (unsigned long long)__shfl_xor((long long)s1, off, kWave);
                s2 += (unsigned long long)__shfl_xor((long long)s2, off, kWave);
            }
            if (threadIdx.x == 0) {
                if (counts) { counts[0] = s1; counts[1] = s2; }
                if (counts_f) { counts_f[0] = (float)s1; counts_f[1] = (float)s2; }
                *reinterpret_cast<unsigned*>(part + 2 * kSampleMaxBlocks) = 0;   // ticket for the next call
            }
        }
    }
}

// ------------------------------------------------------------------ 3-D quadrature (integra3d)
// plane i of the grid: block i sums wy[j] wz[k] u v over its ny x nz points in fp64 (thread-strided, then a
// fixed-order LDS tree), the second kernel adds the nx plane sums times wx[i] in index order
constexpr int kQuadBlock = 256;
__global__ __launch_bounds__(kQuadBlock) void k_quad_planes(const float* __restrict__ u, const float* __restrict__ v,
                                                            const double* __restrict__ wy, const double* __restrict__ wz,
                                                            int64_t ny, int64_t nz, double* __restrict__ plane)
{
    __shared__ double sh[kQuadBlock];
    const unsigned m = (unsigned)(ny * nz), nzu = (unsigned)nz;       // (the host checks ny * nz < 2^31)
    const float* up = u + (size_t)blockIdx.x * m;
    const float* vp = v ? v + (size_t)blockIdx.x * m : up;
    double acc = 0.0;
    for (unsigned t = threadIdx.x; t < m; t += kQuadBlock) {
        const unsigned j = t / nzu, k = t - j * nzu;
        acc += wy[j] * wz[k] * ((double)up[t] * (double)vp[t]);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kQuadBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) plane[blockIdx.x] = sh[0];
}

__global__ void k_quad_final(const double* __restrict__ plane, const double* __restrict__ wx, int64_t nx,
                             double* __restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int64_t i = 0; i < nx; i++) s += wx[i] * plane[i];
        *out = s;
    }
}

}  // namespace

// =================================================================== C ABI
// ------------------------------------------------------------------ E(R), dE/dR, d2E/dR2
// The callers next to the hot path differentiate the E-net w.r.t. R with autograd (forces,
// poc/main.py:1322-1328).  Three forward streams (value, d/dR, d2/dR2) through Lin_E1/E2/Eout;
// one lane per R value, the layer-1 streams parked in a lane-private LDS column
// (sh[(s*HE + k)*64 + lane]: bank = lane, conflict-free).  Tiny and rare: not tuned further.
template <int ACT>
__global__ __launch_bounds__(64) void k_energy_R(const float* __restrict__ thE, int HE,
                                                 const float* __restrict__ R, int64_t n,
                                                 float* __restrict__ E, float* __restrict__ dE,
                                                 float* __restrict__ d2E)
{
    extern __shared__ float sh_e[];
    const RnetLayout Lo{HE, 1};
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const float r = R[i < n ? i : n - 1];
    for (int k = 0; k < HE; k++) {
        const float w = thE[Lo.WE1() + k];
        float s0, s1, s2;
        act012<ACT>(w * r + thE[Lo.bE1() + k], s0, s1, s2);
        sh_e[(0 * HE + k) * 64 + lane] = s0;
        sh_e[(1 * HE + k) * 64 + lane] = s1 * w;
        sh_e[(2 * HE + k) * 64 + lane] = s2 * w * w;
    }
    float e = thE[Lo.bE3()], de = 0.f, d2e = 0.f;
    for (int j = 0; j < HE; j++) {
        const float* W = thE + Lo.WE2() + j * HE;
        float a = thE[Lo.bE2() + j], da = 0.f, d2a = 0.f;
        for (int k = 0; k < HE; k++) {
            const float w = W[k];
            a += w * sh_e[(0 * HE + k) * 64 + lane];
            da += w * sh_e[(1 * HE + k) * 64 + lane];
            d2a += w * sh_e[(2 * HE + k) * 64 + lane];
        }
        float s0, s1, s2;
        act012<ACT>(a, s0, s1, s2);
        const float w3 = thE[Lo.WE3() + j];
        e += w3 * s0;
        de += w3 * s1 * da;
        d2e += w3 * (s2 * da * da + s1 * d2a);
    }
    if (i < n) {
        if (E) E[i] = e;
        if (dE) dE[i] = de;
        if (d2E) d2E[i] = d2e;
    }
}

struct pinn_handle {
    pinn_arch arch;
    int device;
    int64_t max_points;
    int PT, PR, P;
    int nbT, nbR;           // persistent grid sizes
    int cus, bpcT;          // CUs of the device, resident trunk blocks per CU the grid is sized for
    int pend_gT, pend_gR;   // > 0: partial sums of a pinn_residual_fwd_bwd call with out == NULL wait for pinn_adam_fused
    uint64_t pend_mask;
    int strideT, strideR;
    float *E, *g, *Eb, *gb; // per-point scratch
    float *partT, *partR;   // per-block partials
    void* wpk;              // packed bf16 weights (MFMA path)
    float* thT;             // theta with forward matrices input-major (k_prep_theta)
    unsigned long long* samp; // k_sample: per-block partial counts [2 * 256] + its ticket; + k_adam's ticket
    uint8_t* tensor_of;     // [P] tensor index of each parameter
    uint64_t trunk_tensors; // bits of the trunk tensors
    int prof_slots;          // 0 = off
    int64_t prof_calls;      // calls recorded since enable
    hipEvent_t* ev;          // 2 * prof_slots events
    char err[256];
};

static int trunk_size(const pinn_arch* a) { return 3 * a->H + (a->L - 1) * (a->H * a->H + a->H) + a->H + 1; }
static int rnet_size(const pinn_arch* a) { return 4 * a->HE + a->HE * a->HE + 1 + 3 * a->HG + 1; }

static int arch_ok(const pinn_arch* a)
{
    if (!a) return 0;
    if (a->H < 1 || a->L < 2 || a->L > 8 || a->HE < 1 || a->HG < 1) return 0;
    if (a->act != PINN_ACT_SIGMOID && a->act != PINN_ACT_TANH) return 0;
    if (a->n_pass != 1 && a->n_pass != 2) return 0;
    return 1;
}

extern "C" int64_t pinn_param_count(const pinn_arch* a)
{
    if (!arch_ok(a)) return PINN_ERR_ARG;
    return (int64_t)trunk_size(a) + rnet_size(a);
}

extern "C" int32_t pinn_tensor_count(const pinn_arch* a)
{
    if (!arch_ok(a)) return PINN_ERR_ARG;
    return 2 * (a->L + 1) + 6 + 4;
}

extern "C" int pinn_tensor_layout(const pinn_arch* a, int64_t* off, int64_t* sz)
{
    if (!arch_ok(a) || !off || !sz) return PINN_ERR_ARG;
    int t = 0;
    int64_t o = 0;
    const int H = a->H, HE = a->HE, HG = a->HG;
    auto push = [&](int64_t n) { off[t] = o; sz[t] = n; o += n; t++; };
    for (int l = 0; l < a->L; l++) { push((int64_t)H * (l == 0 ? 2 : H)); push(H); }
    push(H); push(1);
    push(HE); push(HE); push((int64_t)HE * HE); push(HE); push(HE); push(1);
    push(HG); push(HG); push(HG); push(1);
    return PINN_OK;
}

extern "C" int pinn_debug_block_tiles(int grid, int block, int64_t ntile, int nclass, const int64_t* bounds, int unit,
                                      int64_t out[3])
{
    if (grid < 1 || block < 0 || block >= grid || ntile < 0 || unit < 1 || !out) return PINN_ERR_ARG;
    TileShares ts{};
    ts.nclass = nclass;
    for (int k = 0; k < 3 && k < nclass - 1 && bounds; k++) ts.bound[k] = bounds[k];
    const BlockTiles t = block_tiles_of(grid, block, ntile, ts, unit);
    out[0] = t.first; out[1] = t.end; out[2] = t.stride;
    return PINN_OK;
}

extern "C" const char* pinn_strerror(int s)
{
    switch (s) {
        case PINN_OK: return "ok";
        case PINN_ERR_ARG: return "invalid argument";
        case PINN_ERR_UNSUPPORTED: return "unsupported architecture or option";
        case PINN_ERR_HIP: return "HIP runtime error";
        case PINN_ERR_CAPACITY: return "batch larger than max_points given at pinn_create";
        default: return "unknown status";
    }
}

extern "C" const char* pinn_last_error(const pinn_handle* h) { return h ? h->err : "null handle"; }

static int fail(pinn_handle* h, int code, const char* what, hipError_t e = hipSuccess)
{
    if (h) {
        if (e != hipSuccess) snprintf(h->err, sizeof h->err, "%s: %s", what, hipGetErrorString(e));
        else snprintf(h->err, sizeof h->err, "%s", what);
    }
    return code;
}

#define HIPCHK(h, call)                                                   \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return fail(h, PINN_ERR_HIP, #call, e_);    \
    } while (0)

// ---- template dispatch ---------------------------------------------------
namespace {

struct TrunkKey { int H, L, act, npass, mode; };

template <int H, int L, int ACT, int NPASS, int MODE>
static hipError_t launch_trunk_t(const float* theta, const float* thT, const TrunkArgs& A, int grid,
                                 hipStream_t st)
{
    hipLaunchKernelGGL((k_trunk<H, L, ACT, NPASS, MODE>), dim3(grid), dim3(kBlock), 0, st, theta, thT, A);
    return hipGetLastError();
}

template <int H, int L>
static int launch_trunk_HL(const TrunkKey& k, const float* theta, const float* thT, const TrunkArgs& A, int grid, hipStream_t st, hipError_t* e)
{
#define PINN_CASE(ACT, NP, MD)                                                     \
    if (k.act == ACT && k.npass == NP && k.mode == MD) {                           \
        *e = launch_trunk_t<H, L, ACT, NP, MD>(theta, thT, A, grid, st);                      \
        return 1;                                                                  \
    }
    PINN_CASE(ACT_SIGMOID, 2, MODE_FWD) PINN_CASE(ACT_SIGMOID, 2, MODE_BWD) PINN_CASE(ACT_SIGMOID, 2, MODE_HEAD)
    PINN_CASE(ACT_SIGMOID, 1, MODE_FWD) PINN_CASE(ACT_SIGMOID, 1, MODE_BWD) PINN_CASE(ACT_SIGMOID, 1, MODE_HEAD)
    PINN_CASE(ACT_TANH, 2, MODE_FWD) PINN_CASE(ACT_TANH, 2, MODE_BWD) PINN_CASE(ACT_TANH, 2, MODE_HEAD)
    PINN_CASE(ACT_TANH, 1, MODE_FWD) PINN_CASE(ACT_TANH, 1, MODE_BWD) PINN_CASE(ACT_TANH, 1, MODE_HEAD)
#undef PINN_CASE
    return 0;
}

static int launch_trunk(const TrunkKey& k, const float* theta, const float* thT, const TrunkArgs& A, int grid, hipStream_t st, hipError_t* e)
{
    if (k.H == 16 && k.L == 2) return launch_trunk_HL<16, 2>(k, theta, thT, A, grid, st, e);
    if (k.H == 16 && k.L == 3) return launch_trunk_HL<16, 3>(k, theta, thT, A, grid, st, e);
    if (k.H == 32 && k.L == 2) return launch_trunk_HL<32, 2>(k, theta, thT, A, grid, st, e);
    return 0;
}

static bool trunk_supported(int H, int L) { return (H == 16 && (L == 2 || L == 3)) || (H == 32 && L == 2); }
// (E-net width, gate width) pairs with a kernel: the reference's (32,10) on the matrix-pipe mat-vec,
// the rest on the packed-VALU formulation
#define PINN_RNET_SHAPES(X) X(32, 10) X(16, 8) X(16, 10) X(32, 16) X(64, 10) X(64, 16)
static bool rnet_supported(int HE, int HG)
{
#define PINN_RNET_OK(he, hg) if (HE == he && HG == hg) return true;
    PINN_RNET_SHAPES(PINN_RNET_OK)
#undef PINN_RNET_OK
    return false;
}

template <int HE, int HG>
static int launch_rnet_fwd_s(int act, const float* th, const float* thT, const float* R, int64_t n,
                             float* E, float* g, float* Eu, int grid, hipStream_t st, hipError_t* e)
{
    if (act == ACT_SIGMOID)
        hipLaunchKernelGGL((k_rnet_fwd<HE, HG, ACT_SIGMOID>), dim3(grid), dim3(kBlock), 0, st, th, thT, R, n, E, g, Eu);
    else
        hipLaunchKernelGGL((k_rnet_fwd<HE, HG, ACT_TANH>), dim3(grid), dim3(kBlock), 0, st, th, thT, R, n, E, g, Eu);
    *e = hipGetLastError();
    return 1;
}

static int launch_rnet_fwd(int HE, int HG, int act, const float* th, const float* thT, const float* R,
                           int64_t n, float* E, float* g, float* Eu, int grid, hipStream_t st,
                           hipError_t* e)
{
#define PINN_RNET_FWD(he, hg) if (HE == he && HG == hg) return launch_rnet_fwd_s<he, hg>(act, th, thT, R, n, E, g, Eu, grid, st, e);
    PINN_RNET_SHAPES(PINN_RNET_FWD)
#undef PINN_RNET_FWD
    return 0;
}

template <int HE, int HG>
static int launch_rnet_bwd_s(int act, const float* th, const float* thT, const float* R, const float* Eb,
                             const float* gb, int64_t n, float* part, int stride, int grid,
                             hipStream_t st, hipError_t* e)
{
    const size_t shm = 0;
    if (act == ACT_SIGMOID)
        hipLaunchKernelGGL((k_rnet_bwd<HE, HG, ACT_SIGMOID>), dim3(grid), dim3(kBlock), shm, st, th, thT, R, Eb, gb, n, part, stride);
    else
        hipLaunchKernelGGL((k_rnet_bwd<HE, HG, ACT_TANH>), dim3(grid), dim3(kBlock), shm, st, th, thT, R, Eb, gb, n, part, stride);
    *e = hipGetLastError();
    return 1;
}

static int launch_rnet_bwd(int HE, int HG, int act, const float* th, const float* thT, const float* R, const float* Eb,
                           const float* gb, int64_t n, float* part, int stride, int grid,
                           hipStream_t st, hipError_t* e)
{
#define PINN_RNET_BWD(he, hg) if (HE == he && HG == hg) return launch_rnet_bwd_s<he, hg>(act, th, thT, R, Eb, gb, n, part, stride, grid, st, e);
    PINN_RNET_SHAPES(PINN_RNET_BWD)
#undef PINN_RNET_BWD
    return 0;
}

}  // namespace

static int grid_for(int64_t n, int cap)
{
    const int64_t tiles = (n + kBlock - 1) / kBlock;
    return (int)(tiles < cap ? tiles : cap);
}

// grid of the 16-points-per-wave R-net kernels: one block = 64 points per trip
static int grid16(int64_t n, int cap)
{
    const int64_t b = (n + 63) / 64;
    return (int)(b < cap ? b : cap);
}

static TileShares rnet_shares(const pinn_handle* h, int64_t n, int grid, bool reverse);

// E(R), g(R) for the batch: matrix-pipe kernel for the reference widths, lane = point kernels otherwise
static int rnet_forward(pinn_handle* h, const float* theta, const float* R, int64_t n, float* Eu,
                        hipStream_t st, hipError_t* e)
{
    const pinn_arch& a = h->arch;
    if (rnet_mfma_supported(a.HE, a.HG)) {
        const int gF = grid16(n, h->nbR * 2);
        *e = launch_rnet_mfma_fwd(a.act, theta + h->PT, R, n, h->E, h->g, Eu, gF, rnet_shares(h, n, gF, false), st);
        return 1;
    }
    return launch_rnet_fwd(a.HE, a.HG, a.act, theta + h->PT, h->thT + h->PT, R, n, h->E, h->g, Eu,
                           grid_for(n, h->nbR * 4), st, e);
}

static hipError_t prep_theta(pinn_handle* h, const float* theta, hipStream_t st)
{
    const pinn_arch& a = h->arch;
    // the input-major copy only feeds the lane = point kernels (k_trunk, k_rnet_fwd/bwd)
    const bool trunk_needs = a.precision == PINN_PREC_F32 && !trunk16_mfma_supported(a.H, a.L);
    if (!trunk_needs && rnet_mfma_supported(a.HE, a.HG)) return hipSuccess;
    hipLaunchKernelGGL(k_prep_theta, dim3((h->P + 255) / 256), dim3(256), 0, st, theta, h->thT, a.H, a.L,
                       a.HE, h->PT, h->P);
    return hipGetLastError();
}

// persistent grid of the trunk kernel for n points (also the number of partial rows k_reduce sums)
static int trunk_grid(const pinn_handle* h, int64_t n, int mode)
{
    const pinn_arch& a = h->arch;
    if (a.precision == PINN_PREC_F32 && trunk16_mfma_supported(a.H, a.L))
        // k_trunk16: the reverse kernel holds two blocks per CU (256 registers), the forward-only
        // sweeps (96-116 registers) four
        return grid16(n, mode == MODE_BWD && h->nbT > 2 * h->cus ? 2 * h->cus : h->nbT);
    if (a.precision != PINN_PREC_F32) {     // wide matrix-core kernels: one 16-point tile per block and trip
        const int64_t t16 = (n + 15) / 16;
        return (int)(t16 < h->nbT ? t16 : h->nbT);
    }
    return grid_for(n, h->nbT);
}

// TileShares (pinn_common.h) of a launch.  Unequal shares only for full grids of two or four resident blocks per CU, where
// blocks b, b + G/K, ... share a CU.  Shares measured per kernel family on MI355X (round 4, tools/share_sweep.sh, 1M points;
// +-0.02 off the optimum costs ~1 %): k_trunk2<64> 0.665 (0.426 -> 0.397 ms), k_trunk2s<64> 0.625 (0.265 -> 0.245),
// k_trunkD<64,4> 0.65 (1.052 -> 0.975 at 0.66 on one box, 0.986 at 0.64-0.65 against 0.997 at 0.66 on another), k_trunk16's reverse kernel 0.72 (0.181 -> 0.165: its waves share no barrier, the
// older block's lead is larger), the R-net reverse kernel 0.68.  Four blocks per CU (k_trunk16's forward-only sweeps, the
// R-net forward kernel): geometric shares 1 : q : q^2 : q^3.  PINN_OLDER_SHARE / PINN_RNET_SHARE / PINN_SHARE_Q override
// (0.5 / 0.5 / 1.0 = rounds 1-3); read once per process, except under PINN_MFMA_VARIANT_LIVE=1 (tests and tools switch them).
// defaults of the four-class shares, tools/q_sweep.sh at 1M points: k_trunk16's forward-only sweep 0.0904 ms at q = 1,
// 0.0861 / 0.0874 / 0.0866 / 0.0854 at 0.65 / 0.5 / 0.4 / 0.3; k_rnet_mfma_fwd 37.1 us at 1, 34.4 / 34.4 / 33.9 / 36.6
constexpr double kShareQ16 = 0.4, kShareQR = 0.45;
static double env_share(const char* name, double lo, double hi)
{
    const char* e = getenv(name);
    const double v = e ? atof(e) : 0.0;
    return (v >= lo && v <= hi) ? v : 0.0;
}
static bool env_live()
{
    static const bool live = [] { const char* e = getenv("PINN_MFMA_VARIANT_LIVE"); return e && e[0] == '1'; }();
    return live;
}
static TileShares shares_two(int64_t n, double share)
{
    TileShares ts{};
    ts.nclass = 2;
    ts.bound[0] = (int64_t)((double)((n + 15) / 16) * share + 0.5);
    return ts;
}
static TileShares shares_four(int64_t n, double q)
{
    TileShares ts{};
    ts.nclass = 4;
    const double tot = 1.0 + q + q * q + q * q * q, nt = (double)((n + 15) / 16);
    ts.bound[0] = (int64_t)(nt / tot + 0.5);
    ts.bound[1] = (int64_t)(nt * (1.0 + q) / tot + 0.5);
    ts.bound[2] = (int64_t)(nt * (1.0 + q + q * q) / tot + 0.5);
    return ts;
}
static TileShares trunk_shares(const pinn_handle* h, int64_t n, int grid, int mode)
{
    static const double once = env_share("PINN_OLDER_SHARE", 0.5, 0.9), once_q = env_share("PINN_SHARE_Q", 0.05, 1.0);
    const double forced = env_live() ? env_share("PINN_OLDER_SHARE", 0.5, 0.9) : once;
    const double forced_q = env_live() ? env_share("PINN_SHARE_Q", 0.05, 1.0) : once_q;
    const pinn_arch& a = h->arch;
    const bool t16 = a.precision == PINN_PREC_F32 && trunk16_mfma_supported(a.H, a.L);
    if (grid == 4 * h->cus && t16 && mode != MODE_BWD)            // k_trunk16's forward-only sweeps: four blocks per CU
        return shares_four(n, forced_q > 0.0 ? forced_q : kShareQ16);
    if (grid != 2 * h->cus || mode != MODE_BWD || (a.precision == PINN_PREC_F32 && !t16)) return TileShares{};
    double share = t16 ? 0.72 : (a.L == 4 ? 0.65 : (a.n_pass == 1 ? 0.625 : 0.665));
    if (forced > 0.0) share = forced;
    return shares_two(n, share);
}

// the same for the R-net kernels on the matrix pipe: reverse two blocks per CU, forward four
static TileShares rnet_shares(const pinn_handle* h, int64_t n, int grid, bool reverse)
{
    static const double once = env_share("PINN_RNET_SHARE", 0.5, 0.9), once_q = env_share("PINN_RNET_Q", 0.05, 1.0);
    const double forced = env_live() ? env_share("PINN_RNET_SHARE", 0.5, 0.9) : once;
    const double forced_q = env_live() ? env_share("PINN_RNET_Q", 0.05, 1.0) : once_q;
    if (reverse) return grid == 2 * h->cus ? shares_two(n, forced > 0.0 ? forced : 0.68) : TileShares{};   // 53.0 -> 50.0 us per 1M points
    return grid == 4 * h->cus ? shares_four(n, forced_q > 0.0 ? forced_q : kShareQR) : TileShares{};
}

static int run_trunk(pinn_handle* h, int mode, const float* theta, const TrunkArgs& A, int grid,
                     hipStream_t st, hipError_t* e)
{
    const pinn_arch& a = h->arch;
    if (a.precision == PINN_PREC_F32) {
        if (trunk16_mfma_supported(a.H, a.L)) {   // the reference trunk on the matrix pipe (k_trunk16)
            *e = launch_trunk16_mfma(a.act, a.n_pass, mode, theta, A, grid, st);
            return 1;
        }
        const TrunkKey k{a.H, a.L, a.act, a.n_pass, mode};
        return launch_trunk(k, theta, h->thT, A, grid, st, e);
    }
    return launch_trunk_mfma(a.H, a.L, a.act, a.n_pass, mode, a.precision == PINN_PREC_BF16X3, theta, h->wpk, A,
                             grid, st, e) ? 1 : 0;
}

extern "C" int pinn_create(const pinn_arch* arch, int device, int64_t max_points, pinn_handle** out)
{
    if (!out) return PINN_ERR_ARG;
    *out = nullptr;
    if (!arch_ok(arch) || max_points < 1 || device < 0) return PINN_ERR_ARG;
    if (arch->precision == PINN_PREC_F32) {
        if (!trunk_supported(arch->H, arch->L)) return PINN_ERR_UNSUPPORTED;
    } else if (arch->precision == PINN_PREC_BF16) {
        if (!mfma_trunk_supported(arch->H, arch->L)) return PINN_ERR_UNSUPPORTED;
    } else if (arch->precision == PINN_PREC_BF16X3) {
        if (!mfma_x3_supported(arch->H, arch->L)) return PINN_ERR_UNSUPPORTED;
    } else {
        return PINN_ERR_UNSUPPORTED;
    }
    if (!rnet_supported(arch->HE, arch->HG)) return PINN_ERR_UNSUPPORTED;
    pinn_handle* h = new (std::nothrow) pinn_handle();
    if (!h) return PINN_ERR_ARG;
    memset(h, 0, sizeof *h);
    h->arch = *arch;
    h->device = device;
    h->max_points = max_points;
    h->PT = trunk_size(arch);
    h->PR = rnet_size(arch);
    h->P = h->PT + h->PR;
    *out = h;
    HIPCHK(h, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, device));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int64_t tiles = (max_points + kBlock - 1) / kBlock;
    // one resident 4-wave block per CU for both trunk kernels (1 wave per SIMD)
    const int bpc = arch->precision != PINN_PREC_F32
                        ? mfma_blocks_per_cu(arch->H, arch->L, arch->precision == PINN_PREC_BF16X3)
                        : (trunk16_mfma_supported(arch->H, arch->L) ? 4 : 1);
    // the wide matrix-core kernels take 16-point tiles one at a time: size their grid (and the partial rows) by those,
    // so that a small batch spreads over the chip instead of queueing 16 tiles behind each other in a few blocks
    // (4 000 points: 16 blocks x 16 trips = 35 us before, 250 blocks x 1 trip now)
    // (the 16-wide reference trunk and the R-net kernels on the matrix pipe: 64 points per block and trip -- round 4: they
    // were sized by 256-point units too, 8 blocks for a 4 000-point reverse sweep)
    const int64_t tiles64 = (max_points + 63) / 64;
    const int64_t tilesT = arch->precision != PINN_PREC_F32 ? (max_points + 15) / 16
                           : (trunk16_mfma_supported(arch->H, arch->L) ? tiles64 : tiles);
    h->nbT = (int)(tilesT < (int64_t)cus * bpc ? tilesT : (int64_t)cus * bpc);
    h->cus = cus;
    h->bpcT = bpc;
    const int64_t tilesR = rnet_mfma_supported(arch->HE, arch->HG) ? tiles64 : tiles;
    h->nbR = (int)(tilesR < (int64_t)cus * 2 ? tilesR : (int64_t)cus * 2);
    h->strideT = h->PT + 3;
    h->strideR = h->PR;
    HIPCHK(h, hipMalloc(&h->E, sizeof(float) * max_points));
    HIPCHK(h, hipMalloc(&h->g, sizeof(float) * max_points));
    HIPCHK(h, hipMalloc(&h->Eb, sizeof(float) * max_points));
    HIPCHK(h, hipMalloc(&h->gb, sizeof(float) * max_points));
    HIPCHK(h, hipMalloc(&h->partT, sizeof(float) * (size_t)h->nbT * h->strideT));
    HIPCHK(h, hipMalloc(&h->partR, sizeof(float) * (size_t)h->nbR * h->strideR));
    HIPCHK(h, hipMalloc(&h->tensor_of, h->P));
    HIPCHK(h, hipMalloc(&h->thT, sizeof(float) * h->P));
    HIPCHK(h, hipMalloc(&h->samp, sizeof(unsigned long long) * (2 * kSampleMaxBlocks + 2)));   // + k_adam's ticket
    HIPCHK(h, hipMemset(h->samp, 0, sizeof(unsigned long long) * (2 * kSampleMaxBlocks + 2)));
    if (arch->precision != PINN_PREC_F32)
        HIPCHK(h, hipMalloc(&h->wpk, mfma_packed_weight_bytes(arch->H, arch->L) +
                                         mfma_stash_bytes(arch->H, arch->L, arch->precision == PINN_PREC_BF16X3, h->nbT)));
    {
        int64_t off[64], sz[64];
        pinn_tensor_layout(arch, off, sz);
        const int nt = pinn_tensor_count(arch);
        uint8_t* tmp = new (std::nothrow) uint8_t[h->P];
        if (!tmp) return fail(h, PINN_ERR_ARG, "out of host memory");
        for (int t = 0; t < nt; t++)
            for (int64_t k = off[t]; k < off[t] + sz[t]; k++) tmp[k] = (uint8_t)t;
        hipError_t e = hipMemcpy(h->tensor_of, tmp, h->P, hipMemcpyHostToDevice);
        delete[] tmp;
        if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "hipMemcpy(tensor_of)", e);
        h->trunk_tensors = ((uint64_t)1 << (2 * (arch->L + 1))) - 1;
    }
    snprintf(h->err, sizeof h->err, "ok");
    return PINN_OK;
}

static void free_events(pinn_handle* h)
{
    if (h->ev) {
        for (int i = 0; i < 2 * h->prof_slots; i++) (void)hipEventDestroy(h->ev[i]);
        delete[] h->ev;
        h->ev = nullptr;
    }
    h->prof_slots = 0;
    h->prof_calls = 0;
}

extern "C" int pinn_profile_enable(pinn_handle* h, int nslots)
{
    if (!h || nslots < 0 || nslots > 1024) return PINN_ERR_ARG;
    free_events(h);
    if (nslots == 0) return PINN_OK;
    h->ev = new (std::nothrow) hipEvent_t[2 * nslots];
    if (!h->ev) return fail(h, PINN_ERR_ARG, "out of host memory");
    for (int i = 0; i < 2 * nslots; i++) {
        hipError_t e = hipEventCreate(&h->ev[i]);
        if (e != hipSuccess) {
            for (int k = 0; k < i; k++) (void)hipEventDestroy(h->ev[k]);
            delete[] h->ev;
            h->ev = nullptr;
            return fail(h, PINN_ERR_HIP, "hipEventCreate", e);
        }
    }
    h->prof_slots = nslots;
    return PINN_OK;
}

extern "C" int pinn_profile_read(pinn_handle* h, int ncalls, float* mean_ms)
{
    if (!h || !mean_ms || ncalls < 1) return PINN_ERR_ARG;
    if (!h->prof_slots) return fail(h, PINN_ERR_ARG, "profiling is not enabled");
    if (ncalls > h->prof_slots || ncalls > h->prof_calls)
        return fail(h, PINN_ERR_ARG, "fewer recorded calls than requested");
    double sum = 0.0;
    for (int k = 0; k < ncalls; k++) {
        const int slot = (int)((h->prof_calls - 1 - k) % h->prof_slots);
        float ms = 0.f;
        HIPCHK(h, hipEventSynchronize(h->ev[2 * slot + 1]));
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev[2 * slot], h->ev[2 * slot + 1]));
        sum += ms;
    }
    *mean_ms = (float)(sum / ncalls);
    return PINN_OK;
}

extern "C" int pinn_destroy(pinn_handle* h)
{
    if (!h) return PINN_ERR_ARG;
    free_events(h);
    (void)hipFree(h->E); (void)hipFree(h->g); (void)hipFree(h->Eb); (void)hipFree(h->gb);
    (void)hipFree(h->partT); (void)hipFree(h->partR); (void)hipFree(h->tensor_of);
    if (h->wpk) (void)hipFree(h->wpk);
    (void)hipFree(h->thT);
    if (h->samp) (void)hipFree(h->samp);
    delete h;
    return PINN_OK;
}

static int check_common(pinn_handle* h, const void* theta, const void* x, const void* y,
                        const void* z, const void* R, int64_t n, const pinn_ham* ham)
{
    if (!h) return PINN_ERR_ARG;
    if (!theta || !x || !y || !z || !R || !ham || n < 1) return fail(h, PINN_ERR_ARG, "null pointer or n < 1");
    if (n > h->max_points) return fail(h, PINN_ERR_CAPACITY, "n exceeds max_points");
    if (ham->Ry != 0.0f || ham->Rz != 0.0f) return fail(h, PINN_ERR_UNSUPPORTED, "Ry/Rz must be 0 (as in the reference)");
    return PINN_OK;
}

extern "C" int pinn_energy_R(pinn_handle* h, const float* theta, const float* R, int64_t n,
                             float* E, float* dE, float* d2E, void* stream)
{
    if (!h) return PINN_ERR_ARG;
    if (!theta || !R) return fail(h, PINN_ERR_ARG, "null pointer argument");
    if (n < 1) return fail(h, PINN_ERR_ARG, "n must be >= 1");
    const pinn_arch& a = h->arch;
    const size_t lds = (size_t)3 * a.HE * 64 * sizeof(float);
    if (lds > 64 * 1024) return fail(h, PINN_ERR_UNSUPPORTED, "E-net wider than 85 neurons");
    const dim3 grid((unsigned)((n + 63) / 64));
    hipStream_t st = (hipStream_t)stream;
    if (a.act == PINN_ACT_SIGMOID)
        hipLaunchKernelGGL(k_energy_R<PINN_ACT_SIGMOID>, grid, dim3(64), lds, st, theta + h->PT, a.HE, R, n, E, dE, d2E);
    else
        hipLaunchKernelGGL(k_energy_R<PINN_ACT_TANH>, grid, dim3(64), lds, st, theta + h->PT, a.HE, R, n, E, dE, d2E);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_energy_R", e);
    return PINN_OK;
}

extern "C" int pinn_quadrature3d(pinn_handle* h, const float* u, const float* v, const double* wx,
                                 const double* wy, const double* wz, int64_t nx, int64_t ny, int64_t nz,
                                 double* scratch, double* out, void* stream)
{
    if (!h) return PINN_ERR_ARG;
    if (!u || !wx || !wy || !wz || !scratch || !out) return fail(h, PINN_ERR_ARG, "null pointer");
    if (nx < 1 || ny < 1 || nz < 1 || nx > 65535 * 32 || ny * nz >= ((int64_t)1 << 31))
        return fail(h, PINN_ERR_ARG, "grid size");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_quad_planes, dim3((unsigned)nx), dim3(kQuadBlock), 0, st, u, v, wy, wz, ny, nz, scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_quad_planes", e);
    hipLaunchKernelGGL(k_quad_final, dim3(1), dim3(64), 0, st, (const double*)scratch, wx, nx, out);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_quad_final", e);
    return PINN_OK;
}

extern "C" int pinn_forward(pinn_handle* h, const float* theta, const float* x, const float* y,
                            const float* z, const float* R, int64_t n, const pinn_ham* ham,
                            float* psi, float* E, float* lap, float* res, void* stream)
{
    int rc = check_common(h, theta, x, y, z, R, n, ham);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    const pinn_arch& a = h->arch;
    e = prep_theta(h, theta, st);
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_prep_theta", e);
    if (!rnet_forward(h, theta, R, n, E, st, &e))
        return fail(h, PINN_ERR_UNSUPPORTED, "no R-net kernel for this (HE, HG)");
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_rnet_fwd", e);
    if (!psi && !lap && !res) return PINN_OK;
    TrunkArgs A{};
    A.x = x; A.y = y; A.z = z; A.R = R; A.E = h->E; A.g = h->g; A.mask = nullptr;
    A.n = n; A.a = ham->a; A.b = ham->b; A.c = ham->c; A.P = ham->P; A.out_scale = a.out_scale;
    A.invN = 0.f; A.invN1 = 0.f; A.invN2 = 0.f;
    A.psi = psi; A.lap = lap; A.res = res;
    A.shares = trunk_shares(h, n, trunk_grid(h, n, MODE_FWD), MODE_FWD);
    if (!run_trunk(h, MODE_FWD, theta, A, trunk_grid(h, n, MODE_FWD), st, &e))
        return fail(h, PINN_ERR_UNSUPPORTED, "no trunk kernel for this (H, L, precision)");
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_trunk<FWD>", e);
    return PINN_OK;
}

extern "C" int pinn_residual_fwd_bwd(pinn_handle* h, const float* theta, const float* x,
                                     const float* y, const float* z, const float* R,
                                     const uint8_t* mask, int64_t n, const pinn_ham* ham,
                                     const int64_t* bc_counts, uint64_t trainable_mask,
                                     float* out, void* stream)
{
    int rc = check_common(h, theta, x, y, z, R, n, ham);
    if (rc) return rc;
    h->pend_gT = h->pend_gR = 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    const pinn_arch& a = h->arch;
    const bool rmfma = rnet_mfma_supported(a.HE, a.HG);
    const bool trunk_trainable = (trainable_mask & h->trunk_tensors) != 0;
    const int gT = trunk_grid(h, n, trunk_trainable ? MODE_BWD : MODE_HEAD);
    const int gR = rmfma ? grid16(n, h->nbR) : grid_for(n, h->nbR);
    e = prep_theta(h, theta, st);
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_prep_theta", e);
    if (!rnet_forward(h, theta, R, n, nullptr, st, &e))
        return fail(h, PINN_ERR_UNSUPPORTED, "no R-net kernel for this (HE, HG)");
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_rnet_fwd", e);
    TrunkArgs A{};
    A.x = x; A.y = y; A.z = z; A.R = R; A.E = h->E; A.g = h->g; A.mask = mask;
    A.n = n; A.a = ham->a; A.b = ham->b; A.c = ham->c; A.P = ham->P; A.out_scale = a.out_scale;
    A.invN = ham->invN; A.invN1 = ham->invN1; A.invN2 = ham->invN2;
    A.bc_counts = (const long long*)bc_counts;
    A.Eb = h->Eb; A.gb = h->gb; A.partial = h->partT; A.stride = h->strideT;
    A.shares = trunk_shares(h, n, gT, trunk_trainable ? MODE_BWD : MODE_HEAD);
#ifdef PINN_STAMPS
    if (const char* sp = getenv("PINN_STAMP_PTR")) A.psi = (float*)strtoull(sp, nullptr, 16);   // diagnostic build only
#endif
    const int slot = h->prof_slots ? (int)(h->prof_calls % h->prof_slots) : 0;
    if (h->prof_slots) HIPCHK(h, hipEventRecord(h->ev[2 * slot], st));
    if (!run_trunk(h, trunk_trainable ? MODE_BWD : MODE_HEAD, theta, A, gT, st, &e))
        return fail(h, PINN_ERR_UNSUPPORTED, "no trunk kernel for this (H, L, precision, mode)");
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_trunk<BWD>", e);
    if (h->prof_slots) {
        HIPCHK(h, hipEventRecord(h->ev[2 * slot + 1], st));
        h->prof_calls++;
    }
    if (rmfma)
        e = launch_rnet_mfma_bwd(a.act, theta + h->PT, R, h->Eb, h->gb, n, h->partR, h->strideR, gR,
                                 rnet_shares(h, n, gR, true), st);
    else if (!launch_rnet_bwd(a.HE, a.HG, a.act, theta + h->PT, h->thT + h->PT, R, h->Eb, h->gb, n, h->partR, h->strideR, gR, st, &e))
        return fail(h, PINN_ERR_UNSUPPORTED, "no R-net kernel for this (HE, HG)");
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_rnet_bwd", e);
    if (!out) {      // the per-block partial sums stay in the handle: pinn_adam_fused reduces them and steps in one launch
        h->pend_gT = gT; h->pend_gR = gR; h->pend_mask = trainable_mask;
        return PINN_OK;
    }
    const int tot = h->P + 3;
    hipLaunchKernelGGL(k_reduce, dim3((tot + 63) / 64), dim3(1024), 0, st, h->partT, gT, h->strideT,
                       h->PT, h->partR, gR, h->strideR, h->PR, h->tensor_of, trainable_mask, out);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_reduce", e);
    return PINN_OK;
}

extern "C" int pinn_adam_dp(pinn_handle* h, float* theta, const float* grad, float* m, float* v,
                            int32_t* step_count, double lr, double beta1, double beta2, double eps,
                            uint64_t trainable_mask, float* best_theta, float* best_loss,
                            const float* msg_counts, int64_t* counts_out, void* stream)
{
    if (!h) return PINN_ERR_ARG;
    if (!theta || !grad || !m || !v || !step_count) return fail(h, PINN_ERR_ARG, "null pointer");
    const int blocks = (h->P + 1023) / 1024;
    hipLaunchKernelGGL(k_adam, dim3(blocks < kAdamMaxBlocks ? blocks : kAdamMaxBlocks), dim3(1024), 0,
                       (hipStream_t)stream, theta, grad, m, v, step_count, h->P, lr, beta1, beta2, eps,
                       h->tensor_of, trainable_mask, best_theta, best_loss,
                       reinterpret_cast<unsigned*>(h->samp + 2 * kSampleMaxBlocks + 1), msg_counts,
                       reinterpret_cast<long long*>(counts_out));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_adam", e);
    return PINN_OK;
}

extern "C" int pinn_adam_fused(pinn_handle* h, float* theta, float* grad_out, float* m, float* v, int32_t* step_count,
                               double lr, double beta1, double beta2, double eps, float* best_theta, float* best_loss,
                               void* stream)
{
    if (!h) return PINN_ERR_ARG;
    if (!theta || !m || !v || !step_count) return fail(h, PINN_ERR_ARG, "null pointer");
    if (h->pend_gT <= 0) return fail(h, PINN_ERR_ARG, "pinn_adam_fused: no pinn_residual_fwd_bwd call with out == NULL before it");
    const int tot = h->P + 3;
    hipLaunchKernelGGL(k_reduce_adam, dim3((tot + 63) / 64), dim3(1024), 0, (hipStream_t)stream, h->partT, h->pend_gT,
                       h->strideT, h->PT, h->partR, h->pend_gR, h->strideR, h->PR, h->tensor_of, h->pend_mask, grad_out,
                       theta, m, v, step_count, lr, beta1, beta2, eps, best_theta, best_loss,
                       reinterpret_cast<unsigned*>(h->samp + 2 * kSampleMaxBlocks + 1));
    h->pend_gT = h->pend_gR = 0;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_reduce_adam", e);
    return PINN_OK;
}

extern "C" int pinn_adam(pinn_handle* h, float* theta, const float* grad, float* m, float* v,
                         int32_t* step_count, double lr, double beta1, double beta2, double eps,
                         uint64_t trainable_mask, float* best_theta, float* best_loss, void* stream)
{
    return pinn_adam_dp(h, theta, grad, m, v, step_count, lr, beta1, beta2, eps, trainable_mask, best_theta, best_loss,
                        nullptr, nullptr, stream);
}

extern "C" int pinn_sample_dp(pinn_handle* h, uint64_t seed, uint64_t step, const int32_t* step_dev,
                              int64_t first, int64_t n, float box, float Rlo, float Rhi, float cutoff,
                              float bc_cutoff, float* x, float* y, float* z, float* R, uint8_t* mask,
                              int64_t* counts, float* counts_f32, void* stream)
{
    if (!h) return PINN_ERR_ARG;
    if (!x || !y || !z || !R || n < 1 || first < 0) return fail(h, PINN_ERR_ARG, "null pointer or bad range");
    hipStream_t st = (hipStream_t)stream;
    const int64_t sblocks = (n + kSampleBlock - 1) / kSampleBlock;
    hipLaunchKernelGGL(k_sample, dim3((unsigned)(sblocks < kSampleMaxBlocks ? sblocks : kSampleMaxBlocks)), dim3(kSampleBlock), 0, st,
                       seed, step, step_dev, first, n, box, Rlo, Rhi, cutoff, bc_cutoff, x, y, z, R, mask,
                       (unsigned long long*)counts, h->samp, counts_f32);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, PINN_ERR_HIP, "k_sample", e);
    return PINN_OK;
}

extern "C" int pinn_sample(pinn_handle* h, uint64_t seed, uint64_t step, const int32_t* step_dev,
                           int64_t first, int64_t n, float box, float Rlo, float Rhi, float cutoff,
                           float bc_cutoff, float* x, float* y, float* z, float* R, uint8_t* mask,
                           int64_t* counts, void* stream)
{
    return pinn_sample_dp(h, seed, step, step_dev, first, n, box, Rlo, Rhi, cutoff, bc_cutoff, x, y, z, R, mask, counts,
                          nullptr, stream);
}
