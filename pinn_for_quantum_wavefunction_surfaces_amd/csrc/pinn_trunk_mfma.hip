// pinn_trunk_mfma.hip -- fused trunk kernels on the bf16 matrix cores.
//
// Same mathematics as k_trunk in pinn_kernels.hip (streams per neuron, hand-derived adjoint;
// train.py:41-57,71 / poc/main.py:247-303,341-355,403), different execution shape: the H x H layers
// are dense GEMMs, so they run on v_mfma_f32_16x16x32_bf16 (16x16x16 for the 16-wide trunk), and the
// streams are FOUR (value, d/dx, d/drho, Laplacian: see NST below), not the five Cartesian ones.
//
// Common to the kernels below:
//   * every GEMM is computed transposed, D[neuron][column] with column = (stream, point): the
//     streams of one point are 16-column tiles, so lane (g = lane>>4, c = lane&15) holds, for point
//     c, the neurons 16*jt + 4*g + r (r = 0..3) of a row tile jt and all streams of them in its own
//     registers: activation and adjoint algebra are purely lane-local.
//   * the accumulator layout of one layer IS the B-operand layout of the next (and of the
//     reverse pass): a lane's 8 bf16 of a k-chunk are its 4 neurons of row tiles 2kc and
//     2kc+1.  That fixes a permutation of the contraction index, applied once to the
//     weights when they are packed (k_pack_weights): W for the forward GEMM, W^T for the
//     input-adjoint GEMM, both in fragment order (lane-linear 16-byte reads).
//   * the weight gradient contracts over columns (points), i.e. across lanes: (adjoint,
//     input) B-fragments are read back transposed with ds_read_b64_tr_b16 as MFMA operands;
//     gradient tiles accumulate in registers for the whole kernel.  Vector gradients (biases,
//     w_out, the 2-column first layer) are lane-local running sums, reduced once at the end.
//   * per-block partials, fixed-order second stage (k_reduce): bitwise reproducible.
//
// k_trunk2      two hidden layers, symmetric double pass, full reverse sweep (the benchmark's shape,
//   H = 64 and 128, bf16): H/16 waves share a 16-point tile, both passes swept together.
// k_trunk2s     the same for train.py's single-pass formulation: two tiles per trip.
// k_trunk2x     k_trunk2 in split-bf16 (2x64).   k_trunkD   deep trunks (L = 4), bf16.   k_trunkDx   4x128 in split-bf16.
// k_trunk_mfma2 every other wide shape and mode (L = 4, single pass, forward-only, frozen trunk,
//   split-bf16): SPLIT waves share a 16-point tile, each owning 16 or 32 neurons, pass by pass;
//   layer outputs and adjoints are exchanged through LDS buffers in B-fragment order, which double
//   as the transposed-read image of the weight gradient.  Options per shape (M2Cfg): resident
//   blocks per CU, both-pass buffers (KEEP), operand prefetch (PIPE), fp32 saved state (SVF32),
//   weights streamed through LDS by LDS-DMA (WSTREAM), split-bf16 operands (X3: hi + lo planes,
//   three MFMAs per product), one layer-output buffer rebuilt in the reverse sweep (XONE).
// k_trunk16     the reference's 16-wide trunk (fp32-grade: three-piece split-bf16 forward).
// k_rnet_mfma_fwd / _bwd   E-net and gate.
//
// Precision: bf16 MFMA operands (activations, adjoints, weights), fp32 accumulation and
// fp32 elementwise math.  Tolerances are stated in tests/test_gpu_parity.py; plain bf16 is
// NOT the 1e-5 parity path, split-bf16 is.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "pinn_math.h"
#include "pinn_common.h"

#ifndef PINN_TRUNK_PART
#define PINN_TRUNK_PART 0
#endif

using namespace pinn;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

size_t mfma_packed_weight_bytes(int H, int L);

namespace {

// Streams per neuron.  The reference differentiates in Cartesian coordinates (value, d/dx, d/dy, d/dz,
// Laplacian: five streams).  Both orbitals' gradients -- hence every neuron's -- lie in the plane spanned
// by x^ and rho^ = (0, y, z)/rho, so |grad a|^2 = a_x^2 + a_rho^2 and FOUR streams (value, d/dx, d/drho,
// Laplacian) carry exactly the same information: 4/5 of the GEMM columns and of the stream algebra, and
// the 64 (stream, point) columns of a 16-point tile are exactly two K = 32 steps of the weight-gradient
// contraction (80 needed three, the last half empty).  Identical to the Cartesian form in exact arithmetic;
// held to the same oracle / reference tolerances.
constexpr int NST = 4;
constexpr int SLP = NST - 1;   // index of the Laplacian stream

// position of neuron slot (kc, g, j) of a B fragment: lane group g holds, as element j of
// k-chunk kc, neuron 32kc + 16(j>>2) + 4g + (j&3)  (= row tile 2kc + (j>>2), register j&3)
__host__ __device__ constexpr int slot_neuron(int kc, int g, int j)
{
    return 32 * kc + 16 * (j >> 2) + 4 * g + (j & 3);
}
// image / gradient-tile order q = 32kc + 8g + j  ->  neuron
__host__ __device__ constexpr int q_neuron(int q) { return slot_neuron(q >> 5, (q >> 3) & 3, q & 7); }

template <int H, int L>
struct MCfg {
    static constexpr int HT = H / 16;       // row tiles
    static constexpr int KC = H / 32;       // k-chunks of 32
    static constexpr int NW = L - 1;        // H x H layers
    static constexpr int PITCH = H + 8;     // bf16 elements per image row (16-B aligned, skews banks)
    static constexpr int IMG = 32 * PITCH;  // one image: 32 columns (2 streams x 16 points)
    static constexpr int WPK = NW * 2 * H * H;                 // packed bf16 weights
    static constexpr int NSMALL = 3 * H + NW * H + H + 4;      // W0(2H) b0(H) b_l wout bout
    static constexpr size_t LDS_BYTES = (size_t)WPK * 2 + NSMALL * 4 + kWavesPerBlock * 2 * IMG * 2;
};

// ------------------------------------------------------------------ weight pack
// fwd: element j of lane (g, row) of fragment (jt, kc) = W[16jt + row][slot_neuron(kc, g, j)]
// bwd: ...                                            = W[slot_neuron(kc, g, j)][16it + row]
// X3: a second plane with the bf16 of the remainder w - bf16(w) follows the first one
template <int H, int L, bool X3 = false>
__global__ void k_pack_weights(const float* __restrict__ theta, __bf16* __restrict__ out)
{
    using C = MCfg<H, L>;
    using Lay = TrunkLayout<H, L>;
    const int per = C::HT * C::KC * 64;  // 16-byte fragments-of-lane per (layer, dir)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= C::NW * 2 * per) return;
    const int l = t / (2 * per) + 1, rem = t % (2 * per), dir = rem / per, f = rem % per;
    const int lane = f & 63, kc = (f >> 6) % C::KC, tt = (f >> 6) / C::KC;
    const int g = lane >> 4, row = lane & 15;
    const float* W = theta + Lay::W(l);
    bf16x8 v, vlo;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int kn = slot_neuron(kc, g, j);
        const float w = dir == 0 ? W[(16 * tt + row) * H + kn] : W[kn * H + 16 * tt + row];
        v[j] = (__bf16)w;
        vlo[j] = (__bf16)(w - (float)v[j]);
    }
    const size_t plane = (size_t)C::NW * 2 * H * H;
    *reinterpret_cast<bf16x8*>(out + ((size_t)((l - 1) * 2 + dir) * H * H) + (size_t)f * 8) = v;
    if (X3) *reinterpret_cast<bf16x8*>(out + plane + ((size_t)((l - 1) * 2 + dir) * H * H) + (size_t)f * 8) = vlo;
}

// ------------------------------------------------------------------ small helpers
__device__ __forceinline__ f32x4 splat(float v) { return f32x4{v, v, v, v}; }

__device__ __forceinline__ u32x2 pack4(f32x4 v)  // 4 fp32 -> 4 bf16 (round to nearest even)
{
    bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    return __builtin_bit_cast(u32x2, b);
}

// hi = bf16(v), lo = bf16(v - hi): hi + lo carries 16 mantissa bits of v
__device__ __forceinline__ void split4(f32x4 v, u32x2& hi, u32x2& lo);

__device__ __forceinline__ f32x4 unpack4(u32x2 p)
{
    f32x4 r;
    r[0] = __builtin_bit_cast(float, p[0] << 16);
    r[1] = __builtin_bit_cast(float, p[0] & 0xffff0000u);
    r[2] = __builtin_bit_cast(float, p[1] << 16);
    r[3] = __builtin_bit_cast(float, p[1] & 0xffff0000u);
    return r;
}

__device__ __forceinline__ void split4(f32x4 v, u32x2& hi, u32x2& lo)
{
    hi = pack4(v);
    lo = pack4(v - unpack4(hi));
}

template <int ACT>
__device__ __forceinline__ void act012v(f32x4 a, f32x4& s, f32x4& s1, f32x4& s2)
{
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float x, y, z;
        act012<ACT, float>(a[r], x, y, z);
        s[r] = x; s1[r] = y; s2[r] = z;
    }
}

// the same from e2 = log2(e) * (-a) (sigmoid) or log2(e) * (-2a) (tanh): the v_exp_f32 argument itself
template <int ACT>
__device__ __forceinline__ void act012v_scaled(f32x4 e2, f32x4& s, f32x4& s1, f32x4& s2)
{
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float rc = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(e2[r]));
        s[r] = ACT == ACT_SIGMOID ? rc : 2.0f * rc - 1.0f;
    }
    if (ACT == ACT_SIGMOID) {
        s1 = s - s * s;
        s2 = s1 - 2.0f * s * s1;
    } else {
        s1 = 1.0f - s * s;
        s2 = -2.0f * s * s1;
    }
}

template <int ACT>
__device__ __forceinline__ void act123v(f32x4 s, f32x4& s1, f32x4& s2, f32x4& s3)
{
    if (ACT == ACT_SIGMOID) {
        s1 = s - s * s;
        s2 = s1 - 2.0f * s * s1;
        s3 = s1 - 6.0f * s1 * s1;
    } else {
        s1 = 1.0f - s * s;
        s2 = -2.0f * s * s1;
        s3 = s1 * (4.0f - 6.0f * s1);
    }
}

__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// Broadcast scalars and packed fp32 arithmetic.  Written as f32x4 expressions, `vector * scalar` makes hipcc
// broadcast a scalar that sits in the ODD register of a pair with v_pk_mul_f32 / v_pk_fma_f32 ... op_sel:[0,1]
// (the LOW result lane reading the HIGH dword of the operand pair).  On MI355X those instructions intermittently
// computed with 0 in place of that dword: frequently when the pair had just been returned by a ds_read_b128, rarely
// with VALU-produced operands when a second workgroup shared the CU -- k_trunk2 gave run-to-run different d/dx-stream
// values at full grids until this was found (profiles/r02_packed_opsel.md, tools/opselbench).  Two remedies here:
// k_trunk2 loads its ring values so that each sits in the LOW half of a pair (ring_pairs: the safe op_sel_hi:[1,0]
// broadcast, all arithmetic stays packed); k_trunkD (no registers to spare for that) and k_trunk16 (orbital values
// are VALU results in arbitrary registers) do that arithmetic on unpacked instructions (lin2*, fmac_bcast*).
// The eight stream values of one point record of the LDS ring (two orbitals x NST streams, consecutive floats at
// LDS byte address `addr`), each as the LOW element of a register pair of its own: ds_read2_b32 with the wanted
// dword first.  A scalar that sits in the low half of an aligned pair is broadcast by hipcc with op_sel_hi:[1,0];
// one in the high half (every second value of a ds_read_b128) with the op_sel:[0,1] form described above.
// tests/test_host_cpu.py checks the compiled kernels for that form.  (Waits for all of the wave's LDS operations.)
__device__ __forceinline__ void ring_pairs(unsigned addr, float (&in)[2][NST])
{
    static_assert(NST == 4, "eight stream values per record");
    typedef unsigned long long u64;
    u64 p0, p1, p2, p3, p4, p5, p6, p7;
    asm volatile("ds_read2_b32 %0, %8 offset0:0 offset1:1\n\tds_read2_b32 %1, %8 offset0:1 offset1:0\n\t"
                 "ds_read2_b32 %2, %8 offset0:2 offset1:3\n\tds_read2_b32 %3, %8 offset0:3 offset1:2\n\t"
                 "ds_read2_b32 %4, %8 offset0:4 offset1:5\n\tds_read2_b32 %5, %8 offset0:5 offset1:4\n\t"
                 "ds_read2_b32 %6, %8 offset0:6 offset1:7\n\tds_read2_b32 %7, %8 offset0:7 offset1:6\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7)
                 : "v"(addr) : "memory");
    in[0][0] = __builtin_bit_cast(float, (unsigned)p0); in[0][1] = __builtin_bit_cast(float, (unsigned)p1);
    in[0][2] = __builtin_bit_cast(float, (unsigned)p2); in[0][3] = __builtin_bit_cast(float, (unsigned)p3);
    in[1][0] = __builtin_bit_cast(float, (unsigned)p4); in[1][1] = __builtin_bit_cast(float, (unsigned)p5);
    in[1][2] = __builtin_bit_cast(float, (unsigned)p6); in[1][3] = __builtin_bit_cast(float, (unsigned)p7);
}

// The same eight values as SPLAT pairs (x, x): ds_read2_b32 with both offsets on the wanted dword.  A packed
// instruction then takes the pair as it is -- no op_sel form of either kind, and none of the v_mov_b32 with which
// hipcc otherwise builds (x, x) pairs for part of the broadcasts (24 per tile in k_trunk2 with ring_pairs).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ring_splats(unsigned addr, f32x2 (&in)[2][NST])
{
    static_assert(NST == 4, "eight stream values per record");
    asm volatile("ds_read2_b32 %0, %8 offset0:0 offset1:0\n\tds_read2_b32 %1, %8 offset0:1 offset1:1\n\t"
                 "ds_read2_b32 %2, %8 offset0:2 offset1:2\n\tds_read2_b32 %3, %8 offset0:3 offset1:3\n\t"
                 "ds_read2_b32 %4, %8 offset0:4 offset1:4\n\tds_read2_b32 %5, %8 offset0:5 offset1:5\n\t"
                 "ds_read2_b32 %6, %8 offset0:6 offset1:6\n\tds_read2_b32 %7, %8 offset0:7 offset1:7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(in[0][0]), "=&v"(in[0][1]), "=&v"(in[0][2]), "=&v"(in[0][3]),
                   "=&v"(in[1][0]), "=&v"(in[1][1]), "=&v"(in[1][2]), "=&v"(in[1][3])
                 : "v"(addr) : "memory");
}
__device__ __forceinline__ f32x4 bc4(f32x2 p) { return f32x4{p[0], p[1], p[0], p[1]}; }   // p = (x, x)

// hipcc's hazard recognizer does not look into inline assembly: the operands of these helpers must not be results
// of MFMA instructions (the callers pass weights, orbital values and results of ordinary VALU arithmetic); each is ONE
// asm block opened by a wait state, because an operand may be the result of the transcendental instruction right
// before it (the orbital values are), which a VALU instruction may read one wait state later at the earliest.
// (lin2 / fmac_bcast: the same on separate, schedulable asm statements without the wait state -- only for operands
// that are not transcendental results: k_trunkD's ring values, where ring_pairs' eight extra registers cost more
// than these instructions do)
__device__ __forceinline__ f32x4 lin2(f32x4 w0, float x0, f32x4 w1, float x1)   // w0 * x0 + w1 * x1
{
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float t;
        asm("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(w1[k]), "v"(x1));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r[k]) : "v"(w0[k]), "v"(x0), "v"(t));
    }
    return r;
}

__device__ __forceinline__ void fmac_bcast(f32x4& acc, f32x4 v, float x)        // acc += v * x
{
#pragma unroll
    for (int k = 0; k < 4; k++) asm("v_fmac_f32 %0, %1, %2" : "+v"(acc[k]) : "v"(v[k]), "v"(x));
}

__device__ __forceinline__ f32x4 mul_bcast(f32x4 v, float x)                     // v * x
{
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; k++) asm("v_mul_f32 %0, %1, %2" : "=v"(r[k]) : "v"(v[k]), "v"(x));
    return r;
}

__device__ __forceinline__ f32x4 lin2_t(f32x4 w0, float x0, f32x4 w1, float x1)
{
    float r0, r1, r2, r3;
    asm("s_nop 0\n\t"
        "v_mul_f32 %0, %8, %13\n\tv_mul_f32 %1, %9, %13\n\tv_mul_f32 %2, %10, %13\n\tv_mul_f32 %3, %11, %13\n\t"
        "v_fmac_f32 %0, %4, %12\n\tv_fmac_f32 %1, %5, %12\n\tv_fmac_f32 %2, %6, %12\n\tv_fmac_f32 %3, %7, %12"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(w0[0]), "v"(w0[1]), "v"(w0[2]), "v"(w0[3]), "v"(w1[0]), "v"(w1[1]), "v"(w1[2]), "v"(w1[3]), "v"(x0), "v"(x1));
    return f32x4{r0, r1, r2, r3};
}

__device__ __forceinline__ void fmac_bcast_t(f32x4& acc, f32x4 v, float x)
{
    float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
    asm("s_nop 0\n\t"
        "v_fmac_f32 %0, %4, %8\n\tv_fmac_f32 %1, %5, %8\n\tv_fmac_f32 %2, %6, %8\n\tv_fmac_f32 %3, %7, %8"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(x));
    acc = f32x4{a0, a1, a2, a3};
}

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float dot4(f32x4 a, f32x4 b)
{
    const f32x4 p = a * b;
    return (p[0] + p[1]) + (p[2] + p[3]);
}

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float xg_sum(float v)  // sum over the 4 lane groups of one point
{
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Sums of TWO per-lane values over the 4 lane groups of a point (lanes l, l^16, l^32, l^48) without an
// LDS round trip (__shfl_xor is ds_bpermute: ~100 cycles of exposed latency per stage).
// v_permlane32_swap(a, b): lanes 32-63 of a <-> lanes 0-31 of b, so a + b afterwards holds
// a.lo + a.hi in lanes 0-31 and b.lo + b.hi in lanes 32-63; v_permlane16_swap(t, t') (odd 16-lane rows
// of t <-> even rows of t') on two copies then adds the two rows of each half.  Result: the total of
// `a` in lane groups 0 and 1, the total of `b` in lane groups 2 and 3 (returned in one register).
// The s_nop are the VALU-write -> permlane-read wait states, which hipcc does not pad inside asm.
__device__ __forceinline__ float xg_sum_pair(float a, float b)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    float t = a + b, u = t;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(t), "+v"(u));
    return t + u;
}

// both totals in every lane: one more half swap of two copies of the paired result
__device__ __forceinline__ void xg_sum_both(float& a, float& b)
{
    float t = xg_sum_pair(a, b), u = t;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(t), "+v"(u));
    a = t;   // [t.lo, t.lo]: the total of a (lane groups 0, 1 held it)
    b = u;   // [t.hi, t.hi]: the total of b
}

__device__ __forceinline__ float row16_sum(float v)  // sum over the 16 points of a lane group
{
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// =====================================================================================
// Second generation: the 16-point tile of a "group" is shared by SPLIT waves, each owning
// 32 neurons (two row tiles = one k-chunk).  Per-wave state shrinks by SPLIT, so H = 64 runs
// two waves per SIMD (latency hiding) and H = 128 fits at all.  Waves of a group exchange
// B fragments (layer outputs, adjoints) through LDS exchange buffers
//     EB[(kc*NST + s)*64 + lane]   (16 bytes per lane, lane-linear: conflict-free b128)
// which every wave of the group reads directly as MFMA B operands, and which double as the
// [column][neuron] image of the weight-gradient contraction (ds_read_b64_tr_b16 straight
// from the exchange buffer: no extra image writes).  Workgroup barriers separate the
// exchange steps; all groups of a block run the same trip count so barriers always match.
// OCC: resident blocks per CU the kernel is built for; 0 = one block, never KEEP
template <int H, int L, int SPLIT, int WAVES, bool X3 = false, int OCC = 1>
struct M2Cfg {
    static constexpr int HT = H / 16, KC = H / 32, NW = L - 1;
    static constexpr int GROUPS = WAVES / SPLIT;
    static constexpr int RT = H / SPLIT / 16;                     // row tiles (16 neurons) per wave
    static_assert(RT == 1 || RT == 2, "a wave owns 16 or 32 neurons");
    static constexpr int NPL = X3 ? 2 : 1;                        // operand planes (hi, lo)
    static constexpr int WPK_ALL = NPL * NW * 2 * H * H;          // bf16 elements
    // packed weights are staged in LDS when they fit beside the exchange buffers; otherwise
    // (4x128: 192 KB) they are streamed through two one-matrix LDS buffers (WSTREAM), or, in
    // split-bf16 mode, the A fragments are read from global memory, where they stay L2-resident
    static constexpr bool WLDS = (size_t)WPK_ALL * 2 <= 64 * 1024;
    // split-bf16 whose two planes do not fit: the hi plane sits in LDS, the lo plane is read from L2
    static constexpr bool WHI = X3 && !WLDS && (size_t)(WPK_ALL / 2) * 2 <= 64 * 1024;
    // WSTREAM: one layer matrix (its hi plane in split-bf16 mode) at a time, double-buffered in LDS by
    // LDS-DMA (global_load_lds_dwordx4) while the previous layer computes; the packed set stays in L2,
    // where the lo plane of split-bf16 is read from
    static constexpr bool WSTREAM = !WLDS && !WHI;
    // XONE (4x128 split-bf16: three two-plane layer-output buffers do not fit): ONE exchange buffer for
    // layer outputs.  The forward overwrites it layer by layer (one more barrier per layer); the reverse
    // sweep rebuilds the outputs of layer l-2 from the saved state just before the weight gradient of
    // layer l-1 needs them as its X operand.
    static constexpr bool XONE = X3 && !WLDS && !WHI && L > 2;
    static constexpr int NXL = XONE ? 1 : L - 1;                  // resident layer-output buffers per pass slot
    static constexpr int WPK = WLDS ? WPK_ALL : (WSTREAM ? 2 * H * H : (WHI ? WPK_ALL / 2 : 0));
    // deep wide trunks keep the k loop rolled: unrolled, the scheduler hoists all 20 B-fragment
    // reads of a layer and the kernel spills
    static constexpr int KCU = (L > 2 && H >= 128) ? 1 : KC;
    // explicit software pipeline of the GEMM phases: the operand fragments of k-chunk kc+1 are read
    // from LDS before the MFMAs of chunk kc issue (scheduling fences keep that order), so the
    // matrix pipe does not sit behind one LDS round trip per instruction
    static constexpr bool PIPE = !X3 && KCU == KC;
    // saved pre-activations as fp32 (no pack on the way in, no unpack in the reverse sweep: ~30 VALU
    // instructions less per row tile and pass) where the registers are there; bf16 pairs otherwise
    static constexpr bool SVF32 = X3 || (H == 64 && L == 2);
    static constexpr int NSMALL = 3 * H + NW * H + H + 4;         // floats
    static constexpr int EB1 = KC * NST * 64 * 16;                // bytes of one plane of an exchange buffer
    static constexpr int EB = NPL * EB1;
    static constexpr int SBYTES = SPLIT * 16 * 2 * 4;             // per-group partial (Sv, Sl)
    static constexpr int ZBYTES = 0;
    static constexpr size_t OFF_SMALL = (size_t)WPK * 2;
    static constexpr size_t OFF_ZERO = OFF_SMALL + NSMALL * 4;
    static constexpr size_t OFF_GRP = OFF_ZERO + ZBYTES;
    // KEEP: every pass of the symmetric double pass has its own XB buffers (and register slot),
    // so the reverse sweeps reuse the forward state instead of recomputing pass A -- when the
    // second set fits the 160 KB of LDS (for all OCC resident blocks of a CU)
    static constexpr bool KEEP = !XONE && OCC >= 1 && OCC * (OFF_GRP + (size_t)GROUPS * (2 * (L - 1) * EB + EB + SBYTES)) <= 160 * 1024;
    static constexpr int NXB = KEEP ? 2 : 1;
    static constexpr int GBYTES = NXB * NXL * EB + EB + SBYTES;   // XB[slot][0..NXL-1], YB, S
    static constexpr size_t LDS_BYTES = OFF_GRP + (size_t)GROUPS * GBYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "exchange buffers + weights exceed the CU's LDS");
};

template <int H, int L, int ACT, int NPASS, int MODE, int SPLIT, int WAVES, bool X3, int OCC>
__global__ __launch_bounds__(WAVES * 64, (OCC > 1 ? OCC : 1) * WAVES / 4) void k_trunk_mfma2(const float* __restrict__ theta,
                                                                      const __bf16* __restrict__ wpk_g,
                                                                      TrunkArgs A)
{
    using C = M2Cfg<H, L, SPLIT, WAVES, X3, OCC>;
    using Lay = TrunkLayout<H, L>;
    constexpr int HT = C::HT, KC = C::KC, NW = C::NW, NT = WAVES * 64, RT = C::RT;
    constexpr size_t WPLANE = (size_t)NW * 2 * H * H;   // elements of one weight plane
    constexpr int EB1 = C::EB1;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* wpk = reinterpret_cast<__bf16*>(lds_raw);
    float* small = reinterpret_cast<float*>(lds_raw + C::OFF_SMALL);
    float* sW0 = small;
    float* sb0 = small + 2 * H;
    float* sbl = small + 3 * H;
    float* swo = small + 3 * H + NW * H;
    float* sbo = swo + H;
    // LDS addresses are 32-bit: every access below is (uniform base) + (one lane register) + immediate
    typedef __attribute__((address_space(3))) unsigned char* lds_p;
    const lds_p lds0 = (lds_p)lds_raw;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave / SPLIT, part = wave % SPLIT;
    const int g = lane >> 4, c = lane & 15;
    const lds_p gbase = lds0 + C::OFF_GRP + grp * C::GBYTES;
    constexpr bool KEEP = C::KEEP && NPASS == 2;
    constexpr bool SVF = C::SVF32;
    constexpr int NS = KEEP ? 2 : 1;                    // pass slots (registers and XB sets)
    constexpr bool XONE = C::XONE;
    constexpr int NXL = C::NXL;
    const lds_p XB = gbase;                             // [slot][NXL] exchange buffers: outputs of layers 0..L-2
    const lds_p YB = gbase + C::NXB * NXL * C::EB;      // adjoint exchange buffer
    float* SB = reinterpret_cast<float*>(lds_raw + C::OFF_GRP + (size_t)grp * C::GBYTES + (size_t)C::NXB * NXL * C::EB + C::EB);   // [SPLIT][16][2]
    const int lane16 = lane * 16;                       // lane-linear 16-byte fragment slot
    // transposed-read lane offset: row (8g + 4hh + (c>>2)) -> stream half (g>>1), point 8(g&1)+(c>>2);
    // 8-byte piece (c&1) of chunk ((c&3)>>1) of the tile: everything else is an immediate
    const int trl = (g >> 1) * 1024 + (8 * (g & 1) + (c >> 2)) * 16 + (c & 1) * 8 + ((c & 3) >> 1) * 256;
    // this wave's piece of B fragment (k-chunk, stream s): all 16 bytes (RT = 2) or one 8-byte half
    const int own_kc = (RT * part) >> 1, own_half = (RT * part) & 1;
    const int own_slot = lane16 + own_kc * NST * 64 * 16 + 8 * own_half;   // byte offset of the own piece
    auto put_own = [&](lds_p eb, int s, const u32x4& v) {
        const lds_p p = eb + own_slot + s * 64 * 16;
        if (RT == 2) *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(p) = v;
        else *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(p) = u32x2{v[0], v[1]};
    };
    // fragment of RT f32x4 row-tile vectors -> hi plane (and lo plane at +EB1 in X3 mode)
    auto put_split = [&](lds_p eb, int s, const f32x4 (&v)[RT]) {
        u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int tt = 0; tt < RT; tt++) {
            u32x2 ph, pl;
            if (X3) split4(v[tt], ph, pl);
            else ph = pack4(v[tt]);
            hi[2 * tt] = ph[0]; hi[2 * tt + 1] = ph[1];
            if (X3) { lo[2 * tt] = pl[0]; lo[2 * tt + 1] = pl[1]; }
        }
        put_own(eb, s, hi);
        if (X3) put_own(eb + EB1, s, lo);
    };
    // one product a*b as 1 (bf16) or 3 (split bf16: hi*hi + hi*lo + lo*hi) MFMAs
    auto mfma3 = [&](bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 c) {
        c = mfma(ah, bh, c);
        if (X3) { c = mfma(ah, bl, c); c = mfma(al, bh, c); }
        return c;
    };

    {
        if (C::WLDS || C::WHI) {
            const uint4* src = reinterpret_cast<const uint4*>(wpk_g);
            uint4* dst = reinterpret_cast<uint4*>(wpk);
            for (int k = threadIdx.x; k < C::WPK / 8; k += NT) dst[k] = src[k];
        }
        for (int k = threadIdx.x; k < 2 * H; k += NT) sW0[k] = theta[Lay::W(0) + k];
        for (int k = threadIdx.x; k < H; k += NT) {
            sb0[k] = theta[Lay::b(0) + k];
            swo[k] = theta[Lay::Wout + k];
#pragma unroll
            for (int l = 1; l < L; l++) sbl[(l - 1) * H + k] = theta[Lay::b(l) + k];
        }
        if (threadIdx.x == 0) sbo[0] = theta[Lay::bout];
    }
    const __bf16* wsrc = (C::WLDS || C::WHI) ? wpk : wpk_g;   // one provenance per instantiation
    // lo plane: behind the hi plane in the same image, or (WHI, WSTREAM) in global memory
    const __bf16* wlo_f = (C::WHI || C::WSTREAM) ? wpk_g + WPLANE : wsrc + WPLANE;
    // ---- weight streaming: matrix use k reads LDS buffer k & 1; at the start of use k every wave
    // copies its share of use k+1's matrix into the other buffer (free since the barrier that ended
    // use k-1); the barrier that ends use k drains the copy (vmcnt(0)) before anyone reads it.
    constexpr bool WSTREAM = C::WSTREAM;
    constexpr int WBUF = H * H * 2;                     // bytes of one matrix
    int wuse = 0;
    auto wstage = [&](int mat, int into) {              // mat = (l-1)*2 + dir, < 0: nothing follows
        if (!WSTREAM || mat < 0) return;
        const __bf16* src = wpk_g + (size_t)mat * H * H;
        const lds_p dst = lds0 + (into & 1) * WBUF;
        for (int f = wave; f < HT * KC; f += WAVES)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + ((size_t)f * 64 + lane) * 8),
                (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
    };
    auto wcur = [&]() { return reinterpret_cast<const __bf16*>(lds_raw + (wuse & 1) * WBUF); };
    wstage(0, 0);                                       // first use: layer 1 forward
    __syncthreads();

    // ---- persistent gradient state (own row tiles RT*part + tt, tt < RT)
    f32x4 wacc[NW][RT][HT];
    f32x4 bacc[L][RT], w0acc[RT][2], woacc[RT];
    float acc_bout = 0.f, acc_pde = 0.f, acc_bc1 = 0.f, acc_bc2 = 0.f;
    if (MODE == MODE_BWD) {
#pragma unroll
        for (int l = 0; l < NW; l++)
#pragma unroll
            for (int t = 0; t < RT; t++)
#pragma unroll
                for (int b = 0; b < HT; b++) wacc[l][t][b] = splat(0.f);
#pragma unroll
        for (int t = 0; t < RT; t++) {
#pragma unroll
            for (int l = 0; l < L; l++) bacc[l][t] = splat(0.f);
            w0acc[t][0] = splat(0.f); w0acc[t][1] = splat(0.f); woacc[t] = splat(0.f);
        }
    }
    float invN1 = A.invN1, invN2 = A.invN2;
    if (MODE != MODE_FWD && A.bc_counts) {
        const long long c1 = A.bc_counts[0], c2 = A.bc_counts[1];
        invN1 = c1 > 0 ? 1.0f / (float)c1 : 0.0f;
        invN2 = c2 > 0 ? 1.0f / (float)c2 : 0.0f;
    }
    // per-point sums are taken by ONE lane group of ONE wave of the group
    const float lead = (g == 0 && part == 0) ? 1.0f : 0.0f;
    const int64_t ntile = (A.n + 15) / 16;
    const int64_t tstride = (int64_t)gridDim.x * C::GROUPS;
    const int64_t iters = (ntile + tstride - 1) / tstride;   // same for every wave: barriers match

    // the point record of the NEXT tile is requested one trip ahead (7 loads per lane that nothing waits
    // for until the next trip): without this every trip opened with an exposed round trip to L2/HBM
    struct PointRec { float x, y, z, R, E, g; unsigned mk; };
    auto load_rec = [&](int64_t itn) {
        const int64_t tl = itn * tstride + (int64_t)blockIdx.x * C::GROUPS + grp;
        int64_t id = tl * 16 + c;
        id = (tl < ntile && id < A.n) ? id : A.n - 1;
        PointRec r;
        r.x = A.x[id]; r.y = A.y[id]; r.z = A.z[id]; r.R = A.R[id]; r.E = A.E[id]; r.g = A.g[id];
        r.mk = (MODE != MODE_FWD && A.mask) ? A.mask[id] : 0u;
        return r;
    };
    PointRec nxt = load_rec(0);
    for (int64_t itx = 0; itx < iters; itx++) {
        const int64_t tile = itx * tstride + (int64_t)blockIdx.x * C::GROUPS + grp;
        const int64_t idx = tile * 16 + c;
        const bool valid = tile < ntile && idx < A.n;
        const PointRec cur = nxt;
        if (itx + 1 < iters) nxt = load_rec(itx + 1);
        const float xv = cur.x, yv = cur.y, zv = cur.z, Rv = cur.R;
        const float Ev = cur.E, gv = cur.g;
        const float m1 = float(cur.mk & 1), m2 = float((cur.mk >> 1) & 1);
        float inA[2][NST], ri1, ri2;
        {
            const float rho2 = yv * yv + zv * zv, rho = __builtin_amdgcn_sqrtf(rho2);
            orbital_cyl<float>(xv - Rv, rho2, rho, inA[0], ri1);
            orbital_cyl<float>(xv + Rv, rho2, rho, inA[1], ri2);
        }
        const float F = inA[0][0] + A.P * inA[1][0];
        const float lapF = inA[0][SLP] + A.P * inA[1][SLP];

        Tail<float> t;
        // saved forward state per pass slot: (h, a_x, a_y, a_z, a_lap) of the own neurons, layers
        // 1..L-1, as fp32 or bf16 pairs (C::SVF32; always fp32 in split-bf16 mode: the adjoint algebra
        // must not be the precision bottleneck) and the layer-0 activation.  With KEEP both passes keep their own
        // slot (and their own XB buffers in LDS), so no pass is evaluated twice.
        u32x2 sv[NS][SVF ? 1 : NW][RT][NST];
        f32x4 svf[NS][SVF ? NW : 1][RT][NST];
        f32x4 h0[NS][RT];
        auto fwd = [&](const bool useB, auto SLc, float& Sv, float& Sl, const int wnext) {
            constexpr int SL = decltype(SLc)::value;
            const lds_p XBs = XB + SL * NXL * C::EB;
            float in[2][NST];
#pragma unroll
            for (int s = 0; s < NST; s++) {
                in[0][s] = useB ? inA[1][s] : inA[0][s];
                in[1][s] = useB ? inA[0][s] : inA[1][s];
            }
            // =========================== forward ===========================
            // saved (h, a_x, a_y, a_z, a_lap) of the own neurons, layers 1..L-1: bf16 pairs, or
            // fp32 in split-bf16 mode (the adjoint algebra must not be the precision bottleneck)
            {   // ---- layer 0 on the VALU (K = 2), own neurons only
                f32x4 o[NST][RT];
#pragma unroll
                for (int tt = 0; tt < RT; tt++) {
                    const int n0 = 16 * (RT * part + tt) + 4 * g;
                    const f32x4 wA = *reinterpret_cast<const f32x4*>(sW0 + n0 * 2);
                    const f32x4 wB = *reinterpret_cast<const f32x4*>(sW0 + n0 * 2 + 4);
                    const f32x4 w0 = {wA[0], wA[2], wB[0], wB[2]}, w1 = {wA[1], wA[3], wB[1], wB[3]};
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sb0 + n0);
                    f32x4 a[NST];
#pragma unroll
                    for (int s = 0; s < NST; s++) a[s] = w0 * in[0][s] + w1 * in[1][s];
                    a[0] += b0;
                    f32x4 s0, s1, s2;
                    act012v<ACT>(a[0], s0, s1, s2);
                    const f32x4 q = a[1] * a[1] + a[2] * a[2];
                    o[0][tt] = s0;
                    o[1][tt] = s1 * a[1];
                    o[2][tt] = s1 * a[2];
                    o[SLP][tt] = s1 * a[SLP] + s2 * q;
                    h0[SL][tt] = s0;
                }
#pragma unroll
                for (int s = 0; s < NST; s++) put_split(XBs, s, o[s]);
            }
            __syncthreads();
            Sv = 0.f; Sl = 0.f;
#pragma unroll
            for (int l = 1; l < L; l++) {
                wstage(l < L - 1 ? l * 2 : wnext, wuse + 1);
                const __bf16* wf = WSTREAM ? wcur() : wsrc + (size_t)((l - 1) * 2 + 0) * H * H;
                const __bf16* wfl = wlo_f + (size_t)((l - 1) * 2 + 0) * H * H;   // lo plane of the same matrix
                const lds_p xin = XBs + (XONE ? 0 : l - 1) * C::EB + lane16;
                f32x4 a[RT][NST];
#pragma unroll
                for (int tt = 0; tt < RT; tt++) {
                    a[tt][0] = *reinterpret_cast<const f32x4*>(sbl + (l - 1) * H + 16 * (RT * part + tt) + 4 * g);
#pragma unroll
                    for (int s = 1; s < NST; s++) a[tt][s] = splat(0.f);
                }
                if constexpr (C::PIPE) {
                    bf16x8 wq[2][RT], xq[2][NST];
                    auto fetch = [&](int kc, bf16x8 (&w)[RT], bf16x8 (&x)[NST]) {
#pragma unroll
                        for (int tt = 0; tt < RT; tt++)
                            w[tt] = *reinterpret_cast<const bf16x8*>(wf + (((RT * part + tt) * KC + kc) * 64 + lane) * 8);
#pragma unroll
                        for (int s = 0; s < NST; s++)
                            x[s] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(xin + (kc * NST + s) * 64 * 16);
                    };
                    fetch(0, wq[0], xq[0]);
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) {
                        if (kc + 1 < KC) fetch(kc + 1, wq[(kc + 1) & 1], xq[(kc + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int s = 0; s < NST; s++)
#pragma unroll
                            for (int tt = 0; tt < RT; tt++) a[tt][s] = mfma(wq[kc & 1][tt], xq[kc & 1][s], a[tt][s]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll C::KCU
                for (int kc = 0; kc < KC; kc++) {
                    bf16x8 wa[RT], wl[RT];
#pragma unroll
                    for (int tt = 0; tt < RT; tt++) {
                        const int fo = (((RT * part + tt) * KC + kc) * 64 + lane) * 8;
                        wa[tt] = *reinterpret_cast<const bf16x8*>(wf + fo);
                        wl[tt] = X3 ? *reinterpret_cast<const bf16x8*>(wfl + fo) : wa[tt];
                    }
#pragma unroll
                    for (int s = 0; s < NST; s++) {
                        const lds_p px = xin + (kc * NST + s) * 64 * 16;
                        const bf16x8 xb = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(px);
                        const bf16x8 xl = X3 ? *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(px + EB1) : xb;
#pragma unroll
                        for (int tt = 0; tt < RT; tt++) a[tt][s] = mfma3(wa[tt], wl[tt], xb, xl, a[tt][s]);
                    }
                }
                }
                f32x4 o[NST][RT];
#pragma unroll
                for (int tt = 0; tt < RT; tt++) {
                    f32x4 s0, s1, s2;
                    act012v<ACT>(a[tt][0], s0, s1, s2);
                    const f32x4 q = a[tt][1] * a[tt][1] + a[tt][2] * a[tt][2];
                    const f32x4 ol = s1 * a[tt][SLP] + s2 * q;
                    if constexpr (SVF) {
                        svf[SL][l - 1][tt][0] = s0;
#pragma unroll
                        for (int s = 1; s < NST; s++) svf[SL][l - 1][tt][s] = a[tt][s];
                    } else {
                        sv[SL][l - 1][tt][0] = pack4(s0);
#pragma unroll
                        for (int s = 1; s < NST; s++) sv[SL][l - 1][tt][s] = pack4(a[tt][s]);
                    }
                    if (l < L - 1) {
                        o[0][tt] = s0;
                        o[1][tt] = s1 * a[tt][1];
                        o[2][tt] = s1 * a[tt][2];
                        o[SLP][tt] = ol;
                    } else {
                        const f32x4 wo = *reinterpret_cast<const f32x4*>(swo + 16 * (RT * part + tt) + 4 * g);
                        const f32x4 pv = wo * s0, pl = wo * ol;
                        Sv += (pv[0] + pv[1]) + (pv[2] + pv[3]);
                        Sl += (pl[0] + pl[1]) + (pl[2] + pl[3]);
                    }
                }
                if (l < L - 1) {
                    const lds_p xout = XBs + (XONE ? 0 : l) * C::EB;
                    if (XONE) __syncthreads();   // one buffer: everyone is done reading layer l-1's outputs
#pragma unroll
                    for (int s = 0; s < NST; s++) put_split(xout, s, o[s]);
                    __syncthreads();
                }
                if (WSTREAM) wuse++;
            }
            // ---- combine the partial output sums of the group's waves (fixed order)
            {
                const float tot = xg_sum_pair(Sv, Sl);      // Sv in lane groups 0,1; Sl in groups 2,3 (no LDS trip)
                if ((g & 1) == 0) SB[(part * 16 + c) * 2 + (g >> 1)] = tot;
            }
            __syncthreads();
            Sv = 0.f; Sl = 0.f;
#pragma unroll
            for (int pp = 0; pp < SPLIT; pp++) { Sv += SB[(pp * 16 + c) * 2]; Sl += SB[(pp * 16 + c) * 2 + 1]; }
        };
        auto rev = [&](const bool useB, auto SLc, const int wnext) {
            constexpr int SL = decltype(SLc)::value;
            const lds_p XBs = XB + SL * NXL * C::EB;
            float in[2][NST];
#pragma unroll
            for (int s = 0; s < NST; s++) {
                in[0][s] = useB ? inA[1][s] : inA[0][s];
                in[1][s] = useB ? inA[0][s] : inA[1][s];
            }
            // =========================== reverse ===========================
            const float pw = useB ? A.P : 1.0f;
            const float cv = A.out_scale * pw * t.Nb, cl = A.out_scale * pw * t.lapNb;
            {   // ---- last hidden layer, own neurons
                f32x4 yo[NST][RT];
#pragma unroll
                for (int tt = 0; tt < RT; tt++) {
                    f32x4 h, a[NST];
                    if constexpr (SVF) {
                        h = svf[SL][NW - 1][tt][0];
#pragma unroll
                        for (int s = 1; s < NST; s++) a[s] = svf[SL][NW - 1][tt][s];
                    } else {
                        h = unpack4(sv[SL][NW - 1][tt][0]);
#pragma unroll
                        for (int s = 1; s < NST; s++) a[s] = unpack4(sv[SL][NW - 1][tt][s]);
                    }
                    f32x4 s1, s2, s3;
                    act123v<ACT>(h, s1, s2, s3);
                    const f32x4 q = a[1] * a[1] + a[2] * a[2];
                    const f32x4 wo = *reinterpret_cast<const f32x4*>(swo + 16 * (RT * part + tt) + 4 * g);
                    woacc[tt] += cv * h + cl * (s1 * a[SLP] + s2 * q);
                    const f32x4 o0 = cv * wo, ol = cl * wo;
                    const f32x4 u = 2.0f * s2 * ol;
                    yo[SLP][tt] = s1 * ol;
                    yo[1][tt] = u * a[1];
                    yo[2][tt] = u * a[2];
                    yo[0][tt] = s1 * o0 + (s2 * a[SLP] + s3 * q) * ol;
                    bacc[L - 1][tt] += yo[0][tt];
                }
#pragma unroll
                for (int s = 0; s < NST; s++) put_split(YB, s, yo[s]);
            }
            __syncthreads();
#pragma unroll
            for (int l = L - 1; l >= 1; l--) {
                const lds_p xin = XBs + (XONE ? 0 : l - 1) * C::EB;
                wstage(l - 1 >= 1 ? (l - 2) * 2 + 1 : wnext, wuse + 1);
                // ---- W_l-bar[own rows][all columns] += Y-bar (x) X, 32 of the 64 (stream, point) columns per
                //      phase, both operands read transposed straight out of the exchange buffers
#pragma unroll
                for (int ph = 0; ph < NST / 2; ph++) {
                    // rows k = 8g + 4hh + (c>>2) of the phase: stream 2ph + (k>>4), point k&15
                    auto frag = [&](lds_p base, int T, int plane) {
                        const int imm = ph * 2048 + (T >> 1) * NST * 64 * 16 + (T & 1) * 512 + plane * EB1;
                        const lds_p p0 = base + trl + imm;
                        const lds_p p1 = base + trl + imm + 64;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p1);
                        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    };
                    bf16x8 fy[RT], fyl[RT];
#pragma unroll
                    for (int tt = 0; tt < RT; tt++) {
                        fy[tt] = frag(YB, RT * part + tt, 0);
                        fyl[tt] = X3 ? frag(YB, RT * part + tt, 1) : fy[tt];
                    }
                    if constexpr (C::PIPE) {
                        bf16x8 fq[2];
                        fq[0] = frag(xin, 0, 0);
#pragma unroll
                        for (int it = 0; it < HT; it++) {
                            if (it + 1 < HT) fq[(it + 1) & 1] = frag(xin, it + 1, 0);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int tt = 0; tt < RT; tt++)
                                wacc[l - 1][tt][it] = mfma(fy[tt], fq[it & 1], wacc[l - 1][tt][it]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
#pragma unroll
                    for (int it = 0; it < HT; it++) {
                        const bf16x8 fx = frag(xin, it, 0);
                        const bf16x8 fxl = X3 ? frag(xin, it, 1) : fx;
#pragma unroll
                        for (int tt = 0; tt < RT; tt++)
                            wacc[l - 1][tt][it] = mfma3(fy[tt], fyl[tt], fx, fxl, wacc[l - 1][tt][it]);
                    }
                    }
                }
                // ---- adjoint of layer l's input for the own neurons, then layer l-1's activation adjoint
                const __bf16* wb = WSTREAM ? wcur() : wsrc + (size_t)((l - 1) * 2 + 1) * H * H;
                const __bf16* wbl = wlo_f + (size_t)((l - 1) * 2 + 1) * H * H;
                f32x4 ob[RT][NST];
#pragma unroll
                for (int tt = 0; tt < RT; tt++)
#pragma unroll
                    for (int s = 0; s < NST; s++) ob[tt][s] = splat(0.f);
                if constexpr (C::PIPE) {
                    bf16x8 wq[2][RT], yq[2][NST];
                    auto fetch = [&](int kc, bf16x8 (&w)[RT], bf16x8 (&y)[NST]) {
#pragma unroll
                        for (int tt = 0; tt < RT; tt++)
                            w[tt] = *reinterpret_cast<const bf16x8*>(wb + (((RT * part + tt) * KC + kc) * 64 + lane) * 8);
#pragma unroll
                        for (int s = 0; s < NST; s++)
                            y[s] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(YB + lane16 + (kc * NST + s) * 64 * 16);
                    };
                    fetch(0, wq[0], yq[0]);
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) {
                        if (kc + 1 < KC) fetch(kc + 1, wq[(kc + 1) & 1], yq[(kc + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int s = 0; s < NST; s++)
#pragma unroll
                            for (int tt = 0; tt < RT; tt++) ob[tt][s] = mfma(wq[kc & 1][tt], yq[kc & 1][s], ob[tt][s]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll C::KCU
                for (int kc = 0; kc < KC; kc++) {
                    bf16x8 wa[RT], wl[RT];
#pragma unroll
                    for (int tt = 0; tt < RT; tt++) {
                        const int fo = (((RT * part + tt) * KC + kc) * 64 + lane) * 8;
                        wa[tt] = *reinterpret_cast<const bf16x8*>(wb + fo);
                        wl[tt] = X3 ? *reinterpret_cast<const bf16x8*>(wbl + fo) : wa[tt];
                    }
#pragma unroll
                    for (int s = 0; s < NST; s++) {
                        const lds_p py = YB + lane16 + (kc * NST + s) * 64 * 16;
                        const bf16x8 yb = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(py);
                        const bf16x8 yl = X3 ? *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(py + EB1) : yb;
#pragma unroll
                        for (int tt = 0; tt < RT; tt++) ob[tt][s] = mfma3(wa[tt], wl[tt], yb, yl, ob[tt][s]);
                    }
                }
                }
                f32x4 yo[NST][RT];
#pragma unroll
                for (int tt = 0; tt < RT; tt++) {
                    f32x4 h, a[NST];
                    if (l - 1 == 0) {
                        const int n0 = 16 * (RT * part + tt) + 4 * g;
                        const f32x4 wA = *reinterpret_cast<const f32x4*>(sW0 + n0 * 2);
                        const f32x4 wB = *reinterpret_cast<const f32x4*>(sW0 + n0 * 2 + 4);
                        const f32x4 w0 = {wA[0], wA[2], wB[0], wB[2]}, w1 = {wA[1], wA[3], wB[1], wB[3]};
#pragma unroll
                        for (int s = 1; s < NST; s++) a[s] = w0 * in[0][s] + w1 * in[1][s];
                        h = h0[SL][tt];
                    } else if constexpr (SVF) {
                        h = svf[SL][l - 2][tt][0];
#pragma unroll
                        for (int s = 1; s < NST; s++) a[s] = svf[SL][l - 2][tt][s];
                    } else {
                        h = unpack4(sv[SL][l - 2][tt][0]);
#pragma unroll
                        for (int s = 1; s < NST; s++) a[s] = unpack4(sv[SL][l - 2][tt][s]);
                    }
                    f32x4 s1, s2, s3;
                    act123v<ACT>(h, s1, s2, s3);
                    const f32x4 q = a[1] * a[1] + a[2] * a[2];
                    const f32x4 u = 2.0f * s2 * ob[tt][SLP];
                    f32x4 ab[NST];
                    ab[SLP] = s1 * ob[tt][SLP];
                    ab[1] = s1 * ob[tt][1] + u * a[1];
                    ab[2] = s1 * ob[tt][2] + u * a[2];
                    ab[0] = s1 * ob[tt][0] + s2 * (a[1] * ob[tt][1] + a[2] * ob[tt][2]) +
                            (s2 * a[SLP] + s3 * q) * ob[tt][SLP];
                    bacc[l - 1][tt] += ab[0];
                    if (l - 1 == 0) {
                        f32x4 g0 = splat(0.f), g1 = splat(0.f);
#pragma unroll
                        for (int s = 0; s < NST; s++) { g0 += ab[s] * in[0][s]; g1 += ab[s] * in[1][s]; }
                        w0acc[tt][0] += g0;
                        w0acc[tt][1] += g1;
                    } else {
#pragma unroll
                        for (int s = 0; s < NST; s++) yo[s][tt] = ab[s];
                    }
                }
                if (l - 1 >= 1) {   // hand the next layer's adjoints to the group
                    __syncthreads();   // everyone is done reading YB (and, XONE, the layer-output buffer)
#pragma unroll
                    for (int s = 0; s < NST; s++) put_split(YB, s, yo[s]);
                    if constexpr (XONE) {
                        // the next weight gradient (layer l-1) contracts with the outputs of layer l-2:
                        // rebuild them from that layer's saved state into the one layer-output buffer
                        f32x4 xo[NST][RT];
#pragma unroll
                        for (int tt = 0; tt < RT; tt++) {
                            f32x4 h, a[NST];
                            if (l - 2 == 0) {
                                const int n0 = 16 * (RT * part + tt) + 4 * g;
                                const f32x4 wA = *reinterpret_cast<const f32x4*>(sW0 + n0 * 2);
                                const f32x4 wB = *reinterpret_cast<const f32x4*>(sW0 + n0 * 2 + 4);
                                const f32x4 w0 = {wA[0], wA[2], wB[0], wB[2]}, w1 = {wA[1], wA[3], wB[1], wB[3]};
#pragma unroll
                                for (int s = 1; s < NST; s++) a[s] = w0 * in[0][s] + w1 * in[1][s];
                                h = h0[SL][tt];
                            } else {
                                h = svf[SL][l - 3 >= 0 ? l - 3 : 0][tt][0];
#pragma unroll
                                for (int s = 1; s < NST; s++) a[s] = svf[SL][l - 3 >= 0 ? l - 3 : 0][tt][s];
                            }
                            f32x4 s1, s2, s3;
                            act123v<ACT>(h, s1, s2, s3);
                            const f32x4 q = a[1] * a[1] + a[2] * a[2];
                            xo[0][tt] = h;
                            xo[1][tt] = s1 * a[1];
                            xo[2][tt] = s1 * a[2];
                            xo[SLP][tt] = s1 * a[SLP] + s2 * q;
                        }
#pragma unroll
                        for (int s = 0; s < NST; s++) put_split(XBs, s, xo[s]);
                    }
                    __syncthreads();
                }
                if (WSTREAM) wuse++;
            }
            // the next forward overwrites the own XB slot, which the other waves of the group
            // read in their weight-gradient phase: wait until they are through
            __syncthreads();
        };
        auto finish = [&](float Sv, float Sl) {
            const float N = A.out_scale * Sv + sbo[0];
            const float lapN = A.out_scale * Sl;
            tail<float>(N, lapN, gv, Ev, F, lapF, ri1 + ri2, A.a, A.b, A.c, m1, m2, A.invN, invN1,
                        invN2, valid ? 1.0f : 0.0f, t);
            if (MODE == MODE_FWD) {
                if (valid && g == 0 && part == 0) {
                    if (A.psi) A.psi[idx] = t.psi;
                    if (A.lap) A.lap[idx] = t.lap;
                    if (A.res) A.res[idx] = t.res;
                }
            } else {
                if (valid && g == 0 && part == 0) { A.Eb[idx] = t.Eb; A.gb[idx] = t.gb; }
                acc_pde += lead * t.l_pde;
                acc_bc1 += lead * t.l_bc1;
                acc_bc2 += lead * t.l_bc2;
                acc_bout += lead * t.Nb;
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, NS - 1>;
        // matrix that the use after a pass's last one needs (weight streaming): layer 1 forward of
        // the next pass / tile, the last layer's transpose when a reverse sweep follows, or nothing
        constexpr int F1 = 0, BL = (L - 2) * 2 + 1;
        const int F1n = itx == iters - 1 ? -1 : F1;     // ... of the next tile
        if (NPASS == 1) {
            float Sv, Sl;
            fwd(false, S0{}, Sv, Sl, MODE == MODE_BWD ? BL : F1n);
            finish(Sv, Sl);
            if (MODE == MODE_BWD) rev(false, S0{}, F1n);
        } else if (KEEP) {
            float SvA, SlA, SvB, SlB;
            fwd(false, S0{}, SvA, SlA, F1);
            fwd(true, S1{}, SvB, SlB, MODE == MODE_BWD ? BL : F1n);
            finish(SvA + A.P * SvB, SlA + A.P * SlB);
            if (MODE == MODE_BWD) {
                rev(true, S1{}, BL);
                rev(false, S0{}, F1n);
            }
        } else {
            // one slot: 0 = forward(f1,f2); 1 = forward(f2,f1) + tail + its reverse; 2 = forward(f1,f2)
            // again + its reverse.  One copy of the code, three trips.
            float SvA = 0.f, SlA = 0.f;
            constexpr int nstage = MODE == MODE_BWD ? 3 : 2;
#pragma unroll 1
            for (int st = 0; st < nstage; st++) {
                const bool useB = st == 1;
                const bool reverse = MODE == MODE_BWD && st >= 1;
                const int after = st == nstage - 1 ? F1n : F1;
                float Sv, Sl;
                fwd(useB, S0{}, Sv, Sl, reverse ? BL : after);
                if (st == 0) { SvA = Sv; SlA = Sl; }
                if (st == 1) finish(SvA + A.P * Sv, SlA + A.P * Sl);
                if (reverse) rev(useB, S0{}, after);
            }
        }
    }
    if (MODE == MODE_FWD) return;
    // =========================== block reduction ===========================
    __syncthreads();
    constexpr int PT = Lay::size;
    float* out = A.partial + (size_t)blockIdx.x * A.stride;
    const float s0 = row16_sum(xg_sum(acc_pde)), s1r = row16_sum(xg_sum(acc_bc1)),
                s2r = row16_sum(xg_sum(acc_bc2)), s3r = row16_sum(xg_sum(acc_bout));
    if (MODE == MODE_BWD) {
#pragma unroll
        for (int tt = 0; tt < RT; tt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int l = 0; l < L; l++) bacc[l][tt][r] = row16_sum(bacc[l][tt][r]);
                w0acc[tt][0][r] = row16_sum(w0acc[tt][0][r]);
                w0acc[tt][1][r] = row16_sum(w0acc[tt][1][r]);
                woacc[tt][r] = row16_sum(woacc[tt][r]);
            }
    }
    if (C::GROUPS == 1) {
        // one group per block: every wave owns distinct gradient rows -> write them straight to
        // the block's partial row (the LDS image of a 4x128 trunk would not fit anyway)
        float* sc = reinterpret_cast<float*>(lds_raw);
        if (threadIdx.x < 4) sc[threadIdx.x] = 0.f;
        __syncthreads();
        if (MODE == MODE_BWD) {
#pragma unroll
            for (int l = 1; l < L; l++)
#pragma unroll
                for (int tt = 0; tt < RT; tt++)
#pragma unroll
                    for (int it = 0; it < HT; it++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int nj = q_neuron(16 * (RT * part + tt) + 4 * g + r), ni = q_neuron(16 * it + c);
                            out[Lay::W(l) + nj * H + ni] = wacc[l - 1][tt][it][r];
                        }
            if (c == 0) {
#pragma unroll
                for (int tt = 0; tt < RT; tt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int nj = 16 * (RT * part + tt) + 4 * g + r;
#pragma unroll
                        for (int l = 0; l < L; l++) out[Lay::b(l) + nj] = bacc[l][tt][r];
                        out[Lay::W(0) + 2 * nj] = w0acc[tt][0][r];
                        out[Lay::W(0) + 2 * nj + 1] = w0acc[tt][1][r];
                        out[Lay::Wout + nj] = woacc[tt][r];
                    }
            }
        } else {
            for (int k = threadIdx.x; k < PT; k += NT) out[k] = 0.0f;
        }
        for (int w = 0; w < WAVES; w++) {   // per-point sums live in part 0 only; fixed order anyway
            if (wave == w && lane == 0) { sc[0] += s0; sc[1] += s1r; sc[2] += s2r; sc[3] += s3r; }
            __syncthreads();
        }
        if (threadIdx.x < 3) out[PT + threadIdx.x] = sc[threadIdx.x];
        if (threadIdx.x == 3 && MODE == MODE_BWD) out[Lay::bout] = sc[3];
        return;
    }
    float* gimg = reinterpret_cast<float*>(lds_raw);
    float* sc = gimg + PT;
    if (MODE == MODE_BWD) {
        // zero the image once (each wave only touches its own rows; several groups add into them)
        for (int k = threadIdx.x; k < PT + 4; k += NT) gimg[k] = 0.f;
    } else {
        if (threadIdx.x < 4) sc[threadIdx.x] = 0.f;
    }
    __syncthreads();
    for (int w = 0; w < WAVES; w++) {
        if (wave == w) {
            if (MODE == MODE_BWD) {
#pragma unroll
                for (int l = 1; l < L; l++)
#pragma unroll
                    for (int tt = 0; tt < RT; tt++)
#pragma unroll
                        for (int it = 0; it < HT; it++)
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int nj = q_neuron(16 * (RT * part + tt) + 4 * g + r), ni = q_neuron(16 * it + c);
                                gimg[Lay::W(l) + nj * H + ni] += wacc[l - 1][tt][it][r];
                            }
                if (c == 0) {
#pragma unroll
                    for (int tt = 0; tt < RT; tt++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int nj = 16 * (RT * part + tt) + 4 * g + r;
#pragma unroll
                            for (int l = 0; l < L; l++) gimg[Lay::b(l) + nj] += bacc[l][tt][r];
                            gimg[Lay::W(0) + 2 * nj] += w0acc[tt][0][r];
                            gimg[Lay::W(0) + 2 * nj + 1] += w0acc[tt][1][r];
                            gimg[Lay::Wout + nj] += woacc[tt][r];
                        }
                }
            }
            if (lane == 0) { sc[0] += s0; sc[1] += s1r; sc[2] += s2r; sc[3] += s3r; }
        }
        __syncthreads();
    }
    if (MODE == MODE_BWD) {
        for (int k = threadIdx.x; k < PT; k += NT) out[k] = k == Lay::bout ? sc[3] : gimg[k];
    } else {
        for (int k = threadIdx.x; k < PT; k += NT) out[k] = 0.0f;
    }
    if (threadIdx.x < 3) out[PT + threadIdx.x] = sc[threadIdx.x];
}

template <int H, int L, int ACT, int NPASS, int MODE, int SPLIT, int WAVES, bool X3, int OCC>
hipError_t launch2_t(const float* theta, __bf16* wpk, const TrunkArgs& A, int grid, hipStream_t st)
{
    using C1 = MCfg<H, L>;
    using C = M2Cfg<H, L, SPLIT, WAVES, X3, OCC>;
    const int nfrag = C1::NW * 2 * C1::HT * C1::KC * 64;
    hipLaunchKernelGGL((k_pack_weights<H, L, X3>), dim3((nfrag + 255) / 256), dim3(256), 0, st, theta, wpk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    auto kern = k_trunk_mfma2<H, L, ACT, NPASS, MODE, SPLIT, WAVES, X3, OCC>;
    size_t lds = C::LDS_BYTES;
    const size_t need = C::GROUPS == 1 ? 64 : (size_t)(TrunkLayout<H, L>::size + 16) * 4;
    if (lds < need) lds = need;
    static bool attr_set[64] = {};   // once per instantiation and device: not a stream operation (graph capture)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, st, theta, (const __bf16*)wpk, A);
    return hipGetLastError();
}

template <int H, int L, int SPLIT, int WAVES, bool X3 = false, int OCC = 1>
bool launch2_HL(int act, int npass, int mode, const float* theta, __bf16* wpk, const TrunkArgs& A,
                int grid, hipStream_t st, hipError_t* e)
{
#define PINN_MCASE2(ACT, NP, MD)                                                        \
    if (act == ACT && npass == NP && mode == MD) {                                      \
        *e = launch2_t<H, L, ACT, NP, MD, SPLIT, WAVES, X3, OCC>(theta, wpk, A, grid, st);   \
        return true;                                                                    \
    }
    PINN_MCASE2(ACT_SIGMOID, 2, MODE_FWD) PINN_MCASE2(ACT_SIGMOID, 2, MODE_BWD) PINN_MCASE2(ACT_SIGMOID, 2, MODE_HEAD)
    PINN_MCASE2(ACT_SIGMOID, 1, MODE_FWD) PINN_MCASE2(ACT_SIGMOID, 1, MODE_BWD) PINN_MCASE2(ACT_SIGMOID, 1, MODE_HEAD)
    PINN_MCASE2(ACT_TANH, 2, MODE_FWD) PINN_MCASE2(ACT_TANH, 2, MODE_BWD) PINN_MCASE2(ACT_TANH, 2, MODE_HEAD)
    PINN_MCASE2(ACT_TANH, 1, MODE_FWD) PINN_MCASE2(ACT_TANH, 1, MODE_BWD) PINN_MCASE2(ACT_TANH, 1, MODE_HEAD)
#undef PINN_MCASE2
    return false;
}


// =====================================================================================
// k_trunk2: two hidden layers, symmetric double pass, full reverse sweep -- the benchmark's shape --
// with BOTH passes of the double pass swept together.  Same decomposition as k_trunk_mfma2 (H/16 waves
// share a 16-point tile, each owning 16 neurons; B fragments exchanged through LDS in operand order),
// but every phase handles the 2 x NST x 16 = 128 (pass, stream, point) columns of the tile at once:
//   P1  point record (orbitals, E, g, mask: from an LDS ring filled 4 tiles per wave at a time, the raw
//       points of the next fill requested a whole ring ahead), layer 0 of both passes -> XB[A], XB[B]
//   P2  layer-1 GEMM over 128 columns (each weight fragment feeds 8 MFMAs), activations of both
//       passes, partial output sums -> SB
//   P3  tail (psi, residual, loss seeds), last-hidden adjoints of both passes -> YB[A], YB[B]
//   P4  weight gradient (K = 128 columns = 4 MFMA steps), input-adjoint GEMM, layer-0 adjoints
// Three workgroup barriers per tile instead of eight (the layer-output buffers alternate between two
// sets from tile to tile, so P4 of tile t runs into P1 of tile t+1 without a barrier), twice the
// independent work between them, half the weight-fragment reads.  Results are those of k_trunk_mfma2
// up to the order of fp32 additions.
template <int H>
struct T2Cfg {
    static constexpr int HT = H / 16, KC = H / 32, WAVES = H / 16;
    static constexpr int OCC = H == 64 ? 2 : 1;                  // resident blocks per CU
    static constexpr int EB = KC * NST * 64 * 16;                 // bytes of one (pass) exchange buffer
    static constexpr int WBYTES = 2 * H * H * 2;                  // forward + transposed matrix, bf16
    static constexpr int NSMALL = 3 * H + H + H + 4;              // W0a W0b b0 | b1 | wout | bout
    // point-record ring: the orbitals (and E, g, mask) of the block's next RB tiles, 12 floats per point,
    // computed 64 points per wave at a time -- without it every one of the tile's WAVES waves (and each of
    // its 4 lane groups) computes the same 16 points' orbitals in every trip
    static constexpr int RB = 4 * WAVES, RECF = 12;
    static constexpr int RBYTES = RB * 16 * RECF * 4;
    static constexpr int XSETS = (OCC * (WBYTES + NSMALL * 4 + 6 * EB + RBYTES + 1024) <= 160 * 1024) ? 2 : 1;
    static constexpr size_t OFF_SMALL = WBYTES;
    static constexpr size_t OFF_X = OFF_SMALL + NSMALL * 4;
    static constexpr size_t OFF_Y = OFF_X + (size_t)XSETS * 2 * EB;
    static constexpr size_t OFF_S = OFF_Y + 2 * EB;
    static constexpr size_t OFF_R = OFF_S + WAVES * 16 * 2 * 4;
    static constexpr size_t LDS_BYTES = OFF_R + RBYTES;
    static_assert(OCC * LDS_BYTES <= 160 * 1024, "k_trunk2 does not fit the CU's LDS");
};

// SWZ: exchange slots of the odd lane groups permuted by c ^ 12 (see k_trunkD: conflict-free transposed reads)
template <int H, int ACT, bool SWZ = true>
__global__ __launch_bounds__(T2Cfg<H>::WAVES * 64, T2Cfg<H>::OCC * T2Cfg<H>::WAVES / 4)
void k_trunk2(const float* __restrict__ theta, const __bf16* __restrict__ wpk_g, TrunkArgs A)
{
    using C = T2Cfg<H>;
    using Lay = TrunkLayout<H, 2>;
    constexpr int HT = C::HT, KC = C::KC, NT = C::WAVES * 64, EB = C::EB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef __attribute__((address_space(3))) unsigned char* lds_p;
    const lds_p lds0 = (lds_p)lds_raw;
    float* small = reinterpret_cast<float*>(lds_raw + C::OFF_SMALL);
    float* sW0a = small;              // W0[:, 0]
    float* sW0b = small + H;          // W0[:, 1]
    float* sb0 = small + 2 * H;
    float* sb1 = small + 3 * H;
    float* swo = small + 4 * H;
    float* sbo = small + 5 * H;
    float* SB = reinterpret_cast<float*>(lds_raw + C::OFF_S);   // [wave][16 points][2]

    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // row tile owned by this wave
    const int g = lane >> 4, c = lane & 15;
    const int lane16 = lane * 16;
    auto slot16 = [](int gg, int cc) { return (gg * 16 + (SWZ ? (cc ^ ((gg & 1) * 12)) : cc)) * 16; };
    const int sw16 = slot16(g, c);
    const int own_slot = sw16 + (part >> 1) * NST * 64 * 16 + 8 * (part & 1);
    const int trl = (g >> 1) * 1024 + slot16((c & 3) >> 1, 8 * (g & 1) + (c >> 2)) + (c & 1) * 8;
    const int trh = SWZ ? (g >> 1) * 1024 + slot16((c & 3) >> 1, 8 * (g & 1) + (c >> 2) + 4) + (c & 1) * 8 : trl + 64;
    const int n0 = 16 * part + 4 * g;                                    // first own neuron of this lane

    {   // weights (fragment order) and small parameters -> LDS
        const uint4* src = reinterpret_cast<const uint4*>(wpk_g);
        uint4* dst = reinterpret_cast<uint4*>(lds_raw);
        for (int k = threadIdx.x; k < C::WBYTES / 16; k += NT) dst[k] = src[k];
        for (int k = threadIdx.x; k < H; k += NT) {
            sW0a[k] = theta[Lay::W(0) + 2 * k];
            sW0b[k] = theta[Lay::W(0) + 2 * k + 1];
            sb0[k] = theta[Lay::b(0) + k];
            sb1[k] = theta[Lay::b(1) + k];
            swo[k] = theta[Lay::Wout + k];
        }
        if (threadIdx.x == 0) sbo[0] = theta[Lay::bout];
    }
    __syncthreads();
    // per-lane constants of the own neurons (registers for the whole kernel)
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(sW0a + n0), w1 = *reinterpret_cast<const f32x4*>(sW0b + n0);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sb0 + n0), b1 = *reinterpret_cast<const f32x4*>(sb1 + n0);
    // layer 0 only needs the ACTIVATION of its value stream, never the pre-activation itself: fold the
    // exponent scaling of the activation (exp(-a) / exp(-2a) as exp2) into its weights and bias once
    constexpr float kEx = (ACT == ACT_SIGMOID ? -1.0f : -2.0f) * 1.4426950408889634f;
    const f32x4 w0e = w0 * kEx, w1e = w1 * kEx, b0e = b0 * kEx;
    const f32x4 wo = *reinterpret_cast<const f32x4*>(swo + n0);
    const float bout = sbo[0];
    const lds_p wfwd = lds0 + (size_t)(part * KC) * 1024 + lane16;             // + kc * 1024
    const lds_p wbwd = lds0 + (size_t)H * H * 2 + (size_t)(part * KC) * 1024 + lane16;

    f32x4 wacc[HT], bacc0 = splat(0.f), bacc1 = splat(0.f), w0acc0 = splat(0.f), w0acc1 = splat(0.f),
          woacc = splat(0.f);
#pragma unroll
    for (int it = 0; it < HT; it++) wacc[it] = splat(0.f);
    float acc_bout = 0.f, acc_pde = 0.f, acc_bc1 = 0.f, acc_bc2 = 0.f;
    float invN1 = A.invN1, invN2 = A.invN2;
    if (A.bc_counts) {
        const long long c1 = A.bc_counts[0], c2 = A.bc_counts[1];
        invN1 = c1 > 0 ? 1.0f / (float)c1 : 0.0f;
        invN2 = c2 > 0 ? 1.0f / (float)c2 : 0.0f;
    }
    const float lead = (g == 0 && part == 0) ? 1.0f : 0.0f;
    const int64_t ntile = (A.n + 15) / 16;
    const int64_t iters = (ntile + gridDim.x - 1) / gridDim.x;   // same for every block: barriers match

    // ring batch b covers trips b*RB .. b*RB + RB-1; this lane's point of it: trip b*RB + 4*part + g, column c
    constexpr int RB = C::RB, RECF = C::RECF;
    float* ring = reinterpret_cast<float*>(lds_raw + C::OFF_R);
    struct PointRec { float x, y, z, R, E, g; unsigned mk; };
    auto load_rec = [&](int64_t b) {
        const int64_t tl = (b * RB + 4 * part + g) * gridDim.x + blockIdx.x;
        int64_t id = tl * 16 + c;
        id = (tl < ntile && id < A.n) ? id : A.n - 1;
        PointRec r;
        r.x = A.x[id]; r.y = A.y[id]; r.z = A.z[id]; r.R = A.R[id]; r.E = A.E[id]; r.g = A.g[id];
        r.mk = A.mask ? A.mask[id] : 0u;
        return r;
    };
    auto fill_ring = [&](const PointRec& r) {
        float i0[NST], i1[NST], ri1, ri2;
        const float rho2 = r.y * r.y + r.z * r.z, rho = __builtin_amdgcn_sqrtf(rho2);
        orbital_cyl<float>(r.x - r.R, rho2, rho, i0, ri1);
        orbital_cyl<float>(r.x + r.R, rho2, rho, i1, ri2);
        float* d = ring + ((4 * part + g) * 16 + c) * RECF;
        *reinterpret_cast<f32x4*>(d) = f32x4{i0[0], i0[1], i0[2], i0[3]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{i1[0], i1[1], i1[2], i1[3]};
        *reinterpret_cast<f32x4*>(d + 8) = f32x4{ri1 + ri2, r.E, r.g, (float)r.mk};   // mask bits as 0.0 .. 3.0
    };
    auto put = [&](lds_p eb, int s, f32x4 v) {      // own 4 neurons of stream s -> 8 bytes of the B fragment
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(eb + own_slot + s * 1024) = pack4(v);
    };
    auto bfrag = [&](lds_p eb, int kc, int s) {
        return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(eb + sw16 + (kc * NST + s) * 1024);
    };
    auto trfrag = [&](lds_p eb, int T, int ph) {     // rows 32ph..32ph+31 of the [column][neuron] image, tile T
        const int o = ph * 2048 + (T >> 1) * NST * 1024 + (T & 1) * 512;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(eb + trl + o));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(eb + trh + o));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    // -DPINN_STAMPS (diagnostic build, tools/stamps.py): per-wave cycle totals of the phases and of the
    // barrier waits between them, written to the buffer passed in A.psi (unused by this mode)
#ifdef PINN_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define T2_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; } while (0)
#else
#define T2_STAMP(i) do { } while (0)
#endif
    // D[own 16 rows][2 passes x NST column tiles] += W[own rows][H] x B, B fragments from the pass buffers
    // at eb, eb + EB: groups of 4 MFMAs (one pass of one k chunk), the operands of group j+1 requested
    // before the MFMAs of group j issue
    auto gemm8 = [&](lds_p wbase, lds_p eb, f32x4 (&d)[2][NST]) {
        constexpr int NG = 2 * KC;
        bf16x8 w[2], x[2][NST];
        auto fetch = [&](int j, bf16x8 (&xx)[NST]) {
            const int kc = j >> 1, ps = j & 1;
            if (ps == 0) w[kc & 1] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(wbase + kc * 1024);
#pragma unroll
            for (int s = 0; s < NST; s++) xx[s] = bfrag(eb + ps * EB, kc, s);
        };
        fetch(0, x[0]);
#pragma unroll
        for (int j = 0; j < NG; j++) {
            if (j + 1 < NG) fetch(j + 1, x[(j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NST; s++) d[j & 1][s] = mfma(w[(j >> 1) & 1], x[j & 1][s], d[j & 1][s]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    PointRec nxt = load_rec(0);
    int xset = 0;
    for (int64_t itx = 0; itx < iters; itx++) {
        T2_STAMP(7);
        const int64_t tile = itx * gridDim.x + blockIdx.x;
        const int64_t idx = tile * 16 + c;
        const bool valid = tile < ntile && idx < A.n;
        if (itx % RB == 0) {
            // refill: the last reads of the previous batch (P1, P3 of the previous trip) lie before that
            // trip's third barrier, which every wave has passed
            fill_ring(nxt);
            nxt = load_rec(itx / RB + 1);       // raw points of the next batch: nothing waits for them for RB trips
            __syncthreads();
        }
        const lds_p XB = lds0 + C::OFF_X + (size_t)(C::XSETS == 2 ? xset : 0) * 2 * EB;   // [pass] at + ps * EB
        const lds_p YB = lds0 + C::OFF_Y;
        // ================= P1: inputs and layer 0 of both passes =================
        f32x2 inA[2][NST];      // splat pairs (x, x)
        float Usum, Ept, gpt;
        unsigned mkpt;
        {
            const float* rp = ring + ((int)(itx % RB) * 16 + c) * RECF;
            const f32x4 r2 = *reinterpret_cast<const f32x4*>(rp + 8);
            ring_splats((unsigned)(uintptr_t)(lds0 + C::OFF_R) + (unsigned)((((int)(itx % RB) * 16 + c) * RECF) * 4), inA);
            Usum = r2[0]; Ept = r2[1]; gpt = r2[2]; mkpt = (unsigned)r2[3];
        }
        const float F = inA[0][0][0] + A.P * inA[1][0][0];
        const float lapF = inA[0][SLP][0] + A.P * inA[1][SLP][0];
        f32x4 h0[2];            // layer-0 activations of the own neurons, pass A / pass B
        f32x4 sv[2][NST];       // (h, a_x, a_rho, a_lap) of layer 1
#pragma unroll
        for (int ps = 0; ps < 2; ps++) {
            f32x4 a[NST];
#pragma unroll
            for (int s = 1; s < NST; s++) a[s] = w0 * bc4(inA[ps][s]) + w1 * bc4(inA[1 - ps][s]);
            f32x4 s0, s1, s2;
            act012v_scaled<ACT>(w0e * bc4(inA[ps][0]) + w1e * bc4(inA[1 - ps][0]) + b0e, s0, s1, s2);
            const f32x4 q = a[1] * a[1] + a[2] * a[2];
            h0[ps] = s0;
            put(XB + ps * EB, 0, s0);
            put(XB + ps * EB, 1, s1 * a[1]);
            put(XB + ps * EB, 2, s1 * a[2]);
            put(XB + ps * EB, SLP, s1 * a[SLP] + s2 * q);
        }
        T2_STAMP(0);
        __syncthreads();
        T2_STAMP(1);
        // ================= P2: layer-1 GEMM over both passes, activations, output partial sums =================
        {
            f32x4 acc[2][NST];
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                acc[ps][0] = b1;
#pragma unroll
                for (int s = 1; s < NST; s++) acc[ps][s] = splat(0.f);
            }
            gemm8(wfwd, XB, acc);
            float Sv = 0.f, Sl = 0.f;
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                f32x4 s0, s1, s2;
                act012v<ACT>(acc[ps][0], s0, s1, s2);
                const f32x4 q = acc[ps][1] * acc[ps][1] + acc[ps][2] * acc[ps][2];
                const f32x4 ol = s1 * acc[ps][SLP] + s2 * q;
                sv[ps][0] = s0;
#pragma unroll
                for (int s = 1; s < NST; s++) sv[ps][s] = acc[ps][s];
                const float pw = ps ? A.P : 1.0f;
                Sv += pw * dot4(wo, s0);
                Sl += pw * dot4(wo, ol);
            }
            const float tot = xg_sum_pair(Sv, Sl);          // Sv in lane groups 0,1; Sl in groups 2,3
            if ((g & 1) == 0) SB[(part * 16 + c) * 2 + (g >> 1)] = tot;
        }
        T2_STAMP(2);
        __syncthreads();
        T2_STAMP(3);
        // ================= P3: tail, last hidden layer's adjoints of both passes =================
        Tail<float> t;
        {
            float Sv = 0.f, Sl = 0.f;
#pragma unroll
            for (int pp = 0; pp < C::WAVES; pp++) { Sv += SB[(pp * 16 + c) * 2]; Sl += SB[(pp * 16 + c) * 2 + 1]; }
            tail<float>(A.out_scale * Sv + bout, A.out_scale * Sl, gpt, Ept, F, lapF, Usum, A.a, A.b, A.c,
                        float(mkpt & 1), float((mkpt >> 1) & 1), A.invN, invN1, invN2, valid ? 1.0f : 0.0f, t);
            if (valid && g == 0 && part == 0) { A.Eb[idx] = t.Eb; A.gb[idx] = t.gb; }
            acc_pde += lead * t.l_pde;
            acc_bc1 += lead * t.l_bc1;
            acc_bc2 += lead * t.l_bc2;
            acc_bout += lead * t.Nb;
        }
#pragma unroll
        for (int ps = 0; ps < 2; ps++) {
            const float pw = ps ? A.P : 1.0f;
            const float cv = A.out_scale * pw * t.Nb, cl = A.out_scale * pw * t.lapNb;
            const f32x4 h = sv[ps][0];
            f32x4 s1, s2, s3;
            act123v<ACT>(h, s1, s2, s3);
            const f32x4 q = sv[ps][1] * sv[ps][1] + sv[ps][2] * sv[ps][2];
            woacc += cv * h + cl * (s1 * sv[ps][SLP] + s2 * q);
            const f32x4 o0 = cv * wo, ol = cl * wo;
            const f32x4 u = 2.0f * s2 * ol;
            const f32x4 y0 = s1 * o0 + (s2 * sv[ps][SLP] + s3 * q) * ol;
            bacc1 += y0;
            put(YB + ps * EB, 0, y0);
            put(YB + ps * EB, 1, u * sv[ps][1]);
            put(YB + ps * EB, 2, u * sv[ps][2]);
            put(YB + ps * EB, SLP, s1 * ol);
        }
        T2_STAMP(4);
        __syncthreads();
        T2_STAMP(5);
        // ================= P4: weight gradient, input-adjoint GEMM, layer-0 adjoints =================
        {   // K = 128 columns = 4 steps (pass, stream pair) x HT column tiles, in groups of 4 MFMAs; the
            // fragments of group j+1 are requested before the MFMAs of group j issue (the fences pin that
            // order: otherwise every MFMA sits behind the LDS round trip of its own operands)
            constexpr int GPK = HT / 4, NG = NST * GPK;          // groups per K step, groups in all
            bf16x8 fy[2], fx[2][4];
            auto fetch = [&](int j, bf16x8 (&x)[4]) {
                const int k = j / GPK, i0 = (j % GPK) * 4;
                if (j % GPK == 0) fy[k & 1] = trfrag(YB + (k >> 1) * EB, part, k & 1);
#pragma unroll
                for (int i = 0; i < 4; i++) x[i] = trfrag(XB + (k >> 1) * EB, i0 + i, k & 1);
            };
            fetch(0, fx[0]);
#pragma unroll
            for (int j = 0; j < NG; j++) {
                if (j + 1 < NG) fetch(j + 1, fx[(j + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const int k = j / GPK, i0 = (j % GPK) * 4;
#pragma unroll
                for (int i = 0; i < 4; i++) wacc[i0 + i] = mfma(fy[k & 1], fx[j & 1][i], wacc[i0 + i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            f32x4 ob[2][NST];
#pragma unroll
            for (int ps = 0; ps < 2; ps++)
#pragma unroll
                for (int s = 0; s < NST; s++) ob[ps][s] = splat(0.f);
            gemm8(wbwd, YB, ob);
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                f32x4 a[NST];
#pragma unroll
                for (int s = 1; s < NST; s++) a[s] = w0 * bc4(inA[ps][s]) + w1 * bc4(inA[1 - ps][s]);
                f32x4 s1, s2, s3;
                act123v<ACT>(h0[ps], s1, s2, s3);
                const f32x4 q = a[1] * a[1] + a[2] * a[2];
                const f32x4 u = 2.0f * s2 * ob[ps][SLP];
                f32x4 ab[NST];
                ab[SLP] = s1 * ob[ps][SLP];
                ab[1] = s1 * ob[ps][1] + u * a[1];
                ab[2] = s1 * ob[ps][2] + u * a[2];
                ab[0] = s1 * ob[ps][0] + s2 * (a[1] * ob[ps][1] + a[2] * ob[ps][2]) + (s2 * a[SLP] + s3 * q) * ob[ps][SLP];
                bacc0 += ab[0];
#pragma unroll
                for (int s = 0; s < NST; s++) { w0acc0 += ab[s] * bc4(inA[ps][s]); w0acc1 += ab[s] * bc4(inA[1 - ps][s]); }
            }
        }
        if (C::XSETS == 1) __syncthreads();   // one set of layer-output buffers: P1 of the next tile overwrites it
        xset ^= 1;
        T2_STAMP(6);
    }
#ifdef PINN_STAMPS
    if (A.psi && lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(A.psi) + ((size_t)blockIdx.x * C::WAVES + part) * 8;
        for (int i = 0; i < 8; i++) dbg[i] = st_acc[i];
    }
#endif
#undef T2_STAMP
    // =========================== block reduction ===========================
    __syncthreads();
    constexpr int PT = Lay::size;
    float* out = A.partial + (size_t)blockIdx.x * A.stride;
    const float s0 = row16_sum(xg_sum(acc_pde)), s1r = row16_sum(xg_sum(acc_bc1)),
                s2r = row16_sum(xg_sum(acc_bc2)), s3r = row16_sum(xg_sum(acc_bout));
#pragma unroll
    for (int r = 0; r < 4; r++) {
        bacc0[r] = row16_sum(bacc0[r]); bacc1[r] = row16_sum(bacc1[r]);
        w0acc0[r] = row16_sum(w0acc0[r]); w0acc1[r] = row16_sum(w0acc1[r]);
        woacc[r] = row16_sum(woacc[r]);
    }
    float* sc = reinterpret_cast<float*>(lds_raw);
    if (threadIdx.x < 4) sc[threadIdx.x] = 0.f;
    __syncthreads();
    // every wave owns distinct gradient rows: straight to the block's partial row
#pragma unroll
    for (int it = 0; it < HT; it++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int nj = q_neuron(16 * part + 4 * g + r), ni = q_neuron(16 * it + c);
            out[Lay::W(1) + nj * H + ni] = wacc[it][r];
        }
    if (c == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int nj = n0 + r;
            out[Lay::b(0) + nj] = bacc0[r];
            out[Lay::b(1) + nj] = bacc1[r];
            out[Lay::W(0) + 2 * nj] = w0acc0[r];
            out[Lay::W(0) + 2 * nj + 1] = w0acc1[r];
            out[Lay::Wout + nj] = woacc[r];
        }
    }
    for (int w = 0; w < C::WAVES; w++) {   // per-point sums live in one wave only; fixed order anyway
        if (part == w && lane == 0) { sc[0] += s0; sc[1] += s1r; sc[2] += s2r; sc[3] += s3r; }
        __syncthreads();
    }
    if (threadIdx.x < 3) out[PT + threadIdx.x] = sc[threadIdx.x];
    if (threadIdx.x == 3) out[Lay::bout] = sc[3];
}

template <int H, int ACT, bool SWZ = true>
hipError_t launch_trunk2_t(const float* theta, __bf16* wpk, const TrunkArgs& A, int grid, hipStream_t st)
{
    using C1 = MCfg<H, 2>;
    using C = T2Cfg<H>;
    const int nfrag = C1::NW * 2 * C1::HT * C1::KC * 64;
    hipLaunchKernelGGL((k_pack_weights<H, 2, false>), dim3((nfrag + 255) / 256), dim3(256), 0, st, theta, wpk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    auto kern = k_trunk2<H, ACT, SWZ>;
    static bool attr_set[64] = {};   // per device (not a stream operation: done outside graph capture)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::WAVES * 64), C::LDS_BYTES, st, theta, (const __bf16*)wpk, A);
    return hipGetLastError();
}

template <int H>
struct T2sCfg {
    static constexpr int HT = H / 16, KC = H / 32, WAVES = H / 16;
    static constexpr int OCC = H == 64 ? 2 : 1;                  // resident blocks per CU
    static constexpr int EB = KC * NST * 64 * 16;                 // bytes of one (pass) exchange buffer
    static constexpr int WBYTES = 2 * H * H * 2;                  // forward + transposed matrix, bf16
    static constexpr int NSMALL = 3 * H + H + H + 4;              // W0a W0b b0 | b1 | wout | bout
    // point-record ring: the orbitals (and E, g, mask) of the block's next RB tiles, 12 floats per point,
    // computed 64 points per wave at a time -- without it every one of the tile's WAVES waves (and each of
    // its 4 lane groups) computes the same 16 points' orbitals in every trip
    static constexpr int RB = 4 * WAVES, RECF = 12;
    static constexpr int RBYTES = RB * 16 * RECF * 4;
    static constexpr int XSETS = (OCC * (WBYTES + NSMALL * 4 + 6 * EB + RBYTES + 2048) <= 160 * 1024) ? 2 : 1;
    static constexpr size_t OFF_SMALL = WBYTES;
    static constexpr size_t OFF_X = OFF_SMALL + NSMALL * 4;
    static constexpr size_t OFF_Y = OFF_X + (size_t)XSETS * 2 * EB;
    static constexpr size_t OFF_S = OFF_Y + 2 * EB;
    static constexpr size_t OFF_R = OFF_S + WAVES * 16 * 2 * 2 * 4;       // S: [wave][16 points][2 tiles][Sv, Sl]
    static constexpr size_t LDS_BYTES = OFF_R + RBYTES;
    static_assert(OCC * LDS_BYTES <= 160 * 1024, "k_trunk2 does not fit the CU's LDS");
};

// k_trunk2s<H, ACT>: k_trunk2 for train.py's formulation (ONE pass per point over (f1, f2), output 2h): the two
// 64-column halves of every phase are two TILES instead of the two passes of one tile -- same GEMMs, barriers and
// buffers; each tile has its own record, output sums and tail, and the layer-0 inputs are not swapped.  A separate
// kernel, not a template flag of k_trunk2: with the flag, hipcc scheduled the symmetric instantiation 3 % slower.
template <int H, int ACT>
__global__ __launch_bounds__(T2sCfg<H>::WAVES * 64, T2sCfg<H>::OCC * T2sCfg<H>::WAVES / 4)
void k_trunk2s(const float* __restrict__ theta, const __bf16* __restrict__ wpk_g, TrunkArgs A)
{
    constexpr bool TT = true;                   // (the `!TT` branches below are k_trunk2's own code, kept for reference)
    constexpr int NTL = TT ? 2 : 1;             // tiles per trip
    using C = T2sCfg<H>;
    using Lay = TrunkLayout<H, 2>;
    constexpr int HT = C::HT, KC = C::KC, NT = C::WAVES * 64, EB = C::EB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef __attribute__((address_space(3))) unsigned char* lds_p;
    const lds_p lds0 = (lds_p)lds_raw;
    float* small = reinterpret_cast<float*>(lds_raw + C::OFF_SMALL);
    float* sW0a = small;              // W0[:, 0]
    float* sW0b = small + H;          // W0[:, 1]
    float* sb0 = small + 2 * H;
    float* sb1 = small + 3 * H;
    float* swo = small + 4 * H;
    float* sbo = small + 5 * H;
    float* SB = reinterpret_cast<float*>(lds_raw + C::OFF_S);   // [wave][16 points][tile][2]

    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // row tile owned by this wave
    const int g = lane >> 4, c = lane & 15;
    const int lane16 = lane * 16;
    // exchange slots: the 16-byte slots of the odd lane groups permuted by c ^ 12 (k_trunkD: conflict-free transposed reads)
    auto slot16 = [](int gg, int cc) { return (gg * 16 + (cc ^ ((gg & 1) * 12))) * 16; };
    const int sw16 = slot16(g, c);
    const int own_slot = sw16 + (part >> 1) * NST * 64 * 16 + 8 * (part & 1);
    const int trl = (g >> 1) * 1024 + slot16((c & 3) >> 1, 8 * (g & 1) + (c >> 2)) + (c & 1) * 8;
    const int trh = (g >> 1) * 1024 + slot16((c & 3) >> 1, 8 * (g & 1) + (c >> 2) + 4) + (c & 1) * 8;
    const int n0 = 16 * part + 4 * g;                                    // first own neuron of this lane

    {   // weights (fragment order) and small parameters -> LDS
        const uint4* src = reinterpret_cast<const uint4*>(wpk_g);
        uint4* dst = reinterpret_cast<uint4*>(lds_raw);
        for (int k = threadIdx.x; k < C::WBYTES / 16; k += NT) dst[k] = src[k];
        for (int k = threadIdx.x; k < H; k += NT) {
            sW0a[k] = theta[Lay::W(0) + 2 * k];
            sW0b[k] = theta[Lay::W(0) + 2 * k + 1];
            sb0[k] = theta[Lay::b(0) + k];
            sb1[k] = theta[Lay::b(1) + k];
            swo[k] = theta[Lay::Wout + k];
        }
        if (threadIdx.x == 0) sbo[0] = theta[Lay::bout];
    }
    __syncthreads();
    // per-lane constants of the own neurons (registers for the whole kernel)
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(sW0a + n0), w1 = *reinterpret_cast<const f32x4*>(sW0b + n0);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sb0 + n0), b1 = *reinterpret_cast<const f32x4*>(sb1 + n0);
    // layer 0 only needs the ACTIVATION of its value stream, never the pre-activation itself: fold the
    // exponent scaling of the activation (exp(-a) / exp(-2a) as exp2) into its weights and bias once
    constexpr float kEx = (ACT == ACT_SIGMOID ? -1.0f : -2.0f) * 1.4426950408889634f;
    const f32x4 w0e = w0 * kEx, w1e = w1 * kEx, b0e = b0 * kEx;
    const f32x4 wo = *reinterpret_cast<const f32x4*>(swo + n0);
    const float bout = sbo[0];
    const lds_p wfwd = lds0 + (size_t)(part * KC) * 1024 + lane16;             // + kc * 1024
    const lds_p wbwd = lds0 + (size_t)H * H * 2 + (size_t)(part * KC) * 1024 + lane16;

    f32x4 wacc[HT], bacc0 = splat(0.f), bacc1 = splat(0.f), w0acc0 = splat(0.f), w0acc1 = splat(0.f),
          woacc = splat(0.f);
#pragma unroll
    for (int it = 0; it < HT; it++) wacc[it] = splat(0.f);
    float acc_bout = 0.f, acc_pde = 0.f, acc_bc1 = 0.f, acc_bc2 = 0.f;
    float invN1 = A.invN1, invN2 = A.invN2;
    if (A.bc_counts) {
        const long long c1 = A.bc_counts[0], c2 = A.bc_counts[1];
        invN1 = c1 > 0 ? 1.0f / (float)c1 : 0.0f;
        invN2 = c2 > 0 ? 1.0f / (float)c2 : 0.0f;
    }
    const float lead = (g == 0 && part == 0) ? 1.0f : 0.0f;
    const int64_t ntile = (A.n + 15) / 16;
    const int64_t iters = (ntile + NTL * gridDim.x - 1) / (NTL * gridDim.x);   // same for every block: barriers match

    // ring batch b covers the block's tiles b*RB .. b*RB + RB-1 (NTL per trip); this lane's point of it: tile
    // b*RB + 4*part + g, column c
    constexpr int RB = C::RB, RECF = C::RECF;
    float* ring = reinterpret_cast<float*>(lds_raw + C::OFF_R);
    struct PointRec { float x, y, z, R, E, g; unsigned mk; };
    auto load_rec = [&](int64_t b) {
        const int64_t tl = (b * RB + 4 * part + g) * gridDim.x + blockIdx.x;
        int64_t id = tl * 16 + c;
        id = (tl < ntile && id < A.n) ? id : A.n - 1;
        PointRec r;
        r.x = A.x[id]; r.y = A.y[id]; r.z = A.z[id]; r.R = A.R[id]; r.E = A.E[id]; r.g = A.g[id];
        r.mk = A.mask ? A.mask[id] : 0u;
        return r;
    };
    auto fill_ring = [&](const PointRec& r) {
        float i0[NST], i1[NST], ri1, ri2;
        const float rho2 = r.y * r.y + r.z * r.z, rho = __builtin_amdgcn_sqrtf(rho2);
        orbital_cyl<float>(r.x - r.R, rho2, rho, i0, ri1);
        orbital_cyl<float>(r.x + r.R, rho2, rho, i1, ri2);
        float* d = ring + ((4 * part + g) * 16 + c) * RECF;
        *reinterpret_cast<f32x4*>(d) = f32x4{i0[0], i0[1], i0[2], i0[3]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{i1[0], i1[1], i1[2], i1[3]};
        *reinterpret_cast<f32x4*>(d + 8) = f32x4{ri1 + ri2, r.E, r.g, (float)r.mk};   // mask bits as 0.0 .. 3.0
    };
    auto put = [&](lds_p eb, int s, f32x4 v) {      // own 4 neurons of stream s -> 8 bytes of the B fragment
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(eb + own_slot + s * 1024) = pack4(v);
    };
    auto bfrag = [&](lds_p eb, int kc, int s) {
        return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(eb + sw16 + (kc * NST + s) * 1024);
    };
    auto trfrag = [&](lds_p eb, int T, int ph) {     // rows 32ph..32ph+31 of the [column][neuron] image, tile T
        const int o = ph * 2048 + (T >> 1) * NST * 1024 + (T & 1) * 512;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(eb + trl + o));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(eb + trh + o));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    // -DPINN_STAMPS (diagnostic build, tools/stamps.py): per-wave cycle totals of the phases and of the
    // barrier waits between them, written to the buffer passed in A.psi (unused by this mode)
#ifdef PINN_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define T2_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; } while (0)
#else
#define T2_STAMP(i) do { } while (0)
#endif
    // D[own 16 rows][2 passes x NST column tiles] += W[own rows][H] x B, B fragments from the pass buffers
    // at eb, eb + EB: groups of 4 MFMAs (one pass of one k chunk), the operands of group j+1 requested
    // before the MFMAs of group j issue
    auto gemm8 = [&](lds_p wbase, lds_p eb, f32x4 (&d)[2][NST]) {
        constexpr int NG = 2 * KC;
        bf16x8 w[2], x[2][NST];
        auto fetch = [&](int j, bf16x8 (&xx)[NST]) {
            const int kc = j >> 1, ps = j & 1;
            if (ps == 0) w[kc & 1] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(wbase + kc * 1024);
#pragma unroll
            for (int s = 0; s < NST; s++) xx[s] = bfrag(eb + ps * EB, kc, s);
        };
        fetch(0, x[0]);
#pragma unroll
        for (int j = 0; j < NG; j++) {
            if (j + 1 < NG) fetch(j + 1, x[(j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NST; s++) d[j & 1][s] = mfma(w[(j >> 1) & 1], x[j & 1][s], d[j & 1][s]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    PointRec nxt = load_rec(0);
    int xset = 0;
    for (int64_t itx = 0; itx < iters; itx++) {
        T2_STAMP(7);
        const int64_t tile = (NTL * itx) * gridDim.x + blockIdx.x;      // (TT: the first of the trip's two tiles)
        const int64_t idx = tile * 16 + c;
        const bool valid = tile < ntile && idx < A.n;
        const int slot0 = (int)((NTL * itx) % RB);                       // ring slot of that tile
        if (slot0 == 0) {
            // refill: the last reads of the previous batch (P1, P3 of the previous trip) lie before that
            // trip's third barrier, which every wave has passed -- except in TT mode, whose P4 reads the records again
            // (keeping both tiles' records in registers instead was measured: spills, no faster)
            if (TT && itx) __syncthreads();
            fill_ring(nxt);
            nxt = load_rec(NTL * itx / RB + 1); // raw points of the next batch: nothing waits for them for RB trips
            __syncthreads();
        }
        const lds_p XB = lds0 + C::OFF_X + (size_t)(C::XSETS == 2 ? xset : 0) * 2 * EB;   // [pass] at + ps * EB
        const lds_p YB = lds0 + C::OFF_Y;
        // ================= P1: inputs and layer 0 of both passes =================
        const unsigned ring_a0 = (unsigned)(uintptr_t)(lds0 + C::OFF_R) + (unsigned)(((slot0 * 16 + c) * RECF) * 4);
        float inA[2][NST];
        float Usum = 0.f, Ept = 0.f, gpt = 0.f, F = 0.f, lapF = 0.f;
        unsigned mkpt = 0;
        if constexpr (!TT) {
            const float* rp = ring + (slot0 * 16 + c) * RECF;
            const f32x4 r2 = *reinterpret_cast<const f32x4*>(rp + 8);
            ring_pairs(ring_a0, inA);
            Usum = r2[0]; Ept = r2[1]; gpt = r2[2]; mkpt = (unsigned)r2[3];
            F = inA[0][0] + A.P * inA[1][0];
            lapF = inA[0][SLP] + A.P * inA[1][SLP];
        }
        // the two inputs of half `ps` of the tile columns: (f1, f2) streams in the order the half's layer 0 takes them
        auto inputs = [&](int ps, float (&f)[2][NST]) {
            if constexpr (TT) {
                ring_pairs(ring_a0 + ps * 16 * RECF * 4, f);              // the record of tile ps of this trip
            } else {
#pragma unroll
                for (int s = 0; s < NST; s++) { f[0][s] = inA[ps][s]; f[1][s] = inA[1 - ps][s]; }
            }
        };
        f32x4 h0[2];            // layer-0 activations of the own neurons, pass A / pass B
        f32x4 a0k[2][NST];      // TT: layer-0 pre-activation streams 1..3 of both tiles, kept for P4
        f32x4 sv[2][NST];       // (h, a_x, a_rho, a_lap) of layer 1
#pragma unroll
        for (int ps = 0; ps < 2; ps++) {
            float f[2][NST];
            inputs(ps, f);
            f32x4 a[NST];
#pragma unroll
            for (int s = 1; s < NST; s++) a[s] = w0 * f[0][s] + w1 * f[1][s];
            f32x4 s0, s1, s2;
            act012v_scaled<ACT>(w0e * f[0][0] + w1e * f[1][0] + b0e, s0, s1, s2);
            const f32x4 q = a[1] * a[1] + a[2] * a[2];
            h0[ps] = s0;
            if constexpr (TT) {
#pragma unroll
                for (int s = 1; s < NST; s++) a0k[ps][s] = a[s];
            }
            put(XB + ps * EB, 0, s0);
            put(XB + ps * EB, 1, s1 * a[1]);
            put(XB + ps * EB, 2, s1 * a[2]);
            put(XB + ps * EB, SLP, s1 * a[SLP] + s2 * q);
        }
        T2_STAMP(0);
        __syncthreads();
        T2_STAMP(1);
        // ================= P2: layer-1 GEMM over both passes, activations, output partial sums =================
        {
            f32x4 acc[2][NST];
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                acc[ps][0] = b1;
#pragma unroll
                for (int s = 1; s < NST; s++) acc[ps][s] = splat(0.f);
            }
            gemm8(wfwd, XB, acc);
            float Sv = 0.f, Sl = 0.f;
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                f32x4 s0, s1, s2;
                act012v<ACT>(acc[ps][0], s0, s1, s2);
                const f32x4 q = acc[ps][1] * acc[ps][1] + acc[ps][2] * acc[ps][2];
                const f32x4 ol = s1 * acc[ps][SLP] + s2 * q;
                sv[ps][0] = s0;
#pragma unroll
                for (int s = 1; s < NST; s++) sv[ps][s] = acc[ps][s];
                if constexpr (TT) {                          // every tile has its own output
                    const float tot = xg_sum_pair(dot4(wo, s0), dot4(wo, ol));
                    if ((g & 1) == 0) SB[((part * 16 + c) * 2 + ps) * 2 + (g >> 1)] = tot;
                } else {
                    const float pw = ps ? A.P : 1.0f;
                    Sv += pw * dot4(wo, s0);
                    Sl += pw * dot4(wo, ol);
                }
            }
            if constexpr (!TT) {
                const float tot = xg_sum_pair(Sv, Sl);      // Sv in lane groups 0,1; Sl in groups 2,3
                if ((g & 1) == 0) SB[((part * 16 + c) * 2) * 2 + (g >> 1)] = tot;
            }
        }
        T2_STAMP(2);
        __syncthreads();
        T2_STAMP(3);
        // ================= P3: tail, last hidden layer's adjoints of both passes =================
        Tail<float> t;
        auto do_tail = [&](int tl) {                 // tail of tile `tl` of the trip (0 unless TT)
            float Sv = 0.f, Sl = 0.f;
#pragma unroll
            for (int pp = 0; pp < C::WAVES; pp++) { Sv += SB[((pp * 16 + c) * 2 + tl) * 2]; Sl += SB[((pp * 16 + c) * 2 + tl) * 2 + 1]; }
            if constexpr (TT) {
                const float* rp = ring + ((slot0 + tl) * 16 + c) * RECF;
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4),
                            r2 = *reinterpret_cast<const f32x4*>(rp + 8);
                const int64_t tile_t = tile + (int64_t)tl * gridDim.x, idx_t = tile_t * 16 + c;
                const bool valid_t = tile_t < ntile && idx_t < A.n;
                const unsigned mk = (unsigned)r2[3];
                tail<float>(A.out_scale * Sv + bout, A.out_scale * Sl, r2[2], r2[1], r0[0] + A.P * r1[0], r0[SLP] + A.P * r1[SLP],
                            r2[0], A.a, A.b, A.c, float(mk & 1), float((mk >> 1) & 1), A.invN, invN1, invN2,
                            valid_t ? 1.0f : 0.0f, t);
                if (valid_t && g == 0 && part == 0) { A.Eb[idx_t] = t.Eb; A.gb[idx_t] = t.gb; }
            } else {
                tail<float>(A.out_scale * Sv + bout, A.out_scale * Sl, gpt, Ept, F, lapF, Usum, A.a, A.b, A.c,
                            float(mkpt & 1), float((mkpt >> 1) & 1), A.invN, invN1, invN2, valid ? 1.0f : 0.0f, t);
                if (valid && g == 0 && part == 0) { A.Eb[idx] = t.Eb; A.gb[idx] = t.gb; }
            }
            acc_pde += lead * t.l_pde;
            acc_bc1 += lead * t.l_bc1;
            acc_bc2 += lead * t.l_bc2;
            acc_bout += lead * t.Nb;
        };
        if constexpr (!TT) do_tail(0);
#pragma unroll
        for (int ps = 0; ps < 2; ps++) {
            if constexpr (TT) do_tail(ps);
            const float pw = (!TT && ps) ? A.P : 1.0f;
            const float cv = A.out_scale * pw * t.Nb, cl = A.out_scale * pw * t.lapNb;
            const f32x4 h = sv[ps][0];
            f32x4 s1, s2, s3;
            act123v<ACT>(h, s1, s2, s3);
            const f32x4 q = sv[ps][1] * sv[ps][1] + sv[ps][2] * sv[ps][2];
            // (TT: the per-tile scalars cv, cl ended up in odd registers and hipcc broadcast them with the unsafe
            // op_sel:[0,1] form -- the CPU test of the assembly caught it; unpacked multiplies there)
            auto smul = [&](f32x4 v, float x) { if constexpr (TT) return mul_bcast(v, x); else return x * v; };
            woacc += smul(h, cv) + smul(s1 * sv[ps][SLP] + s2 * q, cl);
            const f32x4 o0 = smul(wo, cv), ol = smul(wo, cl);
            const f32x4 u = 2.0f * s2 * ol;
            const f32x4 y0 = s1 * o0 + (s2 * sv[ps][SLP] + s3 * q) * ol;
            bacc1 += y0;
            put(YB + ps * EB, 0, y0);
            put(YB + ps * EB, 1, u * sv[ps][1]);
            put(YB + ps * EB, 2, u * sv[ps][2]);
            put(YB + ps * EB, SLP, s1 * ol);
        }
        T2_STAMP(4);
        __syncthreads();
        T2_STAMP(5);
        // ================= P4: weight gradient, input-adjoint GEMM, layer-0 adjoints =================
        {   // K = 128 columns = 4 steps (pass, stream pair) x HT column tiles, in groups of 4 MFMAs; the
            // fragments of group j+1 are requested before the MFMAs of group j issue (the fences pin that
            // order: otherwise every MFMA sits behind the LDS round trip of its own operands)
            constexpr int GPK = HT / 4, NG = NST * GPK;          // groups per K step, groups in all
            bf16x8 fy[2], fx[2][4];
            auto fetch = [&](int j, bf16x8 (&x)[4]) {
                const int k = j / GPK, i0 = (j % GPK) * 4;
                if (j % GPK == 0) fy[k & 1] = trfrag(YB + (k >> 1) * EB, part, k & 1);
#pragma unroll
                for (int i = 0; i < 4; i++) x[i] = trfrag(XB + (k >> 1) * EB, i0 + i, k & 1);
            };
            fetch(0, fx[0]);
#pragma unroll
            for (int j = 0; j < NG; j++) {
                if (j + 1 < NG) fetch(j + 1, fx[(j + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const int k = j / GPK, i0 = (j % GPK) * 4;
#pragma unroll
                for (int i = 0; i < 4; i++) wacc[i0 + i] = mfma(fy[k & 1], fx[j & 1][i], wacc[i0 + i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            f32x4 ob[2][NST];
#pragma unroll
            for (int ps = 0; ps < 2; ps++)
#pragma unroll
                for (int s = 0; s < NST; s++) ob[ps][s] = splat(0.f);
            gemm8(wbwd, YB, ob);
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                float f[2][NST];
                f32x4 a[NST];
                if constexpr (TT) {
                    // the tile's record again as plain 16-byte reads (its values are only multiplied by unpacked
                    // instructions below); the pre-activation streams were kept from P1
                    const float* rp = ring + ((slot0 + ps) * 16 + c) * RECF;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                    for (int s = 0; s < NST; s++) { f[0][s] = r0[s]; f[1][s] = r1[s]; }
#pragma unroll
                    for (int s = 1; s < NST; s++) a[s] = a0k[ps][s];
                } else {
                    inputs(ps, f);
#pragma unroll
                    for (int s = 1; s < NST; s++) a[s] = w0 * f[0][s] + w1 * f[1][s];
                }
                f32x4 s1, s2, s3;
                act123v<ACT>(h0[ps], s1, s2, s3);
                const f32x4 q = a[1] * a[1] + a[2] * a[2];
                const f32x4 u = 2.0f * s2 * ob[ps][SLP];
                f32x4 ab[NST];
                ab[SLP] = s1 * ob[ps][SLP];
                ab[1] = s1 * ob[ps][1] + u * a[1];
                ab[2] = s1 * ob[ps][2] + u * a[2];
                ab[0] = s1 * ob[ps][0] + s2 * (a[1] * ob[ps][1] + a[2] * ob[ps][2]) + (s2 * a[SLP] + s3 * q) * ob[ps][SLP];
                bacc0 += ab[0];
#pragma unroll
                for (int s = 0; s < NST; s++) {
                    if constexpr (TT) { fmac_bcast(w0acc0, ab[s], f[0][s]); fmac_bcast(w0acc1, ab[s], f[1][s]); }
                    else { w0acc0 += ab[s] * f[0][s]; w0acc1 += ab[s] * f[1][s]; }
                }
            }
        }
        if (C::XSETS == 1) __syncthreads();   // one set of layer-output buffers: P1 of the next tile overwrites it
        xset ^= 1;
        T2_STAMP(6);
    }
#ifdef PINN_STAMPS
    if (A.psi && lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(A.psi) + ((size_t)blockIdx.x * C::WAVES + part) * 8;
        for (int i = 0; i < 8; i++) dbg[i] = st_acc[i];
    }
#endif
#undef T2_STAMP
    // =========================== block reduction ===========================
    __syncthreads();
    constexpr int PT = Lay::size;
    float* out = A.partial + (size_t)blockIdx.x * A.stride;
    const float s0 = row16_sum(xg_sum(acc_pde)), s1r = row16_sum(xg_sum(acc_bc1)),
                s2r = row16_sum(xg_sum(acc_bc2)), s3r = row16_sum(xg_sum(acc_bout));
#pragma unroll
    for (int r = 0; r < 4; r++) {
        bacc0[r] = row16_sum(bacc0[r]); bacc1[r] = row16_sum(bacc1[r]);
        w0acc0[r] = row16_sum(w0acc0[r]); w0acc1[r] = row16_sum(w0acc1[r]);
        woacc[r] = row16_sum(woacc[r]);
    }
    float* sc = reinterpret_cast<float*>(lds_raw);
    if (threadIdx.x < 4) sc[threadIdx.x] = 0.f;
    __syncthreads();
    // every wave owns distinct gradient rows: straight to the block's partial row
#pragma unroll
    for (int it = 0; it < HT; it++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int nj = q_neuron(16 * part + 4 * g + r), ni = q_neuron(16 * it + c);
            out[Lay::W(1) + nj * H + ni] = wacc[it][r];
        }
    if (c == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int nj = n0 + r;
            out[Lay::b(0) + nj] = bacc0[r];
            out[Lay::b(1) + nj] = bacc1[r];
            out[Lay::W(0) + 2 * nj] = w0acc0[r];
            out[Lay::W(0) + 2 * nj + 1] = w0acc1[r];
            out[Lay::Wout + nj] = woacc[r];
        }
    }
    for (int w = 0; w < C::WAVES; w++) {   // per-point sums live in one wave only; fixed order anyway
        if (part == w && lane == 0) { sc[0] += s0; sc[1] += s1r; sc[2] += s2r; sc[3] += s3r; }
        __syncthreads();
    }
    if (threadIdx.x < 3) out[PT + threadIdx.x] = sc[threadIdx.x];
    if (threadIdx.x == 3) out[Lay::bout] = sc[3];
}

template <int H, int ACT>
hipError_t launch_trunk2s_t(const float* theta, __bf16* wpk, const TrunkArgs& A, int grid, hipStream_t st)
{
    using C1 = MCfg<H, 2>;
    using C = T2sCfg<H>;
    const int nfrag = C1::NW * 2 * C1::HT * C1::KC * 64;
    hipLaunchKernelGGL((k_pack_weights<H, 2, false>), dim3((nfrag + 255) / 256), dim3(256), 0, st, theta, wpk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    auto kern = k_trunk2s<H, ACT>;
    static bool attr_set[64] = {};   // per device (not a stream operation: done outside graph capture)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::WAVES * 64), C::LDS_BYTES, st, theta, (const __bf16*)wpk, A);
    return hipGetLastError();
}

// =====================================================================================
// k_trunk2x<ACT>: k_trunk2's schedule (both passes swept together, three barriers per tile) for the 2x64
// trunk in split-bf16, the fp32-grade mode: every MFMA operand is hi + lo (two bf16 planes in the exchange
// buffers, the lo weight fragments in 16 registers per wave), every product three MFMAs into one fp32
// accumulator (hi*hi + hi*lo + lo*hi), saved pre-activations fp32.  Two tile groups of four waves per
// block (one block per CU, two waves per SIMD); the groups share the weights and the barriers, each has its
// own exchange buffers; their gradient accumulators are added through LDS at the end (fixed order).
struct T2xCfg {
    static constexpr int H = 64, HT = 4, KC = 2, WPG = 4, GROUPS = 2, WAVES = 8;
    static constexpr int EB1 = KC * NST * 64 * 16;                // one plane of one pass buffer
    static constexpr int EB = 2 * EB1;                            // hi plane, lo plane
    static constexpr int WBYTES = 2 * H * H * 2;                  // hi plane: forward + transposed matrix
    static constexpr int NSMALL = 5 * H + 4;                      // W0a W0b b0 | b1 | wout | bout
    static constexpr size_t OFF_SMALL = WBYTES;
    static constexpr size_t OFF_G = OFF_SMALL + NSMALL * 4;
    static constexpr int GBYTES = 2 * EB + 2 * EB + WPG * 16 * 2 * 4;   // XB[2 passes], YB[2 passes], SB
    static constexpr size_t LDS_BYTES = OFF_G + (size_t)GROUPS * GBYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "k_trunk2x does not fit the CU's LDS");
};

template <int ACT>
__global__ __launch_bounds__(T2xCfg::WAVES * 64, 2)
void k_trunk2x(const float* __restrict__ theta, const __bf16* __restrict__ wpk_g, TrunkArgs A)
{
    using C = T2xCfg;
    constexpr int H = C::H, HT = C::HT, KC = C::KC, NT = C::WAVES * 64, EB = C::EB, EB1 = C::EB1;
    using Lay = TrunkLayout<H, 2>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef __attribute__((address_space(3))) unsigned char* lds_p;
    const lds_p lds0 = (lds_p)lds_raw;
    float* small = reinterpret_cast<float*>(lds_raw + C::OFF_SMALL);
    float* sW0a = small;
    float* sW0b = small + H;
    float* sb0 = small + 2 * H;
    float* sb1 = small + 3 * H;
    float* swo = small + 4 * H;
    float* sbo = small + 5 * H;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave / C::WPG, part = wave % C::WPG;
    const int g = lane >> 4, c = lane & 15;
    const int lane16 = lane * 16;
    // exchange slots: the 16-byte slots of the odd lane groups permuted by c ^ 12 (k_trunkD: conflict-free transposed reads)
    auto slot16 = [](int gg, int cc) { return (gg * 16 + (cc ^ ((gg & 1) * 12))) * 16; };
    const int sw16 = slot16(g, c);
    const int own_slot = sw16 + (part >> 1) * NST * 64 * 16 + 8 * (part & 1);
    const int trl = (g >> 1) * 1024 + slot16((c & 3) >> 1, 8 * (g & 1) + (c >> 2)) + (c & 1) * 8;
    const int trh = (g >> 1) * 1024 + slot16((c & 3) >> 1, 8 * (g & 1) + (c >> 2) + 4) + (c & 1) * 8;
    const int n0 = 16 * part + 4 * g;
    const lds_p gbase = lds0 + C::OFF_G + grp * C::GBYTES;
    const lds_p XB = gbase, YB = gbase + 2 * EB;
    float* SB = reinterpret_cast<float*>(lds_raw + C::OFF_G + (size_t)grp * C::GBYTES + 4 * EB);

    {
        const uint4* src = reinterpret_cast<const uint4*>(wpk_g);     // hi plane
        uint4* dst = reinterpret_cast<uint4*>(lds_raw);
        for (int k = threadIdx.x; k < C::WBYTES / 16; k += NT) dst[k] = src[k];
        for (int k = threadIdx.x; k < H; k += NT) {
            sW0a[k] = theta[Lay::W(0) + 2 * k];
            sW0b[k] = theta[Lay::W(0) + 2 * k + 1];
            sb0[k] = theta[Lay::b(0) + k];
            sb1[k] = theta[Lay::b(1) + k];
            swo[k] = theta[Lay::Wout + k];
        }
        if (threadIdx.x == 0) sbo[0] = theta[Lay::bout];
    }
    __syncthreads();
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(sW0a + n0), w1 = *reinterpret_cast<const f32x4*>(sW0b + n0);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sb0 + n0), b1 = *reinterpret_cast<const f32x4*>(sb1 + n0);
    const f32x4 wo = *reinterpret_cast<const f32x4*>(swo + n0);
    const float bout = sbo[0];
    const lds_p wfwd = lds0 + (size_t)(part * KC) * 1024 + lane16;
    const lds_p wbwd = lds0 + (size_t)H * H * 2 + (size_t)(part * KC) * 1024 + lane16;
    // lo plane of the own weight fragments: registers for the whole kernel (the packed lo plane follows the
    // hi plane: k_pack_weights<64, 2, true>)
    bf16x8 wlo_f[KC], wlo_b[KC];
    {
        const __bf16* lo = wpk_g + (size_t)2 * H * H;
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
            wlo_f[kc] = *reinterpret_cast<const bf16x8*>(lo + ((size_t)(part * KC + kc) * 64 + lane) * 8);
            wlo_b[kc] = *reinterpret_cast<const bf16x8*>(lo + (size_t)H * H + ((size_t)(part * KC + kc) * 64 + lane) * 8);
        }
    }

    f32x4 wacc[HT], bacc0 = splat(0.f), bacc1 = splat(0.f), w0acc0 = splat(0.f), w0acc1 = splat(0.f),
          woacc = splat(0.f);
#pragma unroll
    for (int it = 0; it < HT; it++) wacc[it] = splat(0.f);
    float acc_bout = 0.f, acc_pde = 0.f, acc_bc1 = 0.f, acc_bc2 = 0.f;
    float invN1 = A.invN1, invN2 = A.invN2;
    if (A.bc_counts) {
        const long long c1 = A.bc_counts[0], c2 = A.bc_counts[1];
        invN1 = c1 > 0 ? 1.0f / (float)c1 : 0.0f;
        invN2 = c2 > 0 ? 1.0f / (float)c2 : 0.0f;
    }
    const float lead = (g == 0 && part == 0) ? 1.0f : 0.0f;
    const int64_t ntile = (A.n + 15) / 16;
    const int64_t tstride = (int64_t)gridDim.x * C::GROUPS;
    const int64_t iters = (ntile + tstride - 1) / tstride;       // same for every wave: barriers match

    struct PointRec { float x, y, z, R, E, g; unsigned mk; };
    auto load_rec = [&](int64_t itn) {
        const int64_t tl = itn * tstride + (int64_t)blockIdx.x * C::GROUPS + grp;
        int64_t id = tl * 16 + c;
        id = (tl < ntile && id < A.n) ? id : A.n - 1;
        PointRec r;
        r.x = A.x[id]; r.y = A.y[id]; r.z = A.z[id]; r.R = A.R[id]; r.E = A.E[id]; r.g = A.g[id];
        r.mk = A.mask ? A.mask[id] : 0u;
        return r;
    };
    auto put = [&](lds_p eb, int s, f32x4 v) {      // own 4 neurons of stream s: hi plane, lo plane
        u32x2 ph, pl;
        split4(v, ph, pl);
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(eb + own_slot + s * 1024) = ph;
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(eb + EB1 + own_slot + s * 1024) = pl;
    };
    auto bfrag = [&](lds_p eb, int kc, int s) {
        return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(eb + sw16 + (kc * NST + s) * 1024);
    };
    auto trfrag = [&](lds_p eb, int T, int ph) {
        const int o = ph * 2048 + (T >> 1) * NST * 1024 + (T & 1) * 512;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(eb + trl + o));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(eb + trh + o));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    // D[own 16 rows][2 passes x NST column tiles] += W x B with split operands: per (k chunk, pass) group the
    // four hi*hi MFMAs issue first (their B fragments were requested a group ahead), the lo B fragments are
    // requested at the head of the group and consumed by the hi*lo MFMAs, the lo*hi ones close the group
    auto gemm8 = [&](lds_p wbase, const bf16x8 (&wlo)[KC], lds_p eb, f32x4 (&d)[2][NST]) {
        constexpr int NG = 2 * KC;
        bf16x8 w[2], xh[2][NST], xl[NST];
        auto fetch_hi = [&](int j, bf16x8 (&xx)[NST]) {
            const int kc = j >> 1, ps = j & 1;
            if (ps == 0) w[kc & 1] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(wbase + kc * 1024);
#pragma unroll
            for (int s = 0; s < NST; s++) xx[s] = bfrag(eb + ps * EB, kc, s);
        };
        fetch_hi(0, xh[0]);
#pragma unroll
        for (int j = 0; j < NG; j++) {
            const int kc = j >> 1, ps = j & 1;
#pragma unroll
            for (int s = 0; s < NST; s++) xl[s] = bfrag(eb + ps * EB + EB1, kc, s);
            if (j + 1 < NG) fetch_hi(j + 1, xh[(j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NST; s++) d[ps][s] = mfma(w[kc & 1], xh[j & 1][s], d[ps][s]);
#pragma unroll
            for (int s = 0; s < NST; s++) d[ps][s] = mfma(w[kc & 1], xl[s], d[ps][s]);
#pragma unroll
            for (int s = 0; s < NST; s++) d[ps][s] = mfma(wlo[kc], xh[j & 1][s], d[ps][s]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    PointRec nxt = load_rec(0);
    for (int64_t itx = 0; itx < iters; itx++) {
        const int64_t tile = itx * tstride + (int64_t)blockIdx.x * C::GROUPS + grp;
        const int64_t idx = tile * 16 + c;
        const bool valid = tile < ntile && idx < A.n;
        const PointRec cur = nxt;
        if (itx + 1 < iters) nxt = load_rec(itx + 1);
        // ================= P1: inputs and layer 0 of both passes =================
        float inA[2][NST], ri1, ri2;
        {
            const float rho2 = cur.y * cur.y + cur.z * cur.z, rho = __builtin_amdgcn_sqrtf(rho2);
            orbital_cyl<float>(cur.x - cur.R, rho2, rho, inA[0], ri1);
            orbital_cyl<float>(cur.x + cur.R, rho2, rho, inA[1], ri2);
        }
        const float F = inA[0][0] + A.P * inA[1][0];
        const float lapF = inA[0][SLP] + A.P * inA[1][SLP];
        f32x4 h0[2], sv[2][NST];
#pragma unroll
        for (int ps = 0; ps < 2; ps++) {
            f32x4 a[NST];
#pragma unroll
            for (int s = 0; s < NST; s++) a[s] = w0 * inA[ps][s] + w1 * inA[1 - ps][s];
            a[0] += b0;
            f32x4 s0, s1, s2;
            act012v<ACT>(a[0], s0, s1, s2);
            const f32x4 q = a[1] * a[1] + a[2] * a[2];
            h0[ps] = s0;
            put(XB + ps * EB, 0, s0);
            put(XB + ps * EB, 1, s1 * a[1]);
            put(XB + ps * EB, 2, s1 * a[2]);
            put(XB + ps * EB, SLP, s1 * a[SLP] + s2 * q);
        }
        __syncthreads();
        // ================= P2: layer-1 GEMM over both passes, activations, output partial sums =================
        {
            f32x4 acc[2][NST];
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                acc[ps][0] = b1;
#pragma unroll
                for (int s = 1; s < NST; s++) acc[ps][s] = splat(0.f);
            }
            gemm8(wfwd, wlo_f, XB, acc);
            float Sv = 0.f, Sl = 0.f;
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                f32x4 s0, s1, s2;
                act012v<ACT>(acc[ps][0], s0, s1, s2);
                const f32x4 q = acc[ps][1] * acc[ps][1] + acc[ps][2] * acc[ps][2];
                const f32x4 ol = s1 * acc[ps][SLP] + s2 * q;
                sv[ps][0] = s0;
#pragma unroll
                for (int s = 1; s < NST; s++) sv[ps][s] = acc[ps][s];
                const float pw = ps ? A.P : 1.0f;
                Sv += pw * dot4(wo, s0);
                Sl += pw * dot4(wo, ol);
            }
            const float tot = xg_sum_pair(Sv, Sl);
            if ((g & 1) == 0) SB[(part * 16 + c) * 2 + (g >> 1)] = tot;
        }
        __syncthreads();
        // ================= P3: tail, last hidden layer's adjoints of both passes =================
        Tail<float> t;
        {
            float Sv = 0.f, Sl = 0.f;
#pragma unroll
            for (int pp = 0; pp < C::WPG; pp++) { Sv += SB[(pp * 16 + c) * 2]; Sl += SB[(pp * 16 + c) * 2 + 1]; }
            tail<float>(A.out_scale * Sv + bout, A.out_scale * Sl, cur.g, cur.E, F, lapF, ri1 + ri2, A.a, A.b, A.c,
                        float(cur.mk & 1), float((cur.mk >> 1) & 1), A.invN, invN1, invN2, valid ? 1.0f : 0.0f, t);
            if (valid && g == 0 && part == 0) { A.Eb[idx] = t.Eb; A.gb[idx] = t.gb; }
            acc_pde += lead * t.l_pde;
            acc_bc1 += lead * t.l_bc1;
            acc_bc2 += lead * t.l_bc2;
            acc_bout += lead * t.Nb;
        }
#pragma unroll
        for (int ps = 0; ps < 2; ps++) {
            const float pw = ps ? A.P : 1.0f;
            const float cv = A.out_scale * pw * t.Nb, cl = A.out_scale * pw * t.lapNb;
            const f32x4 h = sv[ps][0];
            f32x4 s1, s2, s3;
            act123v<ACT>(h, s1, s2, s3);
            const f32x4 q = sv[ps][1] * sv[ps][1] + sv[ps][2] * sv[ps][2];
            woacc += cv * h + cl * (s1 * sv[ps][SLP] + s2 * q);
            const f32x4 o0 = cv * wo, ol = cl * wo;
            const f32x4 u = 2.0f * s2 * ol;
            const f32x4 y0 = s1 * o0 + (s2 * sv[ps][SLP] + s3 * q) * ol;
            bacc1 += y0;
            put(YB + ps * EB, 0, y0);
            put(YB + ps * EB, 1, u * sv[ps][1]);
            put(YB + ps * EB, 2, u * sv[ps][2]);
            put(YB + ps * EB, SLP, s1 * ol);
        }
        __syncthreads();
        // ================= P4: weight gradient, input-adjoint GEMM, layer-0 adjoints =================
        {   // K = 128 columns = 4 steps (pass, stream pair); three MFMAs per (step, column tile)
            bf16x8 fyh, fyl, fxh[2][HT], fxl[HT];
            auto fetch_hi = [&](int k, bf16x8 (&x)[HT]) {
#pragma unroll
                for (int it = 0; it < HT; it++) x[it] = trfrag(XB + (k >> 1) * EB, it, k & 1);
            };
            fetch_hi(0, fxh[0]);
#pragma unroll
            for (int k = 0; k < NST; k++) {
                fyh = trfrag(YB + (k >> 1) * EB, part, k & 1);
                fyl = trfrag(YB + (k >> 1) * EB + EB1, part, k & 1);
#pragma unroll
                for (int it = 0; it < HT; it++) fxl[it] = trfrag(XB + (k >> 1) * EB + EB1, it, k & 1);
                if (k + 1 < NST) fetch_hi(k + 1, fxh[(k + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < HT; it++) wacc[it] = mfma(fyh, fxh[k & 1][it], wacc[it]);
#pragma unroll
                for (int it = 0; it < HT; it++) wacc[it] = mfma(fyh, fxl[it], wacc[it]);
#pragma unroll
                for (int it = 0; it < HT; it++) wacc[it] = mfma(fyl, fxh[k & 1][it], wacc[it]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            f32x4 ob[2][NST];
#pragma unroll
            for (int ps = 0; ps < 2; ps++)
#pragma unroll
                for (int s = 0; s < NST; s++) ob[ps][s] = splat(0.f);
            gemm8(wbwd, wlo_b, YB, ob);
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                f32x4 a[NST];
#pragma unroll
                for (int s = 1; s < NST; s++) a[s] = w0 * inA[ps][s] + w1 * inA[1 - ps][s];
                f32x4 s1, s2, s3;
                act123v<ACT>(h0[ps], s1, s2, s3);
                const f32x4 q = a[1] * a[1] + a[2] * a[2];
                const f32x4 u = 2.0f * s2 * ob[ps][SLP];
                f32x4 ab[NST];
                ab[SLP] = s1 * ob[ps][SLP];
                ab[1] = s1 * ob[ps][1] + u * a[1];
                ab[2] = s1 * ob[ps][2] + u * a[2];
                ab[0] = s1 * ob[ps][0] + s2 * (a[1] * ob[ps][1] + a[2] * ob[ps][2]) + (s2 * a[SLP] + s3 * q) * ob[ps][SLP];
                bacc0 += ab[0];
#pragma unroll
                for (int s = 0; s < NST; s++) { w0acc0 += ab[s] * inA[ps][s]; w0acc1 += ab[s] * inA[1 - ps][s]; }
            }
        }
        __syncthreads();   // one set of exchange buffers: P1 of the next tile overwrites XB
    }
    // =========================== block reduction ===========================
    constexpr int PT = Lay::size;
    float* out = A.partial + (size_t)blockIdx.x * A.stride;
    const float s0 = row16_sum(xg_sum(acc_pde)), s1r = row16_sum(xg_sum(acc_bc1)),
                s2r = row16_sum(xg_sum(acc_bc2)), s3r = row16_sum(xg_sum(acc_bout));
#pragma unroll
    for (int r = 0; r < 4; r++) {
        bacc0[r] = row16_sum(bacc0[r]); bacc1[r] = row16_sum(bacc1[r]);
        w0acc0[r] = row16_sum(w0acc0[r]); w0acc1[r] = row16_sum(w0acc1[r]);
        woacc[r] = row16_sum(woacc[r]);
    }
    // group 1 hands its accumulators to group 0 through LDS (the exchange buffers are dead): [wave][36][64 lanes]
    float* xfer = reinterpret_cast<float*>(lds_raw + C::OFF_G) + (size_t)part * 36 * 64 + lane;
    float* sc = reinterpret_cast<float*>(lds_raw);
    if (threadIdx.x < 4) sc[threadIdx.x] = 0.f;
    if (grp == 1) {
#pragma unroll
        for (int it = 0; it < HT; it++)
#pragma unroll
            for (int r = 0; r < 4; r++) xfer[(it * 4 + r) * 64] = wacc[it][r];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            xfer[(16 + r) * 64] = bacc0[r]; xfer[(20 + r) * 64] = bacc1[r]; xfer[(24 + r) * 64] = w0acc0[r];
            xfer[(28 + r) * 64] = w0acc1[r]; xfer[(32 + r) * 64] = woacc[r];
        }
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int it = 0; it < HT; it++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nj = q_neuron(16 * part + 4 * g + r), ni = q_neuron(16 * it + c);
                out[Lay::W(1) + nj * H + ni] = wacc[it][r] + xfer[(it * 4 + r) * 64];
            }
        if (c == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nj = n0 + r;
                out[Lay::b(0) + nj] = bacc0[r] + xfer[(16 + r) * 64];
                out[Lay::b(1) + nj] = bacc1[r] + xfer[(20 + r) * 64];
                out[Lay::W(0) + 2 * nj] = w0acc0[r] + xfer[(24 + r) * 64];
                out[Lay::W(0) + 2 * nj + 1] = w0acc1[r] + xfer[(28 + r) * 64];
                out[Lay::Wout + nj] = woacc[r] + xfer[(32 + r) * 64];
            }
        }
    }
    for (int w = 0; w < C::WAVES; w++) {   // per-point sums live in one wave per group; fixed order
        if (wave == w && lane == 0) { sc[0] += s0; sc[1] += s1r; sc[2] += s2r; sc[3] += s3r; }
        __syncthreads();
    }
    if (threadIdx.x < 3) out[PT + threadIdx.x] = sc[threadIdx.x];
    if (threadIdx.x == 3) out[Lay::bout] = sc[3];
}

template <int ACT>
hipError_t launch_trunk2x_t(const float* theta, __bf16* wpk, const TrunkArgs& A, int grid, hipStream_t st)
{
    using C1 = MCfg<64, 2>;
    using C = T2xCfg;
    const int nfrag = C1::NW * 2 * C1::HT * C1::KC * 64;
    hipLaunchKernelGGL((k_pack_weights<64, 2, true>), dim3((nfrag + 255) / 256), dim3(256), 0, st, theta, wpk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    auto kern = k_trunk2x<ACT>;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::WAVES * 64), C::LDS_BYTES, st, theta, (const __bf16*)wpk, A);
    return hipGetLastError();
}

// =====================================================================================
// k_trunkD<H, L, ACT>: deep trunks (L = 4), symmetric double pass, full reverse sweep, bf16 -- k_trunk2's
// decomposition (H/16 waves share a 16-point tile, each owning 16 neurons, both passes swept together: every
// GEMM phase handles the tile's 2 x NST x 16 = 128 columns) carried through L - 1 dense layers:
//   * LDS holds only what waves exchange: the outputs of layers 0..L-2 (the B operands of the next layer's
//     GEMM and, read transposed, of the weight gradients) and ONE adjoint buffer.  4x128: 96 + 32 KB.
//   * the weights are NOT staged in LDS: in the transposed layout every weight fragment is used by exactly one
//     wave (the owner of its 16 rows), so each wave reads its A fragments of the next GEMM straight from the
//     packed, L2-resident set into registers while the elementwise phase before that GEMM runs.
//   * the L - 1 weight-gradient matrices accumulate in registers for the whole kernel (4x128: 96 registers
//     per lane = 196 KB of the CU's 512 KB).  To run two waves per SIMD beside them (a single wave issues a
//     VALU instruction only every 5.2 cycles, two waves one every 2.6), H = 128 parks the saved state of the
//     middle layers -- (h, a_x, a_rho, a_lap) as bf16 pairs, 16 registers per layer -- in a per-block 64 KB
//     region of global memory that is rewritten every tile and therefore stays in L2 (STASH): stores right
//     after the forward activation, loads requested a GEMM ahead of the reverse step that needs them.
//   * layer 0 is re-evaluated in the reverse sweep (K = 2: a handful of FMAs and one activation per neuron)
//     instead of being saved.
// Barriers per tile: 1 (layer 0) + (L-1) forward + 1 (tail) + per reverse step one after the GEMMs (the adjoint
// buffer is overwritten by the next layer's adjoints) and one after those are written.
template <int H, int L>
struct TDCfg {
    static constexpr int HT = H / 16, KC = H / 32, WAVES = H / 16, NW = L - 1;
    static constexpr int OCC = H == 64 ? 2 : 1;                  // resident blocks per CU
    static constexpr bool STASH = H >= 128;
    static constexpr int EB = KC * NST * 64 * 16;                 // bytes of one (pass) exchange buffer
    static constexpr int NSMALL = 3 * H + NW * H + H + 4;         // W0a W0b b0 | b_1..b_{L-1} | wout | bout
    static constexpr int RB = 4 * WAVES, RECF = 12;
    static constexpr int RBYTES = RB * 16 * RECF * 4;
    static constexpr size_t OFF_X = 0;                            // [NW][pass]
    static constexpr size_t OFF_Y = OFF_X + (size_t)NW * 2 * EB;
    static constexpr size_t OFF_SMALL = OFF_Y + 2 * EB;
    static constexpr size_t OFF_S = OFF_SMALL + NSMALL * 4;
    static constexpr size_t OFF_R = OFF_S + WAVES * 16 * 2 * 4;
    static constexpr size_t LDS_BYTES = OFF_R + RBYTES;
    static constexpr size_t STASH_BLOCK = STASH ? (size_t)(NW - 1) * WAVES * 4 * 1024 : 0;   // bytes per block
    static_assert(OCC * LDS_BYTES <= 160 * 1024, "k_trunkD does not fit the CU's LDS");
    static_assert(OFF_S % 16 == 0 && OFF_R % 16 == 0, "alignment");
};

// VAR (measurement variant, tools/kbench.py): bit 0 = exchange slots lane-linear (round 3's layout) instead of swizzled.
// (s_setprio variants -- the younger half of the block at priority 1; the halves leading one pass of a phase each -- measured
// +0.8 % and +49 % in round 4 and removed: profiles/r04_trunkD_notes.md.)
template <int H, int L, int ACT, int VAR = 0>
__global__ __launch_bounds__((TDCfg<H, L>::WAVES * 64), (TDCfg<H, L>::OCC * TDCfg<H, L>::WAVES / 4))
void k_trunkD(const float* __restrict__ theta, const __bf16* __restrict__ wpk_g, uint4* __restrict__ stash_g, TrunkArgs A)
{
    constexpr bool SWZ = (VAR & 1) == 0;
    using C = TDCfg<H, L>;
    using Lay = TrunkLayout<H, L>;
    constexpr int HT = C::HT, KC = C::KC, NT = C::WAVES * 64, EB = C::EB, NW = C::NW;
    constexpr bool STASH = C::STASH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef __attribute__((address_space(3))) unsigned char* lds_p;
    const lds_p lds0 = (lds_p)lds_raw;
    float* small = reinterpret_cast<float*>(lds_raw + C::OFF_SMALL);
    float* sW0a = small;              // W0[:, 0]
    float* sW0b = small + H;          // W0[:, 1]
    float* sb0 = small + 2 * H;
    float* sbl = small + 3 * H;       // b_l at + (l - 1) * H
    float* swo = small + 3 * H + NW * H;
    float* sbo = swo + H;
    float* SB = reinterpret_cast<float*>(lds_raw + C::OFF_S);   // [wave][16 points][2]

    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // row tile owned by this wave
    const int g = lane >> 4, c = lane & 15;
    const int lane16 = lane * 16;
    // Slot of lane (g, c) inside a 1 KB (k chunk, stream) block of an exchange buffer: lane-linear, or (SWZ) with the
    // 16-byte slots of the odd lane groups permuted by c ^ 12.  The transposed reads of the weight gradient fetch, per
    // 32-lane service group, the slots c' in {0-3, 8-11} (then {4-7, 12-15}) of TWO adjacent lane groups, which the
    // linear layout puts on the same banks (2-way conflict: half the LDS rate in the weight-gradient phases, where
    // eight waves read 2 x 36 fragments per layer and tile); c ^ 12 moves the odd group's to the other banks and keeps
    // the 16-byte fragment reads (hardware groups {0-3, 12-15, 20-27}, ...) and the 8-byte stores conflict-free.
    auto slot16 = [](int gg, int cc) { return (gg * 16 + (SWZ ? (cc ^ ((gg & 1) * 12)) : cc)) * 16; };
    const int sw16 = slot16(g, c);
    const int own_slot = sw16 + (part >> 1) * NST * 64 * 16 + 8 * (part & 1);
    // transposed read: this lane fetches half (c & 1) of the slot of source lane (g' = (c & 3) >> 1 [+ 2 (T & 1)],
    // c' = 8 (g & 1) + (c >> 2) [+ 4 for the second read]) of stream pair member g >> 1
    const int trl = (g >> 1) * 1024 + slot16((c & 3) >> 1, 8 * (g & 1) + (c >> 2)) + (c & 1) * 8;
    const int trh = SWZ ? (g >> 1) * 1024 + slot16((c & 3) >> 1, 8 * (g & 1) + (c >> 2) + 4) + (c & 1) * 8 : trl + 64;
    const int n0 = 16 * part + 4 * g;                                    // first own neuron of this lane

    for (int k = threadIdx.x; k < H; k += NT) {
        sW0a[k] = theta[Lay::W(0) + 2 * k];
        sW0b[k] = theta[Lay::W(0) + 2 * k + 1];
        sb0[k] = theta[Lay::b(0) + k];
#pragma unroll
        for (int l = 1; l < L; l++) sbl[(l - 1) * H + k] = theta[Lay::b(l) + k];
        swo[k] = theta[Lay::Wout + k];
    }
    if (threadIdx.x == 0) sbo[0] = theta[Lay::bout];
    __syncthreads();
    constexpr float kEx = (ACT == ACT_SIGMOID ? -1.0f : -2.0f) * 1.4426950408889634f;
    const float bout = sbo[0];

    // A fragments of matrix m = (l - 1) * 2 + dir of this wave's 16 rows: buffer loads (resource in SGPRs, one
    // 32-bit lane offset, the matrix offset as the scalar offset) -- no per-lane 64-bit addresses for hipcc to
    // hoist out of the tile loop and spill
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wpk_g), 0, NW * 2 * H * H * 2, 0x00020000);
    const int wvoff = part * KC * 1024 + lane16;
    auto loadA = [&](int m, bf16x8 (&a)[KC]) {
#pragma unroll
        for (int kc = 0; kc < KC; kc++)
            a[kc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + kc * 1024, m * H * H * 2, 0));
    };
    // parked saved state of layer l (1..L-2): 8 x u32x2 = 4 x 16 bytes per lane, [k][lane] per wave
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(stash_g, 0, STASH ? (int)(C::STASH_BLOCK * gridDim.x) : 0, 0x00020000);
    const int svoff = ((blockIdx.x * (NW - 1) * C::WAVES + part) * 256 + lane) * 16;
    auto stash_put = [&](int l, int k, u32x4 v) {
        __builtin_amdgcn_raw_buffer_store_b128(v, srs, svoff + k * 1024, (l - 1) * C::WAVES * 4096, 0);
    };
    auto stash_get = [&](int l, int k) {
        return __builtin_amdgcn_raw_buffer_load_b128(srs, svoff + k * 1024, (l - 1) * C::WAVES * 4096, 0);
    };

    f32x4 wacc[NW][HT], bacc[L], w0acc0 = splat(0.f), w0acc1 = splat(0.f), woacc = splat(0.f);
#pragma unroll
    for (int l = 0; l < NW; l++)
#pragma unroll
        for (int it = 0; it < HT; it++) wacc[l][it] = splat(0.f);
#pragma unroll
    for (int l = 0; l < L; l++) bacc[l] = splat(0.f);
    float acc_bout = 0.f, acc_pde = 0.f, acc_bc1 = 0.f, acc_bc2 = 0.f;
    float invN1 = A.invN1, invN2 = A.invN2;
    if (A.bc_counts) {
        const long long c1 = A.bc_counts[0], c2 = A.bc_counts[1];
        invN1 = c1 > 0 ? 1.0f / (float)c1 : 0.0f;
        invN2 = c2 > 0 ? 1.0f / (float)c2 : 0.0f;
    }
    const float lead = (g == 0 && part == 0) ? 1.0f : 0.0f;
    const int64_t ntile = (A.n + 15) / 16;
    const int64_t iters = (ntile + gridDim.x - 1) / gridDim.x;   // same for every block: barriers match

    constexpr int RB = C::RB, RECF = C::RECF;
    float* ring = reinterpret_cast<float*>(lds_raw + C::OFF_R);
    struct PointRec { float x, y, z, R, E, g; unsigned mk; };
    auto load_rec = [&](int64_t b) {
        const int64_t tl = (b * RB + 4 * part + g) * gridDim.x + blockIdx.x;
        int64_t id = tl * 16 + c;
        id = (tl < ntile && id < A.n) ? id : A.n - 1;
        PointRec r;
        r.x = A.x[id]; r.y = A.y[id]; r.z = A.z[id]; r.R = A.R[id]; r.E = A.E[id]; r.g = A.g[id];
        r.mk = A.mask ? A.mask[id] : 0u;
        return r;
    };
    auto fill_ring = [&](const PointRec& r) {
        float i0[NST], i1[NST], ri1, ri2;
        const float rho2 = r.y * r.y + r.z * r.z, rho = __builtin_amdgcn_sqrtf(rho2);
        orbital_cyl<float>(r.x - r.R, rho2, rho, i0, ri1);
        orbital_cyl<float>(r.x + r.R, rho2, rho, i1, ri2);
        float* d = ring + ((4 * part + g) * 16 + c) * RECF;
        *reinterpret_cast<f32x4*>(d) = f32x4{i0[0], i0[1], i0[2], i0[3]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{i1[0], i1[1], i1[2], i1[3]};
        *reinterpret_cast<f32x4*>(d + 8) = f32x4{ri1 + ri2, r.E, r.g, (float)r.mk};   // mask bits as 0.0 .. 3.0
    };
    auto put = [&](lds_p eb, int s, f32x4 v) {      // own 4 neurons of stream s -> 8 bytes of the B fragment
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(eb + own_slot + s * 1024) = pack4(v);
    };
    auto bfrag = [&](lds_p eb, int kc, int s) {
        return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(eb + sw16 + (kc * NST + s) * 1024);
    };
    auto trfrag = [&](lds_p eb, int T, int ph) {     // rows 32ph..32ph+31 of the [column][neuron] image, tile T
        const int o = ph * 2048 + (T >> 1) * NST * 1024 + (T & 1) * 512;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(eb + trl + o));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(eb + trh + o));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    // D[own 16 rows][NST column tiles of ONE pass] += A x B: A fragments in registers, B fragments from the pass
    // buffer at eb; one group of NST MFMAs per k chunk, the operands of chunk kc+1 requested before the MFMAs of
    // chunk kc issue.  The two passes of a phase run one after the other in registers (half the accumulators
    // and elementwise temporaries live at a time; MFMA and VALU work do not overlap on a SIMD anyway), sharing
    // the A fragments and the phase's barriers.
    auto gemmP = [&](const bf16x8 (&a)[KC], lds_p eb, f32x4 (&d)[NST]) {
        constexpr int HS = NST;                     // streams per operand group
        constexpr int NG = KC * (NST / HS);
        bf16x8 x[2][HS];
        auto fetch = [&](int j, bf16x8 (&xx)[HS]) {
            const int kc = j / (NST / HS), s0 = (j % (NST / HS)) * HS;
#pragma unroll
            for (int s = 0; s < HS; s++) xx[s] = bfrag(eb, kc, s0 + s);
        };
        fetch(0, x[0]);
#pragma unroll
        for (int j = 0; j < NG; j++) {
            if (j + 1 < NG) fetch(j + 1, x[(j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int kc = j / (NST / HS), s0 = (j % (NST / HS)) * HS;
#pragma unroll
            for (int s = 0; s < HS; s++) d[s0 + s] = mfma(a[kc], x[j & 1][s], d[s0 + s]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // weight gradient of one layer: K = 128 columns = NST steps (pass, stream pair) x HT column tiles, in groups
    // of WG MFMAs whose operands are requested one group ahead
    auto wgrad = [&](f32x4 (&wa)[HT], lds_p yb, lds_p xb) {
        constexpr int WG = 4;
        constexpr int GPK = HT / WG, NG = NST * GPK;
        bf16x8 fy[2], fx[2][WG];
        auto fetch = [&](int j, bf16x8 (&x)[WG]) {
            const int k = j / GPK, i0 = (j % GPK) * WG;
            if (j % GPK == 0) fy[k & 1] = trfrag(yb + (k >> 1) * EB, part, k & 1);
#pragma unroll
            for (int i = 0; i < WG; i++) x[i] = trfrag(xb + (k >> 1) * EB, i0 + i, k & 1);
        };
        fetch(0, fx[0]);
#pragma unroll
        for (int j = 0; j < NG; j++) {
            if (j + 1 < NG) fetch(j + 1, fx[(j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int k = j / GPK, i0 = (j % GPK) * WG;
#pragma unroll
            for (int i = 0; i < WG; i++) wa[i0 + i] = mfma(fy[k & 1], fx[j & 1][i], wa[i0 + i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto ring_in = [&](int slot, float (&inA)[2][NST]) {
        const float* rp = ring + (slot * 16 + c) * RECF;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
        for (int s = 0; s < NST; s++) { inA[0][s] = r0[s]; inA[1][s] = r1[s]; }
    };

    const lds_p YB = lds0 + C::OFF_Y;
    // the raw points of a ring batch are requested a whole batch ahead (seven registers held for RB trips)
    constexpr bool RPF = true;   // (false: load at the refill, 7 registers less, measured 1.5 % slower at 4x128)
    PointRec nxt;
    if (RPF) nxt = load_rec(0);
    bf16x8 afr[KC];
    loadA(0, afr);
    // -DPINN_STAMPS (diagnostic build, tools/stamps.py): per-wave cycle totals of the phases and barrier waits
#ifdef PINN_STAMPS
    unsigned long long st_acc[24], st_last = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 24; i++) st_acc[i] = 0;
#define TD_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; } while (0)
#else
#define TD_STAMP(i) do { } while (0)
#endif
    for (int64_t itx = 0; itx < iters; itx++) {
        const int64_t tile = itx * gridDim.x + blockIdx.x;
        const int64_t idx = tile * 16 + c;
        const bool valid = tile < ntile && idx < A.n;
        const int slot = (int)(itx % RB);
        if (slot == 0) {
            if (itx) __syncthreads();           // the last reverse step of the previous trip read the old batch
            if (RPF) {
                fill_ring(nxt);
                nxt = load_rec(itx / RB + 1);   // raw points of the next batch: nothing waits for them for RB trips
            } else {
                fill_ring(load_rec(itx / RB));
            }
            __syncthreads();
        }
        TD_STAMP(22);
        // ================= layer 0 of both passes -> X[0] =================
        {
            float inA[2][NST];
            ring_in(slot, inA);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(sW0a + n0), w1 = *reinterpret_cast<const f32x4*>(sW0b + n0);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sb0 + n0);
            const lds_p X0 = lds0 + C::OFF_X;
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                f32x4 a[NST];
#pragma unroll
                for (int s = 1; s < NST; s++) a[s] = lin2(w0, inA[ps][s], w1, inA[1 - ps][s]);
                f32x4 s0, s1, s2;
                act012v_scaled<ACT>((lin2(w0, inA[ps][0], w1, inA[1 - ps][0]) + b0) * kEx, s0, s1, s2);
                const f32x4 q = a[1] * a[1] + a[2] * a[2];
                put(X0 + ps * EB, 0, s0);
                put(X0 + ps * EB, 1, s1 * a[1]);
                put(X0 + ps * EB, 2, s1 * a[2]);
                put(X0 + ps * EB, SLP, s1 * a[SLP] + s2 * q);
            }
        }
        TD_STAMP(0);
        __syncthreads();
        TD_STAMP(1);
        // ================= layers 1 .. L-1 =================
        u32x2 sv[STASH ? 1 : (NW > 1 ? NW - 1 : 1)][2][NST];   // packed saved state of layers 1..L-2 (registers)
        f32x4 svl[2][NST];                                      // (h, a_x, a_rho, a_lap) of the last hidden layer
#pragma unroll
        for (int l = 1; l < L; l++) {
            const lds_p Xin = lds0 + C::OFF_X + (size_t)(l - 1) * 2 * EB;
            const lds_p Xout = lds0 + C::OFF_X + (size_t)l * 2 * EB;      // (not used by the last layer)
            float Sv = 0.f, Sl = 0.f;
            auto fwdG = [&](int ps, f32x4 (&acc)[NST]) {
                acc[0] = *reinterpret_cast<const f32x4*>(sbl + (l - 1) * H + n0);
#pragma unroll
                for (int s = 1; s < NST; s++) acc[s] = splat(0.f);
                gemmP(afr, Xin + ps * EB, acc);
                if (ps == 1) {
                    // the A fragments of the next GEMM: layer l+1 forward, or the transposed last layer
                    loadA(l < L - 1 ? l * 2 : (L - 2) * 2 + 1, afr);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto fwdE = [&](int ps, f32x4 (&acc)[NST]) {
                f32x4 s0, s1, s2;
                act012v<ACT>(acc[0], s0, s1, s2);
                const f32x4 q = acc[1] * acc[1] + acc[2] * acc[2];
                const f32x4 ol = s1 * acc[SLP] + s2 * q;
                if (l < L - 1) {
                    put(Xout + ps * EB, 0, s0);
                    put(Xout + ps * EB, 1, s1 * acc[1]);
                    put(Xout + ps * EB, 2, s1 * acc[2]);
                    put(Xout + ps * EB, SLP, ol);
                    u32x2 p[NST];
                    p[0] = pack4(s0);
#pragma unroll
                    for (int s = 1; s < NST; s++) p[s] = pack4(acc[s]);
                    if constexpr (STASH) {
                        stash_put(l, 2 * ps, u32x4{p[0][0], p[0][1], p[1][0], p[1][1]});
                        stash_put(l, 2 * ps + 1, u32x4{p[2][0], p[2][1], p[3][0], p[3][1]});
                    } else {
#pragma unroll
                        for (int s = 0; s < NST; s++) sv[l - 1][ps][s] = p[s];
                    }
                } else {
                    const f32x4 wo = *reinterpret_cast<const f32x4*>(swo + n0);
                    svl[ps][0] = s0;
#pragma unroll
                    for (int s = 1; s < NST; s++) svl[ps][s] = acc[s];
                    const float pw = ps ? A.P : 1.0f;
                    Sv += pw * dot4(wo, s0);
                    Sl += pw * dot4(wo, ol);
                }
            };
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                f32x4 acc[NST];
                fwdG(ps, acc);
                fwdE(ps, acc);
            }
            if (l == L - 1) {
                const float tot = xg_sum_pair(Sv, Sl);          // Sv in lane groups 0,1; Sl in groups 2,3
                if ((g & 1) == 0) SB[(part * 16 + c) * 2 + (g >> 1)] = tot;
            }
            TD_STAMP(2 * l);
            __syncthreads();
            TD_STAMP(2 * l + 1);
        }
        // ================= tail, adjoints of the last hidden layer -> YB =================
        {
            Tail<float> t;
            {
                float inA[2][NST];
                ring_in(slot, inA);
                const f32x4 r2 = *reinterpret_cast<const f32x4*>(ring + (slot * 16 + c) * RECF + 8);
                const unsigned mkpt = (unsigned)r2[3];
                const float F = inA[0][0] + A.P * inA[1][0];
                const float lapF = inA[0][SLP] + A.P * inA[1][SLP];
                float Sv = 0.f, Sl = 0.f;
#pragma unroll
                for (int pp = 0; pp < C::WAVES; pp++) { Sv += SB[(pp * 16 + c) * 2]; Sl += SB[(pp * 16 + c) * 2 + 1]; }
                tail<float>(A.out_scale * Sv + bout, A.out_scale * Sl, r2[2], r2[1], F, lapF, r2[0], A.a, A.b, A.c,
                            float(mkpt & 1), float((mkpt >> 1) & 1), A.invN, invN1, invN2, valid ? 1.0f : 0.0f, t);
                if (valid && g == 0 && part == 0) { A.Eb[idx] = t.Eb; A.gb[idx] = t.gb; }
                acc_pde += lead * t.l_pde;
                acc_bc1 += lead * t.l_bc1;
                acc_bc2 += lead * t.l_bc2;
                acc_bout += lead * t.Nb;
            }
            const f32x4 wo = *reinterpret_cast<const f32x4*>(swo + n0);
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                const float pw = ps ? A.P : 1.0f;
                const float cv = A.out_scale * pw * t.Nb, cl = A.out_scale * pw * t.lapNb;
                const f32x4 h = svl[ps][0];
                f32x4 s1, s2, s3;
                act123v<ACT>(h, s1, s2, s3);
                const f32x4 q = svl[ps][1] * svl[ps][1] + svl[ps][2] * svl[ps][2];
                woacc += cv * h + cl * (s1 * svl[ps][SLP] + s2 * q);
                const f32x4 o0 = cv * wo, ol = cl * wo;
                const f32x4 u = 2.0f * s2 * ol;
                const f32x4 y0 = s1 * o0 + (s2 * svl[ps][SLP] + s3 * q) * ol;
                bacc[L - 1] += y0;
                put(YB + ps * EB, 0, y0);
                put(YB + ps * EB, 1, u * svl[ps][1]);
                put(YB + ps * EB, 2, u * svl[ps][2]);
                put(YB + ps * EB, SLP, s1 * ol);
            }
        }
        TD_STAMP(8);
        __syncthreads();
        TD_STAMP(9);
        // ================= reverse steps l = L-1 .. 1 =================
#pragma unroll
        for (int l = L - 1; l >= 1; l--) {
            const lds_p Xin = lds0 + C::OFF_X + (size_t)(l - 1) * 2 * EB;
            u32x4 sq[2][2];
            if constexpr (STASH) {
                if (l > 1) {
#pragma unroll
                    for (int ps = 0; ps < 2; ps++) { sq[ps][0] = stash_get(l - 1, 2 * ps); sq[ps][1] = stash_get(l - 1, 2 * ps + 1); }
                }
            }
            wgrad(wacc[l - 1], YB, Xin);
            u32x2 yp[2][NST];               // adjoints of layer l-1, packed: written to YB once every wave has read it
            float inA[2][NST];
            f32x4 w0, w1, b0;
            if (l == 1) {                   // layer 0 is re-evaluated from the point record
                ring_in(slot, inA);
                w0 = *reinterpret_cast<const f32x4*>(sW0a + n0);
                w1 = *reinterpret_cast<const f32x4*>(sW0b + n0);
                b0 = *reinterpret_cast<const f32x4*>(sb0 + n0);
            }
#pragma unroll
            for (int ps = 0; ps < 2; ps++) {
                f32x4 ob[NST];
#pragma unroll
                for (int s = 0; s < NST; s++) ob[s] = splat(0.f);
                gemmP(afr, YB + ps * EB, ob);
                if (ps == 1) {
                    // next A fragments: the transposed matrix of layer l-1, or layer 1 forward for the next tile
                    loadA(l > 1 ? (l - 2) * 2 + 1 : 0, afr);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (l > 1) {
                    u32x2 p[NST];
                    if constexpr (STASH) {
                        p[0] = u32x2{sq[ps][0][0], sq[ps][0][1]}; p[1] = u32x2{sq[ps][0][2], sq[ps][0][3]};
                        p[2] = u32x2{sq[ps][1][0], sq[ps][1][1]}; p[3] = u32x2{sq[ps][1][2], sq[ps][1][3]};
                    } else {
#pragma unroll
                        for (int s = 0; s < NST; s++) p[s] = sv[l - 2][ps][s];
                    }
                    const f32x4 h = unpack4(p[0]), a1 = unpack4(p[1]), a2 = unpack4(p[2]), al = unpack4(p[3]);
                    f32x4 s1, s2, s3;
                    act123v<ACT>(h, s1, s2, s3);
                    const f32x4 q = a1 * a1 + a2 * a2;
                    const f32x4 u = 2.0f * s2 * ob[SLP];
                    const f32x4 y0 = s1 * ob[0] + s2 * (a1 * ob[1] + a2 * ob[2]) + (s2 * al + s3 * q) * ob[SLP];
                    bacc[l - 1] += y0;
                    yp[ps][0] = pack4(y0);
                    yp[ps][1] = pack4(s1 * ob[1] + u * a1);
                    yp[ps][2] = pack4(s1 * ob[2] + u * a2);
                    yp[ps][SLP] = pack4(s1 * ob[SLP]);
                } else {
                    f32x4 a[NST];
#pragma unroll
                    for (int s = 1; s < NST; s++) a[s] = lin2(w0, inA[ps][s], w1, inA[1 - ps][s]);
                    f32x4 h0, s1, s2, s3;
                    act012v_scaled<ACT>((lin2(w0, inA[ps][0], w1, inA[1 - ps][0]) + b0) * kEx, h0, s1, s2);
                    act123v<ACT>(h0, s1, s2, s3);
                    const f32x4 q = a[1] * a[1] + a[2] * a[2];
                    const f32x4 u = 2.0f * s2 * ob[SLP];
                    f32x4 ab[NST];
                    ab[SLP] = s1 * ob[SLP];
                    ab[1] = s1 * ob[1] + u * a[1];
                    ab[2] = s1 * ob[2] + u * a[2];
                    ab[0] = s1 * ob[0] + s2 * (a[1] * ob[1] + a[2] * ob[2]) + (s2 * a[SLP] + s3 * q) * ob[SLP];
                    bacc[0] += ab[0];
#pragma unroll
                    for (int s = 0; s < NST; s++) { fmac_bcast(w0acc0, ab[s], inA[ps][s]); fmac_bcast(w0acc1, ab[s], inA[1 - ps][s]); }
                }
            }
            TD_STAMP(10 + 4 * (L - 1 - l));
            __syncthreads();                    // every wave has read YB (and X[l-1])
            TD_STAMP(11 + 4 * (L - 1 - l));
            if (l > 1) {
#pragma unroll
                for (int ps = 0; ps < 2; ps++)
#pragma unroll
                    for (int s = 0; s < NST; s++)
                        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(YB + ps * EB + own_slot + s * 1024) = yp[ps][s];
                TD_STAMP(12 + 4 * (L - 1 - l));
                __syncthreads();
                TD_STAMP(13 + 4 * (L - 1 - l));
            }
        }
    }
#ifdef PINN_STAMPS
    if (A.psi && lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(A.psi) + ((size_t)blockIdx.x * C::WAVES + part) * 24;
        for (int i = 0; i < 24; i++) dbg[i] = st_acc[i];
    }
#endif
#undef TD_STAMP
    // =========================== block reduction ===========================
    __syncthreads();
    constexpr int PT = Lay::size;
    float* out = A.partial + (size_t)blockIdx.x * A.stride;
    const float s0 = row16_sum(xg_sum(acc_pde)), s1r = row16_sum(xg_sum(acc_bc1)),
                s2r = row16_sum(xg_sum(acc_bc2)), s3r = row16_sum(xg_sum(acc_bout));
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int l = 0; l < L; l++) bacc[l][r] = row16_sum(bacc[l][r]);
        w0acc0[r] = row16_sum(w0acc0[r]); w0acc1[r] = row16_sum(w0acc1[r]);
        woacc[r] = row16_sum(woacc[r]);
    }
    float* sc = reinterpret_cast<float*>(lds_raw);
    if (threadIdx.x < 4) sc[threadIdx.x] = 0.f;
    __syncthreads();
    // every wave owns distinct gradient rows: straight to the block's partial row
#pragma unroll
    for (int l = 1; l < L; l++)
#pragma unroll
        for (int it = 0; it < HT; it++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nj = q_neuron(16 * part + 4 * g + r), ni = q_neuron(16 * it + c);
                out[Lay::W(l) + nj * H + ni] = wacc[l - 1][it][r];
            }
    if (c == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int nj = n0 + r;
#pragma unroll
            for (int l = 0; l < L; l++) out[Lay::b(l) + nj] = bacc[l][r];
            out[Lay::W(0) + 2 * nj] = w0acc0[r];
            out[Lay::W(0) + 2 * nj + 1] = w0acc1[r];
            out[Lay::Wout + nj] = woacc[r];
        }
    }
    for (int w = 0; w < C::WAVES; w++) {   // per-point sums live in one wave only; fixed order anyway
        if (part == w && lane == 0) { sc[0] += s0; sc[1] += s1r; sc[2] += s2r; sc[3] += s3r; }
        __syncthreads();
    }
    if (threadIdx.x < 3) out[PT + threadIdx.x] = sc[threadIdx.x];
    if (threadIdx.x == 3) out[Lay::bout] = sc[3];
}

template <int H, int L, int ACT, int VAR = 0>
hipError_t launch_trunkD_t(const float* theta, __bf16* wpk, const TrunkArgs& A, int grid, hipStream_t st)
{
    using C1 = MCfg<H, L>;
    using C = TDCfg<H, L>;
    const int nfrag = C1::NW * 2 * C1::HT * C1::KC * 64;
    hipLaunchKernelGGL((k_pack_weights<H, L, false>), dim3((nfrag + 255) / 256), dim3(256), 0, st, theta, wpk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    auto kern = k_trunkD<H, L, ACT, VAR>;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    uint4* stash = reinterpret_cast<uint4*>(reinterpret_cast<char*>(wpk) + mfma_packed_weight_bytes(H, L));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::WAVES * 64), C::LDS_BYTES, st, theta, (const __bf16*)wpk, stash, A);
    return hipGetLastError();
}

// =====================================================================================
// k_trunkDx<ACT>: the 4x128 trunk in split-bf16 (fp32-grade: BASELINE config 1), symmetric double pass, full
// reverse sweep -- k_trunkD's decomposition (8 waves share a 16-point tile, each owning 16 neurons; weights from
// L2 into registers; saved layer state parked in a per-block region of global memory; gradient accumulators in
// registers; two waves per SIMD) with every MFMA operand as hi + lo bf16 planes and three MFMAs per product.
// Two operand planes double every exchange buffer, so only ONE pass of a tile (64 columns) fits the LDS: the tile
// goes through   forward A (keeps only its output sums)  ->  forward B  ->  tail  ->  reverse B  ->  forward A
// again  ->  reverse A,   i.e. one extra forward pass (+17 % arithmetic) instead of a second set of buffers.
// The three stages run as ONE loop body (stage 0: forward A; 1: forward B, tail, reverse B; 2: forward A, reverse
// A) so the code of a forward and of a reverse pass exists once.  Saved state is fp32 (as in every split-bf16
// kernel: the adjoint algebra must not be the precision bottleneck): 16 registers per layer, parked for layers 1
// and 2, kept for layer 3; layer 0 is re-evaluated in the reverse sweep.
struct TDxCfg {
    static constexpr int H = 128, L = 4, HT = 8, KC = 4, WAVES = 8, NW = 3;
    static constexpr int EB1 = KC * NST * 64 * 16;                // one plane of one pass buffer
    static constexpr int EB = 2 * EB1;                            // hi plane, lo plane
    static constexpr int NSMALL = 3 * H + NW * H + H + 4;
    static constexpr int RB = 4 * WAVES, RECF = 12;
    static constexpr int RBYTES = RB * 16 * RECF * 4;
    static constexpr size_t OFF_X = 0;                            // [NW] layer-output buffers of the current pass
    static constexpr size_t OFF_Y = OFF_X + (size_t)NW * EB;
    static constexpr size_t OFF_SMALL = OFF_Y + EB;
    static constexpr size_t OFF_S = OFF_SMALL + NSMALL * 4;       // [wave][16 points][pass][Sv, Sl]
    static constexpr size_t OFF_R = OFF_S + WAVES * 16 * 2 * 2 * 4;
    static constexpr size_t LDS_BYTES = OFF_R + RBYTES;
    static constexpr size_t STASH_BLOCK = (size_t)(NW - 1) * WAVES * 4 * 1024;   // bytes per block (one pass at a time)
    static_assert(LDS_BYTES <= 160 * 1024, "k_trunkDx does not fit the CU's LDS");
    static_assert(OFF_S % 16 == 0 && OFF_R % 16 == 0, "alignment");
};

template <int ACT>
__global__ __launch_bounds__(TDxCfg::WAVES * 64, 2)
void k_trunkDx(const float* __restrict__ theta, const __bf16* __restrict__ wpk_g, uint4* __restrict__ stash_g, TrunkArgs A)
{
    using C = TDxCfg;
    constexpr int H = C::H, L = C::L, HT = C::HT, KC = C::KC, NT = C::WAVES * 64, EB = C::EB, EB1 = C::EB1, NW = C::NW;
    using Lay = TrunkLayout<H, L>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef __attribute__((address_space(3))) unsigned char* lds_p;
    const lds_p lds0 = (lds_p)lds_raw;
    float* small = reinterpret_cast<float*>(lds_raw + C::OFF_SMALL);
    float* sW0a = small;
    float* sW0b = small + H;
    float* sb0 = small + 2 * H;
    float* sbl = small + 3 * H;       // b_l at + (l - 1) * H
    float* swo = small + 3 * H + NW * H;
    float* sbo = swo + H;
    float* SB = reinterpret_cast<float*>(lds_raw + C::OFF_S);

    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int lane16 = lane * 16;
    // (lane-linear exchange slots: the swizzle of the other merged-pass kernels costs this one a register it does not
    // have -- 0.906 -> 0.946 ms per 100 000 points with it, A/B on one box, round 4)
    const int own_slot = lane16 + (part >> 1) * NST * 64 * 16 + 8 * (part & 1);
    const int trl = (g >> 1) * 1024 + (8 * (g & 1) + (c >> 2)) * 16 + (c & 1) * 8 + ((c & 3) >> 1) * 256;
    const int n0 = 16 * part + 4 * g;

    for (int k = threadIdx.x; k < H; k += NT) {
        sW0a[k] = theta[Lay::W(0) + 2 * k];
        sW0b[k] = theta[Lay::W(0) + 2 * k + 1];
        sb0[k] = theta[Lay::b(0) + k];
#pragma unroll
        for (int l = 1; l < L; l++) sbl[(l - 1) * H + k] = theta[Lay::b(l) + k];
        swo[k] = theta[Lay::Wout + k];
    }
    if (threadIdx.x == 0) sbo[0] = theta[Lay::bout];
    __syncthreads();
    const float bout = sbo[0];

    // A fragments of matrix m = (l - 1) * 2 + dir of this wave's 16 rows: hi plane, lo plane (one packed plane later)
    constexpr int WPLANE_B = NW * 2 * H * H * 2;                  // bytes of one packed weight plane
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wpk_g), 0, 2 * WPLANE_B, 0x00020000);
    const int wvoff = part * KC * 1024 + lane16;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(stash_g, 0, (int)(C::STASH_BLOCK * gridDim.x), 0x00020000);
    const int svoff = ((blockIdx.x * (NW - 1) * C::WAVES + part) * 256 + lane) * 16;
    auto stash_put = [&](int l, int k, f32x4 v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srs, svoff + k * 1024, (l - 1) * C::WAVES * 4096, 0);
    };
    auto stash_get = [&](int l, int k) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, svoff + k * 1024, (l - 1) * C::WAVES * 4096, 0));
    };

    f32x4 wacc[NW][HT], bacc[L], w0acc0 = splat(0.f), w0acc1 = splat(0.f), woacc = splat(0.f);
#pragma unroll
    for (int l = 0; l < NW; l++)
#pragma unroll
        for (int it = 0; it < HT; it++) wacc[l][it] = splat(0.f);
#pragma unroll
    for (int l = 0; l < L; l++) bacc[l] = splat(0.f);
    float acc_bout = 0.f, acc_pde = 0.f, acc_bc1 = 0.f, acc_bc2 = 0.f;
    float invN1 = A.invN1, invN2 = A.invN2;
    if (A.bc_counts) {
        const long long c1 = A.bc_counts[0], c2 = A.bc_counts[1];
        invN1 = c1 > 0 ? 1.0f / (float)c1 : 0.0f;
        invN2 = c2 > 0 ? 1.0f / (float)c2 : 0.0f;
    }
    const float lead = (g == 0 && part == 0) ? 1.0f : 0.0f;
    const int64_t ntile = (A.n + 15) / 16;
    const int64_t iters = (ntile + gridDim.x - 1) / gridDim.x;

    constexpr int RB = C::RB, RECF = C::RECF;
    float* ring = reinterpret_cast<float*>(lds_raw + C::OFF_R);
    struct PointRec { float x, y, z, R, E, g; unsigned mk; };
    auto load_rec = [&](int64_t b) {
        const int64_t tl = (b * RB + 4 * part + g) * gridDim.x + blockIdx.x;
        int64_t id = tl * 16 + c;
        id = (tl < ntile && id < A.n) ? id : A.n - 1;
        PointRec r;
        r.x = A.x[id]; r.y = A.y[id]; r.z = A.z[id]; r.R = A.R[id]; r.E = A.E[id]; r.g = A.g[id];
        r.mk = A.mask ? A.mask[id] : 0u;
        return r;
    };
    auto fill_ring = [&](const PointRec& r) {
        float i0[NST], i1[NST], ri1, ri2;
        const float rho2 = r.y * r.y + r.z * r.z, rho = __builtin_amdgcn_sqrtf(rho2);
        orbital_cyl<float>(r.x - r.R, rho2, rho, i0, ri1);
        orbital_cyl<float>(r.x + r.R, rho2, rho, i1, ri2);
        float* d = ring + ((4 * part + g) * 16 + c) * RECF;
        *reinterpret_cast<f32x4*>(d) = f32x4{i0[0], i0[1], i0[2], i0[3]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{i1[0], i1[1], i1[2], i1[3]};
        *reinterpret_cast<f32x4*>(d + 8) = f32x4{ri1 + ri2, r.E, r.g, (float)r.mk};
    };
    auto put = [&](lds_p eb, int s, f32x4 v) {      // own 4 neurons of stream s: hi plane, lo plane
        u32x2 ph, pl;
        split4(v, ph, pl);
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(eb + own_slot + s * 1024) = ph;
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(eb + EB1 + own_slot + s * 1024) = pl;
    };
    auto bfrag = [&](lds_p eb, int kc, int s) {
        return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(eb + lane16 + (kc * NST + s) * 1024);
    };
    auto trfrag = [&](lds_p eb, int T, int ph) {
        const lds_p p0 = eb + trl + ph * 2048 + (T >> 1) * NST * 1024 + (T & 1) * 512;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p0 + 64));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    // D[own 16 rows][NST column tiles] += A x B, split operands (hi*hi, hi*lo, lo*hi into the same accumulator).  The B
    // fragments of a k chunk are requested right before its twelve MFMAs (prefetching the next chunk's, 16 more
    // registers, spilled and was 11 % slower; the SIMD's other wave covers the LDS latency); the A fragments (hi and
    // lo) of matrix m are streamed from L2 two k chunks ahead: 16 registers in flight instead of 32 resident, -6 %
    auto loadA1 = [&](int m, int kc, bf16x8& ah, bf16x8& al) {
        ah = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + kc * 1024, m * H * H * 2, 0));
        al = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + kc * 1024, WPLANE_B + m * H * H * 2, 0));
    };
    auto gemmXs = [&](int m, bf16x8 (&ah)[2], bf16x8 (&al)[2], lds_p eb, f32x4 (&d)[NST]) {
        // ah[0..1], al[0..1] hold chunks 0 and 1 on entry (requested by the caller a phase ahead)
        bf16x8 xh[NST], xl[NST];
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
#pragma unroll
            for (int s = 0; s < NST; s++) xh[s] = bfrag(eb, kc, s);
#pragma unroll
            for (int s = 0; s < NST; s++) xl[s] = bfrag(eb + EB1, kc, s);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NST; s++) d[s] = mfma(ah[kc & 1], xh[s], d[s]);
#pragma unroll
            for (int s = 0; s < NST; s++) d[s] = mfma(ah[kc & 1], xl[s], d[s]);
#pragma unroll
            for (int s = 0; s < NST; s++) d[s] = mfma(al[kc & 1], xh[s], d[s]);
            if (kc + 2 < KC) loadA1(m, kc + 2, ah[kc & 1], al[kc & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // weight gradient of one layer from one pass: K = 64 columns = 2 steps (stream pairs) x HT column tiles
    auto wgradX = [&](f32x4 (&wa)[HT], lds_p yb, lds_p xb) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const bf16x8 fyh = trfrag(yb, part, k), fyl = trfrag(yb + EB1, part, k);
#pragma unroll
            for (int i0 = 0; i0 < HT; i0 += 2) {
                bf16x8 fxh[2], fxl[2];
#pragma unroll
                for (int i = 0; i < 2; i++) { fxh[i] = trfrag(xb, i0 + i, k); fxl[i] = trfrag(xb + EB1, i0 + i, k); }
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    wa[i0 + i] = mfma(fyh, fxh[i], wa[i0 + i]);
                    wa[i0 + i] = mfma(fyh, fxl[i], wa[i0 + i]);
                    wa[i0 + i] = mfma(fyl, fxh[i], wa[i0 + i]);
                }
            }
        }
    };
    auto ring_in = [&](int slot, float (&inA)[2][NST]) {
        const float* rp = ring + (slot * 16 + c) * RECF;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
        for (int s = 0; s < NST; s++) { inA[0][s] = r0[s]; inA[1][s] = r1[s]; }
    };
    // layer-0 inputs of pass ps: (f1, f2) streams for pass A, (f2, f1) for pass B
    auto inputs = [&](int slot, int ps, float (&f)[2][NST]) {
        float inA[2][NST];
        ring_in(slot, inA);
#pragma unroll
        for (int s = 0; s < NST; s++) { f[0][s] = ps ? inA[1][s] : inA[0][s]; f[1][s] = ps ? inA[0][s] : inA[1][s]; }
    };

    const lds_p YB = lds0 + C::OFF_Y;
    PointRec nxt = load_rec(0);
    bf16x8 afh[2], afl[2];
    auto loadA2 = [&](int m) { loadA1(m, 0, afh[0], afl[0]); loadA1(m, 1, afh[1], afl[1]); };
    for (int64_t itx = 0; itx < iters; itx++) {
        const int64_t tile = itx * gridDim.x + blockIdx.x;
        const int64_t idx = tile * 16 + c;
        const bool valid = tile < ntile && idx < A.n;
        const int slot = (int)(itx % RB);
        if (slot == 0) {
            if (itx) __syncthreads();
            fill_ring(nxt);
            nxt = load_rec(itx / RB + 1);
            __syncthreads();
        }
        float tNb = 0.f, tlapNb = 0.f;      // adjoints of N and lap N (from the tail), shared by both reverse passes
#pragma unroll 1
        for (int stage = 0; stage < 3; stage++) {
            const int ps = stage == 1 ? 1 : 0;          // stage 0: forward A; 1: forward B, tail, reverse B; 2: forward A, reverse A
            const bool keep = stage > 0;
            // ================= forward pass ps =================
            loadA2(0);
            {
                float f[2][NST];
                inputs(slot, ps, f);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(sW0a + n0), w1 = *reinterpret_cast<const f32x4*>(sW0b + n0);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(sb0 + n0);
                const lds_p X0 = lds0 + C::OFF_X;
                f32x4 a[NST];
#pragma unroll
                for (int s = 0; s < NST; s++) a[s] = lin2(w0, f[0][s], w1, f[1][s]);
                f32x4 s0, s1, s2;
                act012v<ACT>(a[0] + b0, s0, s1, s2);
                const f32x4 q = a[1] * a[1] + a[2] * a[2];
                put(X0, 0, s0);
                put(X0, 1, s1 * a[1]);
                put(X0, 2, s1 * a[2]);
                put(X0, SLP, s1 * a[SLP] + s2 * q);
            }
            __syncthreads();
            f32x4 st3[NST];                 // (h, a_x, a_rho, a_lap) of the last hidden layer of this pass
#pragma unroll
            for (int l = 1; l < L; l++) {
                const lds_p Xin = lds0 + C::OFF_X + (size_t)(l - 1) * EB;
                const lds_p Xout = lds0 + C::OFF_X + (size_t)l * EB;
                f32x4 acc[NST];
                acc[0] = *reinterpret_cast<const f32x4*>(sbl + (l - 1) * H + n0);
#pragma unroll
                for (int s = 1; s < NST; s++) acc[s] = splat(0.f);
                gemmXs((l - 1) * 2, afh, afl, Xin, acc);
                if (l < L - 1) loadA2(l * 2);                             // next layer's forward matrix (first two chunks)
                __builtin_amdgcn_sched_barrier(0);
                f32x4 s0, s1, s2;
                act012v<ACT>(acc[0], s0, s1, s2);
                const f32x4 q = acc[1] * acc[1] + acc[2] * acc[2];
                const f32x4 ol = s1 * acc[SLP] + s2 * q;
                if (l < L - 1) {
                    put(Xout, 0, s0);
                    put(Xout, 1, s1 * acc[1]);
                    put(Xout, 2, s1 * acc[2]);
                    put(Xout, SLP, ol);
                    if (keep) {
                        stash_put(l, 0, s0);
#pragma unroll
                        for (int s = 1; s < NST; s++) stash_put(l, s, acc[s]);
                    }
                } else {
                    const f32x4 wo = *reinterpret_cast<const f32x4*>(swo + n0);
                    st3[0] = s0;
#pragma unroll
                    for (int s = 1; s < NST; s++) st3[s] = acc[s];
                    const float tot = xg_sum_pair(dot4(wo, s0), dot4(wo, ol));
                    if ((g & 1) == 0) SB[((part * 16 + c) * 2 + ps) * 2 + (g >> 1)] = tot;
                }
                __syncthreads();
            }
            if (stage == 0) continue;
            // ================= tail (once, when both passes' output sums are there) =================
            if (stage == 1) {
                float inA[2][NST];
                ring_in(slot, inA);
                const f32x4 r2 = *reinterpret_cast<const f32x4*>(ring + (slot * 16 + c) * RECF + 8);
                const unsigned mkpt = (unsigned)r2[3];
                const float F = inA[0][0] + A.P * inA[1][0];
                const float lapF = inA[0][SLP] + A.P * inA[1][SLP];
                float Sv = 0.f, Sl = 0.f;
#pragma unroll
                for (int pp = 0; pp < C::WAVES; pp++) {
                    Sv += SB[((pp * 16 + c) * 2) * 2] + A.P * SB[((pp * 16 + c) * 2 + 1) * 2];
                    Sl += SB[((pp * 16 + c) * 2) * 2 + 1] + A.P * SB[((pp * 16 + c) * 2 + 1) * 2 + 1];
                }
                Tail<float> t;
                tail<float>(A.out_scale * Sv + bout, A.out_scale * Sl, r2[2], r2[1], F, lapF, r2[0], A.a, A.b, A.c,
                            float(mkpt & 1), float((mkpt >> 1) & 1), A.invN, invN1, invN2, valid ? 1.0f : 0.0f, t);
                if (valid && g == 0 && part == 0) { A.Eb[idx] = t.Eb; A.gb[idx] = t.gb; }
                acc_pde += lead * t.l_pde;
                acc_bc1 += lead * t.l_bc1;
                acc_bc2 += lead * t.l_bc2;
                acc_bout += lead * t.Nb;
                tNb = t.Nb; tlapNb = t.lapNb;
            }
            // ================= adjoints of the last hidden layer of pass ps -> YB =================
            {
                const f32x4 wo = *reinterpret_cast<const f32x4*>(swo + n0);
                const float pw = ps ? A.P : 1.0f;
                const float cv = A.out_scale * pw * tNb, cl = A.out_scale * pw * tlapNb;
                const f32x4 h = st3[0];
                f32x4 s1, s2, s3;
                act123v<ACT>(h, s1, s2, s3);
                const f32x4 q = st3[1] * st3[1] + st3[2] * st3[2];
                woacc += mul_bcast(h, cv) + mul_bcast(s1 * st3[SLP] + s2 * q, cl);
                const f32x4 o0 = mul_bcast(wo, cv), ol = mul_bcast(wo, cl);
                const f32x4 u = 2.0f * s2 * ol;
                const f32x4 y0 = s1 * o0 + (s2 * st3[SLP] + s3 * q) * ol;
                bacc[L - 1] += y0;
                put(YB, 0, y0);
                put(YB, 1, u * st3[1]);
                put(YB, 2, u * st3[2]);
                put(YB, SLP, s1 * ol);
            }
            __syncthreads();
            // ================= reverse steps l = L-1 .. 1 of pass ps =================
#pragma unroll
            for (int l = L - 1; l >= 1; l--) {
                const lds_p Xin = lds0 + C::OFF_X + (size_t)(l - 1) * EB;
                f32x4 sq[NST];
                if (l > 1) {
#pragma unroll
                    for (int s = 0; s < NST; s++) sq[s] = stash_get(l - 1, s);
                }
                // this step's transposed matrix: requested here, behind the weight gradient's 48 MFMAs (not a step
                // earlier: 32 registers less through the elementwise phase, where the pressure peaks)
                loadA2((l - 1) * 2 + 1);
                wgradX(wacc[l - 1], YB, Xin);
                f32x4 ob[NST];
#pragma unroll
                for (int s = 0; s < NST; s++) ob[s] = splat(0.f);
                gemmXs((l - 1) * 2 + 1, afh, afl, YB, ob);
                if (l > 1) {
                    const f32x4 h = sq[0], a1 = sq[1], a2 = sq[2], al_ = sq[3];
                    f32x4 s1, s2, s3;
                    act123v<ACT>(h, s1, s2, s3);
                    const f32x4 q = a1 * a1 + a2 * a2;
                    const f32x4 u = 2.0f * s2 * ob[SLP];
                    const f32x4 y0 = s1 * ob[0] + s2 * (a1 * ob[1] + a2 * ob[2]) + (s2 * al_ + s3 * q) * ob[SLP];
                    const f32x4 y1 = s1 * ob[1] + u * a1, y2 = s1 * ob[2] + u * a2, y3 = s1 * ob[SLP];
                    bacc[l - 1] += y0;
                    __syncthreads();                // every wave has read YB (and X[l-1])
                    put(YB, 0, y0);
                    put(YB, 1, y1);
                    put(YB, 2, y2);
                    put(YB, SLP, y3);
                    __syncthreads();
                } else {
                    float f[2][NST];
                    inputs(slot, ps, f);
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(sW0a + n0), w1 = *reinterpret_cast<const f32x4*>(sW0b + n0);
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sb0 + n0);
                    f32x4 a[NST];
#pragma unroll
                    for (int s = 0; s < NST; s++) a[s] = lin2(w0, f[0][s], w1, f[1][s]);
                    f32x4 h0, s1, s2, s3;
                    act012v<ACT>(a[0] + b0, h0, s1, s2);
                    act123v<ACT>(h0, s1, s2, s3);
                    const f32x4 q = a[1] * a[1] + a[2] * a[2];
                    const f32x4 u = 2.0f * s2 * ob[SLP];
                    f32x4 ab[NST];
                    ab[SLP] = s1 * ob[SLP];
                    ab[1] = s1 * ob[1] + u * a[1];
                    ab[2] = s1 * ob[2] + u * a[2];
                    ab[0] = s1 * ob[0] + s2 * (a[1] * ob[1] + a[2] * ob[2]) + (s2 * a[SLP] + s3 * q) * ob[SLP];
                    bacc[0] += ab[0];
#pragma unroll
                    for (int s = 0; s < NST; s++) { fmac_bcast(w0acc0, ab[s], f[0][s]); fmac_bcast(w0acc1, ab[s], f[1][s]); }
                    __syncthreads();                // X[0] and YB are rewritten by the next forward pass
                }
            }
        }
    }
    // =========================== block reduction ===========================
    __syncthreads();
    constexpr int PT = Lay::size;
    float* out = A.partial + (size_t)blockIdx.x * A.stride;
    const float s0 = row16_sum(xg_sum(acc_pde)), s1r = row16_sum(xg_sum(acc_bc1)),
                s2r = row16_sum(xg_sum(acc_bc2)), s3r = row16_sum(xg_sum(acc_bout));
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int l = 0; l < L; l++) bacc[l][r] = row16_sum(bacc[l][r]);
        w0acc0[r] = row16_sum(w0acc0[r]); w0acc1[r] = row16_sum(w0acc1[r]);
        woacc[r] = row16_sum(woacc[r]);
    }
    float* sc = reinterpret_cast<float*>(lds_raw);
    if (threadIdx.x < 4) sc[threadIdx.x] = 0.f;
    __syncthreads();
#pragma unroll
    for (int l = 1; l < L; l++)
#pragma unroll
        for (int it = 0; it < HT; it++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nj = q_neuron(16 * part + 4 * g + r), ni = q_neuron(16 * it + c);
                out[Lay::W(l) + nj * H + ni] = wacc[l - 1][it][r];
            }
    if (c == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int nj = n0 + r;
#pragma unroll
            for (int l = 0; l < L; l++) out[Lay::b(l) + nj] = bacc[l][r];
            out[Lay::W(0) + 2 * nj] = w0acc0[r];
            out[Lay::W(0) + 2 * nj + 1] = w0acc1[r];
            out[Lay::Wout + nj] = woacc[r];
        }
    }
    for (int w = 0; w < C::WAVES; w++) {
        if (part == w && lane == 0) { sc[0] += s0; sc[1] += s1r; sc[2] += s2r; sc[3] += s3r; }
        __syncthreads();
    }
    if (threadIdx.x < 3) out[PT + threadIdx.x] = sc[threadIdx.x];
    if (threadIdx.x == 3) out[Lay::bout] = sc[3];
}

template <int ACT>
hipError_t launch_trunkDx_t(const float* theta, __bf16* wpk, const TrunkArgs& A, int grid, hipStream_t st)
{
    using C1 = MCfg<128, 4>;
    using C = TDxCfg;
    const int nfrag = C1::NW * 2 * C1::HT * C1::KC * 64;
    hipLaunchKernelGGL((k_pack_weights<128, 4, true>), dim3((nfrag + 255) / 256), dim3(256), 0, st, theta, wpk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    auto kern = k_trunkDx<ACT>;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    uint4* stash = reinterpret_cast<uint4*>(reinterpret_cast<char*>(wpk) + mfma_packed_weight_bytes(128, 4));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::WAVES * 64), C::LDS_BYTES, st, theta, (const __bf16*)wpk, stash, A);
    return hipGetLastError();
}

// =====================================================================================
// R-only nets (E-net 1 -> 32 -> 32 -> 1, gate 1 -> 10 -> 1) in the trunk's transposed layout:
// a wave owns 16 points; lane (g, c) holds, for point c, neurons 16 jt + 4 g + r of both row
// tiles.  The 32 x 32 layer (forward, its transpose in the reverse sweep) is three split-bf16
// MFMAs per row tile (hi*hi + hi*lo + lo*hi, fp32 accumulate: 16 operand mantissa bits, the
// fp32-grade mode of the trunk), with the weight fragments built once per wave straight from
// theta; the weight gradient contracts over the tile's 16 points with v_mfma_f32_16x16x16_bf16 on
// transposed reads of a small per-wave LDS image.  Everything else (first layer, activations,
// output dot products, vector gradients) is lane-local fp32.  Replaces the lane = point kernels
// of pinn_kernels.hip for (HE, HG) = (32, 10), whose f32-MFMA mat-vecs and per-tile LDS staging
// made them 20-55 % of a step (poc/main.py:249-253,262-265).
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct RnetConsts {
    f32x4 w1[2], b1[2], b2[2], w3[2], wg1, bg1, wg2;
    float b3, bg2;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__device__ __forceinline__ void rnet_load_consts(const float* __restrict__ thE, int g, RnetConsts& K)
{
    constexpr RnetLayout Lo{32, 10};
#pragma unroll
    for (int jt = 0; jt < 2; jt++) {
        const int n0 = 16 * jt + 4 * g;
        K.w1[jt] = ld4(thE + Lo.WE1() + n0);
        K.b1[jt] = ld4(thE + Lo.bE1() + n0);
        K.b2[jt] = ld4(thE + Lo.bE2() + n0);
        K.w3[jt] = ld4(thE + Lo.WE3() + n0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {   // gate neurons 4g + r, zero weights beyond the 10 real ones
        const int j = 4 * g + r;
        K.wg1[r] = j < 10 ? thE[Lo.WG1() + j] : 0.f;
        K.bg1[r] = j < 10 ? thE[Lo.bG1() + j] : 0.f;
        K.wg2[r] = j < 10 ? thE[Lo.WG2() + j] : 0.f;
    }
    K.b3 = thE[Lo.bE3()];
    K.bg2 = thE[Lo.bG2()];
}

// hi / lo A fragments of the 32 x 32 layer for row tile t: forward W[16t + c][slot], reverse
// W[slot][16t + c] (slot = slot_neuron(0, g, j): the lane's own B-fragment order)
__device__ __forceinline__ void rnet_weight_frag(const float* __restrict__ W, bool transpose, int t, int g,
                                                 int c, bf16x8& hi, bf16x8& lo)
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int kn = slot_neuron(0, g, j);
        const float w = transpose ? W[kn * 32 + 16 * t + c] : W[(16 * t + c) * 32 + kn];
        hi[j] = (__bf16)w;
        lo[j] = (__bf16)(w - (float)hi[j]);
    }
}

// three pieces (24 mantissa bits): the forward kernel's E(R) is the physics output and must hold
// 1e-5 against the fp64 reference on its own, which two pieces (2^-17 per operand) do not guarantee
__device__ __forceinline__ void rnet_weight_frag3(const float* __restrict__ W, int t, int g, int c,
                                                  bf16x8& hi, bf16x8& mid, bf16x8& lo)
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float w = W[(16 * t + c) * 32 + slot_neuron(0, g, j)];
        hi[j] = (__bf16)w;
        const float r1 = w - (float)hi[j];
        mid[j] = (__bf16)r1;
        lo[j] = (__bf16)(r1 - (float)mid[j]);
    }
}

__device__ __forceinline__ void split4x3(f32x4 v, u32x2& hi, u32x2& mid, u32x2& lo)
{
    hi = pack4(v);
    const f32x4 r1 = v - unpack4(hi);
    mid = pack4(r1);
    lo = pack4(r1 - unpack4(mid));
}

__device__ __forceinline__ f32x4 mfma_x3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 c)
{
    c = mfma(ah, bh, c);
    c = mfma(ah, bl, c);
    return mfma(al, bh, c);
}


// forward of one 16-point tile: returns E, g of point c (all four lane groups hold them)
template <int ACT>
__device__ __forceinline__ void rnet_tile_forward(const RnetConsts& K, const bf16x8 (&Ah)[2], const bf16x8 (&Al)[2],
                                                  float Rv, f32x4 (&h1)[2], f32x4 (&h2)[2], f32x4& hg,
                                                  u32x4& Bh, u32x4& Bl, float& e, float& gg)
{
    u32x2 ph[2], pl[2];
#pragma unroll
    for (int jt = 0; jt < 2; jt++) {
        f32x4 s1, s2;
        act012v<ACT>(K.w1[jt] * Rv + K.b1[jt], h1[jt], s1, s2);
        split4(h1[jt], ph[jt], pl[jt]);
    }
    Bh = u32x4{ph[0][0], ph[0][1], ph[1][0], ph[1][1]};
    Bl = u32x4{pl[0][0], pl[0][1], pl[1][0], pl[1][1]};
    float acc = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; jt++) {
        const f32x4 a2 = mfma_x3(Ah[jt], Al[jt], as_frag(Bh), as_frag(Bl), K.b2[jt]);
        f32x4 s1, s2;
        act012v<ACT>(a2, h2[jt], s1, s2);
        acc += dot4(K.w3[jt], h2[jt]);
    }
    e = xg_sum(acc) + K.b3;
    f32x4 s1, s2;
    act012v<ACT>(K.wg1 * Rv + K.bg1, hg, s1, s2);
    gg = xg_sum(dot4(K.wg2, hg)) + K.bg2;
}

// the same tile forward with three-piece operands: six MFMAs per row tile (all products down to 2^-24)
template <int ACT>
__device__ __forceinline__ void rnet_tile_forward_fine(const RnetConsts& K, const bf16x8 (&Ah)[2],
                                                       const bf16x8 (&Am)[2], const bf16x8 (&Al)[2],
                                                       float Rv, float& e, float& gg)
{
    u32x2 ph[2], pm[2], pl[2];
#pragma unroll
    for (int jt = 0; jt < 2; jt++) {
        f32x4 h1, s1, s2;
        act012v<ACT>(K.w1[jt] * Rv + K.b1[jt], h1, s1, s2);
        split4x3(h1, ph[jt], pm[jt], pl[jt]);
    }
    const bf16x8 Bh = as_frag(u32x4{ph[0][0], ph[0][1], ph[1][0], ph[1][1]});
    const bf16x8 Bm = as_frag(u32x4{pm[0][0], pm[0][1], pm[1][0], pm[1][1]});
    const bf16x8 Bl = as_frag(u32x4{pl[0][0], pl[0][1], pl[1][0], pl[1][1]});
    float acc = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; jt++) {
        f32x4 a2 = mfma(Al[jt], Bh, splat(0.f));     // smallest terms first
        a2 = mfma(Ah[jt], Bl, a2);
        a2 = mfma(Am[jt], Bm, a2);
        a2 = mfma(Am[jt], Bh, a2);
        a2 = mfma(Ah[jt], Bm, a2);
        a2 = mfma(Ah[jt], Bh, a2);
        f32x4 h2, s1, s2;
        act012v<ACT>(a2 + K.b2[jt], h2, s1, s2);
        acc += dot4(K.w3[jt], h2);
    }
    e = xg_sum(acc) + K.b3;
    f32x4 hg, s1, s2;
    act012v<ACT>(K.wg1 * Rv + K.bg1, hg, s1, s2);
    gg = xg_sum(dot4(K.wg2, hg)) + K.bg2;
}

template <int ACT>
__global__ __launch_bounds__(kBlock) void k_rnet_mfma_fwd(const float* __restrict__ thE,
                                                          const float* __restrict__ R, int64_t n,
                                                          float* __restrict__ E, float* __restrict__ gout,
                                                          float* __restrict__ Eu)
{
    constexpr RnetLayout Lo{32, 10};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    RnetConsts K;
    rnet_load_consts(thE, g, K);
    bf16x8 Ah[2], Am[2], Al[2];
#pragma unroll
    for (int jt = 0; jt < 2; jt++) rnet_weight_frag3(thE + Lo.WE2(), jt, g, c, Ah[jt], Am[jt], Al[jt]);
    const int64_t ntile = (n + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < ntile;
         tile += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t idx = tile * 16 + c;
        const bool valid = idx < n;
        const float Rv = R[valid ? idx : n - 1];
        float e, gg;
        rnet_tile_forward_fine<ACT>(K, Ah, Am, Al, Rv, e, gg);
        if (valid && g == 0) {
            E[idx] = e;
            gout[idx] = gg;
            if (Eu) Eu[idx] = e;
        }
    }
}

template <int ACT>
__global__ __launch_bounds__(kBlock, 2) void k_rnet_mfma_bwd(const float* __restrict__ thE,
                                                             const float* __restrict__ R,
                                                             const float* __restrict__ Eb,
                                                             const float* __restrict__ gb, int64_t n,
                                                             float* __restrict__ partial, int stride)
{
    constexpr RnetLayout Lo{32, 10};
    constexpr int PR = Lo.size();
    constexpr int PITCH = 40;                 // bf16 per image row: 32 neurons + 8 (skews the banks)
    constexpr int IMG = 16 * PITCH;           // one image: 16 rows (the tile's points)
    __shared__ __attribute__((aligned(16))) __bf16 imgs[kWavesPerBlock * 4 * IMG];
    __shared__ float gimg[PR + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    __bf16* Yh = imgs + (size_t)wave * 4 * IMG;
    __bf16* Yl = Yh + IMG;
    __bf16* Xh = Yl + IMG;
    __bf16* Xl = Xh + IMG;
    RnetConsts K;
    rnet_load_consts(thE, g, K);
    bf16x8 Ah[2], Al[2], Th[2], Tl[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        rnet_weight_frag(thE + Lo.WE2(), false, t, g, c, Ah[t], Al[t]);
        rnet_weight_frag(thE + Lo.WE2(), true, t, g, c, Th[t], Tl[t]);
    }
    f32x4 wacc[2][2], bE2[2], wE3[2], bE1[2], wE1[2], wg2a = splat(0.f), bg1a = splat(0.f), wg1a = splat(0.f);
    float bE3a = 0.f, bG2a = 0.f;
#pragma unroll
    for (int a = 0; a < 2; a++) {
        bE2[a] = splat(0.f); wE3[a] = splat(0.f); bE1[a] = splat(0.f); wE1[a] = splat(0.f);
        wacc[a][0] = splat(0.f); wacc[a][1] = splat(0.f);
    }
    const float lead = g == 0 ? 1.0f : 0.0f;
    const int64_t ntile = (n + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < ntile;
         tile += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t idx = tile * 16 + c;
        const bool valid = idx < n;
        const int64_t id = valid ? idx : n - 1;
        const float Rv = R[id];
        const float ebv = valid ? Eb[id] : 0.f, gbv = valid ? gb[id] : 0.f;
        f32x4 h1[2], h2[2], hg;
        u32x4 Xbh, Xbl;
        float e, gg;
        rnet_tile_forward<ACT>(K, Ah, Al, Rv, h1, h2, hg, Xbh, Xbl, e, gg);
        // ---- E-net reverse: output layer, then the adjoint of the second pre-activation
        f32x4 d2[2];
        u32x2 dh[2], dl[2];
#pragma unroll
        for (int jt = 0; jt < 2; jt++) {
            f32x4 s1, s2, s3;
            act123v<ACT>(h2[jt], s1, s2, s3);
            wE3[jt] += ebv * h2[jt];
            d2[jt] = (ebv * K.w3[jt]) * s1;
            bE2[jt] += d2[jt];
            split4(d2[jt], dh[jt], dl[jt]);
        }
        bE3a += lead * ebv;
        const u32x4 Dh = {dh[0][0], dh[0][1], dh[1][0], dh[1][1]}, Dl = {dl[0][0], dl[0][1], dl[1][0], dl[1][1]};
        // ---- WE2-bar += d2 (x) h1 over the tile's 16 points: rows of the [point][neuron] images
        *reinterpret_cast<u32x4*>(Yh + c * PITCH + 8 * g) = Dh;
        *reinterpret_cast<u32x4*>(Yl + c * PITCH + 8 * g) = Dl;
        *reinterpret_cast<u32x4*>(Xh + c * PITCH + 8 * g) = Xbh;
        *reinterpret_cast<u32x4*>(Xl + c * PITCH + 8 * g) = Xbl;
        wave_fence();
        {
            auto frag = [&](const __bf16* img, int t) {
                const __bf16* p = img + (4 * g + (c >> 2)) * PITCH + 16 * t + 4 * (c & 3);
                const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
                return __builtin_bit_cast(s16x4, v);
            };
            s16x4 fyh[2], fyl[2], fxh[2], fxl[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                fyh[t] = frag(Yh, t); fyl[t] = frag(Yl, t); fxh[t] = frag(Xh, t); fxl[t] = frag(Xl, t);
            }
#pragma unroll
            for (int jt = 0; jt < 2; jt++)
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    f32x4 a = wacc[jt][it];
                    a = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fyh[jt], fxh[it], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fyh[jt], fxl[it], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fyl[jt], fxh[it], a, 0, 0, 0);
                    wacc[jt][it] = a;
                }
        }
        wave_fence();
        // ---- adjoint of h1 through W2^T, then the first layer
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const f32x4 ob = mfma_x3(Th[it], Tl[it], as_frag(Dh), as_frag(Dl), splat(0.f));
            f32x4 s1, s2, s3;
            act123v<ACT>(h1[it], s1, s2, s3);
            const f32x4 a1b = ob * s1;
            bE1[it] += a1b;
            wE1[it] += a1b * Rv;
        }
        // ---- gate
        {
            f32x4 s1, s2, s3;
            act123v<ACT>(hg, s1, s2, s3);
            wg2a += gbv * hg;
            bG2a += lead * gbv;
            const f32x4 agb = (gbv * K.wg2) * s1;
            bg1a += agb;
            wg1a += agb * Rv;
        }
    }
    // ---- block reduction: lane-local sums over the 16 point columns, waves in fixed order
    const float sE3 = row16_sum(xg_sum(bE3a)), sG2 = row16_sum(xg_sum(bG2a));
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            bE2[t][r] = row16_sum(bE2[t][r]); wE3[t][r] = row16_sum(wE3[t][r]);
            bE1[t][r] = row16_sum(bE1[t][r]); wE1[t][r] = row16_sum(wE1[t][r]);
        }
        wg2a[r] = row16_sum(wg2a[r]); bg1a[r] = row16_sum(bg1a[r]); wg1a[r] = row16_sum(wg1a[r]);
    }
    __syncthreads();
    for (int w = 0; w < kWavesPerBlock; w++) {
        if (wave == w) {
            const bool add = w != 0;
            auto put = [&](int k, float v) { gimg[k] = add ? gimg[k] + v : v; };
#pragma unroll
            for (int jt = 0; jt < 2; jt++)
#pragma unroll
                for (int it = 0; it < 2; it++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        put(Lo.WE2() + q_neuron(16 * jt + 4 * g + r) * 32 + q_neuron(16 * it + c), wacc[jt][it][r]);
            if (c == 0) {
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int j = 16 * t + 4 * g + r;
                        put(Lo.bE2() + j, bE2[t][r]); put(Lo.WE3() + j, wE3[t][r]);
                        put(Lo.bE1() + j, bE1[t][r]); put(Lo.WE1() + j, wE1[t][r]);
                    }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int j = 4 * g + r;
                    if (j < 10) { put(Lo.WG2() + j, wg2a[r]); put(Lo.bG1() + j, bg1a[r]); put(Lo.WG1() + j, wg1a[r]); }
                }
            }
            if (lane == 0) { put(Lo.bE3(), sE3); put(Lo.bG2(), sG2); }
        }
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * stride;
    for (int k = threadIdx.x; k < PR; k += kBlock) out[k] = gimg[k];
}


// =====================================================================================
// The reference's own trunk (H = 16, two hidden layers) on the matrix pipe: k_trunk16.
// One wave owns a 16-point tile and all 16 neurons; lane (g, c) holds neurons 4g + r of point c,
// five streams of them, so the whole per-point algebra is lane-local and the 16 x 16 layer is
// v_mfma_f32_16x16x16_bf16: the lane's 4 values ARE its B-operand k-slots (no permutation at
// K = 16) and nothing leaves registers between layers.  Operands are split bf16 with fp32
// accumulation: THREE pieces (24 mantissa bits, six MFMAs per product) in the forward sweep --
// this is the 1e-5 parity architecture -- and two pieces (three MFMAs) in the reverse sweep's
// adjoint GEMM and weight gradient (gradient bar 1e-4).  The weight gradient contracts over the
// 80 (stream, point) columns with transposed reads of a per-wave LDS image, as everywhere in this
// file.  Both passes of the symmetric double pass keep their forward state (no recompute).
// Against k_trunk of pinn_kernels.hip (lane = point, mat-vecs as packed VALU rank-1 updates): the
// 16 x 16 mat-vecs were 77 % of that kernel's VALU instructions.
__device__ __forceinline__ s16x4 as_s4(u32x2 v) { return __builtin_bit_cast(s16x4, v); }

__device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

template <int ACT, int NPASS, int MODE>
__global__ __launch_bounds__(kBlock, MODE == MODE_BWD ? 2 : 4) void k_trunk16(const float* __restrict__ theta, TrunkArgs A)
{
    using Lay = TrunkLayout<16, 2>;
    constexpr int PT = Lay::size;
    constexpr int PITCH = 24;                 // bf16 per image row: 16 neurons + 8
    constexpr int IMG = NST * 16 * PITCH;     // one image: NST streams x 16 points
    __shared__ __attribute__((aligned(16))) __bf16 imgs[(MODE == MODE_BWD ? kWavesPerBlock * 4 * IMG : 8)];
    __shared__ float gimg[PT + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    __bf16* Yh = imgs + (MODE == MODE_BWD ? (size_t)wave * 4 * IMG : 0);
    __bf16* Yl = Yh + IMG;
    __bf16* Xh = Yl + IMG;
    __bf16* Xl = Xh + IMG;

    // ---- per-lane constants: the own neurons' first-layer rows, biases, output weights
    // (the symmetric reverse kernel sits at its 256-register cap: there the biases and output weights are
    // re-read from a 192-byte LDS table where they are used instead of occupying 12 registers throughout)
    constexpr bool CLDS = MODE == MODE_BWD && NPASS == 2;
    __shared__ __attribute__((aligned(16))) float sconst[CLDS ? 48 : 4];
    f32x4 w0a, w0b, b0r, b1r, wor;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int nj = 4 * g + r;
        w0a[r] = theta[Lay::W(0) + 2 * nj];
        w0b[r] = theta[Lay::W(0) + 2 * nj + 1];
        b0r[r] = theta[Lay::b(0) + nj];
        b1r[r] = theta[Lay::b(1) + nj];
        wor[r] = theta[Lay::Wout + nj];
    }
    if (CLDS) {
        if (threadIdx.x < 16) {
            sconst[threadIdx.x] = theta[Lay::b(0) + threadIdx.x];
            sconst[16 + threadIdx.x] = theta[Lay::b(1) + threadIdx.x];
            sconst[32 + threadIdx.x] = theta[Lay::Wout + threadIdx.x];
        }
        __syncthreads();
    }
#define T16_B0 (CLDS ? *reinterpret_cast<const f32x4*>(sconst + 4 * g) : b0r)
#define T16_B1 (CLDS ? *reinterpret_cast<const f32x4*>(sconst + 16 + 4 * g) : b1r)
#define T16_WO (CLDS ? *reinterpret_cast<const f32x4*>(sconst + 32 + 4 * g) : wor)
    const float bout = theta[Lay::bout];
    // A fragments of the 16 x 16 layer: forward W1[c][4g + j] in three pieces, reverse W1[4g + j][c] in two
    s16x4 Fh, Fm, Fl, Th, Tl;
    {
        u32x2 fh, fm, fl, th, tl;
        f32x4 wf, wt;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            wf[j] = theta[Lay::W(1) + c * 16 + 4 * g + j];
            wt[j] = theta[Lay::W(1) + (4 * g + j) * 16 + c];
        }
        split4x3(wf, fh, fm, fl);
        split4(wt, th, tl);
        Fh = as_s4(fh); Fm = as_s4(fm); Fl = as_s4(fl); Th = as_s4(th); Tl = as_s4(tl);
    }

    f32x4 wacc = splat(0.f), bacc0 = splat(0.f), bacc1 = splat(0.f), w0acc0 = splat(0.f), w0acc1 = splat(0.f),
          woacc = splat(0.f);
    float acc_bout = 0.f, acc_pde = 0.f, acc_bc1 = 0.f, acc_bc2 = 0.f;
    float invN1 = A.invN1, invN2 = A.invN2;
    if (MODE != MODE_FWD && A.bc_counts) {
        const long long c1 = A.bc_counts[0], c2 = A.bc_counts[1];
        invN1 = c1 > 0 ? 1.0f / (float)c1 : 0.0f;
        invN2 = c2 > 0 ? 1.0f / (float)c2 : 0.0f;
    }
    const float lead = g == 0 ? 1.0f : 0.0f;
    const int64_t ntile = (A.n + 15) / 16;

    // the point record of the next tile is requested one trip ahead (nothing waits for it until then)
    struct PointRec { float x, y, z, R, E, g; unsigned mk; };
    const int64_t tstep = (int64_t)gridDim.x * kWavesPerBlock;
    auto load_rec = [&](int64_t tl) {
        int64_t id = tl * 16 + c;
        id = (tl < ntile && id < A.n) ? id : A.n - 1;
        PointRec r;
        r.x = A.x[id]; r.y = A.y[id]; r.z = A.z[id]; r.R = A.R[id]; r.E = A.E[id]; r.g = A.g[id];
        r.mk = (MODE != MODE_FWD && A.mask) ? A.mask[id] : 0u;
        return r;
    };
    PointRec nxt = load_rec((int64_t)blockIdx.x * kWavesPerBlock + wave);
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < ntile; tile += tstep) {
        const int64_t idx = tile * 16 + c;
        const bool valid = idx < A.n;
        const PointRec cur = nxt;
        if (tile + tstep < ntile) nxt = load_rec(tile + tstep);
        const float xv = cur.x, yv = cur.y, zv = cur.z, Rv = cur.R;
        const float Ev = cur.E, gv = cur.g;
        const float m1 = float(cur.mk & 1), m2 = float((cur.mk >> 1) & 1);
        float inA[2][NST], ri1, ri2;
        {
            const float rho2 = yv * yv + zv * zv, rho = __builtin_amdgcn_sqrtf(rho2);
            orbital_cyl<float>(xv - Rv, rho2, rho, inA[0], ri1);
            orbital_cyl<float>(xv + Rv, rho2, rho, inA[1], ri2);
        }
        const float F = inA[0][0] + A.P * inA[1][0];
        const float lapF = inA[0][SLP] + A.P * inA[1][SLP];

        // saved forward state per pass: the layer-0 output streams as their two leading bf16 pieces
        // (exactly the X operand of the weight gradient), the layer-0 activation in fp32, and
        // (h, a_x, a_rho, a_lap) of layer 1
        u32x2 xh[NPASS][NST], xl[NPASS][NST];
        f32x4 h0s[NPASS], sv[NPASS][NST];
        float SvT = 0.f, SlT = 0.f;
#pragma unroll
        for (int ps = 0; ps < NPASS; ps++) {
            const bool useB = ps == 1;
            float in[2][NST];
#pragma unroll
            for (int s = 0; s < NST; s++) {
                in[0][s] = useB ? inA[1][s] : inA[0][s];
                in[1][s] = useB ? inA[0][s] : inA[1][s];
            }
            f32x4 a0[NST], ox[NST];
#pragma unroll
            for (int s = 0; s < NST; s++) a0[s] = lin2_t(w0a, in[0][s], w0b, in[1][s]);
            a0[0] += T16_B0;
            {
                f32x4 s0, s1, s2;
                act012v<ACT>(a0[0], s0, s1, s2);
                const f32x4 q = a0[1] * a0[1] + a0[2] * a0[2];
                ox[0] = s0;
                ox[1] = s1 * a0[1];
                ox[2] = s1 * a0[2];
                ox[SLP] = s1 * a0[SLP] + s2 * q;
                h0s[ps] = s0;
            }
            f32x4 a1[NST];
#pragma unroll
            for (int s = 0; s < NST; s++) {
                u32x2 ph, pm, pl;
                split4x3(ox[s], ph, pm, pl);
                xh[ps][s] = ph;
                xl[ps][s] = pm;
                f32x4 acc = mfma16(Fl, as_s4(ph), splat(0.f));       // smallest terms first
                acc = mfma16(Fh, as_s4(pl), acc);
                acc = mfma16(Fm, as_s4(pm), acc);
                acc = mfma16(Fm, as_s4(ph), acc);
                acc = mfma16(Fh, as_s4(pm), acc);
                a1[s] = mfma16(Fh, as_s4(ph), acc);
            }
            a1[0] += T16_B1;
            f32x4 s0, s1, s2;
            act012v<ACT>(a1[0], s0, s1, s2);
            const f32x4 q = a1[1] * a1[1] + a1[2] * a1[2];
            const f32x4 ol = s1 * a1[SLP] + s2 * q;
            sv[ps][0] = s0;
#pragma unroll
            for (int s = 1; s < NST; s++) sv[ps][s] = a1[s];
            const f32x4 wo = T16_WO;
            SvT += (useB ? A.P : 1.0f) * dot4(wo, s0);    // lane-local partial sums of both passes ...
            SlT += (useB ? A.P : 1.0f) * dot4(wo, ol);
        }
        xg_sum_both(SvT, SlT);                            // ... summed over the point's 4 lane groups once
        Tail<float> t;
        tail<float>(A.out_scale * SvT + bout, A.out_scale * SlT, gv, Ev, F, lapF, ri1 + ri2, A.a, A.b, A.c, m1,
                    m2, A.invN, invN1, invN2, valid ? 1.0f : 0.0f, t);
        if (MODE == MODE_FWD) {
            if (valid && g == 0) {
                if (A.psi) A.psi[idx] = t.psi;
                if (A.lap) A.lap[idx] = t.lap;
                if (A.res) A.res[idx] = t.res;
            }
            continue;
        }
        if (valid && g == 0) { A.Eb[idx] = t.Eb; A.gb[idx] = t.gb; }
        acc_pde += lead * t.l_pde;
        acc_bc1 += lead * t.l_bc1;
        acc_bc2 += lead * t.l_bc2;
        acc_bout += lead * t.Nb;
        if (MODE != MODE_BWD) continue;
        // =========================== reverse ===========================
#pragma unroll
        for (int ps = NPASS - 1; ps >= 0; ps--) {
            const bool useB = ps == 1;
            float in[2][NST];
#pragma unroll
            for (int s = 0; s < NST; s++) {
                in[0][s] = useB ? inA[1][s] : inA[0][s];
                in[1][s] = useB ? inA[0][s] : inA[1][s];
            }
            const float pw = useB ? A.P : 1.0f;
            const float cv = A.out_scale * pw * t.Nb, cl = A.out_scale * pw * t.lapNb;
            f32x4 yo[NST];
            {
                const f32x4 h = sv[ps][0], a1x = sv[ps][1], a1r = sv[ps][2], a1l = sv[ps][SLP];
                f32x4 s1, s2, s3;
                act123v<ACT>(h, s1, s2, s3);
                const f32x4 q = a1x * a1x + a1r * a1r;
                woacc += cv * h + cl * (s1 * a1l + s2 * q);
                const f32x4 wo = T16_WO;
                const f32x4 o0 = cv * wo, olb = cl * wo;
                const f32x4 u = 2.0f * s2 * olb;
                yo[SLP] = s1 * olb;
                yo[1] = u * a1x;
                yo[2] = u * a1r;
                yo[0] = s1 * o0 + (s2 * a1l + s3 * q) * olb;
                bacc1 += yo[0];
            }
            // ---- W1-bar += Y-bar (x) X over the 64 (stream, point) columns
            u32x2 yh[NST], yl[NST];
#pragma unroll
            for (int s = 0; s < NST; s++) {
                split4(yo[s], yh[s], yl[s]);
                const int off = (s * 16 + c) * PITCH + 4 * g;
                *reinterpret_cast<u32x2*>(Yh + off) = yh[s];
                *reinterpret_cast<u32x2*>(Yl + off) = yl[s];
                *reinterpret_cast<u32x2*>(Xh + off) = xh[ps][s];
                *reinterpret_cast<u32x2*>(Xl + off) = xl[ps][s];
            }
            wave_fence();
#pragma unroll
            for (int s = 0; s < NST; s++) {
                auto frag = [&](const __bf16* img) {
                    const __bf16* p = img + (s * 16 + 4 * g + (c >> 2)) * PITCH + 4 * (c & 3);
                    return __builtin_bit_cast(s16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                                         (__attribute__((address_space(3))) bf16x4*)p));
                };
                const s16x4 fyh = frag(Yh), fyl = frag(Yl), fxh = frag(Xh), fxl = frag(Xl);
                wacc = mfma16(fyh, fxh, wacc);
                wacc = mfma16(fyh, fxl, wacc);
                wacc = mfma16(fyl, fxh, wacc);
            }
            wave_fence();
            // ---- adjoint of the layer-0 outputs through W1^T, then layer 0 itself
            f32x4 ob[NST];
#pragma unroll
            for (int s = 0; s < NST; s++) {
                f32x4 acc = mfma16(Th, as_s4(yh[s]), splat(0.f));
                acc = mfma16(Th, as_s4(yl[s]), acc);
                ob[s] = mfma16(Tl, as_s4(yh[s]), acc);
            }
            {
                f32x4 a0[NST];
#pragma unroll
                for (int s = 1; s < NST; s++) a0[s] = lin2_t(w0a, in[0][s], w0b, in[1][s]);
                const f32x4 h = h0s[ps];
                f32x4 s1, s2, s3;
                act123v<ACT>(h, s1, s2, s3);
                const f32x4 q = a0[1] * a0[1] + a0[2] * a0[2];
                const f32x4 u = 2.0f * s2 * ob[SLP];
                f32x4 ab[NST];
                ab[SLP] = s1 * ob[SLP];
                ab[1] = s1 * ob[1] + u * a0[1];
                ab[2] = s1 * ob[2] + u * a0[2];
                ab[0] = s1 * ob[0] + s2 * (a0[1] * ob[1] + a0[2] * ob[2]) +
                        (s2 * a0[SLP] + s3 * q) * ob[SLP];
                bacc0 += ab[0];
#pragma unroll
                for (int s = 0; s < NST; s++) { fmac_bcast_t(w0acc0, ab[s], in[0][s]); fmac_bcast_t(w0acc1, ab[s], in[1][s]); }
            }
        }
    }

    if (MODE == MODE_FWD) return;
    // =========================== block reduction ===========================
    const float s0 = row16_sum(xg_sum(acc_pde)), s1r = row16_sum(xg_sum(acc_bc1)),
                s2r = row16_sum(xg_sum(acc_bc2)), s3r = row16_sum(xg_sum(acc_bout));
    if (MODE == MODE_BWD) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            bacc0[r] = row16_sum(bacc0[r]); bacc1[r] = row16_sum(bacc1[r]);
            w0acc0[r] = row16_sum(w0acc0[r]); w0acc1[r] = row16_sum(w0acc1[r]);
            woacc[r] = row16_sum(woacc[r]);
        }
    }
    float* sc = gimg + PT;
    __syncthreads();
    for (int w = 0; w < kWavesPerBlock; w++) {
        if (wave == w) {
            const bool add = w != 0;
            auto put = [&](float* p, float v) { *p = add ? *p + v : v; };
            if (MODE == MODE_BWD) {
#pragma unroll
                for (int r = 0; r < 4; r++) put(gimg + Lay::W(1) + (4 * g + r) * 16 + c, wacc[r]);
                if (c == 0) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int nj = 4 * g + r;
                        put(gimg + Lay::b(0) + nj, bacc0[r]);
                        put(gimg + Lay::b(1) + nj, bacc1[r]);
                        put(gimg + Lay::W(0) + 2 * nj, w0acc0[r]);
                        put(gimg + Lay::W(0) + 2 * nj + 1, w0acc1[r]);
                        put(gimg + Lay::Wout + nj, woacc[r]);
                    }
                }
            }
            if (lane == 0) { put(sc + 0, s0); put(sc + 1, s1r); put(sc + 2, s2r); put(sc + 3, s3r); }
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
#undef T16_B0
#undef T16_B1
#undef T16_WO

template <int ACT, int NPASS>
hipError_t launch_trunk16_t(int mode, const float* theta, const TrunkArgs& A, int grid, hipStream_t st)
{
    if (mode == MODE_FWD)
        hipLaunchKernelGGL((k_trunk16<ACT, NPASS, MODE_FWD>), dim3(grid), dim3(kBlock), 0, st, theta, A);
    else if (mode == MODE_BWD)
        hipLaunchKernelGGL((k_trunk16<ACT, NPASS, MODE_BWD>), dim3(grid), dim3(kBlock), 0, st, theta, A);
    else
        hipLaunchKernelGGL((k_trunk16<ACT, NPASS, MODE_HEAD>), dim3(grid), dim3(kBlock), 0, st, theta, A);
    return hipGetLastError();
}

}  // namespace

#if PINN_TRUNK_PART == 0

// ---- R-only nets on the matrix pipe (see k_rnet_mfma_fwd / k_rnet_mfma_bwd)
bool rnet_mfma_supported(int HE, int HG)
{
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("PINN_RNET_MFMA");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on && HE == 32 && HG == 10;
}

hipError_t launch_rnet_mfma_fwd(int act, const float* thE, const float* R, int64_t n, float* E, float* g,
                                float* Eu, int grid, hipStream_t st)
{
    if (act == ACT_SIGMOID)
        hipLaunchKernelGGL(k_rnet_mfma_fwd<ACT_SIGMOID>, dim3(grid), dim3(kBlock), 0, st, thE, R, n, E, g, Eu);
    else
        hipLaunchKernelGGL(k_rnet_mfma_fwd<ACT_TANH>, dim3(grid), dim3(kBlock), 0, st, thE, R, n, E, g, Eu);
    return hipGetLastError();
}

hipError_t launch_rnet_mfma_bwd(int act, const float* thE, const float* R, const float* Eb, const float* gb,
                                int64_t n, float* part, int stride, int grid, hipStream_t st)
{
    if (act == ACT_SIGMOID)
        hipLaunchKernelGGL(k_rnet_mfma_bwd<ACT_SIGMOID>, dim3(grid), dim3(kBlock), 0, st, thE, R, Eb, gb, n, part, stride);
    else
        hipLaunchKernelGGL(k_rnet_mfma_bwd<ACT_TANH>, dim3(grid), dim3(kBlock), 0, st, thE, R, Eb, gb, n, part, stride);
    return hipGetLastError();
}

// ---- the 16-wide reference trunk on the matrix pipe (see k_trunk16)
bool trunk16_mfma_supported(int H, int L)
{
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("PINN_TRUNK16_MFMA");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on && H == 16 && L == 2;
}

hipError_t launch_trunk16_mfma(int act, int npass, int mode, const float* theta, const TrunkArgs& A, int grid,
                               hipStream_t st)
{
    if (act == ACT_SIGMOID)
        return npass == 2 ? launch_trunk16_t<ACT_SIGMOID, 2>(mode, theta, A, grid, st)
                          : launch_trunk16_t<ACT_SIGMOID, 1>(mode, theta, A, grid, st);
    return npass == 2 ? launch_trunk16_t<ACT_TANH, 2>(mode, theta, A, grid, st)
                      : launch_trunk16_t<ACT_TANH, 1>(mode, theta, A, grid, st);
}

bool mfma_trunk_supported(int H, int L) { return (H == 64 || H == 128) && (L == 2 || L == 4); }
bool mfma_x3_supported(int H, int L) { return (H == 64 || H == 128) && (L == 2 || L == 4); }

// resident blocks per CU the chosen decomposition is built for (pinn_create sizes the persistent
// grid and the per-block partial buffer with it)
// measurement switch (tools/kbench.py): PINN_MFMA_VARIANT = 0 (default), 3 (the per-pass kernel k_trunk_mfma2 where a
// merged-pass kernel is the default) or 11.. (variants of k_trunkD).  Read ONCE per process (round 4; it was a getenv per
// launch) -- except under PINN_MFMA_VARIANT_LIVE=1, which tools/kbench.py and the tests set to switch variants inside
// one process.
static int mfma_variant()
{
    auto read = [] {
        const char* e = getenv("PINN_MFMA_VARIANT");
        return (e && e[0] >= '0' && e[0] <= '9') ? atoi(e) : 0;
    };
    static const bool live = [] { const char* e = getenv("PINN_MFMA_VARIANT_LIVE"); return e && e[0] == '1'; }();
    static const int once = read();
    return live ? read() : once;
}

int mfma_blocks_per_cu(int H, int L, bool x3)
{
    if (H == 64 && (L == 2 || L == 4) && !x3) return 2;
    return 1;
}

// bytes behind the packed weights for k_trunkD's parked saved state (nblocks = the persistent grid)
size_t mfma_stash_bytes(int H, int L, bool x3, int nblocks)
{
    if (L != 4) return 0;
    if (x3) return H == 128 ? TDxCfg::STASH_BLOCK * (size_t)nblocks : 0;
    return (H == 128 ? TDCfg<128, 4>::STASH_BLOCK : H == 64 ? TDCfg<64, 4>::STASH_BLOCK : 0) * (size_t)nblocks;
}

size_t mfma_packed_weight_bytes(int H, int L) { return (size_t)2 * (L - 1) * 2 * H * H * 2; }  // hi + lo planes

int mfma_partial_stride(int H, int L) { return 3 * H + (L - 1) * (H * H + H) + H + 1 + 3; }

bool launch_trunk_mfma(int H, int L, int act, int npass, int mode, bool x3, const float* theta,
                       void* packed_weights, const TrunkArgs& A, int grid, hipStream_t st,
                       hipError_t* err)
{
    __bf16* wpk = reinterpret_cast<__bf16*>(packed_weights);
    if (H == 128 && L == 4)   // compiled in the second translation unit of this file (PINN_TRUNK_PART == 1)
        return launch_trunk_mfma_128x4(act, npass, mode, x3, mfma_variant(), theta, packed_weights, A, grid, st, err);
    // <H, L, SPLIT, WAVES, X3, OCC>: SPLIT waves share a 16-point tile, WAVES per block, OCC resident
    // blocks per CU.  One decomposition per shape, the fastest measured (round 1, 1M points, kernel ms):
    //   2x64 : <4,4> x 2 blocks/CU 0.72 (<4,8> 0.74, <2,8> spills)     2x128: <8,8> 1.88 (<4,4> 2.7)
    //   4x64 : <2,4> 2.27 (<4,8> 4.2)                                   4x128: <4,4> 11.8 (<8,8> 15.4)
    // (round 2, four streams: 2x64 0.57, 3 blocks/CU 0.69, <4,8> 0.58; 4x128 <4,4> 7.6, <8,8> 8.3)
    if (x3) {   // split-bf16 operands (3 MFMAs per product)
        if (H == 64 && L == 2 && npass == 2 && mode == MODE_BWD && mfma_variant() != 3) {
            // the benchmark's shape in fp32-grade mode: both passes swept together (k_trunk2x)
            *err = act == ACT_TANH ? launch_trunk2x_t<ACT_TANH>(theta, wpk, A, grid, st)
                                   : launch_trunk2x_t<ACT_SIGMOID>(theta, wpk, A, grid, st);
            return true;
        }
        if (H == 64 && L == 2) {
            return launch2_HL<64, 2, 4, 8, true>(act, npass, mode, theta, wpk, A, grid, st, err);
        }
        if (H == 64 && L == 4)   // deep, fp32-grade: one group per block, hi plane in LDS, lo plane from L2
            return launch2_HL<64, 4, 4, 4, true, 0>(act, npass, mode, theta, wpk, A, grid, st, err);
        if (H == 128 && L == 2) {
            // hi weight plane in LDS (64 KB), lo plane from L2, one pass resident: 4.3 ms
            return launch2_HL<128, 2, 8, 8, true, 0>(act, npass, mode, theta, wpk, A, grid, st, err);
        }
        return false;
    }
    if (L == 2 && npass == 1 && mode == MODE_BWD && mfma_variant() != 3 && (H == 64 || H == 128)) {
        // train.py's formulation (one pass per point): k_trunk2's schedule with two tiles per trip (k_trunk2s)
        if (H == 64) *err = act == ACT_TANH ? launch_trunk2s_t<64, ACT_TANH>(theta, wpk, A, grid, st)
                                            : launch_trunk2s_t<64, ACT_SIGMOID>(theta, wpk, A, grid, st);
        else *err = act == ACT_TANH ? launch_trunk2s_t<128, ACT_TANH>(theta, wpk, A, grid, st)
                                    : launch_trunk2s_t<128, ACT_SIGMOID>(theta, wpk, A, grid, st);
        return true;
    }
    if (L == 2 && npass == 2 && mode == MODE_BWD && mfma_variant() == 12 && act == ACT_TANH && (H == 64 || H == 128)) {
        // measurement variant (tools/kbench.py): k_trunk2 with round 3's lane-linear exchange slots
        *err = H == 64 ? launch_trunk2_t<64, ACT_TANH, false>(theta, wpk, A, grid, st)
                       : launch_trunk2_t<128, ACT_TANH, false>(theta, wpk, A, grid, st);
        return true;
    }
    if (L == 2 && npass == 2 && mode == MODE_BWD && mfma_variant() != 3 && (H == 64 || H == 128)) {
        // the benchmark's shape: both passes swept together (k_trunk2)
        if (H == 64) *err = act == ACT_TANH ? launch_trunk2_t<64, ACT_TANH>(theta, wpk, A, grid, st)
                                            : launch_trunk2_t<64, ACT_SIGMOID>(theta, wpk, A, grid, st);
        else *err = act == ACT_TANH ? launch_trunk2_t<128, ACT_TANH>(theta, wpk, A, grid, st)
                                    : launch_trunk2_t<128, ACT_SIGMOID>(theta, wpk, A, grid, st);
        return true;
    }
    if (L == 4 && npass == 2 && mode == MODE_BWD && mfma_variant() != 3 && H == 64) {
        // deep trunks: both passes swept together, weights from L2 into registers (k_trunkD)
        *err = act == ACT_TANH ? launch_trunkD_t<64, 4, ACT_TANH>(theta, wpk, A, grid, st)
                               : launch_trunkD_t<64, 4, ACT_SIGMOID>(theta, wpk, A, grid, st);
        return true;
    }
    if (H == 64 && L == 2) return launch2_HL<64, 2, 4, 4, false, 2>(act, npass, mode, theta, wpk, A, grid, st, err);
    if (H == 128 && L == 2) return launch2_HL<128, 2, 8, 8>(act, npass, mode, theta, wpk, A, grid, st, err);
    if (H == 64 && L == 4) return launch2_HL<64, 4, 2, 4>(act, npass, mode, theta, wpk, A, grid, st, err);
    return false;
}

#else   // ---- PINN_TRUNK_PART == 1: the 4x128 kernels, built WITHOUT the SLP vectorizer
// This file is compiled twice (build.py).  The second pass defines only this launcher and instantiates only
// the 4x128 kernels, with -fno-slp-vectorize: at that register pressure the SLP pass costs more in spills
// than its packed instructions save (k_trunkD<128> 3.94 -> 3.78 ms per 1M points, 432 -> 96 B of scratch; the
// split-bf16 kernel of BASELINE config 1 1.80 -> 1.46 ms per 100 000 points), while the 2-layer kernels are
// 2-4 % faster with it.
bool launch_trunk_mfma_128x4(int act, int npass, int mode, bool x3, int variant, const float* theta, void* packed_weights,
                             const TrunkArgs& A, int grid, hipStream_t st, hipError_t* err)
{
    __bf16* wpk = reinterpret_cast<__bf16*>(packed_weights);
    if (x3 && npass == 2 && mode == MODE_BWD && variant != 3) {   // BASELINE config 1 (4x128 fp32-grade): k_trunkDx
        *err = act == ACT_TANH ? launch_trunkDx_t<ACT_TANH>(theta, wpk, A, grid, st)
                               : launch_trunkDx_t<ACT_SIGMOID>(theta, wpk, A, grid, st);
        return true;
    }
    if (x3)   // the other modes of 4x128 split-bf16: hi plane streamed through LDS by LDS-DMA, lo plane from L2, one
              // layer-output buffer (XONE)
        return launch2_HL<128, 4, 4, 4, true, 0>(act, npass, mode, theta, wpk, A, grid, st, err);
    if (npass == 2 && mode == MODE_BWD && variant != 3) {
        if (act == ACT_TANH && variant == 11) {   // measurement variant (tools/kbench.py): round 3's lane-linear exchange slots
            *err = launch_trunkD_t<128, 4, ACT_TANH, 1>(theta, wpk, A, grid, st);
            return true;
        }
        *err = act == ACT_TANH ? launch_trunkD_t<128, 4, ACT_TANH>(theta, wpk, A, grid, st)
                               : launch_trunkD_t<128, 4, ACT_SIGMOID>(theta, wpk, A, grid, st);
        return true;
    }
    return launch2_HL<128, 4, 4, 4>(act, npass, mode, theta, wpk, A, grid, st, err);
}
#endif
