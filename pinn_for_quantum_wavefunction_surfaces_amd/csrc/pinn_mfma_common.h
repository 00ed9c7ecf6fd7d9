// pinn_mfma_common.h -- what every matrix-core translation unit of libpinn_hip.so shares: vector types, the
// four-stream convention, the B-fragment order of the packed weights (k_pack_weights) and the lane-local
// activation / bf16 helpers.  Included by pinn_trunk_general.hip (k_trunk_mfma2), pinn_trunk_merged.hip (k_trunk2,
// k_trunk2s, k_trunk2x), pinn_trunk_deep.hip (k_trunkD, k_trunkDx) and pinn_rnet_mfma.hip (R-only nets, k_trunk16);
// pinn_trunk_mfma.hip picks the kernel for a shape.  Everything here sits in an anonymous namespace on purpose:
// each translation unit gets its own internal copies (kernel names stay `(anonymous namespace)::k_...` in profiles).
//
// Same mathematics as k_trunk in pinn_kernels.hip (streams per neuron, hand-derived adjoint;
// train.py:41-57,71 / poc/main.py:247-303,341-355,403), different execution shape: the H x H layers
// are dense GEMMs, so they run on v_mfma_f32_16x16x32_bf16 (16x16x16 for the 16-wide trunk), and the
// streams are FOUR (value, d/dx, d/drho, Laplacian: see NST below), not the five Cartesian ones.
//
// Common to the kernels:
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
//
// Precision: bf16 MFMA operands (activations, adjoints, weights), fp32 accumulation and
// fp32 elementwise math.  Tolerances are stated in tests/test_gpu_parity.py; plain bf16 is
// NOT the 1e-5 parity path, split-bf16 is.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "pinn_math.h"
#include "pinn_common.h"

// pinn_trunk_general.hip and pinn_trunk_deep.hip are compiled twice (build.py): part 0 holds every shape but
// 4x128, part 1 (-DPINN_TRUNK_PART=1 -fno-slp-vectorize) the 4x128 kernels
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

// ------------------------------------------------------------------ life cycle of a wave (diagnostic build only)
// -DPINN_STAMPS, tools/lifecycle.py: entry, first tile, last tile done, exit on the chip-wide 100 MHz clock, and the
// hardware id of the CU the wave ran on; record (block * waves + wave) behind the phase-stamp region of the buffer in A.psi
#ifdef PINN_STAMPS
#define PINN_LIFE_NOW() __builtin_amdgcn_s_memrealtime()
__device__ __forceinline__ void life_write(float* psi, int waves, int wave, unsigned long long t0, unsigned long long t1,
                                           unsigned long long t2)
{
    if (!psi || (threadIdx.x & 63) != 0) return;
    unsigned long long* lf = reinterpret_cast<unsigned long long*>(psi) + (size_t)4096 * 8 * 32
                             + ((size_t)blockIdx.x * waves + wave) * 6;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lf[0] = t0; lf[1] = t1; lf[2] = t2; lf[3] = __builtin_amdgcn_s_memrealtime(); lf[4] = hw; lf[5] = xcc;
}
#else
#define PINN_LIFE_NOW() 0ull
__device__ __forceinline__ void life_write(float*, int, int, unsigned long long, unsigned long long, unsigned long long) {}
#endif

// ------------------------------------------------------------------ which tiles a block takes
// The grid is K classes of G / K consecutive blocks (the last class also takes the G % K left over); each class stripes
// its own range of the ntile tiles: block b of a class takes first, first + stride, ... below end.  K and the ranges come
// from TileShares (pinn_common.h: why); nclass = 0 gives two classes with shares proportional to their sizes, i.e. the
// plain striping of rounds 1-3 up to the order of the tiles.  `unit` = points per tile / 16 (bounds are in 16-point tiles).
__device__ __forceinline__ BlockTiles block_tiles(int64_t ntile, const TileShares& ts, int unit = 1)
{
    return block_tiles_of((int)gridDim.x, (int)blockIdx.x, ntile, ts, unit);
}

}  // namespace
