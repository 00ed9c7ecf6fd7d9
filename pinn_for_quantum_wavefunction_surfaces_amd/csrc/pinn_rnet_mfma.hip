// pinn_rnet_mfma.hip -- the small nets on the matrix pipe:
//   k_rnet_mfma_fwd / _bwd   E-net (1 -> 32 -> 32 -> 1) and gate (1 -> 10 -> 1), split-bf16
//   k_trunk16                the reference's 16-wide trunk (fp32-grade: three-piece split-bf16 forward)
// Conventions: pinn_mfma_common.h.  (poc/main.py:247-303,341-355,403)
#include "pinn_mfma_common.h"

namespace {

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
                                                          float* __restrict__ Eu, TileShares shares)
{
    constexpr RnetLayout Lo{32, 10};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    RnetConsts K;
    rnet_load_consts(thE, g, K);
    bf16x8 Ah[2], Am[2], Al[2];
#pragma unroll
    for (int jt = 0; jt < 2; jt++) rnet_weight_frag3(thE + Lo.WE2(), jt, g, c, Ah[jt], Am[jt], Al[jt]);
    // a block takes kWavesPerBlock consecutive tiles per trip (one per wave); which ones: block_tiles (TileShares)
    const int64_t ntile0 = (n + 15) / 16;
    const BlockTiles bt = block_tiles((ntile0 + kWavesPerBlock - 1) / kWavesPerBlock, shares, kWavesPerBlock);
    const int64_t tstep = (int64_t)bt.stride * kWavesPerBlock;
    const int64_t ntile = bt.end * kWavesPerBlock < ntile0 ? bt.end * kWavesPerBlock : ntile0;
    // the next trip's R is requested before this trip's arithmetic: one exposed load per wave, not one per tile
    auto clampi = [n](int64_t i) { return i < n ? i : n - 1; };
    int64_t tile = bt.first * kWavesPerBlock + wave;
    float Rnext = tile < ntile ? R[clampi(tile * 16 + c)] : 0.f;
    for (; tile < ntile; tile += tstep) {
        const int64_t idx = tile * 16 + c;
        const bool valid = idx < n;
        const float Rv = Rnext;
        if (tile + tstep < ntile) Rnext = R[clampi((tile + tstep) * 16 + c)];
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
                                                             float* __restrict__ partial, int stride,
                                                             TileShares shares)
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
    // a block takes kWavesPerBlock consecutive tiles per trip (one per wave); which ones: block_tiles (TileShares)
    const int64_t ntile0 = (n + 15) / 16;
    const BlockTiles bt = block_tiles((ntile0 + kWavesPerBlock - 1) / kWavesPerBlock, shares, kWavesPerBlock);
    const int64_t tstep = (int64_t)bt.stride * kWavesPerBlock;
    const int64_t ntile = bt.end * kWavesPerBlock < ntile0 ? bt.end * kWavesPerBlock : ntile0;
    // the next trip's point record (R, E-bar, g-bar) is requested before this trip's arithmetic
    int64_t tile = bt.first * kWavesPerBlock + wave;
    float Rn = 0.f, ebn = 0.f, gbn = 0.f;
    auto request = [&](int64_t t) {
        const int64_t i = t * 16 + c;
        const bool ok = i < n;
        const int64_t id = ok ? i : n - 1;
        Rn = R[id];
        ebn = ok ? Eb[id] : 0.f;
        gbn = ok ? gb[id] : 0.f;
    };
    if (tile < ntile) request(tile);
    for (; tile < ntile; tile += tstep) {
        const float Rv = Rn, ebv = ebn, gbv = gbn;
        if (tile + tstep < ntile) request(tile + tstep);
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
    const unsigned long long life0 = PINN_LIFE_NOW();
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
    // a block takes kWavesPerBlock consecutive tiles per trip (one per wave): block_tiles in units of that many tiles
    const BlockTiles bt = block_tiles((ntile + kWavesPerBlock - 1) / kWavesPerBlock, A.shares, kWavesPerBlock);
    const int64_t tstep = (int64_t)bt.stride * kWavesPerBlock, tlast = bt.end * kWavesPerBlock < ntile ? bt.end * kWavesPerBlock : ntile;
    auto load_rec = [&](int64_t tl) {
        int64_t id = tl * 16 + c;
        id = (tl < ntile && id < A.n) ? id : A.n - 1;
        PointRec r;
        r.x = A.x[id]; r.y = A.y[id]; r.z = A.z[id]; r.R = A.R[id]; r.E = A.E[id]; r.g = A.g[id];
        r.mk = (MODE != MODE_FWD && A.mask) ? A.mask[id] : 0u;
        return r;
    };
    PointRec nxt = load_rec(bt.first * kWavesPerBlock + wave);
    const unsigned long long life1 = PINN_LIFE_NOW();
    for (int64_t tile = bt.first * kWavesPerBlock + wave; tile < tlast; tile += tstep) {
        const int64_t idx = tile * 16 + c;
        const bool valid = idx < A.n;
        const PointRec cur = nxt;
        if (tile + tstep < tlast) nxt = load_rec(tile + tstep);
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
    const unsigned long long life2 = PINN_LIFE_NOW();
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
    life_write(A.psi, kWavesPerBlock, wave, life0, life1, life2);
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
                                float* Eu, int grid, TileShares shares, hipStream_t st)
{
    if (act == ACT_SIGMOID)
        hipLaunchKernelGGL(k_rnet_mfma_fwd<ACT_SIGMOID>, dim3(grid), dim3(kBlock), 0, st, thE, R, n, E, g, Eu, shares);
    else
        hipLaunchKernelGGL(k_rnet_mfma_fwd<ACT_TANH>, dim3(grid), dim3(kBlock), 0, st, thE, R, n, E, g, Eu, shares);
    return hipGetLastError();
}

hipError_t launch_rnet_mfma_bwd(int act, const float* thE, const float* R, const float* Eb, const float* gb,
                                int64_t n, float* part, int stride, int grid, TileShares shares, hipStream_t st)
{
    if (act == ACT_SIGMOID)
        hipLaunchKernelGGL(k_rnet_mfma_bwd<ACT_SIGMOID>, dim3(grid), dim3(kBlock), 0, st, thE, R, Eb, gb, n, part, stride, shares);
    else
        hipLaunchKernelGGL(k_rnet_mfma_bwd<ACT_TANH>, dim3(grid), dim3(kBlock), 0, st, thE, R, Eb, gb, n, part, stride, shares);
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
