// pinn_trunk_merged.hip -- the two-hidden-layer trunk kernels that sweep both passes of the symmetric double pass
// together (the benchmark's shapes):
//   k_trunk2    H = 64 and 128, bf16: H/16 waves share a 16-point tile (BASELINE's headline kernel)
//   k_trunk2s   the same for train.py's single-pass formulation: two tiles per trip
//   k_trunk2x   k_trunk2 in split-bf16 (2x64, the fp32-grade mode)
// Conventions: pinn_mfma_common.h.  (train.py:41-57,71 / poc/main.py:247-303,341-355,403)
#include "pinn_mfma_common.h"

namespace {

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
    const unsigned long long life0 = PINN_LIFE_NOW();
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
    const BlockTiles bt = block_tiles(ntile, A.shares);      // this block's tiles: first, first + stride, ... < end
    const int64_t tfirst = bt.first, tend = bt.end, iters = bt.count();
    const int tstride = bt.stride;

    // ring batch b covers trips b*RB .. b*RB + RB-1; this lane's point of it: trip b*RB + 4*part + g, column c
    constexpr int RB = C::RB, RECF = C::RECF;
    float* ring = reinterpret_cast<float*>(lds_raw + C::OFF_R);
    struct PointRec { float x, y, z, R, E, g; unsigned mk; };
    auto load_rec = [&](int64_t b) {
        const int64_t tl = tfirst + (b * RB + 4 * part + g) * tstride;
        int64_t id = tl * 16 + c;
        id = (tl < tend && id < A.n) ? id : A.n - 1;
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
    const unsigned long long life1 = PINN_LIFE_NOW();
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
        const int64_t tile = tfirst + itx * tstride;
        const int64_t idx = tile * 16 + c;
        const bool valid = tile < tend && idx < A.n;
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
    const unsigned long long life2 = PINN_LIFE_NOW();
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
    life_write(A.psi, C::WAVES, part, life0, life1, life2);
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
    const BlockTiles bt = block_tiles(ntile, A.shares);      // this block's tiles: first, first + stride, ... < end
    const int64_t tfirst = bt.first, tend = bt.end, iters = (bt.count() + NTL - 1) / NTL;   // NTL tiles per trip
    const int tstride = bt.stride;

    // ring batch b covers the block's tiles b*RB .. b*RB + RB-1 (NTL per trip); this lane's point of it: tile
    // b*RB + 4*part + g, column c
    constexpr int RB = C::RB, RECF = C::RECF;
    float* ring = reinterpret_cast<float*>(lds_raw + C::OFF_R);
    struct PointRec { float x, y, z, R, E, g; unsigned mk; };
    auto load_rec = [&](int64_t b) {
        const int64_t tl = tfirst + (b * RB + 4 * part + g) * tstride;
        int64_t id = tl * 16 + c;
        id = (tl < tend && id < A.n) ? id : A.n - 1;
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
        const int64_t tile = tfirst + (NTL * itx) * tstride;           // (TT: the first of the trip's two tiles)
        const int64_t idx = tile * 16 + c;
        const bool valid = tile < tend && idx < A.n;
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
                const int64_t tile_t = tile + (int64_t)tl * tstride, idx_t = tile_t * 16 + c;
                const bool valid_t = tile_t < tend && idx_t < A.n;
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
}  // namespace

// the merged-pass kernel of (H, 2 hidden layers, act, npass, MODE_BWD); false if there is none for these arguments
// (pinn_trunk_mfma.hip then falls back to k_trunk_mfma2).  variant: PINN_MFMA_VARIANT (12 = k_trunk2 with round 3's
// lane-linear exchange slots, tanh only: tools/kbench.py)
bool launch_trunk_merged(int H, int act, int npass, int mode, bool x3, int variant, const float* theta,
                         void* packed_weights, const TrunkArgs& A, int grid, hipStream_t st, hipError_t* err)
{
    __bf16* wpk = reinterpret_cast<__bf16*>(packed_weights);
    if (mode != MODE_BWD || (H != 64 && H != 128)) return false;
    if (x3) {
        if (H != 64 || npass != 2) return false;
        // the benchmark's shape in fp32-grade mode: both passes swept together (k_trunk2x)
        *err = act == ACT_TANH ? launch_trunk2x_t<ACT_TANH>(theta, wpk, A, grid, st)
                               : launch_trunk2x_t<ACT_SIGMOID>(theta, wpk, A, grid, st);
        return true;
    }
    if (npass == 1) {
        // train.py's formulation (one pass per point): k_trunk2's schedule with two tiles per trip (k_trunk2s)
        if (H == 64) *err = act == ACT_TANH ? launch_trunk2s_t<64, ACT_TANH>(theta, wpk, A, grid, st)
                                            : launch_trunk2s_t<64, ACT_SIGMOID>(theta, wpk, A, grid, st);
        else *err = act == ACT_TANH ? launch_trunk2s_t<128, ACT_TANH>(theta, wpk, A, grid, st)
                                    : launch_trunk2s_t<128, ACT_SIGMOID>(theta, wpk, A, grid, st);
        return true;
    }
    if (npass != 2) return false;
    if (variant == 12 && act == ACT_TANH) {
        *err = H == 64 ? launch_trunk2_t<64, ACT_TANH, false>(theta, wpk, A, grid, st)
                       : launch_trunk2_t<128, ACT_TANH, false>(theta, wpk, A, grid, st);
        return true;
    }
    // the benchmark's shape: both passes swept together (k_trunk2)
    if (H == 64) *err = act == ACT_TANH ? launch_trunk2_t<64, ACT_TANH>(theta, wpk, A, grid, st)
                                        : launch_trunk2_t<64, ACT_SIGMOID>(theta, wpk, A, grid, st);
    else *err = act == ACT_TANH ? launch_trunk2_t<128, ACT_TANH>(theta, wpk, A, grid, st)
                                : launch_trunk2_t<128, ACT_SIGMOID>(theta, wpk, A, grid, st);
    return true;
}
