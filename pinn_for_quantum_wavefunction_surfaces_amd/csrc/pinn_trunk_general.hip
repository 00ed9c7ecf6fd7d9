// pinn_trunk_general.hip -- k_trunk_mfma2: the trunk kernel for every wide shape and mode that has no
// merged-pass kernel (L = 4 single pass, forward-only, frozen trunk, split-bf16 outside the two benchmark
// shapes): SPLIT waves share a 16-point tile, each owning 16 or 32 neurons, pass by pass; layer outputs and
// adjoints are exchanged through LDS buffers in B-fragment order, which double as the transposed-read image of
// the weight gradient.  Options per shape (M2Cfg): resident blocks per CU, both-pass buffers (KEEP), operand
// prefetch (PIPE), fp32 saved state (SVF32), weights streamed through LDS by LDS-DMA (WSTREAM), split-bf16
// operands (X3: hi + lo planes, three MFMAs per product), one layer-output buffer rebuilt in the reverse sweep
// (XONE).  Conventions: pinn_mfma_common.h.  (train.py:41-57,71 / poc/main.py:247-303,341-355,403)
#include "pinn_mfma_common.h"

namespace {

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
    // a block takes GROUPS consecutive tiles per trip (one per group): block_tiles in units of GROUPS tiles
    const BlockTiles bt = block_tiles((ntile + C::GROUPS - 1) / C::GROUPS, A.shares, C::GROUPS);
    const int64_t iters = bt.count();   // same for every wave of the block: barriers match

    // the point record of the NEXT tile is requested one trip ahead (7 loads per lane that nothing waits
    // for until the next trip): without this every trip opened with an exposed round trip to L2/HBM
    struct PointRec { float x, y, z, R, E, g; unsigned mk; };
    auto load_rec = [&](int64_t itn) {
        const int64_t tl = (bt.first + itn * bt.stride) * C::GROUPS + grp;
        int64_t id = tl * 16 + c;
        id = (tl < ntile && id < A.n) ? id : A.n - 1;
        PointRec r;
        r.x = A.x[id]; r.y = A.y[id]; r.z = A.z[id]; r.R = A.R[id]; r.E = A.E[id]; r.g = A.g[id];
        r.mk = (MODE != MODE_FWD && A.mask) ? A.mask[id] : 0u;
        return r;
    };
    PointRec nxt = load_rec(0);
    for (int64_t itx = 0; itx < iters; itx++) {
        const int64_t tile = (bt.first + itx * bt.stride) * C::GROUPS + grp;
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
}  // namespace

#if PINN_TRUNK_PART == 0

// <H, L, SPLIT, WAVES, X3, OCC>: SPLIT waves share a 16-point tile, WAVES per block, OCC resident
// blocks per CU.  One decomposition per shape, the fastest measured (round 1, 1M points, kernel ms):
//   2x64 : <4,4> x 2 blocks/CU 0.72 (<4,8> 0.74, <2,8> spills)     2x128: <8,8> 1.88 (<4,4> 2.7)
//   4x64 : <2,4> 2.27 (<4,8> 4.2)                                   4x128: <4,4> 11.8 (<8,8> 15.4)
// (round 2, four streams: 2x64 0.57, 3 blocks/CU 0.69, <4,8> 0.58; 4x128 <4,4> 7.6, <8,8> 8.3)
bool launch_trunk_general(int H, int L, int act, int npass, int mode, bool x3, const float* theta, void* packed_weights,
                          const TrunkArgs& A, int grid, hipStream_t st, hipError_t* err)
{
    __bf16* wpk = reinterpret_cast<__bf16*>(packed_weights);
    if (H == 128 && L == 4)   // compiled in this file's second translation unit
        return launch_trunk_general_128x4(act, npass, mode, x3, theta, packed_weights, A, grid, st, err);
    if (x3) {   // split-bf16 operands (3 MFMAs per product)
        if (H == 64 && L == 2) return launch2_HL<64, 2, 4, 8, true>(act, npass, mode, theta, wpk, A, grid, st, err);
        if (H == 64 && L == 4)   // deep, fp32-grade: one group per block, hi plane in LDS, lo plane from L2
            return launch2_HL<64, 4, 4, 4, true, 0>(act, npass, mode, theta, wpk, A, grid, st, err);
        if (H == 128 && L == 2)   // hi weight plane in LDS (64 KB), lo plane from L2, one pass resident: 4.3 ms
            return launch2_HL<128, 2, 8, 8, true, 0>(act, npass, mode, theta, wpk, A, grid, st, err);
        return false;
    }
    if (H == 64 && L == 2) return launch2_HL<64, 2, 4, 4, false, 2>(act, npass, mode, theta, wpk, A, grid, st, err);
    if (H == 128 && L == 2) return launch2_HL<128, 2, 8, 8>(act, npass, mode, theta, wpk, A, grid, st, err);
    if (H == 64 && L == 4) return launch2_HL<64, 4, 2, 4>(act, npass, mode, theta, wpk, A, grid, st, err);
    return false;
}

#else   // ---- PINN_TRUNK_PART == 1: the 4x128 shapes, built WITHOUT the SLP vectorizer (see pinn_trunk_deep.hip)

bool launch_trunk_general_128x4(int act, int npass, int mode, bool x3, const float* theta, void* packed_weights,
                                const TrunkArgs& A, int grid, hipStream_t st, hipError_t* err)
{
    __bf16* wpk = reinterpret_cast<__bf16*>(packed_weights);
    if (x3)   // split-bf16: hi plane streamed through LDS by LDS-DMA, lo plane from L2, one layer-output buffer (XONE)
        return launch2_HL<128, 4, 4, 4, true, 0>(act, npass, mode, theta, wpk, A, grid, st, err);
    return launch2_HL<128, 4, 4, 4>(act, npass, mode, theta, wpk, A, grid, st, err);
}

#endif
