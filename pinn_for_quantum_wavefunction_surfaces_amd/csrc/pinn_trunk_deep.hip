// pinn_trunk_deep.hip -- deep trunks (four hidden layers), symmetric double pass, full reverse sweep:
//   k_trunkD<H, 4>   bf16, H = 64 and 128 (BASELINE config 1's shape in the fast mode)
//   k_trunkDx        4x128 in split-bf16 (BASELINE config 1 in the fp32-grade mode)
// Compiled twice (build.py): part 0 holds H = 64 and keeps the SLP vectorizer, part 1 (-DPINN_TRUNK_PART=1
// -fno-slp-vectorize) holds the 4x128 kernels: at that register pressure the SLP pass costs more in spills than
// its packed instructions save (k_trunkD<128> 3.94 -> 3.78 ms per 1M points, 432 -> 96 B of scratch; k_trunkDx
// 1.80 -> 1.46 ms per 100 000 points), while the narrower kernels are 2-4 % faster with it.
// Conventions: pinn_mfma_common.h.  (train.py:41-57,71 / poc/main.py:247-303,341-355,403)
#include "pinn_mfma_common.h"

namespace {

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
#ifdef PINN_STASH_PAD
constexpr int kStashPad = PINN_STASH_PAD;      // (tools/build_flags.py: experiments)
#else
constexpr int kStashPad = 4096;
#endif
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
    // Parked-state slot of a block: [layer][wave][4 KB], slots kStashPad bytes apart beyond their size.  With slots at a
    // power-of-two stride (64 KB, rounds 2-4) the L2 kept missing on the parked state (FETCH_SIZE 1.84 GB per 1M points of
    // the 4 GB loaded) and the blocks whose slot address has bit 17 ^ bit 22 ^ bit 23 set finished 6.5 % after the others,
    // whatever CU they ran on (tools/lifecycle.py --save: per-block times correlate 0.99 between runs); 4 KB of padding per
    // slot: FETCH_SIZE 0.43 GB, kernel 3.75 -> 3.60 ms (steady A/B, tools/ab_libs.sh; 1 / 2 / 8 / 12 / 20 / 36 KB: 3.66 /
    // 3.44 / 3.46 / 3.60 / 3.61 / 3.53 on the box where 4 KB gave 3.43; pieces interleaved over all blocks: FETCH 0.16 GB,
    // 3.65 ms).  Every store still reaches the memory side (WRITE_SIZE 3.4-4.0 GB either way).
    static constexpr int STASH_LAYER = WAVES * 4 * 1024;                                           // bytes per layer of a block
    static constexpr size_t STASH_BLOCK = STASH ? (size_t)(NW - 1) * STASH_LAYER + kStashPad : 0;  // bytes from slot to slot
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
    const unsigned long long life0 = PINN_LIFE_NOW();
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
    // Slot of lane (g, c) inside a 1 KB (k chunk, stream) block of an exchange buffer: lane-linear, or (SWZ) with the
    // 16-byte slots of the odd lane groups permuted by c ^ 12.  The transposed reads of the weight gradient fetch, per
    // 32-lane service group, the slots c' in {0-3, 8-11} (then {4-7, 12-15}) of TWO adjacent lane groups, which the
    // linear layout puts on the same banks (2-way conflict: half the LDS rate in the weight-gradient phases, where
    // eight waves read 2 x 36 fragments per layer and tile); c ^ 12 moves the odd group's to the other banks and keeps
    // the 16-byte fragment reads (hardware groups {0-3, 12-15, 20-27}, ...) and the 8-byte stores conflict-free.
    auto slot16 = [](int gg, int cc) { return (gg * 16 + (SWZ ? (cc ^ ((gg & 1) * 12)) : cc)) * 16; };
    // LEAN (H = 128, where the kernel spills): address values that can be rebuilt from sw16 / trl with one to three VALU
    // instructions are rebuilt where they are used instead of occupying registers for the whole kernel -- scratch
    // 128 -> 76 B/lane, 3.59 -> 3.34 ms per 1M points (steady A/B, round 4); at H = 64 (no spills) the same costs 1 %.
    constexpr bool LEAN = C::STASH;
    const int sw16 = SWZ ? slot16(g, c) : lane * 16;
    const int own_slot = sw16 + (part >> 1) * NST * 64 * 16 + 8 * (part & 1);     // (rebuilding this one too: no gain)
    const int lane16_r = lane * 16;
    // the lane's column c and lane group g where the tile loop needs them: v_mbcnt (2 instructions) instead of a register
    // that hipcc spills before the loop and reloads in every trip (LEAN)
    auto lane_now = [&]() {
        int l = lane;
        if constexpr (LEAN) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // transposed read: this lane fetches half (c & 1) of the slot of source lane (g' = (c & 3) >> 1 [+ 2 (T & 1)],
    // c' = 8 (g & 1) + (c >> 2) [+ 4 for the second read]) of stream pair member g >> 1
    const int trl = (g >> 1) * 1024 + slot16((c & 3) >> 1, 8 * (g & 1) + (c >> 2)) + (c & 1) * 8;
    // (the second transposed read of a fragment is at slot c' + 4: 64 bytes above the first for even source groups, 64
    // below for odd ones with swizzled slots -- the first address with bit 6 flipped either way: LEAN forms it by one
    // v_xor per weight-gradient phase instead of keeping it in a register)
    const int trh_r = SWZ ? trl ^ 64 : trl;
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
    // (the wave's part of every buffer offset is uniform and rides in the scalar offset: the only per-lane offset
    // register of the weight and stash accesses is lane16)
    const int wsoff = part * KC * 1024;
    auto loadA = [&](int m, bf16x8 (&a)[KC]) {
        int lane16 = lane16_r;  // the packed weights are lane-linear: with swizzled slots the offset is rebuilt from sw16 here
        if constexpr (SWZ && LEAN)
            asm volatile("v_bfe_u32 %0, %1, 8, 1\n\tv_mul_u32_u24 %0, 0xc0, %0\n\tv_xor_b32 %0, %0, %1" : "=&v"(lane16) : "v"(sw16));
#pragma unroll
        for (int kc = 0; kc < KC; kc++)
            a[kc] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + kc * 1024, wsoff + m * H * H * 2, 0));
    };
    // parked saved state of layer l (1..L-2): 8 x u32x2 = 4 x 16 bytes per lane, [k][lane] per wave
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(stash_g, 0, STASH ? (int)(C::STASH_BLOCK * gridDim.x) : 0, 0x00020000);
    // slot of this block: [layer][wave][4 KB], slots STASH_BLOCK bytes apart (see TDCfg)
    const int ssoff = (int)(blockIdx.x * C::STASH_BLOCK) + part * 4096;
    auto stash_so = [&](int l) { return ssoff + (l - 1) * C::STASH_LAYER; };
    auto stash_put = [&](int l, int k, u32x4 v) {
        __builtin_amdgcn_raw_buffer_store_b128(v, srs, sw16 + k * 1024, stash_so(l), 0);
    };
    auto stash_get = [&](int l, int k) {
        return __builtin_amdgcn_raw_buffer_load_b128(srs, sw16 + k * 1024, stash_so(l), 0);
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
    const BlockTiles bt = block_tiles(ntile, A.shares);      // this block's tiles: first, first + stride, ... < end
    const int64_t tfirst = bt.first, tend = bt.end, iters = bt.count();
    const int tstride = bt.stride;

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
    auto trfrag = [&](lds_p eb, int trh, int T, int ph) {     // rows 32ph..32ph+31 of the [column][neuron] image, tile T
        const int o = ph * 2048 + (T >> 1) * NST * 1024 + (T & 1) * 512;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(eb + trl + o));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(eb + trh + o + (SWZ ? 0 : 64)));
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
        int trh = trh_r;
        if constexpr (SWZ && LEAN) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(trh) : "v"(trl));   // not hoisted out of the tile loop
        auto fetch = [&](int j, bf16x8 (&x)[WG]) {
            const int k = j / GPK, i0 = (j % GPK) * WG;
            if (j % GPK == 0) fy[k & 1] = trfrag(yb + (k >> 1) * EB, trh, part, k & 1);
#pragma unroll
            for (int i = 0; i < WG; i++) x[i] = trfrag(xb + (k >> 1) * EB, trh, i0 + i, k & 1);
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
        const float* rp = ring + (slot * 16 + (lane_now() & 15)) * RECF;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
        for (int s = 0; s < NST; s++) { inA[0][s] = r0[s]; inA[1][s] = r1[s]; }
    };

    const lds_p YB = lds0 + C::OFF_Y;
    // the raw points of a ring batch are requested a whole batch ahead (seven registers held for RB trips)
    // (H = 128: loaded at the refill instead -- seven registers less through sixteen trips, scratch 68 -> 24 B/lane, 3.21 ->
    // 3.17 ms; before the other register savings of round 4 the same switch measured 1.5 % SLOWER.  H = 64: no difference.)
    constexpr bool RPF = !C::STASH;
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
    const unsigned long long life1 = PINN_LIFE_NOW();
    for (int64_t itx = 0; itx < iters; itx++) {
        const int64_t tile = tfirst + itx * tstride;
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
                {
                    const int ln = lane_now(), cc = ln & 15, gg = ln >> 4;
                    if ((gg & 1) == 0) SB[(part * 16 + cc) * 2 + (gg >> 1)] = tot;
                }
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
                const int ln = lane_now(), c = ln & 15, g = ln >> 4;
                const int64_t idx = tile * 16 + c;
                const bool valid = tile < tend && idx < A.n;
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
    const unsigned long long life2 = PINN_LIFE_NOW();
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
    life_write(A.psi, C::WAVES, part, life0, life1, life2);
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
    static constexpr size_t STASH_BLOCK = (size_t)(NW - 1) * WAVES * 4 * 1024 + kStashPad;   // bytes from slot to slot (one pass at a time; padding: TDCfg)
    static_assert(LDS_BYTES <= 160 * 1024, "k_trunkDx does not fit the CU's LDS");
    static_assert(OFF_S % 16 == 0 && OFF_R % 16 == 0, "alignment");
};

// VAR (measurement variants, tools/kbench.py): bit 0 = swizzled exchange slots (as in k_trunkD) instead of lane-linear;
// bit 1 = operand fragments requested per k chunk right before its MFMAs (rounds 3-4) instead of pipelined in place
template <int ACT, int VAR = 0>
__global__ __launch_bounds__(TDxCfg::WAVES * 64, 2)
void k_trunkDx(const float* __restrict__ theta, const __bf16* __restrict__ wpk_g, uint4* __restrict__ stash_g, TrunkArgs A)
{
    const unsigned long long life0 = PINN_LIFE_NOW();
    using C = TDxCfg;
    constexpr bool SWZ = (VAR & 1) != 0, PIPE = (VAR & 2) == 0;
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
    // Exchange slots: lane-linear.  SWZ (measurement variant) permutes them as in k_trunkD (odd lane groups by c ^ 12:
    // conflict-free transposed reads): the second transposed read of a fragment, at slot c' + 4, then sits 64 bytes
    // ABOVE the first for even source groups and 64 bytes BELOW for odd ones -- in both cases the first address with
    // bit 6 flipped, so it is formed by one v_xor per weight-gradient phase instead of living in a register, and the
    // lane-linear offset of the weight loads is rebuilt from the slot offset where it is used.  Measured (round 4,
    // steady A/B on one box): bank conflicts -64 %, LDS-array cycles -18 %, kernel +1.5 % (0.850 -> 0.866 ms per
    // 100 000 points; with a resident second address +4.4 %): the array is busy half the kernel's time either way, what
    // the waves wait for is its latency under eight waves' simultaneous requests.  Not the default.
    auto slot16 = [](int gg, int cc) { return (gg * 16 + (SWZ ? (cc ^ ((gg & 1) * 12)) : cc)) * 16; };
    const int sw16 = SWZ ? slot16(g, c) : lane * 16;
    const int own_slot = sw16 + (part >> 1) * NST * 64 * 16 + 8 * (part & 1);
    const int trl = (g >> 1) * 1024 + slot16((c & 3) >> 1, 8 * (g & 1) + (c >> 2)) + (c & 1) * 8;
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
    const int wsoff = part * KC * 1024;         // the wave's part of the buffer offsets is uniform: scalar offset
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(stash_g, 0, (int)(C::STASH_BLOCK * gridDim.x), 0x00020000);
    const int ssoff = (int)(blockIdx.x * C::STASH_BLOCK) + part * 4096;
    auto stash_put = [&](int l, int k, f32x4 v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srs, sw16 + k * 1024, ssoff + (l - 1) * C::WAVES * 4096, 0);
    };
    auto stash_get = [&](int l, int k) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, sw16 + k * 1024, ssoff + (l - 1) * C::WAVES * 4096, 0));
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
        return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(eb + sw16 + (kc * NST + s) * 1024);
    };
    auto trfrag = [&](lds_p eb, int trh, int T, int ph) {
        const int o = ph * 2048 + (T >> 1) * NST * 1024 + (T & 1) * 512;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(eb + trl + o));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(eb + trh + o + (SWZ ? 0 : 64)));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    // D[own 16 rows][NST column tiles] += A x B, split operands (hi*hi, lo*hi, hi*lo into the same accumulator).  Operand
    // fragments are software-pipelined IN PLACE (no second register set -- prefetching into one spilled and was 11 %
    // slower): the hi-plane fragments of the next k chunk are requested right after the eight MFMAs that read this
    // chunk's (hi*hi, lo*hi), the lo-plane ones after the four that read those (hi*lo), so each request has four to eight
    // MFMAs of this wave (and the partner wave's) to land behind instead of stalling the chunk's first MFMA.  The A
    // fragments (hi and lo) of matrix m are streamed from L2 two k chunks ahead: 16 registers in flight instead of 32
    // resident, -6 %.  VAR bit 1 = round 3's schedule (all eight fragments of a chunk requested right before its MFMAs).
    auto loadA1 = [&](int m, int kc, bf16x8& ah, bf16x8& al) {
        // the packed weights are in lane-linear fragment order; with swizzled slots the lane-linear offset is rebuilt from
        // sw16 on the spot (three VALU instructions per pair of loads) instead of occupying a register for the whole kernel
        int lane16 = sw16;
        if constexpr (SWZ)
            asm volatile("v_bfe_u32 %0, %1, 8, 1\n\tv_mul_u32_u24 %0, 0xc0, %0\n\tv_xor_b32 %0, %0, %1" : "=&v"(lane16) : "v"(sw16));
        ah = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + kc * 1024, wsoff + m * H * H * 2, 0));
        al = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + kc * 1024, wsoff + WPLANE_B + m * H * H * 2, 0));
    };
    auto gemmXs = [&](int m, bf16x8 (&ah)[2], bf16x8 (&al)[2], lds_p eb, f32x4 (&d)[NST]) {
        // ah[0..1], al[0..1] hold chunks 0 and 1 on entry (requested by the caller a phase ahead)
        bf16x8 xh[NST], xl[NST];
        if constexpr (PIPE) {
#pragma unroll
            for (int s = 0; s < NST; s++) xh[s] = bfrag(eb, 0, s);
#pragma unroll
            for (int s = 0; s < NST; s++) xl[s] = bfrag(eb + EB1, 0, s);
#pragma unroll
            for (int kc = 0; kc < KC; kc++) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < NST; s++) d[s] = mfma(ah[kc & 1], xh[s], d[s]);
#pragma unroll
                for (int s = 0; s < NST; s++) d[s] = mfma(al[kc & 1], xh[s], d[s]);
                __builtin_amdgcn_sched_barrier(0);
                if (kc + 1 < KC) {
#pragma unroll
                    for (int s = 0; s < NST; s++) xh[s] = bfrag(eb, kc + 1, s);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < NST; s++) d[s] = mfma(ah[kc & 1], xl[s], d[s]);
                __builtin_amdgcn_sched_barrier(0);
                if (kc + 1 < KC) {
#pragma unroll
                    for (int s = 0; s < NST; s++) xl[s] = bfrag(eb + EB1, kc + 1, s);
                }
                if (kc + 2 < KC) loadA1(m, kc + 2, ah[kc & 1], al[kc & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
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
        }
    };
    // weight gradient of one layer from one pass: K = 64 columns = 2 steps (stream pairs) x HT column tiles, three MFMAs
    // per (step, tile).  Pipelined in place like the GEMMs: two fragment pairs (hi, lo) in flight, the pair of step
    // s + 2 requested into the registers of step s right after its three MFMAs.
    auto wgradX = [&](f32x4 (&wa)[HT], lds_p yb, lds_p xb) {
        int trh = trl;
        if constexpr (SWZ) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(trh) : "v"(trl));   // not hoisted: see above
        if constexpr (PIPE) {
            bf16x8 fyh = trfrag(yb, trh, part, 0), fyl = trfrag(yb + EB1, trh, part, 0);
            bf16x8 xs[2][2];
#pragma unroll
            for (int q = 0; q < 2; q++) { xs[q][0] = trfrag(xb, trh, q, 0); xs[q][1] = trfrag(xb + EB1, trh, q, 0); }
#pragma unroll
            for (int st = 0; st < 2 * HT; st++) {
                const int T = st & (HT - 1), q = st & 1;
                __builtin_amdgcn_sched_barrier(0);
                if (st == HT) { fyh = trfrag(yb, trh, part, 1); fyl = trfrag(yb + EB1, trh, part, 1); }
                wa[T] = mfma(fyh, xs[q][0], wa[T]);
                wa[T] = mfma(fyl, xs[q][0], wa[T]);
                wa[T] = mfma(fyh, xs[q][1], wa[T]);
                __builtin_amdgcn_sched_barrier(0);
                if (st + 2 < 2 * HT) {
                    const int T2 = (st + 2) & (HT - 1), k2 = (st + 2) / HT;
                    xs[q][0] = trfrag(xb, trh, T2, k2);
                    xs[q][1] = trfrag(xb + EB1, trh, T2, k2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const bf16x8 fyh = trfrag(yb, trh, part, k), fyl = trfrag(yb + EB1, trh, part, k);
#pragma unroll
                for (int i0 = 0; i0 < HT; i0 += 2) {
                    bf16x8 fxh[2], fxl[2];
#pragma unroll
                    for (int i = 0; i < 2; i++) { fxh[i] = trfrag(xb, trh, i0 + i, k); fxl[i] = trfrag(xb + EB1, trh, i0 + i, k); }
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        wa[i0 + i] = mfma(fyh, fxh[i], wa[i0 + i]);
                        wa[i0 + i] = mfma(fyh, fxl[i], wa[i0 + i]);
                        wa[i0 + i] = mfma(fyl, fxh[i], wa[i0 + i]);
                    }
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
    // -DPINN_STAMPS (diagnostic build, tools/stamps.py): per-wave cycle totals by kind of phase, summed over the three stages
#ifdef PINN_STAMPS
    unsigned long long st_acc[24], st_last = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 24; i++) st_acc[i] = 0;
#define TDX_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; } while (0)
#else
#define TDX_STAMP(i) do { } while (0)
#endif
    const unsigned long long life1 = PINN_LIFE_NOW();
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
        TDX_STAMP(23);
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
            TDX_STAMP(0);
            __syncthreads();
            TDX_STAMP(1);
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
                TDX_STAMP(1 + l);
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
                TDX_STAMP(4 + l);
                __syncthreads();
                TDX_STAMP(8);
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
            TDX_STAMP(9);
            __syncthreads();
            TDX_STAMP(10);
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
                TDX_STAMP(14 - l);
                f32x4 ob[NST];
#pragma unroll
                for (int s = 0; s < NST; s++) ob[s] = splat(0.f);
                gemmXs((l - 1) * 2 + 1, afh, afl, YB, ob);
                TDX_STAMP(17 - l);
                if (l > 1) {
                    const f32x4 h = sq[0], a1 = sq[1], a2 = sq[2], al_ = sq[3];
                    f32x4 s1, s2, s3;
                    act123v<ACT>(h, s1, s2, s3);
                    const f32x4 q = a1 * a1 + a2 * a2;
                    const f32x4 u = 2.0f * s2 * ob[SLP];
                    const f32x4 y0 = s1 * ob[0] + s2 * (a1 * ob[1] + a2 * ob[2]) + (s2 * al_ + s3 * q) * ob[SLP];
                    const f32x4 y1 = s1 * ob[1] + u * a1, y2 = s1 * ob[2] + u * a2, y3 = s1 * ob[SLP];
                    bacc[l - 1] += y0;
                    TDX_STAMP(20 - l);
                    __syncthreads();                // every wave has read YB (and X[l-1])
                    TDX_STAMP(20);
                    put(YB, 0, y0);
                    put(YB, 1, y1);
                    put(YB, 2, y2);
                    put(YB, SLP, y3);
                    TDX_STAMP(21);
                    __syncthreads();
                    TDX_STAMP(22);
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
                    TDX_STAMP(19);
                    __syncthreads();                // X[0] and YB are rewritten by the next forward pass
                    TDX_STAMP(22);
                }
            }
        }
    }
#ifdef PINN_STAMPS
    if (A.psi && lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(A.psi) + ((size_t)blockIdx.x * C::WAVES + part) * 24;
        for (int i = 0; i < 24; i++) dbg[i] = st_acc[i];
    }
#endif
#undef TDX_STAMP
    const unsigned long long life2 = PINN_LIFE_NOW();
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
    life_write(A.psi, C::WAVES, part, life0, life1, life2);
}

template <int ACT, int VAR = 0>
hipError_t launch_trunkDx_t(const float* theta, __bf16* wpk, const TrunkArgs& A, int grid, hipStream_t st)
{
    using C1 = MCfg<128, 4>;
    using C = TDxCfg;
    const int nfrag = C1::NW * 2 * C1::HT * C1::KC * 64;
    hipLaunchKernelGGL((k_pack_weights<128, 4, true>), dim3((nfrag + 255) / 256), dim3(256), 0, st, theta, wpk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    auto kern = k_trunkDx<ACT, VAR>;
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
}  // namespace

#if PINN_TRUNK_PART == 0

// bytes of k_trunkD's / k_trunkDx's parked saved state per block (behind the packed weights)
size_t trunk_deep_stash_block_bytes(int H, bool x3)
{
    if (x3) return H == 128 ? TDxCfg::STASH_BLOCK : 0;
    return H == 128 ? TDCfg<128, 4>::STASH_BLOCK : H == 64 ? TDCfg<64, 4>::STASH_BLOCK : 0;
}

// the merged-pass kernel of (H, 4 hidden layers, act, 2 passes, MODE_BWD); false if there is none for these arguments
// (pinn_trunk_mfma.hip then falls back to k_trunk_mfma2)
bool launch_trunk_deep(int H, int act, int npass, int mode, bool x3, int variant, const float* theta,
                       void* packed_weights, const TrunkArgs& A, int grid, hipStream_t st, hipError_t* err)
{
    __bf16* wpk = reinterpret_cast<__bf16*>(packed_weights);
    if (npass != 2 || mode != MODE_BWD) return false;
    if (H == 128) return launch_trunk_deep_128(act, x3, variant, theta, packed_weights, A, grid, st, err);
    if (H != 64 || x3) return false;
    // weights from L2 into registers (k_trunkD)
    *err = act == ACT_TANH ? launch_trunkD_t<64, 4, ACT_TANH>(theta, wpk, A, grid, st)
                           : launch_trunkD_t<64, 4, ACT_SIGMOID>(theta, wpk, A, grid, st);
    return true;
}

#else   // ---- PINN_TRUNK_PART == 1: the 4x128 kernels

bool launch_trunk_deep_128(int act, bool x3, int variant, const float* theta, void* packed_weights,
                           const TrunkArgs& A, int grid, hipStream_t st, hipError_t* err)
{
    __bf16* wpk = reinterpret_cast<__bf16*>(packed_weights);
    if (x3) {   // BASELINE config 1 (4x128 fp32-grade): k_trunkDx
        if (act == ACT_TANH && (variant == 13 || variant == 14)) {   // measurement variants (tools/kbench.py): VAR = variant - 12
            *err = variant == 13 ? launch_trunkDx_t<ACT_TANH, 1>(theta, wpk, A, grid, st)
                                 : launch_trunkDx_t<ACT_TANH, 2>(theta, wpk, A, grid, st);
            return true;
        }
        *err = act == ACT_TANH ? launch_trunkDx_t<ACT_TANH>(theta, wpk, A, grid, st)
                               : launch_trunkDx_t<ACT_SIGMOID>(theta, wpk, A, grid, st);
        return true;
    }
    if (act == ACT_TANH && variant == 11) {   // measurement variant (tools/kbench.py): round 3's lane-linear exchange slots
        *err = launch_trunkD_t<128, 4, ACT_TANH, 1>(theta, wpk, A, grid, st);
        return true;
    }
    *err = act == ACT_TANH ? launch_trunkD_t<128, 4, ACT_TANH>(theta, wpk, A, grid, st)
                           : launch_trunkD_t<128, 4, ACT_SIGMOID>(theta, wpk, A, grid, st);
    return true;
}

#endif
