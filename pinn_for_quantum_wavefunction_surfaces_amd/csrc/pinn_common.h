// pinn_common.h -- declarations shared by the translation units of libpinn_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int kWave = 64;
constexpr int kBlock = 256;             // 4 waves
constexpr int kWavesPerBlock = kBlock / kWave;

enum { MODE_FWD = 0, MODE_BWD = 1, MODE_HEAD = 2 };

// Which share of the 16-point tiles each CLASS of blocks takes.  With K resident blocks per CU the dispatcher puts blocks
// b, b + G/K, b + 2G/K, ... of a grid of G on one CU, and the waves that arrived first win every issue arbitration on their
// SIMDs: of two blocks with equal shares of 1M points the first finishes after 302 us and then leaves its partner alone on
// the CU, at a lone block's efficiency, until 425 us (k_trunk2<64>); of four (k_trunk16's forward-only sweep) the blocks
// finish after 29, 48, 69 and 87 us (tools/lifecycle.py, profiles/r04_block_shares.md).  Class c = blocks [c G/K, (c+1) G/K)
// stripes the tiles [bound[c-1], bound[c]) (bound[-1] = 0, bound[K-1] = all); the host sets the bounds that let the
// classes finish together when the grid is the full K-per-CU one, nclass = 0 (equal shares) otherwise.  The map tile ->
// block stays a pure function of (n, grid, shares): results remain bitwise reproducible.
struct TileShares {
    int nclass;               // 0: equal shares; 2 or 4
    int64_t bound[3];         // upper tile bounds of classes 0 .. nclass-2, in 16-point tiles
};

// tiles of block b of a grid of G: first, first + stride, ... below end.  The grid is K classes of G / K consecutive blocks
// (the last class also takes the G % K left over); nclass = 0 gives two classes with shares proportional to their sizes,
// i.e. the plain striping of rounds 1-3 up to the order of the tiles.  `unit` = points per tile / 16 (the bounds of
// TileShares are in 16-point tiles).  Host and device (the CPU test drives it through pinn_debug_block_tiles).
struct BlockTiles {
    int64_t first, end;
    int stride;
    __host__ __device__ int64_t count() const { return end > first ? (end - first + stride - 1) / stride : 0; }
};
__host__ __device__ inline BlockTiles block_tiles_of(int G, int b, int64_t ntile, const TileShares& ts, int unit)
{
    const bool given = ts.nclass == 2 || ts.nclass == 4;
    const int K = G < 2 ? 1 : (given && G >= ts.nclass ? ts.nclass : 2);
    const int per = G / K;
    int c = b / per;
    if (c >= K) c = K - 1;
    const int lo_b = c * per, nb = c == K - 1 ? G - lo_b : per;
    auto raw = [&](int k) -> int64_t {            // upper tile bound of class k as given
        if (k >= K - 1) return ntile;
        int64_t t = given && K == ts.nclass ? (ts.bound[k] + unit - 1) / unit : ntile * ((int64_t)(k + 1) * per) / G;
        if (t < 0) t = 0;
        return t > ntile ? ntile : t;
    };
    // bounds out of order never make ranges overlap: the bound of a class is at least that of the class before it
    int64_t lo = 0, hi = 0;
    for (int k = 0; k <= c; k++) {
        lo = hi;
        const int64_t r = raw(k);
        hi = r > hi ? r : hi;
    }
    BlockTiles t;
    t.first = lo + (b - lo_b);
    t.end = hi;
    t.stride = nb;
    return t;
}

struct TrunkArgs {
    const float *x, *y, *z, *R;
    const float *E, *g;       // from k_rnet_fwd
    const uint8_t* mask;      // may be null
    int64_t n;
    float a, b, c, P, out_scale, invN, invN1, invN2;
    const long long* bc_counts;  // optional device int64[2]: overrides invN1/invN2
    float *psi, *lap, *res;   // MODE_FWD outputs (nullable)
    float *Eb, *gb;           // MODE_BWD / MODE_HEAD per-point adjoints
    float* partial;           // [gridDim.x][stride]: trunk grads then 3 loss terms
    int stride;
    TileShares shares;        // which tiles a block takes (see TileShares)
};

// ---- MFMA trunk: bf16 operands, fp32 accumulate (dispatch in pinn_trunk_mfma.hip, kernels in pinn_trunk_merged.hip,
// pinn_trunk_deep.hip, pinn_trunk_general.hip; conventions in pinn_mfma_common.h)
bool mfma_trunk_supported(int H, int L);
bool mfma_x3_supported(int H, int L);            // split-bf16 (fp32-grade) operands
// packed bf16 weights for one call: [(L-1)][2 (fwd, bwd)][H*H], hi and lo planes (pack order: k_pack_weights)
size_t mfma_packed_weight_bytes(int H, int L);
int mfma_partial_stride(int H, int L);           // floats per block partial row (trunk grads + 3)
size_t mfma_stash_bytes(int H, int L, bool x3, int nblocks);   // k_trunkD's parked state, behind the packed weights
int mfma_blocks_per_cu(int H, int L, bool x3);   // resident blocks per CU of the chosen kernel
// enqueue pack + trunk kernel; returns false when (H, L, act, npass, mode) has no instantiation
bool launch_trunk_mfma(int H, int L, int act, int npass, int mode, bool x3, const float* theta,
                       void* packed_weights, const TrunkArgs& A, int grid, hipStream_t st,
                       hipError_t* err);
// the per-family entry points launch_trunk_mfma chooses from (same contract; `variant` = PINN_MFMA_VARIANT).  The
// 4x128 kernels live in second translation units of their files, built without SLP vectorization (build.py)
bool launch_trunk_merged(int H, int act, int npass, int mode, bool x3, int variant, const float* theta,
                         void* packed_weights, const TrunkArgs& A, int grid, hipStream_t st, hipError_t* err);
bool launch_trunk_deep(int H, int act, int npass, int mode, bool x3, int variant, const float* theta,
                       void* packed_weights, const TrunkArgs& A, int grid, hipStream_t st, hipError_t* err);
bool launch_trunk_deep_128(int act, bool x3, int variant, const float* theta, void* packed_weights,
                           const TrunkArgs& A, int grid, hipStream_t st, hipError_t* err);
size_t trunk_deep_stash_block_bytes(int H, bool x3);
bool launch_trunk_general(int H, int L, int act, int npass, int mode, bool x3, const float* theta, void* packed_weights,
                          const TrunkArgs& A, int grid, hipStream_t st, hipError_t* err);
bool launch_trunk_general_128x4(int act, int npass, int mode, bool x3, const float* theta, void* packed_weights,
                                const TrunkArgs& A, int grid, hipStream_t st, hipError_t* err);

// ---- R-only nets on the matrix pipe (pinn_rnet_mfma.hip): (HE, HG) = (32, 10), split-bf16 operands
bool rnet_mfma_supported(int HE, int HG);
hipError_t launch_rnet_mfma_fwd(int act, const float* thE, const float* R, int64_t n, float* E, float* g,
                                float* Eu, int grid, TileShares shares, hipStream_t st);
hipError_t launch_rnet_mfma_bwd(int act, const float* thE, const float* R, const float* Eb, const float* gb,
                                int64_t n, float* part, int stride, int grid, TileShares shares, hipStream_t st);

// ---- the 16-wide reference trunk on the matrix pipe (pinn_rnet_mfma.hip, k_trunk16): two blocks per CU,
// 64 points per block and trip
bool trunk16_mfma_supported(int H, int L);
hipError_t launch_trunk16_mfma(int act, int npass, int mode, const float* theta, const TrunkArgs& A, int grid,
                               hipStream_t st);
