// pinn_trunk_mfma.hip -- which matrix-core trunk kernel runs a shape.  The kernels themselves:
//   pinn_trunk_merged.hip    k_trunk2 (two hidden layers, symmetric double pass, full reverse sweep: the benchmark's
//                            shape, H = 64 and 128, bf16), k_trunk2s (train.py's single-pass formulation),
//                            k_trunk2x (k_trunk2 in split-bf16, 2x64)
//   pinn_trunk_deep.hip      k_trunkD (deep trunks, L = 4, bf16), k_trunkDx (4x128 in split-bf16)
//   pinn_trunk_general.hip   k_trunk_mfma2: every other wide shape and mode (L = 4 single pass, forward-only,
//                            frozen trunk, split-bf16 outside the two benchmark shapes)
//   pinn_rnet_mfma.hip       k_rnet_mfma_fwd / _bwd (E-net and gate), k_trunk16 (the reference's 16-wide trunk)
//   pinn_mfma_common.h       layout conventions, weight pack, lane-local helpers
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "pinn_math.h"
#include "pinn_common.h"

using namespace pinn;

bool mfma_trunk_supported(int H, int L) { return (H == 64 || H == 128) && (L == 2 || L == 4); }
bool mfma_x3_supported(int H, int L) { return (H == 64 || H == 128) && (L == 2 || L == 4); }

// measurement switch (tools/kbench.py): PINN_MFMA_VARIANT = 0 (default), 3 (the per-pass kernel k_trunk_mfma2 where a
// merged-pass kernel is the default) 11 / 12 (k_trunkD / k_trunk2 with lane-linear exchange slots), 13 / 14 (k_trunkDx with swizzled slots / without the
// in-place operand pipeline).  Read ONCE per process (round 4; it was a getenv per
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


// resident blocks per CU the chosen decomposition is built for (pinn_create sizes the persistent
// grid and the per-block partial buffer with it)
int mfma_blocks_per_cu(int H, int L, bool x3)
{
    if (H == 64 && (L == 2 || L == 4) && !x3) return 2;
    return 1;
}

// bytes behind the packed weights for k_trunkD's parked saved state (nblocks = the persistent grid)
size_t mfma_stash_bytes(int H, int L, bool x3, int nblocks)
{
    if (L != 4) return 0;
    return trunk_deep_stash_block_bytes(H, x3) * (size_t)nblocks;
}

size_t mfma_packed_weight_bytes(int H, int L) { return (size_t)2 * (L - 1) * 2 * H * H * 2; }  // hi + lo planes

int mfma_partial_stride(int H, int L) { return 3 * H + (L - 1) * (H * H + H) + H + 1 + 3; }

// merged-pass kernels where one exists for the arguments (k_trunk2 / k_trunk2s / k_trunk2x, k_trunkD / k_trunkDx),
// k_trunk_mfma2 otherwise -- and always under PINN_MFMA_VARIANT=3
bool launch_trunk_mfma(int H, int L, int act, int npass, int mode, bool x3, const float* theta,
                       void* packed_weights, const TrunkArgs& A, int grid, hipStream_t st,
                       hipError_t* err)
{
    const int variant = mfma_variant();
    if (variant != 3) {
        if (L == 2 && launch_trunk_merged(H, act, npass, mode, x3, variant, theta, packed_weights, A, grid, st, err))
            return true;
        if (L == 4 && launch_trunk_deep(H, act, npass, mode, x3, variant, theta, packed_weights, A, grid, st, err))
            return true;
    }
    return launch_trunk_general(H, L, act, npass, mode, x3, theta, packed_weights, A, grid, st, err);
}
