// mfmadep -- what does a chain of DEPENDENT MFMAs (same accumulator, back to back) cost on gfx950?
//   hipcc -O2 --offload-arch=gfx950 -o mfmadep mfmadep.cpp && ./mfmadep
// The other micro-benchmarks of this directory rotate four accumulators (dependency distance 4).  The split-bf16
// kernels issue the three MFMAs of one product (hi*hi, hi*lo, lo*hi) into the SAME accumulator back to back
// (distance 1) in their weight-gradient phases.  One block per CU, 8 waves; waves 0-3 (one per SIMD) run `reps` x 12
// MFMAs with the accumulators rotating over D registers, waves 4-7 either idle or the same.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int D, bool BOTH>
__global__ __launch_bounds__(512) void bench(float* out, unsigned long long* cyc, int reps)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 c[12];
    for (int i = 0; i < 12; i++) c[i] = f32x4{0, 0, 0, 0};
    bf16x8 fa, fb;
    for (int j = 0; j < 8; j++) { fa[j] = (__bf16)(0.01f * (lane + j)); fb[j] = (__bf16)(0.02f * (lane - j)); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4 || BOTH) {
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                // distance D: MFMA i accumulates into register (i / G) where runs of G = 12 / ... -> simply i % D rotates;
                // D = 1: one accumulator; D = 3 "runs": three in a row on one accumulator, then the next (the kernels' order)
                const int k = D > 0 ? i % D : i / 3;
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[k]) : "v"(fa), "v"(fb));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 12; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int D, bool BOTH>
void run(const char* what, int reps = 3000)
{
    const int blocks = 256, threads = 512;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 8);
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL((bench<D, BOTH>), dim3(blocks), dim3(threads), 0, 0, out, cyc, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    std::vector<double> ta, tb;
    for (int bk = 0; bk < blocks; bk++) {
        unsigned long long ma = 0, mb = 0;
        for (int w = 0; w < 4; w++) { ma = std::max(ma, h[bk * 8 + w]); mb = std::max(mb, h[bk * 8 + 4 + w]); }
        ta.push_back((double)ma); tb.push_back((double)mb);
    }
    std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
    const double n = reps * 12.0;
    printf("%-44s %s : waves 0-3 %6.2f cycles per MFMA", what, BOTH ? "two waves per SIMD" : "one wave per SIMD ", ta[ta.size() / 2] / n);
    if (BOTH) printf(", waves 4-7 %6.2f, SIMD %6.2f per MFMA", tb[tb.size() / 2] / n, std::max(ta[ta.size() / 2], tb[tb.size() / 2]) / (2 * n));
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<12, false>("12 accumulators in rotation");
    run<4, false>("4 accumulators in rotation");
    run<3, false>("3 accumulators in rotation");
    run<2, false>("2 accumulators in rotation");
    run<1, false>("1 accumulator (every MFMA depends on the last)");
    run<0, false>("runs of 3 on one accumulator, then the next");
    run<12, true>("12 accumulators in rotation");
    run<4, true>("4 accumulators in rotation");
    run<2, true>("2 accumulators in rotation");
    run<1, true>("1 accumulator");
    run<0, true>("runs of 3 on one accumulator, then the next");
    return 0;
}
