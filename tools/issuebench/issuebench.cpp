// issuebench -- how fast does one SIMD of gfx950 issue VALU / transcendental / MFMA work, by waves per SIMD?
//   hipcc -O2 --offload-arch=gfx950 -o issuebench issuebench.cpp && ./issuebench
// One block per CU (256 blocks), W waves per SIMD (block = 256*W threads), each wave runs REPS x 64
// independent instructions of one kind (8 accumulator chains, so dependent latency does not bind).
// Prints cycles per wave-instruction per SIMD (s_memtime, median over blocks) -- the numbers behind the
// VALU floor quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { K_FMA = 0, K_PKFMA = 1, K_EXP = 2, K_MFMA = 3, K_MIX = 4, K_MIXW = 5, K_CVT = 6, K_PKF16 = 7, K_EXPF16 = 8, K_CVTF16 = 9, K_MIX16 = 10, K_MIX32 = 11, K_PKMUL = 12 };

template <int KIND>
__global__ void bench(float* out, unsigned long long* cyc, int reps)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a[8];
    f32x2 p[8];
    f32x4 c[4];
    for (int i = 0; i < 8; i++) { a[i] = 1.0f + 0.001f * (lane + i); p[i] = f32x2{a[i], a[i] + 1}; }
    for (int i = 0; i < 4; i++) c[i] = f32x4{0, 0, 0, 0};
    bf16x8 fa, fb;
    for (int j = 0; j < 8; j++) { fa[j] = (__bf16)(0.01f * (lane + j)); fb[j] = (__bf16)(0.02f * (lane - j)); }
    const float m = 1.0001f, b = 0.0001f;
    f32x16 big;
    for (int i = 0; i < 16; i++) big[i] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == K_FMA || (KIND == K_MIXW && ((wave >> 2) & 1))) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b));
            } else if (KIND == K_PKFMA) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f32x2{m, m}), "v"(f32x2{b, b}));
            } else if (KIND == K_EXP) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            } else if (KIND == K_CVT) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            } else if (KIND == K_PKF16) {      // two f16 FMAs per lane: does it issue at the plain-VALU rate?
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b));
            } else if (KIND == K_PKMUL) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            } else if (KIND == K_EXPF16) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
            } else if (KIND == K_CVTF16) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            } else if (KIND == K_MIX16) {      // one MFMA + 7 packed-f16 FMAs, same wave
                c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c[u & 3], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 7; i++) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b));
            } else if (KIND == K_MIX32) {      // one 32x32x16 MFMA (32 cycles of matrix pipe) + 14 FMAs, same wave
                big = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, big, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 14; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(m), "v"(b));
            } else if (KIND == K_MFMA || KIND == K_MIXW) {
#pragma unroll
                for (int i = 0; i < 8; i++) c[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c[i & 3], 0, 0, 0);
            } else if (KIND == K_MIX) {   // one MFMA + 7 FMAs, same wave
                c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c[u & 3], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 7; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i][0] + p[i][1];
    for (int i = 0; i < 4; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    for (int i = 0; i < 16; i++) s += big[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, int wps)
{
    const int blocks = 256, threads = 256 * wps, reps = 2000;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * threads / 64);
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, reps);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, reps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double inst_per_wave = (double)reps * 64;
    // per SIMD: wps waves each issued inst_per_wave instructions in `med` cycles
    printf("%-8s %d waves/SIMD: %7.2f cycles per wave-instruction per wave, %6.2f cycles per instruction per SIMD, kernel %.3f ms (%.2f GHz eq.)\n",
           name, wps, med / inst_per_wave, med / (inst_per_wave * wps), ms, med / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int wps : {1, 2, 4}) {
        run<K_FMA>("fma", wps);
        run<K_PKFMA>("pk_fma", wps);
        run<K_EXP>("exp", wps);
        run<K_CVT>("cvt_pk", wps);
        run<K_MFMA>("mfma", wps);
        run<K_MIX>("mfma+7fma", wps);
        run<K_PKF16>("pk_fma_f16", wps);
        run<K_PKMUL>("pk_mul_f16", wps);
        run<K_EXPF16>("exp_f16", wps);
        run<K_CVTF16>("cvt_pk_f16", wps);
        run<K_MIX16>("mfma+7pkf16", wps);
        run<K_MIX32>("mfma32+14fma", wps);    // cycles are per 15 instructions/15: x15 = one (MFMA, 14 FMA) group
        if (wps > 1) run<K_MIXW>("mfma|fma", wps);   // SIMD partners: waves 4..7 (12..15) FMA, the others MFMA
    }
    return 0;
}
