// coissue -- when do matrix-pipe and vector instructions of the TWO waves of a SIMD overlap on gfx950?
//   hipcc -O2 --offload-arch=gfx950 -o coissue coissue.cpp && ./coissue
// Round 2's issuebench found "1 MFMA + 7 FMAs per wave at two waves = 34.4 cycles = 16.1 + 7 x 2.62" and an
// MFMA-only wave beside an FMA-only wave at 21.6 cycles per (MFMA, FMA) pair: no overlap -- although ONE wave per
// SIMD hides its MFMAs behind its own FMAs completely.  Hypothesis tested here: an MFMA that waits for the busy
// matrix pipe holds the SIMD's vector issue port (head-of-line), so a partner's VALU work only overlaps with MFMAs
// that are spaced at least a pipe time apart in the issuing wave's own stream.
// One block per CU, 8 waves (two per SIMD: waves w and w+4 share one), role A = waves 0-3, role B = waves 4-7.
// Each role runs its own instruction pattern `reps` times; cycles (s_memtime) are reported PER ROLE (median
// over the blocks of the role's slowest wave).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum Pat {
    P_IDLE = 0,       // nothing (the role exits at once)
    P_FMA,            // 8 independent v_fma_f32
    P_MFMA,           // 8 back-to-back MFMA 16x16x32 (4 accumulators)
    P_MFMA_NOP,       // 8 x (MFMA, s_nop 15)          -- MFMAs spaced a pipe time apart, no VALU of its own
    P_MFMA_NOP2,      // 8 x (MFMA, s_nop 7, s_nop 3)  -- 12 cycles of padding
    P_MFMA_3FMA,      // 8 x (MFMA, 3 FMA)
    P_MFMA_7FMA,      // 8 x (MFMA, 7 FMA)
    P_MFMA32,         // 8 back-to-back MFMA 32x32x16 (one accumulator)
    P_MFMA32_NOP,     // 8 x (MFMA 32x32x16, 2 x s_nop 15)
    P_PKFMA,          // 8 independent v_pk_fma_f32
    P_EXP,            // 8 independent v_exp_f32
    P_MFMA_SLEEP,     // 8 x (MFMA, s_sleep 1)
    P_MFMA4_28FMA,    // 2 x (4 MFMA back to back, 28 FMA)
    P_N
};
const char* kName[P_N] = {"idle", "fma", "mfma", "mfma+nop16", "mfma+nop12", "mfma+3fma", "mfma+7fma", "mfma32", "mfma32+nop32",
                          "pk_fma", "exp", "mfma+sleep1", "4mfma+28fma"};
// instructions of the measured kind per pattern repetition: {MFMA (16x16x32 equivalents), VALU}
const int kMfma[P_N] = {0, 0, 8, 8, 8, 8, 8, 16, 16, 0, 0, 8, 8};
const int kValu[P_N] = {0, 8, 0, 0, 0, 24, 56, 0, 0, 8, 8, 0, 56};

template <int PAT>
__device__ __forceinline__ void body(float (&a)[8], f32x4 (&c)[4], f32x16& big, bf16x8 fa, bf16x8 fb, float m, float b, int reps)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p[8];
    for (int i = 0; i < 8; i++) p[i] = f32x2{a[i], a[i] + 1};
    for (int r = 0; r < reps; r++) {
        if (PAT == P_FMA) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b));
        } else if (PAT == P_PKFMA) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f32x2{m, m}), "v"(f32x2{b, b}));
        } else if (PAT == P_EXP) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if (PAT == P_MFMA) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[i & 3]) : "v"(fa), "v"(fb));
        } else if (PAT == P_MFMA_NOP) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+v"(c[i & 3]) : "v"(fa), "v"(fb));
        } else if (PAT == P_MFMA_NOP2) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 3" : "+v"(c[i & 3]) : "v"(fa), "v"(fb));
        } else if (PAT == P_MFMA_SLEEP) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_sleep 1" : "+v"(c[i & 3]) : "v"(fa), "v"(fb));
        } else if (PAT == P_MFMA_3FMA) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[i & 3]) : "v"(fa), "v"(fb));
#pragma unroll
                for (int j = 0; j < 3; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(3 * i + j) & 7]) : "v"(m), "v"(b));
            }
        } else if (PAT == P_MFMA_7FMA) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[i & 3]) : "v"(fa), "v"(fb));
#pragma unroll
                for (int j = 0; j < 7; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(m), "v"(b));
            }
        } else if (PAT == P_MFMA4_28FMA) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
#pragma unroll
                for (int j = 0; j < 4; j++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(fa), "v"(fb));
#pragma unroll
                for (int j = 0; j < 28; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j & 7]) : "v"(m), "v"(b));
            }
        } else if (PAT == P_MFMA32) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big) : "v"(fa), "v"(fb));
        } else if (PAT == P_MFMA32_NOP) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15" : "+v"(big) : "v"(fa), "v"(fb));
        }
    }
    for (int i = 0; i < 8; i++) a[i] += p[i][0] + p[i][1];
}

template <int PA, int PB, int PRIO_A, int PRIO_B>
__global__ __launch_bounds__(512) void bench(float* out, unsigned long long* cyc, int repsA, int repsB)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a[8];
    f32x4 c[4];
    f32x16 big;
    for (int i = 0; i < 8; i++) a[i] = 1.0f + 0.001f * (lane + i);
    for (int i = 0; i < 4; i++) c[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 16; i++) big[i] = 0.f;
    bf16x8 fa, fb;
    for (int j = 0; j < 8; j++) { fa[j] = (__bf16)(0.01f * (lane + j)); fb[j] = (__bf16)(0.02f * (lane - j)); }
    const float m = 1.0001f, b = 0.0001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (PRIO_A) __builtin_amdgcn_s_setprio(PRIO_A);
        body<PA>(a, c, big, fa, fb, m, b, repsA);
    } else {
        if (PRIO_B) __builtin_amdgcn_s_setprio(PRIO_B);
        body<PB>(a, c, big, fa, fb, m, b, repsB);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    for (int i = 0; i < 16; i++) s += big[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int PA, int PB, int PRIO_A = 0, int PRIO_B = 0>
void run(int repsA = 4000, int repsB = 4000)
{
    const int blocks = 256, threads = 512;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 8);
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL((bench<PA, PB, PRIO_A, PRIO_B>), dim3(blocks), dim3(threads), 0, 0, out, cyc, repsA, repsB);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((bench<PA, PB, PRIO_A, PRIO_B>), dim3(blocks), dim3(threads), 0, 0, out, cyc, repsA, repsB);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    std::vector<double> ta, tb;
    for (int bk = 0; bk < blocks; bk++) {
        unsigned long long ma = 0, mb = 0;
        for (int w = 0; w < 4; w++) { ma = std::max(ma, h[bk * 8 + w]); mb = std::max(mb, h[bk * 8 + 4 + w]); }
        ta.push_back((double)ma); tb.push_back((double)mb);
    }
    std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
    const double ca = ta[ta.size() / 2], cb = tb[tb.size() / 2];
    printf("A %-13s prio %d | B %-13s prio %d : A %9.0f cyc", kName[PA], PRIO_A, kName[PB], PRIO_B, ca);
    if (kMfma[PA]) printf(" (%.2f per MFMA16)", ca / (repsA * (double)kMfma[PA]));
    if (kValu[PA]) printf(" (%.2f per VALU)", ca / (repsA * (double)kValu[PA]));
    printf(" ; B %9.0f cyc", cb);
    if (kMfma[PB]) printf(" (%.2f per MFMA16)", cb / (repsB * (double)kMfma[PB]));
    if (kValu[PB]) printf(" (%.2f per VALU)", cb / (repsB * (double)kValu[PB]));
    printf(" ; kernel %.3f ms\n", ms);
    hipFree(out); hipFree(cyc);
}

int main()
{
    // baselines: one role alone on its SIMD
    run<P_MFMA, P_IDLE>();
    run<P_MFMA_NOP, P_IDLE>();
    run<P_MFMA_NOP2, P_IDLE>();
    run<P_MFMA_SLEEP, P_IDLE>();
    run<P_MFMA32, P_IDLE>();
    run<P_MFMA32_NOP, P_IDLE>();
    run<P_FMA, P_IDLE>();
    run<P_PKFMA, P_IDLE>();
    run<P_EXP, P_IDLE>();
    run<P_MFMA_3FMA, P_IDLE>();
    run<P_MFMA_7FMA, P_IDLE>();
    run<P_MFMA4_28FMA, P_IDLE>();
    run<P_FMA, P_FMA>();
    run<P_PKFMA, P_PKFMA>();
    // matrix wave beside vector wave: same number of pattern repetitions, so B (8 VALU per rep) finishes first when it overlaps
    run<P_MFMA, P_FMA>();
    run<P_MFMA, P_FMA, 0, 1>();
    run<P_MFMA, P_FMA, 1, 0>();
    run<P_MFMA_NOP, P_FMA>();
    run<P_MFMA_NOP, P_FMA>(4000, 12000);     // B long enough to run beside all of A
    run<P_MFMA_NOP2, P_FMA>(4000, 12000);
    run<P_MFMA_SLEEP, P_FMA>(4000, 12000);
    run<P_MFMA_NOP, P_PKFMA>(4000, 12000);
    run<P_MFMA_NOP, P_EXP>(4000, 8000);
    run<P_MFMA_NOP, P_FMA, 1, 0>(4000, 12000);
    run<P_MFMA_NOP, P_FMA, 0, 1>(4000, 12000);
    run<P_MFMA32, P_FMA>(4000, 24000);
    run<P_MFMA32_NOP, P_FMA>(4000, 24000);
    run<P_MFMA, P_FMA>(4000, 12000);
    // both waves the same mixed stream
    run<P_MFMA_3FMA, P_MFMA_3FMA>();
    run<P_MFMA_7FMA, P_MFMA_7FMA>();
    run<P_MFMA4_28FMA, P_MFMA4_28FMA>();
    run<P_MFMA_7FMA, P_MFMA_7FMA, 0, 1>();
    // mixed stream beside a pure vector wave
    run<P_MFMA_3FMA, P_FMA>(4000, 16000);
    run<P_MFMA_7FMA, P_FMA>(4000, 32000);
    run<P_MFMA, P_MFMA>();
    run<P_MFMA_NOP, P_MFMA_NOP>();
    return 0;
}
