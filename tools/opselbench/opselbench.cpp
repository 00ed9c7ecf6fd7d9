// opselbench -- does v_pk_mul_f32 / v_pk_fma_f32 with op_sel:[0,1] (the LOW result lane reading the HIGH dword
// of an operand pair) always deliver the product of its operands on MI355X?  Background: k_trunk2 (csrc/
// pinn_trunk_merged.hip) gave run-to-run different d/dx-stream values with two workgroups per CU until its
// ring-fed broadcast arithmetic was moved to unpacked VALU instructions (profiles/r02_packed_opsel.md).
// This program tries to isolate that: every lane reads a 16-byte record from LDS (ds_read_b128, as the kernel
// does), multiplies a per-lane weight pair by the record's SECOND dword with the packed op_sel form and with
// two plain v_mul_f32, and counts lanes where the two differ -- with MFMAs and LDS traffic interleaved, for
// 1, 2 and 4 workgroups of 4 waves per CU.
//   hipcc -O2 --offload-arch=gfx950 -o opselbench opselbench.cpp && ./opselbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MFMA = 0 (round 3): the same loop without the two MFMAs -- the regime of the exact-fp32 lane = point kernels of
// csrc/pinn_kernels.hip (k_trunk for H = 32 / L = 3, the R-net fallbacks), which still contain the op_sel forms.
template <int MODE, int MFMA = 1>   // MODE 0: operands from an LDS record; 1: operands produced by VALU instructions
__global__ __launch_bounds__(256) void k_opsel(const float* __restrict__ in, unsigned long long* __restrict__ bad, int iters,
                                               float* __restrict__ sink)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* rec = lds + threadIdx.x * 4;                        // one 16-byte record per thread
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (wave + j)); }
    unsigned long long nbad = 0;
    float chk = 0.f;
    for (int it = 0; it < iters; it++) {
        const float base = in[(blockIdx.x * 256 + threadIdx.x + it * 7919) & 0xffff];
        *reinterpret_cast<f32x4*>(rec) = f32x4{base, base * 1.7f + 0.3f, base * 0.9f - 0.2f, base * 2.3f};
        __syncthreads();
        f32x4 r;
        if (MODE == 0) {
            r = *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x + 64) & 255) * 4);   // another wave's record
        } else {
            r = f32x4{base + 1.f, base * 1.7f + 0.3f, base * 0.9f - 0.2f, base * 2.3f};
            asm volatile("" : "+v"(r));
        }
        const float w0 = base * 0.5f + 1.0f, w1 = base * 0.25f - 1.0f;
        typedef unsigned long long u64;
        auto pair = [](float lo, float hi) { return (u64)__builtin_bit_cast(unsigned, lo) | ((u64)__builtin_bit_cast(unsigned, hi) << 32); };
        const u64 w = pair(w0, w1), x = pair(r[0], r[1]);
        u64 pp, qq;
        // packed: p.lo = w.lo * x.hi, p.hi = w.hi * x.hi ; fma form: q = w * x.hi + p
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(pp) : "v"(w), "v"(x));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(qq) : "v"(w), "v"(x), "v"(pp));
        const float p[2] = {__builtin_bit_cast(float, (unsigned)pp), __builtin_bit_cast(float, (unsigned)(pp >> 32))};
        const float q[2] = {__builtin_bit_cast(float, (unsigned)qq), __builtin_bit_cast(float, (unsigned)(qq >> 32))};
        // a few MFMAs in between, as in the kernel's surroundings
        if (MFMA) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        }
        float s0, s1, t0, t1;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(w0), "v"(r[1]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(w1), "v"(r[1]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(w0), "v"(r[1]), "v"(s0));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(w1), "v"(r[1]), "v"(s1));
        const bool m0 = __builtin_bit_cast(unsigned, p[0]) != __builtin_bit_cast(unsigned, s0);
        const bool m1 = __builtin_bit_cast(unsigned, p[1]) != __builtin_bit_cast(unsigned, s1);
        const bool m2 = __builtin_bit_cast(unsigned, q[0]) != __builtin_bit_cast(unsigned, t0);
        const bool m3 = __builtin_bit_cast(unsigned, q[1]) != __builtin_bit_cast(unsigned, t1);
        nbad += (unsigned long long)m0 + ((unsigned long long)m1 << 16) + ((unsigned long long)m2 << 32) + ((unsigned long long)m3 << 48);
        if ((m0 || m1 || m2 || m3) && sink && atomicAdd(reinterpret_cast<unsigned*>(sink), 1u) == 0) {
            sink[1] = w0; sink[2] = w1; sink[3] = r[0]; sink[4] = r[1]; sink[5] = p[0]; sink[6] = p[1]; sink[7] = s0; sink[8] = s1;
            sink[9] = q[0]; sink[10] = q[1]; sink[11] = t0; sink[12] = t1;
        }
        chk += p[0] + q[1];
        __syncthreads();
    }
    if (nbad) { atomicAdd(bad, nbad & 0xffff); atomicAdd(bad + 1, (nbad >> 16) & 0xffff); atomicAdd(bad + 2, (nbad >> 32) & 0xffff); atomicAdd(bad + 3, nbad >> 48); }
    if (sink && chk + acc[0] == 12345.678f) sink[13] = chk;
}

int main()
{
    float* in;
    unsigned long long* bad;
    hipMalloc(&in, 65536 * sizeof(float));
    hipMalloc(&bad, 4 * sizeof(*bad));
    float* sink;
    hipMalloc(&sink, 64);
    float* h = (float*)malloc(65536 * sizeof(float));
    srand(7);
    for (int i = 0; i < 65536; i++) h[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
    hipMemcpy(in, h, 65536 * sizeof(float), hipMemcpyHostToDevice);
    const int iters = 20000;   // < 65536: the four per-lane counters are 16-bit fields
    for (int mfma = 1; mfma >= 0; mfma--)
    for (int mode = 0; mode < 2; mode++)
        for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
            // LDS per workgroup sized so that exactly wg_per_cu of them fit a CU's 160 KB
            const size_t ldsb = wg_per_cu == 1 ? 100 * 1024 : wg_per_cu == 2 ? 70 * 1024 : 36 * 1024;
            auto kern = mfma ? (mode == 0 ? k_opsel<0, 1> : k_opsel<1, 1>) : (mode == 0 ? k_opsel<0, 0> : k_opsel<1, 0>);
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
            hipMemset(bad, 0, 4 * sizeof(*bad));
            hipMemset(sink, 0, 64);
            hipLaunchKernelGGL(kern, dim3(256 * wg_per_cu), dim3(256), ldsb, 0, in, bad, iters, sink);
            hipError_t e = hipDeviceSynchronize();
            unsigned long long nb[4];
            float ex[16];
            hipMemcpy(nb, bad, sizeof(nb), hipMemcpyDeviceToHost);
            hipMemcpy(ex, sink, 64, hipMemcpyDeviceToHost);
            printf("%s operands from %-5s %d workgroup(s) of 4 waves per CU, %.3g products of each kind: mismatches pk_mul.lo %llu pk_mul.hi %llu "
                   "pk_fma.lo %llu pk_fma.hi %llu (%s)\n", mfma ? "MFMAs between," : "no MFMA,      ", mode == 0 ? "LDS," : "VALU,", wg_per_cu, 256.0 * 256 * wg_per_cu * iters,
                   nb[0], nb[1], nb[2], nb[3], hipGetErrorString(e));
            if (nb[0] + nb[1] + nb[2] + nb[3])
                printf("    first: w = (%.9g, %.9g) x = (%.9g, %.9g): pk_mul (%.9g, %.9g) scalar (%.9g, %.9g); pk_fma (%.9g, %.9g) scalar (%.9g, %.9g)\n",
                       ex[1], ex[2], ex[3], ex[4], ex[5], ex[6], ex[7], ex[8], ex[9], ex[10], ex[11], ex[12]);
        }
    return 0;
}
