// mfmaprec -- how many bits does v_mfma_f32_16x16x32_bf16 keep when it adds its 32 products and the accumulator?
//   hipcc -O2 --offload-arch=gfx950 -o mfmaprec mfmaprec.cpp && ./mfmaprec
// Row 0 of A times column 0 of B: product 0 is BIG = 1.0 (1 x 1), product j is 2^-e (2^-a x 2^-b, exact in bf16), the
// accumulator C is c0.  The exact result is representable in fp32 for every case below; printed is what comes back.
// (Why: the split-bf16 kernels rely on fp32 accumulation inside the matrix pipe; their Laplacian streams carry terms
// ~170 next to terms ~1e-3 at collocation points next to a nucleus.)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// lane (m = lane & 15, kg = lane >> 4) holds A[m][8 kg .. 8 kg + 7]; same for B columns; D[4 (lane >> 4) + r][lane & 15]
__global__ void k(const float* a_row, const float* b_col, float c0, float* out)
{
    const int lane = threadIdx.x, m = lane & 15, kg = lane >> 4;
    bf16x8 fa, fb;
    for (int j = 0; j < 8; j++) {
        fa[j] = (__bf16)(m == 0 ? a_row[8 * kg + j] : 0.f);
        fb[j] = (__bf16)(m == 0 ? b_col[8 * kg + j] : 0.f);
    }
    f32x4 c = {0, 0, 0, 0};
    if (lane == 0) c[0] = c0;
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
}

static float run(const float* a, const float* b, float c0)
{
    float *da, *db, *dout, r;
    hipMalloc(&da, 128); hipMalloc(&db, 128); hipMalloc(&dout, 4);
    hipMemcpy(da, a, 128, hipMemcpyHostToDevice); hipMemcpy(db, b, 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, c0, dout);
    hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    return r;
}

int main()
{
    printf("(1) one big product 1.0 + one small product 2^-e at k position p, C = 0: result - 1.0 (exact: 2^-e)\n");
    for (int p : {1, 9, 31}) {
        printf("  small product at k = %2d:", p);
        for (int e = 16; e <= 24; e++) {
            float a[32] = {0}, b[32] = {0};
            a[0] = 1.f; b[0] = 1.f;
            a[p] = ldexpf(1.f, -(e / 2)); b[p] = ldexpf(1.f, -(e - e / 2));
            printf("  e=%d: %.3g", e, (double)run(a, b, 0.f) - 1.0);
        }
        printf("\n");
    }
    printf("(2) C = 1.0, ONE product 2^-e (everything else 0): result - 1.0\n   ");
    for (int e = 16; e <= 24; e++) {
        float a[32] = {0}, b[32] = {0};
        a[5] = ldexpf(1.f, -(e / 2)); b[5] = ldexpf(1.f, -(e - e / 2));
        printf("  e=%d: %.3g", e, (double)run(a, b, 1.f) - 1.0);
    }
    printf("\n(3) 32 products of 2^-e each, C = 1.0: result - 1.0 (exact: 32 x 2^-e = 2^(5-e))\n   ");
    for (int e = 20; e <= 30; e += 2) {
        float a[32], b[32];
        for (int j = 0; j < 32; j++) { a[j] = ldexpf(1.f, -(e / 2)); b[j] = ldexpf(1.f, -(e - e / 2)); }
        printf("  e=%d: %.3g (exact %.3g)", e, (double)run(a, b, 1.f) - 1.0, ldexp(1.0, 5 - e));
    }
    printf("\n(4) cancellation: products +1.0 and -1.0 and one 2^-e, C = 0: result (exact: 2^-e)\n   ");
    for (int e = 16; e <= 40; e += 4) {
        float a[32] = {0}, b[32] = {0};
        a[0] = 1.f; b[0] = 1.f; a[1] = -1.f; b[1] = 1.f;
        a[2] = ldexpf(1.f, -(e / 2)); b[2] = ldexpf(1.f, -(e - e / 2));
        printf("  e=%d: %.3g", e, (double)run(a, b, 0.f));
    }
    printf("\n(5) round or truncate? C = 1.0, one product 1.5 x 2^-24 (exact 1 + 1.5 ulp/2 -> RNE gives 1 + 2^-23): result - 1.0 = %.3g;"
           " product 0.75 x 2^-23... C = 1.0 + product 2^-24 + 2^-25 (k=3, k=7): %.3g\n",
           [] { float a[32] = {0}, b[32] = {0}; a[3] = 1.5f; b[3] = ldexpf(1.f, -24); return (double)run(a, b, 1.f) - 1.0; }(),
           [] { float a[32] = {0}, b[32] = {0}; a[3] = 1.f; b[3] = ldexpf(1.f, -24); a[7] = 1.f; b[7] = ldexpf(1.f, -25); return (double)run(a, b, 1.f) - 1.0; }());
    return 0;
}
