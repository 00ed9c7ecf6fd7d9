// tools/ldsbench/ldsbench.cpp -- throughput of the LDS access patterns used by the MFMA kernels (gfx950).
// One block of W waves on one CU; every wave issues REP x 8 back-to-back accesses with the per-lane byte
// offset of the named pattern; prints cycles per wave-instruction (s_memtime, 100 MHz-independent: we use
// wall_clock64 ticks scaled by the measured ratio against a known busy loop -- here simply clock64()).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char* lds_p;

enum { W64 = 0, W128, R64, R128, TR64 };

template <int KIND>
__global__ __launch_bounds__(512) void k(const int* __restrict__ offs, long long* __restrict__ out, int rep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const lds_p base = (lds_p)lds + (threadIdx.x >> 6) * 8192;      // 8 KB private window per wave
    const lds_p p = base + offs[lane];
    for (int i = threadIdx.x; i < 512 * 16; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    u32x4 acc = {0, 0, 0, 0};
    const unsigned addr = (unsigned)(size_t)p;       // LDS byte address of this lane
    u32x2 d2[8];
    u32x4 d4[8];
    const long long t0 = clock64();
    for (int r = 0; r < rep; r++) {
        // eight independent accesses in flight (distinct 1 KB sub-windows), one wait per batch
#define OFFS(u) ((u) * 1024 % 4096)
#define BATCH(INS, ARR, CONS)                                                                      \
        asm volatile(INS " %0, %8 offset:0\n\t" INS " %1, %8 offset:1024\n\t" INS " %2, %8 offset:2048\n\t"   \
                     INS " %3, %8 offset:3072\n\t" INS " %4, %8 offset:0\n\t" INS " %5, %8 offset:1024\n\t"   \
                     INS " %6, %8 offset:2048\n\t" INS " %7, %8 offset:3072\n\ts_waitcnt lgkmcnt(0)"           \
                     : CONS(ARR[0]), CONS(ARR[1]), CONS(ARR[2]), CONS(ARR[3]), CONS(ARR[4]), CONS(ARR[5]),      \
                       CONS(ARR[6]), CONS(ARR[7])                                                              \
                     : "v"(addr))
#define OUT(x) "=v"(x)
        if (KIND == R64) { BATCH("ds_read_b64", d2, OUT); acc[0] ^= d2[0][0] ^ d2[7][1]; }
        if (KIND == R128) { BATCH("ds_read_b128", d4, OUT); acc[0] ^= d4[0][0] ^ d4[7][3]; }
        if (KIND == TR64) { BATCH("ds_read_b64_tr_b16", d2, OUT); acc[0] ^= d2[0][0] ^ d2[7][1]; }
        if (KIND == W64) {
            const u32x2 v = {acc[0] + (unsigned)r, acc[1]};
            asm volatile("ds_write_b64 %0, %1 offset:0\n\tds_write_b64 %0, %1 offset:1024\n\tds_write_b64 %0, %1 offset:2048\n\t"
                         "ds_write_b64 %0, %1 offset:3072\n\tds_write_b64 %0, %1 offset:0\n\tds_write_b64 %0, %1 offset:1024\n\t"
                         "ds_write_b64 %0, %1 offset:2048\n\tds_write_b64 %0, %1 offset:3072\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(v) : "memory");
        }
        if (KIND == W128) {
            const u32x4 v = {acc[0] + (unsigned)r, acc[1], acc[2], acc[3]};
            asm volatile("ds_write_b128 %0, %1 offset:0\n\tds_write_b128 %0, %1 offset:1024\n\tds_write_b128 %0, %1 offset:2048\n\t"
                         "ds_write_b128 %0, %1 offset:3072\n\tds_write_b128 %0, %1 offset:0\n\tds_write_b128 %0, %1 offset:1024\n\t"
                         "ds_write_b128 %0, %1 offset:2048\n\tds_write_b128 %0, %1 offset:3072\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(v) : "memory");
        }
    }
    const long long t1 = clock64();
    if (lane == 0) out[threadIdx.x >> 6] = t1 - t0;
    if (acc[0] == 0x12345 && acc[3] == 7) out[63] = acc[1] + acc[2];   // keep the loads alive
}

struct Pat { const char* name; int kind; int (*off)(int lane); };
static int g(int l) { return l >> 4; }
static int c(int l) { return l & 15; }
static int swz(int s) { return s ^ (((s >> 4) & 1) << 2); }
static Pat pats[] = {
    {"write b64  lane*8 (contiguous)", W64, [](int l) { return l * 8; }},
    {"write b64  lane*16 (half slots: RT=1 exchange)", W64, [](int l) { return l * 16; }},
    {"write b64  lane*16+8 (other half)", W64, [](int l) { return l * 16 + 8; }},
    {"write b64  c*48+8g (k_trunk16 image, pitch 24)", W64, [](int l) { return c(l) * 48 + 8 * g(l); }},
    {"write b64  c*32+8g (pitch 16)", W64, [](int l) { return c(l) * 32 + 8 * g(l); }},
    {"write b128 lane*16 (contiguous)", W128, [](int l) { return l * 16; }},
    {"write b128 c*80+16g (R-net image, pitch 40)", W128, [](int l) { return c(l) * 80 + 16 * g(l); }},
    {"write b128 c*144+16g (gen-1 image, pitch 72)", W128, [](int l) { return c(l) * 144 + 16 * g(l); }},
    {"read  b64  lane*8", R64, [](int l) { return l * 8; }},
    {"read  b128 lane*16", R128, [](int l) { return l * 16; }},
    {"tr16  b64  exchange-slot pattern (gen-2 trl)", TR64, [](int l) { return (g(l) >> 1) * 1024 + (8 * (g(l) & 1) + (c(l) >> 2)) * 16 + (c(l) & 1) * 8 + ((c(l) & 3) >> 1) * 256; }},
    {"tr16  b64  exchange-slot pattern, XOR-swizzled", TR64, [](int l) { return (g(l) >> 1) * 1024 + swz(16 * ((c(l) & 3) >> 1) + 8 * (g(l) & 1) + (c(l) >> 2)) * 16 + (c(l) & 1) * 8; }},
    {"tr16  b64  (4g+(c>>2))*48+8(c&3) (k_trunk16, pitch 24)", TR64, [](int l) { return (4 * g(l) + (c(l) >> 2)) * 48 + 8 * (c(l) & 3); }},
    {"tr16  b64  (4g+(c>>2))*32+8(c&3) (pitch 16)", TR64, [](int l) { return (4 * g(l) + (c(l) >> 2)) * 32 + 8 * (c(l) & 3); }},
    {"tr16  b64  (4g+(c>>2))*80+8(c&3) (R-net, pitch 40)", TR64, [](int l) { return (4 * g(l) + (c(l) >> 2)) * 80 + 8 * (c(l) & 3); }},
    {"tr16  b64  (8g+(c>>2))*144+8(c&3) (gen-1, pitch 72)", TR64, [](int l) { return (8 * g(l) + (c(l) >> 2)) * 144 + 8 * (c(l) & 3); }},
    {"tr16  b64  lane*8 (contiguous)", TR64, [](int l) { return l * 8; }},
};

int main()
{
    int* d_off; long long* d_out;
    hipMalloc(&d_off, 64 * 4); hipMalloc(&d_out, 64 * 8);
    const int rep = 2000;
    std::printf("%-58s %10s %10s\n", "pattern (per-lane byte offset)", "1 wave", "8 waves");
    for (const Pat& P : pats) {
        int off[64];
        for (int l = 0; l < 64; l++) off[l] = P.off(l);
        hipMemcpy(d_off, off, sizeof off, hipMemcpyHostToDevice);
        double res[2];
        int wv[2] = {1, 8};
        for (int t = 0; t < 2; t++) {
            const int threads = 64 * wv[t];
            for (int pass = 0; pass < 2; pass++) {
                switch (P.kind) {
                    case W64: hipLaunchKernelGGL(k<W64>, dim3(1), dim3(threads), 65536, 0, d_off, d_out, rep); break;
                    case W128: hipLaunchKernelGGL(k<W128>, dim3(1), dim3(threads), 65536, 0, d_off, d_out, rep); break;
                    case R64: hipLaunchKernelGGL(k<R64>, dim3(1), dim3(threads), 65536, 0, d_off, d_out, rep); break;
                    case R128: hipLaunchKernelGGL(k<R128>, dim3(1), dim3(threads), 65536, 0, d_off, d_out, rep); break;
                    default: hipLaunchKernelGGL(k<TR64>, dim3(1), dim3(threads), 65536, 0, d_off, d_out, rep); break;
                }
                hipDeviceSynchronize();
            }
            long long o[8];
            hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost);
            long long mx = 0;
            for (int w = 0; w < wv[t]; w++) mx = o[w] > mx ? o[w] : mx;
            res[t] = (double)mx / (rep * 8.0);                 // clock ticks per wave-instruction (per wave)
        }
        std::printf("%-58s %10.2f %10.2f\n", P.name, res[0], res[1] / 8.0 * 1.0);
    }
    std::printf("(ticks of clock64 per wave-instruction; the 8-wave column is per instruction of the whole block, i.e. "
                "block time / (8 waves x instructions per wave))\n");
    return 0;
}
