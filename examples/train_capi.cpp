// examples/train_capi.cpp -- the C ABI (include/pinn.h) driven from plain C++/HIP: no Python, no torch.
//
// Trains the reference architecture (poc/main.py:225-245 shapes, nn.Linear-style seeded init) for a few
// steps on freshly sampled points each step, exactly the loop of train.py:24-73 / poc/main.py:389-417:
//     pinn_sample -> pinn_residual_fwd_bwd -> pinn_adam          (all on one stream, no host sync)
// and prints the loss terms every 50 steps.  Build and run (tests/test_gpu_parity.py does both):
//     hipcc -O2 --offload-arch=gfx950 -Iinclude examples/train_capi.cpp -L<pkg> -lpinn_hip -o train_capi
//     LD_LIBRARY_PATH=<pkg> ./train_capi [points] [steps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "pinn.h"

#define HIP_OK(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));               \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)
#define PINN_OK_OR_DIE(call)                                                              \
    do {                                                                                  \
        int s_ = (call);                                                                  \
        if (s_ != PINN_OK) {                                                              \
            std::fprintf(stderr, "%s: %s (%s)\n", #call, pinn_strerror(s_), pinn_last_error(h)); \
            return 3;                                                                     \
        }                                                                                 \
    } while (0)

int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 100000;   // poc/main.py:39 n_train
    const int steps = argc > 2 ? std::atoi(argv[2]) : 200;
    pinn_arch arch = {16, 2, 32, 10, PINN_ACT_SIGMOID, 2, 1.0f, PINN_PREC_F32};
    const int64_t P = pinn_param_count(&arch);
    const int nt = pinn_tensor_count(&arch);
    std::vector<int64_t> off(nt), sz(nt);
    pinn_tensor_layout(&arch, off.data(), sz.data());

    // nn.Linear default init: U(+-1/sqrt(fan_in)) for weight and bias; Lin_Eout.bias = -1 (poc/main.py:240)
    const int fan[16] = {2, 2, 16, 16, 16, 16, 1, 1, 32, 32, 32, 32, 1, 1, 10, 10};
    std::vector<float> theta(P);
    std::mt19937 gen(12345);
    for (int t = 0; t < nt; t++) {
        std::uniform_real_distribution<float> u(-1.0f / std::sqrt((float)fan[t]), 1.0f / std::sqrt((float)fan[t]));
        for (int64_t k = 0; k < sz[t]; k++) theta[off[t] + k] = u(gen);
    }
    theta[off[11]] = -1.0f;

    pinn_handle* h = nullptr;
    int st = pinn_create(&arch, 0, n, &h);
    if (st != PINN_OK) { std::fprintf(stderr, "pinn_create: %s\n", pinn_strerror(st)); return 1; }

    float *d_theta, *d_m, *d_v, *d_out, *d_best, *d_bestloss, *d_x, *d_y, *d_z, *d_R;
    uint8_t* d_mask;
    int64_t* d_counts;
    int32_t* d_step;
    HIP_OK(hipMalloc(&d_theta, P * 4)); HIP_OK(hipMalloc(&d_m, P * 4)); HIP_OK(hipMalloc(&d_v, P * 4));
    HIP_OK(hipMalloc(&d_out, (P + 3) * 4)); HIP_OK(hipMalloc(&d_best, P * 4)); HIP_OK(hipMalloc(&d_bestloss, 4));
    HIP_OK(hipMalloc(&d_x, n * 4)); HIP_OK(hipMalloc(&d_y, n * 4)); HIP_OK(hipMalloc(&d_z, n * 4));
    HIP_OK(hipMalloc(&d_R, n * 4)); HIP_OK(hipMalloc(&d_mask, n)); HIP_OK(hipMalloc(&d_counts, 16));
    HIP_OK(hipMalloc(&d_step, 4));
    HIP_OK(hipMemcpy(d_theta, theta.data(), P * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_m, 0, P * 4)); HIP_OK(hipMemset(d_v, 0, P * 4)); HIP_OK(hipMemset(d_step, 0, 4));
    const float inf = INFINITY;
    HIP_OK(hipMemcpy(d_bestloss, &inf, 4, hipMemcpyHostToDevice));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    // NN_ion constants (poc/main.py:118-120,345); invN1/invN2 come from the device-side counts
    pinn_ham ham = {-0.5f, -1.0f, -1.0f, 1.0f, 0.0f, 0.0f, 1.0f / (float)n, 0.0f, 0.0f};
    const uint64_t all = ~0ull;
    float first = 0.f, last = 0.f;
    for (int t = 0; t < steps; t++) {
        // box +-18, R in [0.2, 4], nucleus clamp 0.005, boundary shell r >= 17.5 (poc/main.py:18-30);
        // the Philox step is read from the device-side Adam counter
        PINN_OK_OR_DIE(pinn_sample(h, 12345, 0, d_step, 0, n, 18.0f, 0.2f, 4.0f, 0.005f, 17.5f, d_x, d_y, d_z,
                                   d_R, d_mask, d_counts, stream));
        PINN_OK_OR_DIE(pinn_residual_fwd_bwd(h, d_theta, d_x, d_y, d_z, d_R, d_mask, n, &ham, d_counts, all,
                                             d_out, stream));
        PINN_OK_OR_DIE(pinn_adam(h, d_theta, d_out, d_m, d_v, d_step, 8e-3, 0.9, 0.999, 1e-8, all, d_best,
                                 d_bestloss, stream));
        if (t % 50 == 0 || t == steps - 1) {
            float L[3];
            HIP_OK(hipMemcpyAsync(L, d_out + P, 12, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
            std::printf("step %4d  Ltot %.6e  Lpde %.6e  Lbc %.6e\n", t, L[0] + L[1] + L[2], L[0], L[1] + L[2]);
            if (t == 0) first = L[0] + L[1] + L[2];
            last = L[0] + L[1] + L[2];
        }
    }
    HIP_OK(hipStreamSynchronize(stream));
    std::printf("loss %.6e -> %.6e\n", first, last);
    pinn_destroy(h);
    return (std::isfinite(last) && last < first) ? 0 : 4;
}
