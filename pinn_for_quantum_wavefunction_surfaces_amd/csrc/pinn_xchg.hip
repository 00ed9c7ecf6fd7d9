// pinn_xchg.hip -- one-shot all-reduce of the per-step gradient message over peer-mapped memory.
//
// The data-parallel step exchanges ONE small fp32 vector per step, [grad(P) | 3 loss terms | 2 counts]:
// 6 KB (reference architecture) ... 205 KB (4x128).  A ring all-reduce is the wrong shape for that: 14
// hops at 8 ranks, each latency-bound.  MI355X nodes are a full xGMI mesh (7 links per GPU), so every
// rank can push its vector straight into a slot in every peer's memory (7 concurrent link writes) and
// each rank sums the N slots locally: one hop, two small kernels on the caller's stream, no host
// involvement.  The reference has nothing to match here (it is single device).
//
// Protocol (all device-side, per call; seq = 1, 2, 3, ... identical on every rank, kept in DEVICE memory:
// both kernels read state[0] = number of completed calls, the last block of the sum kernel advances it,
// so a call captured into a HIP graph counts on with every replay exactly like an eager call):
//   memory of rank p:  recv[2][N][cap] floats  +  flag[2][N] words (one 128-byte line each)
//   push kernel (N blocks):   block q copies buf -> recv_q[seq&1][me][0..n), system-scope fence, then
//                             flag_q[seq&1][me] = seq (release)
//   sum kernel:               waits until flag_me[seq&1][q] == seq for every q (bounded spin), then
//                             buf[i] = recv_me[seq&1][0][i] + recv_me[seq&1][1][i] + ... in RANK ORDER
// Every rank adds the same N values in the same order, so the sums are bitwise identical on all ranks
// (theta stays replica-identical under Adam).  Two parities suffice: a rank can only be one call ahead
// of the slowest one (its call seq+1 completes only after every peer has pushed seq+1, i.e. finished
// summing seq), so slot seq&1 is never overwritten (by call seq+2) before its readers are done.
// The exchange memory is allocated uncached (hipDeviceMallocUncached), so neither the writer's nor the
// owner's L2 can hold a stale line of it; peers map it with hipIpcOpenMemHandle.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "../../include/pinn.h"

namespace {

constexpr int kFlagStride = 32;          // uint32 words per flag: one 128-byte line each
constexpr int kPushThreads = 1024;
constexpr int kSumThreads = 256;
constexpr int kMaxWorld = 16;

struct PeerTable {
    float* recv[kMaxWorld];              // base of each rank's recv[2][N][cap]
    unsigned* flag[kMaxWorld];           // base of each rank's flag[2][N][kFlagStride]
};

// state[0] calls completed (= seq of the last one), state[1] blocks of the running sum kernel that are
// done, state[2] sticky status: 1 after a timed-out wait, 2 after a sum that was not finite (a peer that timed out
// earlier pushes the NaN message it was left with, so its partners learn of it with the next call; a diverged
// gradient reports the same way), state[3] this call's decision, taken ONCE for all blocks of the sum kernel (0
// undecided, 1 every peer arrived, 2 timed out): the first block through its wait publishes it, the others follow
// it, so a peer that arrives right at the deadline cannot leave buf part sum, part NaN.
// What status() does NOT prove: a rank whose own waits all succeeded has state[2] == 0 even if a PEER timed out on
// that same call (the peer computed NaN, this rank a valid sum: the replicas differ until the next call's non-finite
// sum flags it here too); a caller that needs agreement at a given step checks after the following call.
__global__ __launch_bounds__(kPushThreads) void k_xchg_push(PeerTable T, const float* __restrict__ buf, int64_t n,
                                                           int64_t cap, int me, int world,
                                                           const unsigned* __restrict__ state)
{
    const int q = blockIdx.x;            // destination rank
    const unsigned seq = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    const int par = seq & 1;
    float* dst = T.recv[q] + ((size_t)par * world + me) * cap;
    const int64_t n4 = n / 4;
    const float4* s4 = reinterpret_cast<const float4*>(buf);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int64_t k = threadIdx.x; k < n4; k += kPushThreads) d4[k] = s4[k];
    for (int64_t k = 4 * n4 + threadIdx.x; k < n; k += kPushThreads) dst[k] = buf[k];
    __threadfence_system();              // every thread's stores have left for the owner's memory
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(T.flag[q] + ((size_t)par * world + me) * kFlagStride, seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(kSumThreads) void k_xchg_sum(const float* __restrict__ recv, const unsigned* flag,
                                                         float* __restrict__ buf, int64_t n, int64_t cap, int world,
                                                         unsigned long long max_wait_ticks, unsigned* state)
{
    __shared__ int ok;
    __shared__ unsigned seq_s;
    if (threadIdx.x == 0) {
        // every block reads state[0] before it counts itself done, and only the last block to be done
        // advances it: all blocks of this launch see the same seq
        const unsigned seq = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        const int par = seq & 1;
        seq_s = seq;
        int good = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        for (int q = 0; q < world && good; q++) {
            const unsigned* f = flag + ((size_t)par * world + q) * kFlagStride;
            while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > max_wait_ticks) { good = 0; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        // one decision per call: whoever gets here first publishes it, everybody follows it
        const unsigned mine = good ? 1u : 2u;
        const unsigned prev = atomicCAS(state + 3, 0u, mine);
        ok = (prev ? prev : mine) == 1u;
        if (!ok) atomicCAS(state + 2, 0u, 1u);  // a peer never arrived: report, and poison buf below
    }
    __syncthreads();
    const unsigned seq = seq_s;
    const int par = seq & 1;
    if (ok) {
        __threadfence_system();
        const float* base = recv + (size_t)par * world * cap;
        for (int64_t i = (int64_t)blockIdx.x * kSumThreads + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * kSumThreads) {
            float s = __builtin_nontemporal_load(base + i);
            for (int q = 1; q < world; q++) s += __builtin_nontemporal_load(base + (size_t)q * cap + i);
            buf[i] = s;
            if (!(__builtin_fabsf(s) <= 3.402823466e38f)) atomicCAS(state + 2, 0u, 2u);   // NaN or inf: see state[2]
        }
    } else {
        // no silent local-only gradient: whatever consumes buf (loss terms, Adam) sees NaN
        for (int64_t i = (int64_t)blockIdx.x * kSumThreads + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * kSumThreads)
            buf[i] = __builtin_nanf("");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(state + 1, 1u) == gridDim.x - 1) {     // last block of this call
            __hip_atomic_store(state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(state + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next call: undecided
            __hip_atomic_store(state, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

struct pinn_xchg {
    int device, rank, world;
    int64_t cap;
    void* local;                 // own allocation: recv then flags
    size_t recv_bytes, flag_bytes;
    void* mapped[kMaxWorld];     // peers' allocations as mapped here (own: == local)
    PeerTable table;
    unsigned* state;             // device words: [0] calls completed, [1] sum blocks done, [2] status, [3] decision
    bool connected;
    double timeout_s;
    char err[256];
};

static int xfail(pinn_xchg* x, int code, const char* what, hipError_t e = hipSuccess)
{
    if (x) {
        if (e != hipSuccess) snprintf(x->err, sizeof x->err, "%s: %s", what, hipGetErrorString(e));
        else snprintf(x->err, sizeof x->err, "%s", what);
    }
    return code;
}

#define XCHK(x, call)                                                       \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return xfail(x, PINN_ERR_HIP, #call, e_);     \
    } while (0)

extern "C" const char* pinn_xchg_last_error(const pinn_xchg* x) { return x ? x->err : "null exchange"; }

extern "C" int pinn_xchg_create(int device, int rank, int world, int64_t max_floats, pinn_xchg** out)
{
    if (!out) return PINN_ERR_ARG;
    *out = nullptr;
    if (device < 0 || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || max_floats < 1)
        return PINN_ERR_ARG;
    pinn_xchg* x = new (std::nothrow) pinn_xchg();
    if (!x) return PINN_ERR_ARG;
    memset(x, 0, sizeof *x);
    x->device = device; x->rank = rank; x->world = world;
    x->cap = (max_floats + 63) / 64 * 64;            // slots start on 256-byte boundaries
    x->timeout_s = 20.0;
    if (const char* t = getenv("PINN_XCHG_TIMEOUT_S")) {
        const double v = atof(t);
        if (v > 0) x->timeout_s = v;
    }
    *out = x;
    XCHK(x, hipSetDevice(device));
    x->recv_bytes = (size_t)2 * world * x->cap * sizeof(float);
    x->flag_bytes = (size_t)2 * world * kFlagStride * sizeof(unsigned);
    // uncached: no L2 (writer's or owner's) may keep a line of the exchange memory.  PINN_XCHG_MEM =
    // finegrained | default selects the other allocation kinds for measurements.
    {
        const char* kind = getenv("PINN_XCHG_MEM");
        const size_t bytes = x->recv_bytes + x->flag_bytes;
        if (kind && !strcmp(kind, "default")) XCHK(x, hipMalloc(&x->local, bytes));
        else if (kind && !strcmp(kind, "finegrained"))
            XCHK(x, hipExtMallocWithFlags(&x->local, bytes, hipDeviceMallocFinegrained));
        else XCHK(x, hipExtMallocWithFlags(&x->local, bytes, hipDeviceMallocUncached));
    }
    XCHK(x, hipMemset(x->local, 0, x->recv_bytes + x->flag_bytes));
    XCHK(x, hipMalloc(&x->state, 4 * sizeof(unsigned)));
    XCHK(x, hipMemset(x->state, 0, 4 * sizeof(unsigned)));
    XCHK(x, hipDeviceSynchronize());
    snprintf(x->err, sizeof x->err, "ok");
    return PINN_OK;
}

extern "C" int pinn_xchg_local_handle(pinn_xchg* x, void* handle64)
{
    if (!x || !handle64) return PINN_ERR_ARG;
    static_assert(sizeof(hipIpcMemHandle_t) == PINN_XCHG_HANDLE_BYTES, "IPC handle size");
    hipIpcMemHandle_t h;
    XCHK(x, hipIpcGetMemHandle(&h, x->local));
    memcpy(handle64, &h, sizeof h);
    return PINN_OK;
}

extern "C" int pinn_xchg_connect(pinn_xchg* x, const void* handles)
{
    if (!x || !handles) return PINN_ERR_ARG;
    if (x->connected) return xfail(x, PINN_ERR_ARG, "already connected");
    XCHK(x, hipSetDevice(x->device));
    for (int q = 0; q < x->world; q++) {
        void* base = x->local;
        if (q != x->rank) {
            hipIpcMemHandle_t h;
            memcpy(&h, (const char*)handles + (size_t)q * sizeof h, sizeof h);
            XCHK(x, hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
        }
        x->mapped[q] = base;
        x->table.recv[q] = (float*)base;
        x->table.flag[q] = (unsigned*)((char*)base + x->recv_bytes);
    }
    x->connected = true;
    return PINN_OK;
}

extern "C" int pinn_xchg_allreduce(pinn_xchg* x, float* buf, int64_t n, void* stream)
{
    if (!x || !buf) return PINN_ERR_ARG;
    if (!x->connected) return xfail(x, PINN_ERR_ARG, "pinn_xchg_connect has not been called");
    if (n < 1 || n > x->cap) return xfail(x, PINN_ERR_CAPACITY, "message longer than max_floats");
    if (((uintptr_t)buf & 15) != 0) return xfail(x, PINN_ERR_ARG, "buf must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_xchg_push, dim3(x->world), dim3(kPushThreads), 0, st, x->table, (const float*)buf, n, x->cap,
                       x->rank, x->world, (const unsigned*)x->state);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xfail(x, PINN_ERR_HIP, "k_xchg_push", e);
    const int blocks = (int)((n + kSumThreads * 4 - 1) / (kSumThreads * 4));
    hipLaunchKernelGGL(k_xchg_sum, dim3(blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks)), dim3(kSumThreads), 0, st,
                       (const float*)x->table.recv[x->rank], (const unsigned*)x->table.flag[x->rank], buf, n, x->cap,
                       x->world, (unsigned long long)(x->timeout_s * 1e8), x->state);
    e = hipGetLastError();
    if (e != hipSuccess) return xfail(x, PINN_ERR_HIP, "k_xchg_sum", e);
    return PINN_OK;
}

extern "C" int pinn_xchg_set_timeout(pinn_xchg* x, double seconds)
{
    if (!x || !(seconds > 0)) return PINN_ERR_ARG;
    x->timeout_s = seconds;
    return PINN_OK;
}

extern "C" int pinn_xchg_status(pinn_xchg* x, void* stream)
{
    if (!x) return PINN_ERR_ARG;
    unsigned s = 0;
    XCHK(x, hipMemcpyAsync(&s, x->state + 2, sizeof s, hipMemcpyDeviceToHost, (hipStream_t)stream));
    XCHK(x, hipStreamSynchronize((hipStream_t)stream));
    if (s == 1) return xfail(x, PINN_ERR_HIP, "a peer did not arrive within the exchange timeout");
    if (s) return xfail(x, PINN_ERR_HIP, "an exchanged sum was not finite (a peer that timed out earlier pushed its NaN "
                                         "message, or the gradient itself diverged)");
    return PINN_OK;
}

extern "C" int pinn_xchg_destroy(pinn_xchg* x)
{
    if (!x) return PINN_ERR_ARG;
    (void)hipSetDevice(x->device);
    (void)hipDeviceSynchronize();
    if (x->connected)
        for (int q = 0; q < x->world; q++)
            if (q != x->rank && x->mapped[q]) (void)hipIpcCloseMemHandle(x->mapped[q]);
    if (x->local) (void)hipFree(x->local);
    if (x->state) (void)hipFree(x->state);
    delete x;
    return PINN_OK;
}
