// pinn_math.h -- per-point mathematics of the PINN Schroedinger-residual step.
//
// One collocation point per SIMD lane.  Everything a point needs is carried as
// five "streams" per neuron (value, d/dx, d/dy, d/dz, Laplacian): the closed
// form of what the reference obtains with nested torch.autograd.grad
// (train.py:8-10,54; poc/main.py:82-97), and its hand-derived adjoint, which
// the reference gets from Ltot.backward() (train.py:71; poc/main.py:403).
//
// The code is templated on the scalar type and on a gradient "sink", so the
// very same functions compile (a) into the HIP kernels, where the sink feeds
// f32 MFMA tiles, and (b) with g++ into tests/native/point_math_check, where
// the sink is a plain array and the result is compared with the CPU oracle.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define PINN_HD __host__ __device__ __forceinline__
#else
#define PINN_HD inline
#endif

// The per-point code is fully unrolled into one huge basic block.  Left alone,
// the machine scheduler hoists every scalar weight load to the top of it and
// spills thousands of SGPRs.  A scheduling fence per neuron row keeps each row's
// s_load next to the FMAs that consume it.
#if defined(__HIP_DEVICE_COMPILE__)
#define PINN_ROW_FENCE() __builtin_amdgcn_sched_barrier(0)
// sched_barrier orders machine instructions only; instruction selection is still free to sink
// pure arithmetic below later loads.  An empty volatile asm on a region's result pins the
// FMAs that produce it inside the region (no instruction is emitted).
#define PINN_PIN(x) asm volatile("" : "+v"(x))
#else
#define PINN_ROW_FENCE() ((void)0)
#define PINN_PIN(x) ((void)0)
#endif

namespace pinn {

enum { ACT_SIGMOID = 0, ACT_TANH = 1 };

// ---------------------------------------------------------------- scalar ops
#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 transcendental unit: v_exp_f32 / v_rcp_f32 / v_rsq_f32 (1 ulp each).
PINN_HD float p_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
PINN_HD float p_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
PINN_HD float p_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#else
PINN_HD float p_exp(float x) { return std::exp(x); }
PINN_HD float p_rcp(float x) { return 1.0f / x; }
PINN_HD float p_rsqrt(float x) { return 1.0f / std::sqrt(x); }
#endif
// exp(k x) for a compile-time constant k: the constant is folded into the exponent scaling, one
// multiply in front of v_exp_f32 instead of two (tanh's exp(-2a))
#if defined(__HIP_DEVICE_COMPILE__)
PINN_HD float p_expk(float x, float k) { return __builtin_amdgcn_exp2f(x * (k * 1.4426950408889634f)); }
#else
PINN_HD float p_expk(float x, float k) { return std::exp(k * x); }
#endif
PINN_HD double p_expk(double x, double k) { return std::exp(k * x); }
PINN_HD double p_exp(double x) { return std::exp(x); }
PINN_HD double p_rcp(double x) { return 1.0 / x; }
PINN_HD double p_rsqrt(double x) { return 1.0 / std::sqrt(x); }

// Two neurons side by side.  The fp32 vector peak of gfx950 (157 TFLOP/s) is the PACKED rate:
// a wave64 VALU instruction takes 4 cycles whether it is v_fmac_f32 or v_pk_fma_f32, so every
// mat-vec below is written as rank-1 updates on neuron PAIRS: acc2[p] += w2[p] * splat(x), which
// hipcc turns into v_pk_fma_f32 v[..], s[pair], v[..] op_sel_hi:[1,0,0] (weights stay in SGPRs).
template <class T> struct V2;
template <> struct V2<float> { typedef float type __attribute__((vector_size(8))); };
template <> struct V2<double> { typedef double type __attribute__((vector_size(16))); };
template <class T> PINN_HD typename V2<T>::type splat2(T x) { typename V2<T>::type v = {x, x}; return v; }
template <class T> PINN_HD typename V2<T>::type make2(T x, T y) { typename V2<T>::type v = {x, y}; return v; }
// Broadcast of ONE element of a pair to both halves.  Written as splat2(v[i & 1]), hipcc reads the odd element in place
// with v_pk_fma_f32 ... op_sel:[1,..] (the LOW result lane reading the HIGH dword of the operand pair) -- the form that
// intermittently computed with 0 in place of that dword on MI355X when a matrix-pipe instruction was in flight
// (profiles/r02_packed_opsel.md, r03_packed_opsel.md; cause open).  The odd element therefore takes a detour through a
// register of its own (one v_mov_b32 per broadcast, against H/2 packed FMAs that use it), from which the broadcast is
// the low-half form op_sel_hi:[0,..]; tests/test_host_cpu.py compiles this unit too and checks that none is left.
PINN_HD float own_register(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mov_b32 %0, %1" : "=v"(x) : "v"(x));
#endif
    return x;
}
PINN_HD double own_register(double x) { return x; }
template <class T> PINN_HD typename V2<T>::type bcast_elem(typename V2<T>::type v, int i)
{
    return splat2<T>((i & 1) ? own_register(v[1]) : v[0]);
}

// thT = theta with every matrix that the FORWARD pass contracts over its input index stored
// input-major: W_l (l >= 1) and WE2 transposed in place, W_0 as [2][H].  Everything else is
// copied.  Built once per call (k_prep_theta on the device, prep_theta_host in the CPU check).

// value, first and second derivative of the hidden activation
// (sigmoid: poc/main.py:231, train.py:45; tanh: build extension)
template <int ACT, class T>
PINN_HD void act012(T a, T& s, T& s1, T& s2)
{
    if (ACT == ACT_SIGMOID) {
        s = p_rcp(T(1) + p_exp(-a));
        s1 = s - s * s;
        s2 = s1 - T(2) * s * s1;
    } else {
        s = T(2) * p_rcp(T(1) + p_expk(a, T(-2))) - T(1);
        s1 = T(1) - s * s;
        s2 = T(-2) * s * s1;
    }
}

// derivatives 1..3 from the stored activation value
template <int ACT, class T>
PINN_HD void act123(T s, T& s1, T& s2, T& s3)
{
    if (ACT == ACT_SIGMOID) {
        s1 = s - s * s;
        s2 = s1 - T(2) * s * s1;
        s3 = s1 - T(6) * s1 * s1;
    } else {
        s1 = T(1) - s * s;
        s2 = T(-2) * s * s1;
        s3 = s1 * (T(4) - T(6) * s1);
    }
}

template <int ACT, class T>
PINN_HD T act1_from_value(T s) { return ACT == ACT_SIGMOID ? s - s * s : T(1) - s * s; }

// --------------------------------------------------------------- parameters
// Offsets into theta (state_dict order; include/pinn.h).
template <int H, int L>
struct TrunkLayout {
    static constexpr int W(int l) { return l == 0 ? 0 : 3 * H + (l - 1) * (H * H + H); }
    static constexpr int b(int l) { return W(l) + (l == 0 ? 2 * H : H * H); }
    static constexpr int Wout = 3 * H + (L - 1) * (H * H + H);
    static constexpr int bout = Wout + H;
    static constexpr int size = bout + 1;
};

struct RnetLayout {  // relative to the start of the E-net block (= trunk size)
    int HE, HG;
    PINN_HD constexpr int WE1() const { return 0; }
    PINN_HD constexpr int bE1() const { return HE; }
    PINN_HD constexpr int WE2() const { return 2 * HE; }
    PINN_HD constexpr int bE2() const { return 2 * HE + HE * HE; }
    PINN_HD constexpr int WE3() const { return 3 * HE + HE * HE; }
    PINN_HD constexpr int bE3() const { return 4 * HE + HE * HE; }
    PINN_HD constexpr int WG1() const { return 4 * HE + HE * HE + 1; }
    PINN_HD constexpr int bG1() const { return WG1() + HG; }
    PINN_HD constexpr int WG2() const { return WG1() + 2 * HG; }
    PINN_HD constexpr int bG2() const { return WG1() + 3 * HG; }
    PINN_HD constexpr int size() const { return WG1() + 3 * HG + 1; }
};

// ------------------------------------------------------------------ orbital
// exp(-|d|) and its gradient / Laplacian streams (poc/main.py:64-78,269-284;
// train.py:41-44).  Also returns 1/r for the Coulomb term.
template <class T>
PINN_HD void orbital(T dx, T dy, T dz, T st[5], T& rinv)
{
    T r2 = dx * dx + dy * dy + dz * dz;
    rinv = p_rsqrt(r2);
    T r = r2 * rinv;
    T f = p_exp(-r);
    T fr = -f * rinv;
    st[0] = f;
    st[1] = fr * dx;
    st[2] = fr * dy;
    st[3] = fr * dz;
    st[4] = f + T(2) * fr;  // f (1 - 2/r)
}

// The same orbital in the cylindrical frame of its point: (value, d/dx, d/drho, Laplacian) with
// rho = sqrt(y^2 + z^2).  Both orbitals' gradients lie in the plane spanned by x^ and rho^, so these four
// streams carry what the five Cartesian ones do (|grad|^2 = d_x^2 + d_rho^2; the Laplacian stream keeps
// the full 3-D Laplacian f (1 - 2/r)).  rho = 0 is regular: d/drho = -f rho / r = 0.
template <class T>
PINN_HD void orbital_cyl(T dx, T rho2, T rho, T st[4], T& rinv)
{
    T r2 = dx * dx + rho2;
    rinv = p_rsqrt(r2);
    T r = r2 * rinv;
    T f = p_exp(-r);
    T fr = -f * rinv;
    st[0] = f;
    st[1] = fr * dx;
    st[2] = fr * rho;
    st[3] = f + T(2) * fr;  // f (1 - 2/r)
}

// -------------------------------------------------------------------- trunk
// pair versions of the activation helpers: the transcendental unit works per half, everything
// else is packed arithmetic
template <int ACT, class T>
PINN_HD void act012_2(typename V2<T>::type a, typename V2<T>::type& s, typename V2<T>::type& s1,
                      typename V2<T>::type& s2)
{
    using V = typename V2<T>::type;
    if (ACT == ACT_SIGMOID) {
        const V d = make2<T>(T(1) + p_exp(-a[0]), T(1) + p_exp(-a[1]));
        s = make2<T>(p_rcp(d[0]), p_rcp(d[1]));
        s1 = s - s * s;
        s2 = s1 - T(2) * s * s1;
    } else {
        const V d = make2<T>(T(1) + p_exp(T(-2) * a[0]), T(1) + p_exp(T(-2) * a[1]));
        s = T(2) * make2<T>(p_rcp(d[0]), p_rcp(d[1])) - T(1);
        s1 = T(1) - s * s;
        s2 = T(-2) * s * s1;
    }
}

template <int ACT, class T>
PINN_HD void act123_2(typename V2<T>::type s, typename V2<T>::type& s1, typename V2<T>::type& s2,
                      typename V2<T>::type& s3)
{
    if (ACT == ACT_SIGMOID) {
        s1 = s - s * s;
        s2 = s1 - T(2) * s * s1;
        s3 = s1 - T(6) * s1 * s1;
    } else {
        s1 = T(1) - s * s;
        s2 = T(-2) * s * s1;
        s3 = s1 * (T(4) - T(6) * s1);
    }
}

// State of one symmetric pass for one point, held as neuron PAIRS.  o[l] = post-activation
// streams of layers 0..L-2 (inputs of the next layer, also the V operand of the weight
// gradient); a[l-1] = (h, a_x, a_y, a_z, a_lap) of layers 1..L-1 (what the activation adjoint
// needs).  Layer 0's pre-activations are recomputed in the reverse pass (K = 2), which keeps
// the live set at 10*H floats for L = 2.
template <int H, int L, int ACT, class T>
struct TrunkPass {
    static_assert(L >= 2, "trunk needs at least two hidden layers");
    static_assert(H % 2 == 0, "trunk width must be even");
    using Lay = TrunkLayout<H, L>;
    using V = typename V2<T>::type;
    static constexpr int HP = H / 2;
    V o[L - 1][5][HP];
    V a[L - 1][5][HP];

    // Forward.  Adds sum_j wout[j]*B_v[j] to Sv and sum_j wout[j]*B_lap[j] to Sl
    // (Lin_out, poc/main.py:261; train.py:47).
    template <class W>
    PINN_HD void forward(const W* __restrict__ th, const W* __restrict__ thT, const T (&in)[2][5],
                         T& Sv, T& Sl)
    {
        {   // ---- layer 0 (K = 2): two rank-1 updates on neuron pairs
            V A[5][HP];
            {
                V w0[HP], w1[HP], b0[HP];
#pragma unroll
                for (int p = 0; p < HP; p++) {
                    w0[p] = make2<T>(T(thT[Lay::W(0) + 2 * p]), T(thT[Lay::W(0) + 2 * p + 1]));
                    w1[p] = make2<T>(T(thT[Lay::W(0) + H + 2 * p]), T(thT[Lay::W(0) + H + 2 * p + 1]));
                    b0[p] = make2<T>(T(th[Lay::b(0) + 2 * p]), T(th[Lay::b(0) + 2 * p + 1]));
                }
                PINN_ROW_FENCE();
#pragma unroll
                for (int s = 0; s < 5; s++) {
                    const V x0 = splat2<T>(in[0][s]), x1 = splat2<T>(in[1][s]);
#pragma unroll
                    for (int p = 0; p < HP; p++) {
                        A[s][p] = w0[p] * x0 + w1[p] * x1;
                        if (s == 0) A[s][p] += b0[p];
                    }
                }
                PINN_ROW_FENCE();
            }
#pragma unroll
            for (int p = 0; p < HP; p++) {
                V s, s1, s2;
                act012_2<ACT, T>(A[0][p], s, s1, s2);
                o[0][0][p] = s;
                o[0][1][p] = s1 * A[1][p];
                o[0][2][p] = s1 * A[2][p];
                o[0][3][p] = s1 * A[3][p];
                o[0][4][p] = s1 * A[4][p] + s2 * (A[1][p] * A[1][p] + A[2][p] * A[2][p] + A[3][p] * A[3][p]);
            }
            PINN_ROW_FENCE();
        }
#pragma unroll
        for (int l = 1; l < L; l++) {
            // A[s][p] accumulates the pre-activations of neuron pair p; input i contributes the
            // rank-1 update (row i of W_l^T) x o_{l-1,s}[i].  Rows are double-buffered so the
            // scalar-load latency hides behind the 5*H/2 packed FMAs of a row.
            V A[5][HP];
#pragma unroll
            for (int p = 0; p < HP; p++) {
                A[0][p] = make2<T>(T(th[Lay::b(l) + 2 * p]), T(th[Lay::b(l) + 2 * p + 1]));
#pragma unroll
                for (int s = 1; s < 5; s++) A[s][p] = splat2<T>(T(0));
            }
            V wc[HP], wn[HP];
#pragma unroll
            for (int p = 0; p < HP; p++)
                wc[p] = make2<T>(T(thT[Lay::W(l) + 2 * p]), T(thT[Lay::W(l) + 2 * p + 1]));
            PINN_ROW_FENCE();
#pragma unroll
            for (int i = 0; i < H; i++) {
                if (i + 1 < H) {
#pragma unroll
                    for (int p = 0; p < HP; p++)
                        wn[p] = make2<T>(T(thT[Lay::W(l) + (i + 1) * H + 2 * p]),
                                         T(thT[Lay::W(l) + (i + 1) * H + 2 * p + 1]));
                }
                PINN_ROW_FENCE();
#pragma unroll
                for (int s = 0; s < 5; s++) {
                    const V xs = bcast_elem<T>(o[l - 1][s][i >> 1], i);
#pragma unroll
                    for (int p = 0; p < HP; p++) {
                        A[s][p] += wc[p] * xs;
                        PINN_PIN(A[s][p]);
                    }
                }
                PINN_ROW_FENCE();
#pragma unroll
                for (int p = 0; p < HP; p++) wc[p] = wn[p];
            }
            V wo2[HP];
            if (l == L - 1) {
#pragma unroll
                for (int p = 0; p < HP; p++)
                    wo2[p] = make2<T>(T(th[Lay::Wout + 2 * p]), T(th[Lay::Wout + 2 * p + 1]));
                PINN_ROW_FENCE();
            }
            V accv = splat2<T>(T(0)), accl = splat2<T>(T(0));
#pragma unroll
            for (int p = 0; p < HP; p++) {
                V s, s1, s2;
                act012_2<ACT, T>(A[0][p], s, s1, s2);
                const V ol = s1 * A[4][p] + s2 * (A[1][p] * A[1][p] + A[2][p] * A[2][p] + A[3][p] * A[3][p]);
                a[l - 1][0][p] = s;
                a[l - 1][1][p] = A[1][p];
                a[l - 1][2][p] = A[2][p];
                a[l - 1][3][p] = A[3][p];
                a[l - 1][4][p] = A[4][p];
                if (l < L - 1) {
                    o[l][0][p] = s;
                    o[l][1][p] = s1 * A[1][p];
                    o[l][2][p] = s1 * A[2][p];
                    o[l][3][p] = s1 * A[3][p];
                    o[l][4][p] = ol;
                } else {
                    accv += wo2[p] * s;
                    accl += wo2[p] * ol;
                }
            }
            if (l == L - 1) {
                Sv += accv[0] + accv[1];
                Sl += accl[0] + accl[1];
            }
            PINN_ROW_FENCE();
        }
    }

    // Reverse.  cv, cl = adjoints of (sum_j wout[j] B_v[j]) and (sum_j wout[j]
    // B_lap[j]) for this pass (out_scale, pass weight and N-bar folded in).
    // The sink receives, per layer, U = pre-activation adjoint streams and
    // V = that layer's input streams; it owns the reduction over points.
    template <class W, class Sink>
    PINN_HD void backward(const W* __restrict__ th, const W* __restrict__ thT, const T (&in)[2][5],
                          T cv, T cl, Sink& sink)
    {
        V ob[5][HP];  // adjoint of the current layer's post-activation streams (pairs)
        V ab[5][HP];  // adjoint of its pre-activation streams
        // ---- last hidden layer: only the value and Laplacian streams feed Lin_out
        {
            V wo2[HP];
#pragma unroll
            for (int p = 0; p < HP; p++)
                wo2[p] = make2<T>(T(th[Lay::Wout + 2 * p]), T(th[Lay::Wout + 2 * p + 1]));
            PINN_ROW_FENCE();
            T t[H];
#pragma unroll
            for (int p = 0; p < HP; p++) {
                const V s = a[L - 2][0][p], ax = a[L - 2][1][p], ay = a[L - 2][2][p], az = a[L - 2][3][p],
                        al = a[L - 2][4][p];
                V s1, s2, s3;
                act123_2<ACT, T>(s, s1, s2, s3);
                const V q = ax * ax + ay * ay + az * az;
                const V tt = cv * s + cl * (s1 * al + s2 * q);  // d/d wout
                t[2 * p] = tt[0];
                t[2 * p + 1] = tt[1];
                const V o0 = cv * wo2[p], ol = cl * wo2[p];
                const V u = T(2) * s2 * ol;
                ab[4][p] = s1 * ol;
                ab[1][p] = u * ax;
                ab[2][p] = u * ay;
                ab[3][p] = u * az;
                ab[0][p] = s1 * o0 + (s2 * al + s3 * q) * ol;
            }
            PINN_ROW_FENCE();
            sink.out_weight(t);
        }
#pragma unroll
        for (int l = L - 1; l >= 1; l--) {
            if (l < L - 1) {  // middle layer: full five-stream adjoint
#pragma unroll
                for (int p = 0; p < HP; p++) {
                    const V s = a[l - 1][0][p], ax = a[l - 1][1][p], ay = a[l - 1][2][p],
                            az = a[l - 1][3][p], al = a[l - 1][4][p];
                    V s1, s2, s3;
                    act123_2<ACT, T>(s, s1, s2, s3);
                    const V q = ax * ax + ay * ay + az * az;
                    const V u = T(2) * s2 * ob[4][p];
                    ab[4][p] = s1 * ob[4][p];
                    ab[1][p] = s1 * ob[1][p] + u * ax;
                    ab[2][p] = s1 * ob[2][p] + u * ay;
                    ab[3][p] = s1 * ob[3][p] + u * az;
                    ab[0][p] = s1 * ob[0][p] + s2 * (ax * ob[1][p] + ay * ob[2][p] + az * ob[3][p]) +
                               (s2 * al + s3 * q) * ob[4][p];
                }
            }
#pragma unroll
            for (int s = 0; s < 5; s++) {
                // W_l-bar += ab_s (x) o_{l-1,s}; the s == 0 call also carries b_l-bar += ab_0
                T us[H], vs[H];
#pragma unroll
                for (int j = 0; j < H; j++) { us[j] = ab[s][j >> 1][j & 1]; vs[j] = o[l - 1][s][j >> 1][j & 1]; }
                sink.weight(l, s, us, vs);
            }
            // adjoint of the layer input: ob_s = W_l^T ab_s as rank-1 updates on input-neuron
            // pairs: row j of W_l (contiguous, double-buffered scalar loads) x ab_s[j]
#pragma unroll
            for (int s = 0; s < 5; s++)
#pragma unroll
                for (int p = 0; p < HP; p++) ob[s][p] = splat2<T>(T(0));
            V wc[HP], wn[HP];
#pragma unroll
            for (int p = 0; p < HP; p++)
                wc[p] = make2<T>(T(th[Lay::W(l) + 2 * p]), T(th[Lay::W(l) + 2 * p + 1]));
            PINN_ROW_FENCE();
#pragma unroll
            for (int j = 0; j < H; j++) {
                if (j + 1 < H) {
#pragma unroll
                    for (int p = 0; p < HP; p++)
                        wn[p] = make2<T>(T(th[Lay::W(l) + (j + 1) * H + 2 * p]),
                                         T(th[Lay::W(l) + (j + 1) * H + 2 * p + 1]));
                }
                PINN_ROW_FENCE();
#pragma unroll
                for (int s = 0; s < 5; s++) {
                    const V xs = bcast_elem<T>(ab[s][j >> 1], j);
#pragma unroll
                    for (int p = 0; p < HP; p++) {
                        ob[s][p] += wc[p] * xs;
                        PINN_PIN(ob[s][p]);
                    }
                }
                PINN_ROW_FENCE();
#pragma unroll
                for (int p = 0; p < HP; p++) wc[p] = wn[p];
            }
        }
        // ---- layer 0: recompute its pre-activation derivative streams (K = 2)
        {
            V w0[HP], w1[HP];
#pragma unroll
            for (int p = 0; p < HP; p++) {
                w0[p] = make2<T>(T(thT[Lay::W(0) + 2 * p]), T(thT[Lay::W(0) + 2 * p + 1]));
                w1[p] = make2<T>(T(thT[Lay::W(0) + H + 2 * p]), T(thT[Lay::W(0) + H + 2 * p + 1]));
            }
            PINN_ROW_FENCE();
#pragma unroll
            for (int p = 0; p < HP; p++) {
                const V ax = w0[p] * in[0][1] + w1[p] * in[1][1];
                const V ay = w0[p] * in[0][2] + w1[p] * in[1][2];
                const V az = w0[p] * in[0][3] + w1[p] * in[1][3];
                const V al = w0[p] * in[0][4] + w1[p] * in[1][4];
                V s1, s2, s3;
                act123_2<ACT, T>(o[0][0][p], s1, s2, s3);
                const V q = ax * ax + ay * ay + az * az;
                const V u = T(2) * s2 * ob[4][p];
                ab[4][p] = s1 * ob[4][p];
                ab[1][p] = s1 * ob[1][p] + u * ax;
                ab[2][p] = s1 * ob[2][p] + u * ay;
                ab[3][p] = s1 * ob[3][p] + u * az;
                ab[0][p] = s1 * ob[0][p] + s2 * (ax * ob[1][p] + ay * ob[2][p] + az * ob[3][p]) +
                           (s2 * al + s3 * q) * ob[4][p];
            }
            PINN_ROW_FENCE();
        }
#pragma unroll
        for (int s = 0; s < 5; s++) {
            T us[H];
#pragma unroll
            for (int j = 0; j < H; j++) us[j] = ab[s][j >> 1][j & 1];
            sink.weight0(s, us, in[0][s], in[1][s]);  // s == 0: + bias
        }
    }
};

// ------------------------------------------------------------- scalar tail
// psi assembly, Hamiltonian residual and loss seeds for one point
// (poc/main.py:265-266,118-120,345-351; train.py:53-57).
template <class T>
struct Tail {
    T psi, lap, res;     // forward results
    T Nb, lapNb;         // adjoints of N and lap N
    T Eb, gb;            // adjoints of E and of the gate value
    T l_pde, l_bc1, l_bc2;  // this point's scaled loss contributions
};

// keeps hipcc's SLP vectorizer from pairing a value with its neighbour into one packed instruction (device, fp32
// only; a no-op elsewhere).  Used where the pairing produces v_pk_mul_f32 ... op_sel:[0,1] -- see
// profiles/r02_packed_opsel.md.
template <class T>
PINN_HD void slp_opaque(T& x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(x));
#else
    (void)x;
#endif
}

template <class T>
PINN_HD void tail(T N, T lapN, T g, T E, T F, T lapF, T U, T a, T b, T c, T m1, T m2, T invN,
                  T invN1, T invN2, T live, Tail<T>& t)
{
    t.psi = g * N + F;
    t.lap = g * lapN + lapF;
    T Vc = b * U + c * E;
    slp_opaque(Vc);
    t.res = a * t.lap + Vc * t.psi;
    T resb = T(2) * t.res * invN * live;
    T bcw = (m1 * invN1 + m2 * invN2) * live;
    T psib = resb * Vc + T(2) * t.psi * bcw;
    T lapb = a * resb;
    t.Eb = c * t.psi * resb;
    t.Nb = g * psib;
    t.lapNb = g * lapb;
    t.gb = N * psib + lapN * lapb;
    t.l_pde = t.res * t.res * invN * live;
    t.l_bc1 = t.psi * t.psi * m1 * invN1 * live;
    t.l_bc2 = t.psi * t.psi * m2 * invN2 * live;
}

// ------------------------------------------------------ E-net and gate (R only)
// E(R): 1 -> HE -> HE -> 1 (poc/main.py:249-253; train.py:50-52)
// g(R): 1 -> HG -> 1       (poc/main.py:262-264; train.py:48-49)
template <int HE, int HG, int ACT, class T>
struct Rnet {
    T h1[HE], h2[HE], hg[HG];

    static constexpr int CH = HE % 16 == 0 ? 16 : HE;  // scalar-load chunk (one s_load_dwordx16)
    static constexpr int NCH = HE / CH;                // chunks per weight row

    template <class W>
    PINN_HD void forward(const W* __restrict__ th, const W* __restrict__ thT, T R, T& E, T& g)
    {
        constexpr RnetLayout Lo{HE, HG};
        T s1, s2;
        // first E layer and the gate: small vectors, fetched in batches of 16 scalars
#pragma unroll
        for (int c0 = 0; c0 < HE; c0 += CH) {
            T w1[CH], b1[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) { w1[j] = T(th[Lo.WE1() + c0 + j]); b1[j] = T(th[Lo.bE1() + c0 + j]); }
            PINN_ROW_FENCE();
#pragma unroll
            for (int j = 0; j < CH; j++) act012<ACT>(w1[j] * R + b1[j], h1[c0 + j], s1, s2);
            PINN_ROW_FENCE();
        }
        {
            T wg1[HG], bg1[HG], wg2[HG];
#pragma unroll
            for (int j = 0; j < HG; j++) {
                wg1[j] = T(th[Lo.WG1() + j]); bg1[j] = T(th[Lo.bG1() + j]); wg2[j] = T(th[Lo.WG2() + j]);
            }
            T gg = T(th[Lo.bG2()]);
            PINN_ROW_FENCE();
#pragma unroll
            for (int j = 0; j < HG; j++) {
                act012<ACT>(wg1[j] * R + bg1[j], hg[j], s1, s2);
                gg += wg2[j] * hg[j];
            }
            g = gg;
            PINN_ROW_FENCE();
        }
        // second E layer: pre-activations of output-neuron pairs accumulate rank-1 updates
        // (row i of WE2^T) x h1[i]; rows are walked in 16-float chunks, double-buffered
        using V = typename V2<T>::type;
        constexpr int CP = CH / 2;
        static_assert(CH % 2 == 0, "E-net width must be even");
        V A2[HE / 2];
#pragma unroll
        for (int p = 0; p < HE / 2; p++)
            A2[p] = make2<T>(T(th[Lo.bE2() + 2 * p]), T(th[Lo.bE2() + 2 * p + 1]));
        V wc[CP], wn[CP];
#pragma unroll
        for (int p = 0; p < CP; p++)
            wc[p] = make2<T>(T(thT[Lo.WE2() + 2 * p]), T(thT[Lo.WE2() + 2 * p + 1]));
        PINN_ROW_FENCE();
#pragma unroll
        for (int q = 0; q < HE * NCH; q++) {
            const int i = q / NCH, part = q % NCH;
            if (q + 1 < HE * NCH) {
#pragma unroll
                for (int p = 0; p < CP; p++)
                    wn[p] = make2<T>(T(thT[Lo.WE2() + (q + 1) * CH + 2 * p]),
                                     T(thT[Lo.WE2() + (q + 1) * CH + 2 * p + 1]));
            }
            PINN_ROW_FENCE();
            const V xs = splat2<T>(h1[i]);
#pragma unroll
            for (int p = 0; p < CP; p++) {
                A2[part * CP + p] += wc[p] * xs;
                PINN_PIN(A2[part * CP + p]);
            }
            PINN_ROW_FENCE();
#pragma unroll
            for (int p = 0; p < CP; p++) wc[p] = wn[p];
        }
        T e = T(th[Lo.bE3()]);
#pragma unroll
        for (int c0 = 0; c0 < HE; c0 += CH) {
            T w3[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) w3[j] = T(th[Lo.WE3() + c0 + j]);
            PINN_ROW_FENCE();
#pragma unroll
            for (int j = 0; j < CH; j++) {
                act012<ACT>(A2[(c0 + j) >> 1][(c0 + j) & 1], h2[c0 + j], s1, s2);
                e += w3[j] * h2[c0 + j];
            }
            PINN_ROW_FENCE();
        }
        E = e;
    }

    // after forward(): push Eb, gb back to the parameters through the sink
    template <class W, class Sink>
    PINN_HD void backward(const W* __restrict__ th, T R, T Eb, T gb, Sink& sink)
    {
        constexpr RnetLayout Lo{HE, HG};
        T a2[HE], t3[HE];
#pragma unroll
        for (int c0 = 0; c0 < HE; c0 += CH) {
            T w3[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) w3[j] = T(th[Lo.WE3() + c0 + j]);
            PINN_ROW_FENCE();
#pragma unroll
            for (int j = 0; j < CH; j++) {
                t3[c0 + j] = Eb * h2[c0 + j];
                a2[c0 + j] = Eb * w3[j] * act1_from_value<ACT>(h2[c0 + j]);
            }
            PINN_ROW_FENCE();
        }
        sink.e3(t3, Eb);          // WE3-bar, bE3-bar
        sink.e2(a2, h1);          // WE2-bar = a2 (x) h1, bE2-bar = a2
        T a1[HE];
        {   // a1 = WE2^T a2: rank-1 updates on input-neuron pairs, natural rows of WE2 in
            // double-buffered 16-float chunks
            using V = typename V2<T>::type;
            constexpr int CP = CH / 2;
            V A1[HE / 2];
#pragma unroll
            for (int p = 0; p < HE / 2; p++) A1[p] = splat2<T>(T(0));
            V wc[CP], wn[CP];
#pragma unroll
            for (int p = 0; p < CP; p++)
                wc[p] = make2<T>(T(th[Lo.WE2() + 2 * p]), T(th[Lo.WE2() + 2 * p + 1]));
            PINN_ROW_FENCE();
#pragma unroll
            for (int q = 0; q < HE * NCH; q++) {
                const int j = q / NCH, part = q % NCH;
                if (q + 1 < HE * NCH) {
#pragma unroll
                    for (int p = 0; p < CP; p++)
                        wn[p] = make2<T>(T(th[Lo.WE2() + (q + 1) * CH + 2 * p]),
                                         T(th[Lo.WE2() + (q + 1) * CH + 2 * p + 1]));
                }
                PINN_ROW_FENCE();
                const V xs = splat2<T>(a2[j]);
#pragma unroll
                for (int p = 0; p < CP; p++) {
                    A1[part * CP + p] += wc[p] * xs;
                    PINN_PIN(A1[part * CP + p]);
                }
                PINN_ROW_FENCE();
#pragma unroll
                for (int p = 0; p < CP; p++) wc[p] = wn[p];
            }
#pragma unroll
            for (int i = 0; i < HE; i++) a1[i] = A1[i >> 1][i & 1] * act1_from_value<ACT>(h1[i]);
        }
        sink.e1(a1, R);           // WE1-bar = a1*R, bE1-bar = a1
        T tg[HG], ag[HG];
        {
            T wg2[HG];
#pragma unroll
            for (int j = 0; j < HG; j++) wg2[j] = T(th[Lo.WG2() + j]);
            PINN_ROW_FENCE();
#pragma unroll
            for (int j = 0; j < HG; j++) {
                tg[j] = gb * hg[j];
                ag[j] = gb * wg2[j] * act1_from_value<ACT>(hg[j]);
            }
            PINN_ROW_FENCE();
        }
        sink.g2(tg, gb);          // WG2-bar, bG2-bar
        sink.g1(ag, R);           // WG1-bar = ag*R, bG1-bar = ag
    }
};

}  // namespace pinn
