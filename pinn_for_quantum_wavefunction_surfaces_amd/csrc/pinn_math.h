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
#else
#define PINN_ROW_FENCE() ((void)0)
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
PINN_HD double p_exp(double x) { return std::exp(x); }
PINN_HD double p_rcp(double x) { return 1.0 / x; }
PINN_HD double p_rsqrt(double x) { return 1.0 / std::sqrt(x); }

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
        s = T(2) * p_rcp(T(1) + p_exp(T(-2) * a)) - T(1);
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

// -------------------------------------------------------------------- trunk
// State of one symmetric pass for one point.  o[l] = post-activation streams
// of layers 0..L-2 (inputs of the next layer, also the V operand of the weight
// gradient); a[l-1] = (h, a_x, a_y, a_z, a_lap) of layers 1..L-1 (what the
// activation adjoint needs).  Layer 0's pre-activations are recomputed in the
// reverse pass (K = 2), which keeps the live set at 10*H floats for L = 2.
template <int H, int L, int ACT, class T>
struct TrunkPass {
    static_assert(L >= 2, "trunk needs at least two hidden layers");
    using Lay = TrunkLayout<H, L>;
    T o[L - 1][5][H];
    T a[L - 1][5][H];

    // Forward.  Adds sum_j wout[j]*B_v[j] to Sv and sum_j wout[j]*B_lap[j] to Sl
    // (Lin_out, poc/main.py:261; train.py:47).
    template <class W>
    PINN_HD void forward(const W* __restrict__ th, const T (&in)[2][5], T& Sv, T& Sl)
    {
#pragma unroll
        for (int j = 0; j < H; j++) {
            T w0 = T(th[Lay::W(0) + 2 * j]), w1 = T(th[Lay::W(0) + 2 * j + 1]);
            T a0 = w0 * in[0][0] + w1 * in[1][0] + T(th[Lay::b(0) + j]);
            T ax = w0 * in[0][1] + w1 * in[1][1];
            T ay = w0 * in[0][2] + w1 * in[1][2];
            T az = w0 * in[0][3] + w1 * in[1][3];
            T al = w0 * in[0][4] + w1 * in[1][4];
            T s, s1, s2;
            act012<ACT>(a0, s, s1, s2);
            o[0][0][j] = s;
            o[0][1][j] = s1 * ax;
            o[0][2][j] = s1 * ay;
            o[0][3][j] = s1 * az;
            o[0][4][j] = s1 * al + s2 * (ax * ax + ay * ay + az * az);
            PINN_ROW_FENCE();
        }
#pragma unroll
        for (int l = 1; l < L; l++) {
#pragma unroll
            for (int j = 0; j < H; j++) {
                const W* __restrict__ wr = th + Lay::W(l) + j * H;
                T c0 = T(th[Lay::b(l) + j]), cx = T(0), cy = T(0), cz = T(0), cl = T(0);
#pragma unroll
                for (int i = 0; i < H; i++) {
                    T w = T(wr[i]);
                    c0 += w * o[l - 1][0][i];
                    cx += w * o[l - 1][1][i];
                    cy += w * o[l - 1][2][i];
                    cz += w * o[l - 1][3][i];
                    cl += w * o[l - 1][4][i];
                }
                T s, s1, s2;
                act012<ACT>(c0, s, s1, s2);
                a[l - 1][0][j] = s;
                a[l - 1][1][j] = cx;
                a[l - 1][2][j] = cy;
                a[l - 1][3][j] = cz;
                a[l - 1][4][j] = cl;
                T ol = s1 * cl + s2 * (cx * cx + cy * cy + cz * cz);
                if (l < L - 1) {
                    o[l][0][j] = s;
                    o[l][1][j] = s1 * cx;
                    o[l][2][j] = s1 * cy;
                    o[l][3][j] = s1 * cz;
                    o[l][4][j] = ol;
                } else {
                    T wo = T(th[Lay::Wout + j]);
                    Sv += wo * s;
                    Sl += wo * ol;
                }
                PINN_ROW_FENCE();
            }
        }
    }

    // Reverse.  cv, cl = adjoints of (sum_j wout[j] B_v[j]) and (sum_j wout[j]
    // B_lap[j]) for this pass (out_scale, pass weight and N-bar folded in).
    // The sink receives, per layer, U = pre-activation adjoint streams and
    // V = that layer's input streams; it owns the reduction over points.
    template <class W, class Sink>
    PINN_HD void backward(const W* __restrict__ th, const T (&in)[2][5], T cv, T cl, Sink& sink)
    {
        T ob[5][H];  // adjoint of the current layer's post-activation streams
        T ab[5][H];  // adjoint of its pre-activation streams
        // ---- last hidden layer: only the value and Laplacian streams feed Lin_out
        {
            T t[H];
#pragma unroll
            for (int j = 0; j < H; j++) {
                T s = a[L - 2][0][j], ax = a[L - 2][1][j], ay = a[L - 2][2][j],
                  az = a[L - 2][3][j], al = a[L - 2][4][j];
                T s1, s2, s3;
                act123<ACT>(s, s1, s2, s3);
                T q = ax * ax + ay * ay + az * az;
                t[j] = cv * s + cl * (s1 * al + s2 * q);  // d/d wout[j]
                T wo = T(th[Lay::Wout + j]);
                T o0 = cv * wo, ol = cl * wo;
                T u = T(2) * s2 * ol;
                ab[4][j] = s1 * ol;
                ab[1][j] = u * ax;
                ab[2][j] = u * ay;
                ab[3][j] = u * az;
                ab[0][j] = s1 * o0 + (s2 * al + s3 * q) * ol;
                PINN_ROW_FENCE();
            }
            sink.out_weight(t);
        }
#pragma unroll
        for (int l = L - 1; l >= 1; l--) {
            if (l < L - 1) {  // middle layer: full five-stream adjoint
#pragma unroll
                for (int j = 0; j < H; j++) {
                    T s = a[l - 1][0][j], ax = a[l - 1][1][j], ay = a[l - 1][2][j],
                      az = a[l - 1][3][j], al = a[l - 1][4][j];
                    T s1, s2, s3;
                    act123<ACT>(s, s1, s2, s3);
                    T q = ax * ax + ay * ay + az * az;
                    T o0 = ob[0][j], ox = ob[1][j], oy = ob[2][j], oz = ob[3][j], ol = ob[4][j];
                    T u = T(2) * s2 * ol;
                    ab[4][j] = s1 * ol;
                    ab[1][j] = s1 * ox + u * ax;
                    ab[2][j] = s1 * oy + u * ay;
                    ab[3][j] = s1 * oz + u * az;
                    ab[0][j] = s1 * o0 + s2 * (ax * ox + ay * oy + az * oz) + (s2 * al + s3 * q) * ol;
                }
            }
#pragma unroll
            for (int s = 0; s < 5; s++) {
                // W_l-bar += ab_s (x) o_{l-1,s}; the s == 0 call also carries b_l-bar += ab_0
                sink.weight(l, s, ab[s], o[l - 1][s]);
            }
            // adjoint of the layer input: ob_s = W_l^T ab_s, walked row by row so that
            // each weight row is one contiguous scalar load feeding 5*H FMAs
#pragma unroll
            for (int s = 0; s < 5; s++)
#pragma unroll
                for (int i = 0; i < H; i++) ob[s][i] = T(0);
#pragma unroll
            for (int j = 0; j < H; j++) {
                const W* __restrict__ wr = th + Lay::W(l) + j * H;
#pragma unroll
                for (int i = 0; i < H; i++) {
                    T w = T(wr[i]);
#pragma unroll
                    for (int s = 0; s < 5; s++) ob[s][i] += w * ab[s][j];
                }
                PINN_ROW_FENCE();
            }
        }
        // ---- layer 0: recompute its pre-activation derivative streams (K = 2)
#pragma unroll
        for (int j = 0; j < H; j++) {
            T w0 = T(th[Lay::W(0) + 2 * j]), w1 = T(th[Lay::W(0) + 2 * j + 1]);
            T ax = w0 * in[0][1] + w1 * in[1][1];
            T ay = w0 * in[0][2] + w1 * in[1][2];
            T az = w0 * in[0][3] + w1 * in[1][3];
            T al = w0 * in[0][4] + w1 * in[1][4];
            T s = o[0][0][j], s1, s2, s3;
            act123<ACT>(s, s1, s2, s3);
            T q = ax * ax + ay * ay + az * az;
            T o0 = ob[0][j], ox = ob[1][j], oy = ob[2][j], oz = ob[3][j], ol = ob[4][j];
            T u = T(2) * s2 * ol;
            ab[4][j] = s1 * ol;
            ab[1][j] = s1 * ox + u * ax;
            ab[2][j] = s1 * oy + u * ay;
            ab[3][j] = s1 * oz + u * az;
            ab[0][j] = s1 * o0 + s2 * (ax * ox + ay * oy + az * oz) + (s2 * al + s3 * q) * ol;
        }
#pragma unroll
        for (int s = 0; s < 5; s++) sink.weight0(s, ab[s], in[0][s], in[1][s]);  // s == 0: + bias
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

template <class T>
PINN_HD void tail(T N, T lapN, T g, T E, T F, T lapF, T U, T a, T b, T c, T m1, T m2, T invN,
                  T invN1, T invN2, T live, Tail<T>& t)
{
    t.psi = g * N + F;
    t.lap = g * lapN + lapF;
    T Vc = b * U + c * E;
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

    template <class W>
    PINN_HD void forward(const W* __restrict__ th, T R, T& E, T& g)
    {
        const RnetLayout Lo{HE, HG};
        T s1, s2;
#pragma unroll
        for (int j = 0; j < HE; j++)
            act012<ACT>(T(th[Lo.WE1() + j]) * R + T(th[Lo.bE1() + j]), h1[j], s1, s2);
        T e = T(th[Lo.bE3()]);
#pragma unroll
        for (int j = 0; j < HE; j++) {
            T acc = T(th[Lo.bE2() + j]);
#pragma unroll
            for (int i = 0; i < HE; i++) acc += T(th[Lo.WE2() + j * HE + i]) * h1[i];
            act012<ACT>(acc, h2[j], s1, s2);
            e += T(th[Lo.WE3() + j]) * h2[j];
            PINN_ROW_FENCE();
        }
        E = e;
        T gg = T(th[Lo.bG2()]);
#pragma unroll
        for (int j = 0; j < HG; j++) {
            act012<ACT>(T(th[Lo.WG1() + j]) * R + T(th[Lo.bG1() + j]), hg[j], s1, s2);
            gg += T(th[Lo.WG2() + j]) * hg[j];
        }
        g = gg;
    }

    // after forward(): push Eb, gb back to the parameters through the sink
    template <class W, class Sink>
    PINN_HD void backward(const W* __restrict__ th, T R, T Eb, T gb, Sink& sink)
    {
        const RnetLayout Lo{HE, HG};
        T a2[HE], t3[HE];
#pragma unroll
        for (int j = 0; j < HE; j++) {
            t3[j] = Eb * h2[j];
            a2[j] = Eb * T(th[Lo.WE3() + j]) * act1_from_value<ACT>(h2[j]);
        }
        sink.e3(t3, Eb);          // WE3-bar, bE3-bar
        sink.e2(a2, h1);          // WE2-bar = a2 (x) h1, bE2-bar = a2
        T a1[HE];
#pragma unroll
        for (int i = 0; i < HE; i++) a1[i] = T(0);
#pragma unroll
        for (int j = 0; j < HE; j++) {  // row-major walk of WE2 (contiguous scalar loads)
#pragma unroll
            for (int i = 0; i < HE; i++) a1[i] += T(th[Lo.WE2() + j * HE + i]) * a2[j];
            PINN_ROW_FENCE();
        }
#pragma unroll
        for (int i = 0; i < HE; i++) a1[i] *= act1_from_value<ACT>(h1[i]);
        sink.e1(a1, R);           // WE1-bar = a1*R, bE1-bar = a1
        T tg[HG], ag[HG];
#pragma unroll
        for (int j = 0; j < HG; j++) {
            tg[j] = gb * hg[j];
            ag[j] = gb * T(th[Lo.WG2() + j]) * act1_from_value<ACT>(hg[j]);
        }
        sink.g2(tg, gb);          // WG2-bar, bG2-bar
        sink.g1(ag, R);           // WG1-bar = ag*R, bG1-bar = ag
    }
};

}  // namespace pinn
