/*
 * oracle/pinn_oracle.c -- TEST INFRASTRUCTURE ONLY (see pinn_oracle.h).
 *
 * fp64 CPU restatement of the reference hot path.  The reference obtains
 * d2psi/dx2 + d2psi/dy2 + d2psi/dz2 by nested torch.autograd.grad calls
 * (train.py:8-10, poc/main.py:82-97) and the parameter gradient by a third
 * autograd sweep (train.py:71).  Points are independent, so the same numbers
 * follow in closed form by carrying five streams (value, d/dx, d/dy, d/dz,
 * Laplacian) through every neuron and reversing that computation by hand.
 */
#include "pinn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ layout */

typedef struct {
    int64_t W[8], b[8]; /* trunk hidden layers, up to 8 */
    int64_t Wout, bout;
    int64_t WE1, bE1, WE2, bE2, WE3, bE3;
    int64_t WG1, bG1, WG2, bG2;
    int64_t P;
} layout_t;

static void make_layout(const oracle_arch *A, layout_t *Lo)
{
    int64_t o = 0;
    int H = A->H;
    for (int l = 0; l < A->L; l++) {
        int in = l == 0 ? 2 : H;
        Lo->W[l] = o; o += (int64_t)H * in;
        Lo->b[l] = o; o += H;
    }
    Lo->Wout = o; o += H;
    Lo->bout = o; o += 1;
    Lo->WE1 = o; o += A->HE;
    Lo->bE1 = o; o += A->HE;
    Lo->WE2 = o; o += (int64_t)A->HE * A->HE;
    Lo->bE2 = o; o += A->HE;
    Lo->WE3 = o; o += A->HE;
    Lo->bE3 = o; o += 1;
    Lo->WG1 = o; o += A->HG;
    Lo->bG1 = o; o += A->HG;
    Lo->WG2 = o; o += A->HG;
    Lo->bG2 = o; o += 1;
    Lo->P = o;
}

int64_t oracle_param_count(const oracle_arch *A)
{
    layout_t Lo;
    make_layout(A, &Lo);
    return Lo.P;
}

int32_t oracle_tensor_count(const oracle_arch *A) { return 2 * (A->L + 1) + 6 + 4; }

void oracle_tensor_layout(const oracle_arch *A, int64_t *off, int64_t *sz)
{
    layout_t Lo;
    make_layout(A, &Lo);
    int t = 0, H = A->H;
    for (int l = 0; l < A->L; l++) {
        off[t] = Lo.W[l]; sz[t++] = (int64_t)H * (l == 0 ? 2 : H);
        off[t] = Lo.b[l]; sz[t++] = H;
    }
    off[t] = Lo.Wout; sz[t++] = H;
    off[t] = Lo.bout; sz[t++] = 1;
    off[t] = Lo.WE1; sz[t++] = A->HE;
    off[t] = Lo.bE1; sz[t++] = A->HE;
    off[t] = Lo.WE2; sz[t++] = (int64_t)A->HE * A->HE;
    off[t] = Lo.bE2; sz[t++] = A->HE;
    off[t] = Lo.WE3; sz[t++] = A->HE;
    off[t] = Lo.bE3; sz[t++] = 1;
    off[t] = Lo.WG1; sz[t++] = A->HG;
    off[t] = Lo.bG1; sz[t++] = A->HG;
    off[t] = Lo.WG2; sz[t++] = A->HG;
    off[t] = Lo.bG2; sz[t++] = 1;
}

/* -------------------------------------------------------------- activation */

/* value and first three derivatives of the hidden activation.
 * sigmoid (poc/main.py:231, train.py:45): s1 = s(1-s), s2 = s1(1-2s), s3 = s1(1-6 s1)
 * tanh:                                    s1 = 1-t^2,  s2 = -2 t s1,  s3 = s1(6t^2-2) */
static void act4(int kind, double a, double *s, double *s1, double *s2, double *s3)
{
    if (kind == 0) {
        double v = 1.0 / (1.0 + exp(-a));
        double d1 = v * (1.0 - v);
        *s = v; *s1 = d1; *s2 = d1 * (1.0 - 2.0 * v); *s3 = d1 * (1.0 - 6.0 * d1);
    } else {
        double t = tanh(a);
        double d1 = 1.0 - t * t;
        *s = t; *s1 = d1; *s2 = -2.0 * t * d1; *s3 = d1 * (6.0 * t * t - 2.0);
    }
}

/* ---------------------------------------------------------- per-point work */

typedef struct {
    /* per pass p, layer l, stream s (0 value, 1..3 gradient, 4 Laplacian):
     * a = pre-activation streams, o = post-activation streams */
    double *a, *o;   /* [2][L][5][H] */
    double in[2][2][5]; /* [pass][input c][stream] */
    double hE1[64 * 8], hE2[64 * 8], hG[64 * 8];
    double r1, r2, f1, f2, lapf1, lapf2;
    double N, lapN, g, E, psi, lap, res, U;
} work_t;

#define IDX(p, l, s) ((((size_t)(p) * A->L + (l)) * 5 + (s)) * A->H)

/* exp(-|d|) with streams; sx = d(d_x)/dx = +-1 (atomicUnit(-x), poc/main.py:256) */
static void orbital(double dx, double dy, double dz, double sx, double *r_out, double st[5])
{
    double r = sqrt(dx * dx + dy * dy + dz * dz);
    double f = exp(-r);
    st[0] = f;
    st[1] = -f * sx * dx / r;
    st[2] = -f * dy / r;
    st[3] = -f * dz / r;
    st[4] = f * (1.0 - 2.0 / r);
    *r_out = r;
}

static void trunk_forward(const oracle_arch *A, const layout_t *Lo, const double *th,
                          work_t *w, int p)
{
    int H = A->H;
    for (int l = 0; l < A->L; l++) {
        int K = l == 0 ? 2 : H;
        const double *W = th + Lo->W[l], *b = th + Lo->b[l];
        for (int j = 0; j < H; j++) {
            double acc[5];
            for (int s = 0; s < 5; s++) {
                double t = s == 0 ? b[j] : 0.0;
                for (int i = 0; i < K; i++) {
                    double xin = l == 0 ? w->in[p][i][s] : w->o[IDX(p, l - 1, s) + i];
                    t += W[(size_t)j * K + i] * xin;
                }
                acc[s] = t;
                w->a[IDX(p, l, s) + j] = t;
            }
            double s0, s1, s2, s3;
            act4(A->act, acc[0], &s0, &s1, &s2, &s3);
            double q = acc[1] * acc[1] + acc[2] * acc[2] + acc[3] * acc[3];
            w->o[IDX(p, l, 0) + j] = s0;
            w->o[IDX(p, l, 1) + j] = s1 * acc[1];
            w->o[IDX(p, l, 2) + j] = s1 * acc[2];
            w->o[IDX(p, l, 3) + j] = s1 * acc[3];
            w->o[IDX(p, l, 4) + j] = s1 * acc[4] + s2 * q;
        }
    }
}

static void point_forward(const oracle_arch *A, const oracle_ham *Hm, const layout_t *Lo,
                          const double *th, double x, double y, double z, double R, work_t *w)
{
    int H = A->H, HE = A->HE, HG = A->HG;
    double st1[5], st2[5], r;
    /* poc/main.py:269-284 (atomicUnit), train.py:41-44 */
    orbital(x - R, y - Hm->Ry, z - Hm->Rz, 1.0, &w->r1, st1);
    orbital(x + R, y + Hm->Ry, z + Hm->Rz, 1.0, &w->r2, st2);
    w->f1 = st1[0]; w->f2 = st2[0]; w->lapf1 = st1[4]; w->lapf2 = st2[4];
    memcpy(w->in[0][0], st1, sizeof st1);
    memcpy(w->in[0][1], st2, sizeof st2);
    double sumB = 0.0, sumLB = 0.0;
    const double *wo = th + Lo->Wout;
    for (int p = 0; p < A->n_pass; p++) {
        if (p == 1) { /* atomicUnit(-x, y, z, R): poc/main.py:256 */
            orbital(-x - R, y - Hm->Ry, z - Hm->Rz, -1.0, &r, w->in[1][0]);
            orbital(-x + R, y + Hm->Ry, z + Hm->Rz, -1.0, &r, w->in[1][1]);
        }
        trunk_forward(A, Lo, th, w, p);
        double pw = p == 0 ? 1.0 : Hm->P; /* poc/main.py:260 */
        for (int j = 0; j < H; j++) {
            sumB += pw * wo[j] * w->o[IDX(p, A->L - 1, 0) + j];
            sumLB += pw * wo[j] * w->o[IDX(p, A->L - 1, 4) + j];
        }
    }
    w->N = A->out_scale * sumB + th[Lo->bout]; /* poc/main.py:261, train.py:47 */
    w->lapN = A->out_scale * sumLB;
    /* gate, poc/main.py:262-264; train.py:48-49 */
    double g = th[Lo->bG2];
    for (int j = 0; j < HG; j++) {
        double s0, s1, s2, s3;
        act4(A->act, th[Lo->WG1 + j] * R + th[Lo->bG1 + j], &s0, &s1, &s2, &s3);
        w->hG[j] = s0;
        g += th[Lo->WG2 + j] * s0;
    }
    w->g = g;
    /* E-net, poc/main.py:249-253; train.py:50-52 */
    for (int j = 0; j < HE; j++) {
        double s0, s1, s2, s3;
        act4(A->act, th[Lo->WE1 + j] * R + th[Lo->bE1 + j], &s0, &s1, &s2, &s3);
        w->hE1[j] = s0;
    }
    double E = th[Lo->bE3];
    for (int j = 0; j < HE; j++) {
        double t = th[Lo->bE2 + j], s0, s1, s2, s3;
        for (int i = 0; i < HE; i++) t += th[Lo->WE2 + (size_t)j * HE + i] * w->hE1[i];
        act4(A->act, t, &s0, &s1, &s2, &s3);
        w->hE2[j] = s0;
        E += th[Lo->WE3 + j] * s0;
    }
    w->E = E;
    /* psi = NN*f + N_LCAO (poc/main.py:265-266, 292); g, E depend on R only */
    w->psi = g * w->N + w->f1 + Hm->P * w->f2;
    w->lap = g * w->lapN + w->lapf1 + Hm->P * w->lapf2;
    w->U = 1.0 / w->r1 + 1.0 / w->r2;
    /* poc/main.py:118-120,345: -1/2 lap + V psi - E psi; train.py:54 */
    w->res = Hm->a * w->lap + (Hm->b * w->U + Hm->c * E) * w->psi;
}

static work_t *work_new(const oracle_arch *A)
{
    work_t *w = (work_t *)calloc(1, sizeof *w);
    size_t n = (size_t)2 * A->L * 5 * A->H;
    w->a = (double *)calloc(n, sizeof(double));
    w->o = (double *)calloc(n, sizeof(double));
    return w;
}

static void work_free(work_t *w)
{
    free(w->a); free(w->o); free(w);
}

void oracle_forward(const oracle_arch *A, const oracle_ham *Hm, const double *theta,
                    const double *x, const double *y, const double *z, const double *R,
                    int64_t n, double *psi, double *E, double *lap, double *res)
{
    layout_t Lo;
    make_layout(A, &Lo);
    work_t *w = work_new(A);
    for (int64_t i = 0; i < n; i++) {
        point_forward(A, Hm, &Lo, theta, x[i], y[i], z[i], R[i], w);
        if (psi) psi[i] = w->psi;
        if (E) E[i] = w->E;
        if (lap) lap[i] = w->lap;
        if (res) res[i] = w->res;
    }
    work_free(w);
}

/* E(R) with its first and second R-derivative: what the reference gets from
 * grad([E],[R],create_graph=True) twice over Lin_E1/E2/Eout (poc/main.py:249-253,
 * 1322-1328), written as three forward streams (value, d/dR, d2/dR2). */
void oracle_energy_R(const oracle_arch *A, const double *theta, const double *R, int64_t n,
                     double *E, double *dE, double *d2E)
{
    layout_t Lo;
    make_layout(A, &Lo);
    int HE = A->HE;
    double *h = (double *)malloc(sizeof(double) * 3 * HE);
    for (int64_t i = 0; i < n; i++) {
        for (int k = 0; k < HE; k++) {
            double w = theta[Lo.WE1 + k], s0, s1, s2, s3;
            act4(A->act, w * R[i] + theta[Lo.bE1 + k], &s0, &s1, &s2, &s3);
            h[k] = s0; h[HE + k] = s1 * w; h[2 * HE + k] = s2 * w * w;
        }
        double e = theta[Lo.bE3], de = 0.0, d2e = 0.0;
        for (int j = 0; j < HE; j++) {
            const double *W = theta + Lo.WE2 + (size_t)j * HE;
            double a = theta[Lo.bE2 + j], da = 0.0, d2a = 0.0, s0, s1, s2, s3;
            for (int k = 0; k < HE; k++) {
                a += W[k] * h[k]; da += W[k] * h[HE + k]; d2a += W[k] * h[2 * HE + k];
            }
            act4(A->act, a, &s0, &s1, &s2, &s3);
            double w3 = theta[Lo.WE3 + j];
            e += w3 * s0; de += w3 * s1 * da; d2e += w3 * (s2 * da * da + s1 * d2a);
        }
        if (E) E[i] = e;
        if (dE) dE[i] = de;
        if (d2E) d2E[i] = d2e;
    }
    free(h);
}

/* ---------------------------------------------------------------- backward */

/* reverse sweep through one trunk pass; ob[s][j] = adjoint of the last hidden
 * layer's post-activation stream s (only s = 0 and 4 are non-zero there). */
static void trunk_backward(const oracle_arch *A, const layout_t *Lo, const double *th,
                           work_t *w, int p, double *ob /*[5][H]*/, double *tmp /*[5][H]*/,
                           double *g)
{
    int H = A->H;
    double *ab = tmp; /* adjoint of pre-activation streams [5][H] */
    for (int l = A->L - 1; l >= 0; l--) {
        int K = l == 0 ? 2 : H;
        for (int j = 0; j < H; j++) {
            double a0 = w->a[IDX(p, l, 0) + j], ax = w->a[IDX(p, l, 1) + j],
                   ay = w->a[IDX(p, l, 2) + j], az = w->a[IDX(p, l, 3) + j],
                   al = w->a[IDX(p, l, 4) + j];
            double s0, s1, s2, s3;
            act4(A->act, a0, &s0, &s1, &s2, &s3);
            double o0 = ob[0 * H + j], ox = ob[1 * H + j], oy = ob[2 * H + j],
                   oz = ob[3 * H + j], ol = ob[4 * H + j];
            double q = ax * ax + ay * ay + az * az;
            ab[4 * H + j] = s1 * ol;
            ab[1 * H + j] = s1 * ox + 2.0 * s2 * ax * ol;
            ab[2 * H + j] = s1 * oy + 2.0 * s2 * ay * ol;
            ab[3 * H + j] = s1 * oz + 2.0 * s2 * az * ol;
            ab[0 * H + j] = s1 * o0 + s2 * (ax * ox + ay * oy + az * oz)
                          + (s2 * al + s3 * q) * ol;
        }
        const double *W = th + Lo->W[l];
        for (int j = 0; j < H; j++) {
            g[Lo->b[l] + j] += ab[0 * H + j];
            for (int i = 0; i < K; i++) {
                double t = 0.0;
                for (int s = 0; s < 5; s++) {
                    double xin = l == 0 ? w->in[p][i][s] : w->o[IDX(p, l - 1, s) + i];
                    t += ab[s * H + j] * xin;
                }
                g[Lo->W[l] + (size_t)j * K + i] += t;
            }
        }
        if (l > 0) {
            for (int s = 0; s < 5; s++)
                for (int i = 0; i < H; i++) {
                    double t = 0.0;
                    for (int j = 0; j < H; j++) t += W[(size_t)j * H + i] * ab[s * H + j];
                    ob[s * H + i] = t;
                }
        }
    }
}

void oracle_loss_grad(const oracle_arch *A, const oracle_ham *Hm, const double *theta,
                      const double *x, const double *y, const double *z, const double *R,
                      const uint8_t *mask, int64_t n, double invN, double invN1,
                      double invN2, uint64_t trainable_mask, double *grad, double *sums)
{
    layout_t Lo;
    make_layout(A, &Lo);
    int H = A->H, HE = A->HE, HG = A->HG;
    work_t *w = work_new(A);
    double *ob = (double *)calloc((size_t)5 * H, sizeof(double));
    double *tmp = (double *)calloc((size_t)5 * H, sizeof(double));
    double *aE2 = (double *)calloc((size_t)HE, sizeof(double));
    const double *th = theta;
    double S0 = 0, S1 = 0, S2 = 0;
    if (grad) memset(grad, 0, sizeof(double) * (size_t)Lo.P);
    for (int64_t i = 0; i < n; i++) {
        point_forward(A, Hm, &Lo, th, x[i], y[i], z[i], R[i], w);
        int m1 = mask ? (mask[i] & 1) : 0, m2 = mask ? ((mask[i] >> 1) & 1) : 0;
        S0 += w->res * w->res;
        if (m1) S1 += w->psi * w->psi;
        if (m2) S2 += w->psi * w->psi;
        if (!grad) continue;
        /* adjoint of the scalar tail */
        double Vc = Hm->b * w->U + Hm->c * w->E;
        double resb = 2.0 * w->res * invN;
        double psib = resb * Vc + 2.0 * w->psi * (m1 * invN1 + m2 * invN2);
        double lapb = Hm->a * resb;
        double Eb = Hm->c * w->psi * resb;
        double Nb = w->g * psib, lapNb = w->g * lapb;
        double gb = w->N * psib + w->lapN * lapb;
        /* output layer */
        grad[Lo.bout] += Nb;
        for (int p = 0; p < A->n_pass; p++) {
            double pw = p == 0 ? 1.0 : Hm->P;
            for (int j = 0; j < H; j++) {
                double Bv = w->o[IDX(p, A->L - 1, 0) + j], Bl = w->o[IDX(p, A->L - 1, 4) + j];
                grad[Lo.Wout + j] += A->out_scale * pw * (Bv * Nb + Bl * lapNb);
                double wj = A->out_scale * pw * th[Lo.Wout + j];
                ob[0 * H + j] = wj * Nb;
                ob[1 * H + j] = ob[2 * H + j] = ob[3 * H + j] = 0.0;
                ob[4 * H + j] = wj * lapNb;
            }
            trunk_backward(A, &Lo, th, w, p, ob, tmp, grad);
        }
        /* gate */
        grad[Lo.bG2] += gb;
        for (int j = 0; j < HG; j++) {
            double s0, s1, s2, s3;
            act4(A->act, th[Lo.WG1 + j] * R[i] + th[Lo.bG1 + j], &s0, &s1, &s2, &s3);
            grad[Lo.WG2 + j] += gb * s0;
            double ab = gb * th[Lo.WG2 + j] * s1;
            grad[Lo.WG1 + j] += ab * R[i];
            grad[Lo.bG1 + j] += ab;
        }
        /* E-net */
        grad[Lo.bE3] += Eb;
        for (int j = 0; j < HE; j++) {
            double h = w->hE2[j];
            double s1 = A->act == 0 ? h * (1.0 - h) : 1.0 - h * h;
            grad[Lo.WE3 + j] += Eb * h;
            aE2[j] = Eb * th[Lo.WE3 + j] * s1;
            grad[Lo.bE2 + j] += aE2[j];
            for (int k = 0; k < HE; k++)
                grad[Lo.WE2 + (size_t)j * HE + k] += aE2[j] * w->hE1[k];
        }
        for (int k = 0; k < HE; k++) {
            double t = 0.0;
            for (int j = 0; j < HE; j++) t += th[Lo.WE2 + (size_t)j * HE + k] * aE2[j];
            double h = w->hE1[k];
            double s1 = A->act == 0 ? h * (1.0 - h) : 1.0 - h * h;
            double ab = t * s1;
            grad[Lo.WE1 + k] += ab * R[i];
            grad[Lo.bE1 + k] += ab;
        }
    }
    if (sums) { sums[0] = S0; sums[1] = S1; sums[2] = S2; }
    if (grad) {
        int nt = oracle_tensor_count(A);
        int64_t off[64], sz[64];
        oracle_tensor_layout(A, off, sz);
        for (int t = 0; t < nt; t++)
            if (!((trainable_mask >> t) & 1))
                memset(grad + off[t], 0, sizeof(double) * (size_t)sz[t]);
    }
    free(ob); free(tmp); free(aE2);
    work_free(w);
}

/* -------------------------------------------------------------------- Adam */

void oracle_adam(const oracle_arch *A, double *theta, const double *grad, double *m,
                 double *v, int64_t step, double lr, double b1, double b2, double eps,
                 uint64_t trainable_mask)
{
    int nt = oracle_tensor_count(A);
    int64_t off[64], sz[64];
    oracle_tensor_layout(A, off, sz);
    double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    double step_size = lr / bc1, bc2s = sqrt(bc2);
    for (int t = 0; t < nt; t++) {
        if (!((trainable_mask >> t) & 1)) continue;
        for (int64_t k = off[t]; k < off[t] + sz[t]; k++) {
            double g = grad[k];
            m[k] = b1 * m[k] + (1.0 - b1) * g;
            v[k] = b2 * v[k] + (1.0 - b2) * g * g;
            double denom = sqrt(v[k]) / bc2s + eps;
            theta[k] -= step_size * (m[k] / denom);
        }
    }
}

/* ----------------------------------------------------------------- sampler */

void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static float u01(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

void oracle_sample(uint64_t seed, uint64_t step, int64_t first, int64_t n, float box,
                   float Rlo, float Rhi, float cutoff, float bc_cutoff, float *x, float *y,
                   float *z, float *R, uint8_t *mask, int64_t counts[2])
{
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    int64_t c1 = 0, c2 = 0;
    float two_box = box + box, dR = Rhi - Rlo;
    float cut2 = cutoff * cutoff, bc2 = bc_cutoff * bc_cutoff;
    for (int64_t i = 0; i < n; i++) {
        uint64_t gi = (uint64_t)(first + i);
        uint32_t ctr[4] = { (uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)step,
                            (uint32_t)(step >> 32) };
        uint32_t o[4];
        oracle_philox4x32_10(ctr, key, o);
        /* x.uniform_(-L, L) etc, train.py:28-31 */
        float xv = fmaf(two_box, u01(o[0]), -box);
        float yv = fmaf(two_box, u01(o[1]), -box);
        float zv = fmaf(two_box, u01(o[2]), -box);
        float Rv = fmaf(dR, u01(o[3]), Rlo);
        /* both clamp tests use pre-clamp radii, train.py:32-35 */
        float yz = fmaf(yv, yv, zv * zv);
        float d1 = xv - Rv, d2 = xv + Rv;
        float r1sq = fmaf(d1, d1, yz), r2sq = fmaf(d2, d2, yz);
        if (r1sq < cut2 || r2sq < cut2) xv = cutoff;
        /* BC sets from post-clamp radii, train.py:36-39 */
        d1 = xv - Rv; d2 = xv + Rv;
        r1sq = fmaf(d1, d1, yz); r2sq = fmaf(d2, d2, yz);
        uint8_t mk = (uint8_t)((r1sq >= bc2 ? 1 : 0) | (r2sq >= bc2 ? 2 : 0));
        x[i] = xv; y[i] = yv; z[i] = zv; R[i] = Rv;
        if (mask) mask[i] = mk;
        c1 += mk & 1; c2 += (mk >> 1) & 1;
    }
    if (counts) { counts[0] = c1; counts[1] = c2; }
}
