/*
 * Structured (O(n) predict/append, O(n^2) correct) C restatement of the reference's EKF-SLAM hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- not part of the product.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this (as the checker / the timed CPU baseline, kind "port").
 * PARITY UNPINNED: the reference (pure MATLAB) holds no golden vectors and cannot run in the image;
 * this file is pinned by hand-derived KATs and by agreeing with oracle/ekf_dense.py (the literal-dense
 * NumPy restatement) to <= 1e-12 relative.
 *
 * P is kept DENSE and FULL (both triangles, row-major, leading dimension ld) and symmetry is NOT
 * assumed, exactly like the reference; only the sparsity of F, Q and H_k is exploited:
 *
 *   predict   EKF_SLAM.m:40-51,56-65   F = I + a e1 e3' + b e2 e3'  ->  two row axpys + two column axpys
 *   append    EKF_SLAM.m:67-98         (== append.m:1-27)
 *   correct   EKF_SLAM.m:124-145       HP = H_s P(S,:), PHt = P(:,S) H_s', phi = HP(:,S) H_s' + R,
 *                                      K = PHt inv(phi), x += K nu, P -= K HP   (rank-2, all n^2 entries)
 *   associate Correspondence.m:28-88   per landmark: phi_k from the 5x5 sub-block P(S_k,S_k)
 *
 * Angles are degrees.  Built by oracle/Makefile (gcc -O2 -fopenmp) into oracle/libekf_oracle.so.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int64_t N;       /* landmarks in the state */
    int64_t cap;     /* landmark capacity      */
    int64_t ld;      /* 3 + 2*cap              */
    double *x;       /* ld                     */
    double *P;       /* ld * ld, row-major     */
    double *s;       /* cap                    */
    double *hp;      /* 2 * ld scratch: H P    */
    double *pht;     /* 2 * ld scratch: P H'   */
    double C;
    double Q[9];     /* last process-noise 3x3 block (EKF_SLAM.m:43-44) */
} oekf;

#define D2R 0.017453292519943295
#define R2D 57.29577951308232

/* ---- MathWorks built-in semantics (see oracle/matlab_compat.py for the assumptions) ---- */
static void reduce90(double a, double *r, int *m) {
    const double q = a / 90.0;
    const double n = copysign(floor(fabs(q) + 0.5), q);   /* MATLAB round(): half away from zero */
    *r = a - n * 90.0;
    double mm = fmod(n, 4.0);
    if (mm < 0) mm += 4.0;
    *m = (int)mm;
}
double oekf_sind(double a) {
    if (!isfinite(a)) return NAN;
    double r; int m; reduce90(fmod(a, 360.0), &r, &m);
    switch (m) { case 0: return sin(D2R * r); case 1: return cos(D2R * r);
                 case 2: return -sin(D2R * r); default: return -cos(D2R * r); }
}
double oekf_cosd(double a) {
    if (!isfinite(a)) return NAN;
    double r; int m; reduce90(fmod(a, 360.0), &r, &m);
    switch (m) { case 0: return cos(D2R * r); case 1: return -sin(D2R * r);
                 case 2: return -cos(D2R * r); default: return sin(D2R * r); }
}
double oekf_wrapTo360(double a) {
    if (!isfinite(a)) return NAN;
    double w = fmod(a, 360.0);
    if (w < 0.0) w += 360.0;
    if (w == 0.0 && a > 0.0) w = 360.0;
    return w;
}
static double atan2d(double y, double x) { return atan2(y, x) * R2D; }

/* inv() of a 2x2 (row-major a[4]) by LU with partial pivoting */
static void inv2(const double a[4], double o[4]) {
    double p = a[0], q = a[1], r = a[2], t = a[3];
    int swap = fabs(r) > fabs(p);
    if (swap) { double tp = p, tq = q; p = r; q = t; r = tp; t = tq; }
    double l = r / p, u22 = t - l * q;
    double i11 = 1.0 / p, i12 = -q / (p * u22), i22 = 1.0 / u22;
    double m11 = i11 + i12 * (-l), m12 = i12, m21 = i22 * (-l), m22 = i22;
    if (swap) { o[0] = m12; o[1] = m11; o[2] = m22; o[3] = m21; }
    else      { o[0] = m11; o[1] = m12; o[2] = m21; o[3] = m22; }
}

/* ---- lifecycle ---- */
oekf *oekf_create(int64_t cap, double C) {
    oekf *h = (oekf *)calloc(1, sizeof(oekf));
    if (!h) return NULL;
    h->cap = cap; h->ld = 3 + 2 * cap; h->C = C; h->N = 0;
    h->x = (double *)calloc((size_t)h->ld, sizeof(double));
    h->P = (double *)calloc((size_t)h->ld * (size_t)h->ld, sizeof(double));
    h->s = (double *)calloc((size_t)(cap > 0 ? cap : 1), sizeof(double));
    h->hp = (double *)calloc((size_t)2 * h->ld, sizeof(double));
    h->pht = (double *)calloc((size_t)2 * h->ld, sizeof(double));
    if (!h->x || !h->P || !h->s || !h->hp || !h->pht) return NULL;
    h->P[0] = h->P[h->ld + 1] = h->P[2 * h->ld + 2] = 0.1;   /* EKF_SLAM.m:28-31 */
    return h;
}
void oekf_destroy(oekf *h) {
    if (!h) return;
    free(h->x); free(h->P); free(h->s); free(h->hp); free(h->pht); free(h);
}
int64_t oekf_num_landmarks(const oekf *h) { return h->N; }
int64_t oekf_ld(const oekf *h) { return h->ld; }
double *oekf_x(oekf *h) { return h->x; }
double *oekf_P(oekf *h) { return h->P; }
double *oekf_s(oekf *h) { return h->s; }
double *oekf_Q(oekf *h) { return h->Q; }
void oekf_set_num_landmarks(oekf *h, int64_t N) { h->N = N; }
/* OpenMP is used only above OEKF_PAR_MIN rows (small states stay single-threaded: a parallel region costs
 * more than the whole update there, and far more when the host exposes more hardware threads than the
 * process may use). */
#define OEKF_PAR_MIN 1024
static int g_threads = 1;
void oekf_set_threads(int t) { g_threads = t < 1 ? 1 : t; }
int oekf_threads(void) {
#ifdef _OPENMP
    return g_threads;
#else
    return 1;
#endif
}

/* ---- predict: EKF_SLAM.m:40-51 with f() :56-65 ---- */
void oekf_predict(oekf *h, const double u[2]) {
    const int64_t n = 3 + 2 * h->N, ld = h->ld;
    double *x = h->x, *P = h->P;
    const double th = x[2];
    const double W[3] = { u[0] * oekf_cosd(th), u[0] * oekf_sind(th), u[1] };
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) h->Q[3 * i + j] = (W[i] * h->C) * W[j];
    const double a = -1 * u[0] * oekf_sind(th), b = u[0] * oekf_cosd(th);
    x[0] = x[0] + u[0] * oekf_cosd(th + u[1]);
    x[1] = x[1] + u[0] * oekf_sind(th + u[1]);
    x[2] = th + u[1];
    /* F*P: rows 0,1 += {a,b} * row 2 */
    for (int64_t j = 0; j < n; ++j) {
        const double p2 = P[2 * ld + j];
        P[0 * ld + j] += a * p2;
        P[1 * ld + j] += b * p2;
    }
    /* (F*P)*F': cols 0,1 += {a,b} * col 2 */
    for (int64_t i = 0; i < n; ++i) {
        const double p2 = P[i * ld + 2];
        P[i * ld + 0] += a * p2;
        P[i * ld + 1] += b * p2;
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) P[i * ld + j] += h->Q[3 * i + j];
    x[2] = oekf_wrapTo360(x[2]);
}

/* ---- append: EKF_SLAM.m:67-98 ---- */
int oekf_append(oekf *h, const double u[2], const double R[4], const double pos[2], double signature) {
    if (h->N >= h->cap) return 1;
    const int64_t n = 3 + 2 * h->N, ld = h->ld;
    double *x = h->x, *P = h->P;
    h->s[h->N] = signature;
    x[n] = pos[0]; x[n + 1] = pos[1];
    const double jxr[2][3] = { { 1, 0, -u[0] * oekf_sind(x[2]) }, { 0, 1, u[0] * oekf_cosd(x[2]) } };
    const double jz[2][2] = { { oekf_cosd(u[1]), -u[0] * oekf_sind(u[1]) },
                              { oekf_sind(u[1]),  u[0] * oekf_cosd(u[1]) } };
    /* C: jxr*Prr*jxr' + jz*R*jz' */
    double t[2][3], c1[2][2], t2[2][2], c2[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) {
        double acc = 0; for (int k = 0; k < 3; ++k) acc += jxr[i][k] * P[k * ld + j]; t[i][j] = acc; }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
        double acc = 0; for (int k = 0; k < 3; ++k) acc += t[i][k] * jxr[j][k]; c1[i][j] = acc; }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
        double acc = 0; for (int k = 0; k < 2; ++k) acc += jz[i][k] * R[2 * k + j]; t2[i][j] = acc; }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
        double acc = 0; for (int k = 0; k < 2; ++k) acc += t2[i][k] * jz[j][k]; c2[i][j] = acc; }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) P[(n + i) * ld + n + j] = c1[i][j] + c2[i][j];
    /* I: P(1:3,new) = Prr*jxr' ; H: mirror */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) {
        double acc = 0; for (int k = 0; k < 3; ++k) acc += P[i * ld + k] * jxr[j][k];
        P[i * ld + n + j] = acc; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) P[(n + j) * ld + i] = P[i * ld + n + j];
    /* F: P(new, lm k) = jxr * P(lm k, 1:3)' ; G: mirror */
    for (int64_t k = 0; k < h->N; ++k) {
        const int64_t c = 3 + 2 * k;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
            double acc = 0; for (int r = 0; r < 3; ++r) acc += jxr[i][r] * P[(c + j) * ld + r];
            P[(n + i) * ld + c + j] = acc; }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) P[(c + j) * ld + n + i] = P[(n + i) * ld + c + j];
    }
    h->N += 1;
    return 0;
}

/* innovation terms for 0-based landmark k: zhat[2], Hs[2][5]  (EKF_SLAM.m:125-138) */
static void innovation_terms(const double *x, int64_t k, double zhat[2], double Hs[2][5]) {
    const int64_t j = 3 + 2 * k;
    const double d0 = x[j] - x[0], d1 = x[j + 1] - x[1];
    const double q = d0 * d0 + d1 * d1, sq = sqrt(q);
    zhat[0] = sq;
    zhat[1] = oekf_wrapTo360(atan2d(d1, d0) - x[2]);
    const double iq = 1 / q;
    const double e[2][5] = { { -sq * d0, -sq * d1, 0, sq * d0, sq * d1 }, { d1, -d0, -q, -d1, d0 } };
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 5; ++b) Hs[a][b] = iq * e[a][b];
}

/* ---- correct: EKF_SLAM.m:124-145 for 1-based landmark idx ---- */
int oekf_correct(oekf *h, const double z[2], const double R[4], int64_t idx) {
    if (idx < 1 || idx > h->N) return 1;
    const int64_t n = 3 + 2 * h->N, ld = h->ld, j = 3 + 2 * (idx - 1);
    double *x = h->x, *P = h->P, *hp = h->hp, *pht = h->pht;
    double zhat[2], Hs[2][5];
    innovation_terms(x, idx - 1, zhat, Hs);
    const int64_t S[5] = { 0, 1, 2, j, j + 1 };
    /* HP = H_s P(S,:) (rows), PHt = P(:,S) H_s' (columns) */
#pragma omp parallel for schedule(static) num_threads(g_threads) if (n >= OEKF_PAR_MIN)
    for (int64_t c = 0; c < n; ++c) {
        for (int a = 0; a < 2; ++a) {
            double r = 0, cc = 0;
            for (int b = 0; b < 5; ++b) { r += Hs[a][b] * P[S[b] * ld + c]; cc += P[c * ld + S[b]] * Hs[a][b]; }
            hp[a * ld + c] = r; pht[a * ld + c] = cc;
        }
    }
    double phi[4], iphi[4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
        double acc = 0; for (int t = 0; t < 5; ++t) acc += hp[a * ld + S[t]] * Hs[b][t];
        phi[2 * a + b] = acc + R[2 * a + b]; }
    inv2(phi, iphi);
    const double nu0 = z[0] - zhat[0], nu1 = z[1] - zhat[1];
    /* K = PHt * inv(phi) ; x += K nu ; P -= K * HP */
#pragma omp parallel for schedule(static) num_threads(g_threads) if (n >= OEKF_PAR_MIN)
    for (int64_t i = 0; i < n; ++i) {
        const double k0 = pht[i] * iphi[0] + pht[ld + i] * iphi[2];
        const double k1 = pht[i] * iphi[1] + pht[ld + i] * iphi[3];
        x[i] += k0 * nu0 + k1 * nu1;
        double *row = P + i * ld;
        for (int64_t c = 0; c < n; ++c) row[c] -= k0 * hp[c] + k1 * hp[ld + c];
    }
    return 0;
}

/* ---- associate: Correspondence.m:28-88 ; returns is_new, 1-based index ---- */
int oekf_associate(oekf *h, const double z[3], const double R[4], double s_cost, double s_thresh, double w_pos,
                   int32_t *is_new, int64_t *index, double *pos_cost, double *sig_cost) {
    const int64_t N = h->N, ld = h->ld;
    const double *x = h->x, *P = h->P;
    *is_new = 1; *index = N + 1;
    double best = INFINITY;
    const double inv_cost = 1.0 / s_cost;
    for (int64_t k = 0; k < N; ++k) {
        const int64_t j = 3 + 2 * k;
        const int64_t S[5] = { 0, 1, 2, j, j + 1 };
        double zhat[2], Hs[2][5], hps[2][5], phi[4], iphi[4];
        innovation_terms(x, k, zhat, Hs);
        for (int a = 0; a < 2; ++a) for (int t = 0; t < 5; ++t) {
            double acc = 0; for (int b = 0; b < 5; ++b) acc += Hs[a][b] * P[S[b] * ld + S[t]];
            hps[a][t] = acc; }
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
            double acc = 0; for (int t = 0; t < 5; ++t) acc += hps[a][t] * Hs[b][t];
            phi[2 * a + b] = acc + R[2 * a + b]; }
        inv2(phi, iphi);
        const double n0 = z[0] - zhat[0], n1 = z[1] - zhat[1];
        const double pc = (n0 * iphi[0] + n1 * iphi[2]) * n0 + (n0 * iphi[1] + n1 * iphi[3]) * n1;
        const double d = z[2] - h->s[k];
        const double sc = d * inv_cost * d;
        if (pos_cost) pos_cost[k] = pc;
        if (sig_cost) sig_cost[k] = sc;
        const double ll = (w_pos != 0.0) ? (w_pos * pc + sc) : sc;     /* Correspondence.m:74-75 */
        if (ll <= s_thresh && ll < best) { *is_new = 0; best = ll; *index = k + 1; }
    }
    return 0;
}
