/* TEST SUPPORT: a recording stand-in for libekfslam (every entry point matlab/ekfslam_mex.c calls), so that the gateway's
 * argument marshalling -- which prhs goes to which parameter, 1-based -> 0-based indices, output shapes, status -> MATLAB
 * error -- can be executed and checked without a GPU.  Every call appends one line to stdout: "ABI <name> <args...>". */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ekfslam.h"

struct ekf_handle { int64_t N, cap; int fail_next; char err[64]; int rank, world; };

#define LOG(...) do { printf("ABI " __VA_ARGS__); printf("\n"); } while (0)
static int32_t st(ekf_handle *h) { if (h && h->fail_next) { h->fail_next = 0; snprintf(h->err, sizeof h->err, "injected failure"); return EKF_ERR_STATE; } return EKF_OK; }

const char *ekf_status_string(int32_t s) { return s == EKF_OK ? "ok" : s == EKF_ERR_STATE ? "call not valid in the current state" : "error"; }
const char *ekf_last_error(const ekf_handle *h) { return h ? h->err : "null handle"; }
int32_t ekf_config_default(ekf_config *cfg, int32_t mode) {
    if (mode != 0 && mode != 1) return EKF_ERR_INVALID_ARG;
    memset(cfg, 0, sizeof *cfg); cfg->mode = mode; cfg->batch = 1; cfg->capacity_landmarks = 1024;
    LOG("ekf_config_default mode=%d", mode); return EKF_OK;
}
int32_t ekf_create(const ekf_config *cfg, ekf_handle **out) {
    ekf_handle *h = calloc(1, sizeof *h);
    h->cap = cfg->capacity_landmarks; h->N = 0;
    *out = h;
    h->rank = cfg->rank; h->world = cfg->world;
    if (cfg->storage || cfg->pass_arith) LOG("ekf_create storage=%d pass_arith=%d", cfg->storage, cfg->pass_arith);
    if (cfg->world > 1) LOG("ekf_create mode=%d cap=%lld tile=%d batch=%d device=%d rank=%d world=%d", cfg->mode,
                            (long long)cfg->capacity_landmarks, cfg->tile, cfg->batch, cfg->device, cfg->rank, cfg->world);
    else
    LOG("ekf_create mode=%d cap=%lld tile=%d batch=%d", cfg->mode, (long long)cfg->capacity_landmarks, cfg->tile, cfg->batch);
    if (cfg->capacity_landmarks == 666) { snprintf(h->err, sizeof h->err, "no device"); return EKF_ERR_NO_DEVICE; }
    return EKF_OK;
}
int32_t ekf_destroy(ekf_handle *h) { LOG("ekf_destroy"); free(h); return EKF_OK; }
int32_t ekf_set_params(ekf_handle *h, double C, const double Rc[2], double s_cost, double s_thresh, double w_pos) {
    LOG("ekf_set_params C=%g Rc=%g,%g s_cost=%g s_thresh=%g w_pos=%g", C, Rc[0], Rc[1], s_cost, s_thresh, w_pos); return st(h); }
int32_t ekf_predict(ekf_handle *h, const double u[2]) { LOG("ekf_predict u=%g,%g", u[0], u[1]); return st(h); }
int32_t ekf_motion_model(const double *x, int64_t n, const double u[2], double *x_new, double *F) {
    LOG("ekf_motion_model n=%lld x0=%g u=%g,%g F=%s", (long long)n, x[0], u[0], u[1], F ? "yes" : "null");
    for (int64_t i = 0; i < n; ++i) x_new[i] = x[i] + 100.0;
    if (F) for (int64_t i = 0; i < n * n; ++i) F[i] = (double)i;
    return EKF_OK;
}
int32_t ekf_append(ekf_handle *h, const double u[2], const double R[4], const double pos[2], double signature) {
    LOG("ekf_append u=%g,%g R=%g,%g,%g,%g pos=%g,%g sig=%g", u[0], u[1], R[0], R[1], R[2], R[3], pos[0], pos[1], signature);
    const int32_t rc = st(h); if (!rc) h->N += 1; return rc; }
int32_t ekf_correct(ekf_handle *h, const double z[2], const double R[4], int64_t idx) {
    LOG("ekf_correct z=%g,%g R=%g,%g,%g,%g idx0=%lld", z[0], z[1], R[0], R[1], R[2], R[3], (long long)idx); return st(h); }
int32_t ekf_associate(ekf_handle *h, const double z[3], const double R[4], int32_t *is_new, int64_t *idx, double *pc, double *sc) {
    LOG("ekf_associate z=%g,%g,%g R=%g,%g,%g,%g costs=%s", z[0], z[1], z[2], R[0], R[1], R[2], R[3], (pc || sc) ? "yes" : "null");
    *is_new = 0; *idx = 6; return st(h); }
int32_t ekf_measure(ekf_handle *h, const double *obs, int64_t m, const double u[2], const double *lm_index, const double *lm_loc, int64_t L) {
    LOG("ekf_measure m=%lld obs_r0=%g,%g,%g obs_last=%g u=%g,%g L=%lld idx0=%g loc0=%g,%g", (long long)m, obs[0], obs[m], obs[2 * m],
        obs[3 * m - 1], u[0], u[1], (long long)L, lm_index[0], lm_loc[0], lm_loc[L]);
    return st(h); }
int32_t ekf_num_landmarks(ekf_handle *h, int64_t *N) { *N = h->N; return EKF_OK; }
int32_t ekf_get_x(ekf_handle *h, double *x) { LOG("ekf_get_x"); for (int64_t i = 0; i < 3 + 2 * h->N; ++i) x[i] = 1000.0 + i; return st(h); }
int32_t ekf_set_x(ekf_handle *h, const double *x, int64_t n) { LOG("ekf_set_x n=%lld x0=%g", (long long)n, x[0]); if (n < 3 || (n - 3) % 2) return EKF_ERR_INVALID_ARG; h->N = (n - 3) / 2; return st(h); }
int32_t ekf_get_s(ekf_handle *h, double *s) { LOG("ekf_get_s"); for (int64_t i = 0; i < h->N; ++i) s[i] = 1.0 + i; return st(h); }
int32_t ekf_set_s(ekf_handle *h, const double *s, int64_t N) { LOG("ekf_set_s N=%lld s0=%g", (long long)N, (s && N) ? s[0] : -1.0); return N == h->N ? st(h) : EKF_ERR_INVALID_ARG; }
int32_t ekf_get_P(ekf_handle *h, double *P) { LOG("ekf_get_P"); const int64_t n = 3 + 2 * h->N; for (int64_t i = 0; i < n * n; ++i) P[i] = 0.5 * i; return st(h); }
int32_t ekf_set_P(ekf_handle *h, const double *P, int64_t n) { LOG("ekf_set_P n=%lld P0=%g", (long long)n, P[0]); return n == 3 + 2 * h->N ? st(h) : EKF_ERR_INVALID_ARG; }
int32_t ekf_get_P_block(ekf_handle *h, int64_t r0, int64_t c0, int64_t nr, int64_t nc, double *out) {
    LOG("ekf_get_P_block r0=%lld c0=%lld nr=%lld nc=%lld", (long long)r0, (long long)c0, (long long)nr, (long long)nc);
    for (int64_t i = 0; i < nr * nc; ++i) out[i] = 7.0 + i;
    return st(h); }
int32_t ekf_get_P_diag_blocks(ekf_handle *h, double *out) { LOG("ekf_get_P_diag_blocks"); for (int64_t i = 0; i < 4 * (h->N + 1); ++i) out[i] = 0.25 * i; return st(h); }
int32_t ekf_get_Q(ekf_handle *h, double Q[9]) { LOG("ekf_get_Q"); for (int i = 0; i < 9; ++i) Q[i] = 10.0 + i; return st(h); }

int32_t ekf_correct_begin(ekf_handle *h, const double z[2], const double R[4], int64_t idx) {
    LOG("ekf_correct_begin rank=%d z=%g,%g R=%g,%g,%g,%g idx0=%lld", h->rank, z[0], z[1], R[0], R[1], R[2], R[3], (long long)idx); return st(h); }
int32_t ekf_correct_finish(ekf_handle *h) { LOG("ekf_correct_finish rank=%d", h->rank); return st(h); }
int32_t ekf_hint_next(ekf_handle *h, int64_t idx) { LOG("ekf_hint_next rank=%d idx0=%lld", h->rank, (long long)idx); return st(h); }
int32_t ekf_associate_begin(ekf_handle *h, const double z[3], const double R[4], int32_t want_costs) {
    LOG("ekf_associate_begin rank=%d z=%g,%g,%g R=%g,%g,%g,%g costs=%d", h->rank, z[0], z[1], z[2], R[0], R[1], R[2], R[3], want_costs); return st(h); }
int32_t ekf_associate_finish(ekf_handle *h, int32_t *is_new, int64_t *idx, double *pc, double *sc) {
    LOG("ekf_associate_finish rank=%d costs=%s", h->rank, (pc || sc) ? "yes" : "null"); *is_new = 0; *idx = 4; return st(h); }
int32_t ekf_exchange_local(ekf_handle **hs, int32_t world) {
    printf("ABI ekf_exchange_local world=%d ranks=", world);
    for (int r = 0; r < world; ++r) printf("%s%d", r ? "," : "", hs[r]->rank);
    printf("\n");
    return st(hs[0]); }
int32_t ekf_flush(ekf_handle *h) { LOG("ekf_flush"); return st(h); }

void stub_fail_next(ekf_handle *h) { h->fail_next = 1; }
