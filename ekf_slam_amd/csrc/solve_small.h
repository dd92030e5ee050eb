// Fragment of kernels.hip (included there, inside its anonymous namespace, after kernels.h / device_math.h): the 5x5 solve of a correction / an association, entry by entry (EKF_SLAM.m:125-143, Correspondence.m:49-69).
#pragma once

// ---------------------------------------------------------------------------------------------------
// gather + solve.  One thread per landmark column c:
//     G(:,c) = H_s * P(S,c),  K(c,:) = G(:,c)' * inv(phi),  x(c) += K(c,:) nu,  strip(:,c) -= K_r G(:,c)
// The 5x5 sub-block P(S,S) that phi needs is fetched by every workgroup (19 doubles, L2-resident), so
// there is no inter-workgroup dependency and the whole correction is two launches.
// ---------------------------------------------------------------------------------------------------
struct SmallSolve {
    double Hs[2][5];
    double Phi[4];     // inv(phi), row-major
    double nu[2];
    double Kr[3][2];
    double Gr[2][3];
};

// pss: 0..8 Prr row-major; 9+2t+b = P(t, j+b), t<3, b<2; 15+2t+b = canonical P(j+t, j+b); 19..21 x_r; 22..23 x_j
// the measurement Jacobian block H_s from delta = landmark - robot  (EKF_SLAM.m:125-127,137-138)
__device__ __forceinline__ void solve_hs(double d0, double d1, double &sq, double Hs[2][5]) {
    const double q = d0 * d0 + d1 * d1;
    sq = sqrt(q);
    const double iq = 1 / q;
    const double e[2][5] = { { -sq * d0, -sq * d1, 0, sq * d0, sq * d1 }, { d1, -d0, -q, -d1, d0 } };
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 5; ++b) Hs[a][b] = iq * e[a][b];
}

// Per-entry forms of the small solve (same idea as predict_*_entry above).  h / g are ROWS of H_s / G(:,S), so that a lane-parallel
// caller can select its row without indexing a register array dynamically (that would put the array in scratch memory);
// `pss` may be a register array (serial callers, static b) or the LDS copy (lane-parallel caller, b = f(lane)).
// G(a, S(b)) = H_s(a,:) * P(S, S(b))  (EKF_SLAM.m:141, first product); b < 3: robot columns (P(j+t, b) is stored as
// strip(b, j+t)), b >= 3: columns j, j+1
__device__ __forceinline__ double solve_gs_entry(const double *pss, const double h[5], int b) {
    double acc = 0;
    if (b < 3) {
        for (int t = 0; t < 3; ++t) acc += h[t] * pss[3 * t + b];
        for (int t = 0; t < 2; ++t) acc += h[3 + t] * pss[9 + 2 * b + t];
    } else {
        for (int t = 0; t < 3; ++t) acc += h[t] * pss[9 + 2 * t + (b - 3)];
        for (int t = 0; t < 2; ++t) acc += h[3 + t] * pss[15 + 2 * t + (b - 3)];
    }
    return acc;
}
// phi(a,b) = G(a,S) * H_s(b,:)' + R(a,b)  (EKF_SLAM.m:141)
__device__ __forceinline__ double solve_phi_entry(const double g[5], const double h[5], double Rab) {
    double acc = 0;
    for (int t = 0; t < 5; ++t) acc += g[t] * h[t];
    return acc + Rab;
}
// K_r(b,cc) = G_r(:,b)' * inv(phi)(:,cc)
__device__ __forceinline__ double solve_kr_entry(double g0b, double g1b, double phi_c, double phi_2c) {
    return g0b * phi_c + g1b * phi_2c;
}

// everything after H_s and the predicted measurement (zhat0 = range, zhat1 = bearing): G(:,S), phi, inv(phi), nu, K_r
__device__ __forceinline__ void solve_rest(const double *pss, double zhat0, double zhat1, double z0, double z1, double R00,
                                           double R01, double R10, double R11, SmallSolve &o) {
    double GS[2][5];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 5; ++b) {
            GS[a][b] = solve_gs_entry(pss, o.Hs[a], b);
            if (b < 3) o.Gr[a][b] = GS[a][b];
        }
    const double R[2][2] = { { R00, R01 }, { R10, R11 } };
    double phi[4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) phi[2 * a + b] = solve_phi_entry(GS[a], o.Hs[b], R[a][b]);   // :141
    ekfm::inv2(phi, o.Phi);                                                            // :143 phi_k^-1
    o.nu[0] = z0 - zhat0;                                                              // :144 (bearing NOT wrapped)
    o.nu[1] = z1 - zhat1;
    for (int b = 0; b < 3; ++b) for (int cc = 0; cc < 2; ++cc) o.Kr[b][cc] = solve_kr_entry(o.Gr[0][b], o.Gr[1][b], o.Phi[cc], o.Phi[2 + cc]);
}

// serial composition (association kernel: one lane per landmark)
__device__ __forceinline__ void solve_small(const double *pss, double z0, double z1, double R00, double R01, double R10,
                                            double R11, SmallSolve &o) {
    const double d0 = pss[22] - pss[19], d1 = pss[23] - pss[20];                       // EKF_SLAM.m:125-126
    double sq;
    solve_hs(d0, d1, sq, o.Hs);
    const double bearing = bearing_ni(d1, d0, pss[21]);
    solve_rest(pss, sq, bearing, z0, z1, R00, R01, R10, R11, o);
}
