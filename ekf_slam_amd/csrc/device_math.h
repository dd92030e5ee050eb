// Degree-trig / wrap / 2x2 inverse with the semantics the reference's MATLAB built-ins are documented to
// have at its call sites (cosd/sind: EKF_SLAM.m:42,58-59,63-64,84-88; atan2d/wrapTo360: EKF_SLAM.m:130,
// Correspondence.m:56; mpower(.,-1): EKF_SLAM.m:143).  Host+device so ekf_motion_model (host) and the
// kernels agree.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define EKF_MHD __host__ __device__ inline
#else
#define EKF_MHD inline
#endif

// These run on ONE wavefront in the latency chain of every update-step, where a taken-or-not branch costs more than both of its
// sides: value selections are marked unpredictable so that the compiler emits v_cndmask instead of exec-mask branches.
#if defined(__clang__)
#define EKF_SEL(c) __builtin_unpredictable(c)
#else
#define EKF_SEL(c) (c)
#endif

namespace ekfm {

constexpr double kD2R = 0.017453292519943295;
constexpr double kR2D = 57.29577951308232;

// a (deg) = 90*n + r with r in [-45,45]; quad = n mod 4; n = round(a / 90), halves away from zero.
// No fmod: for |a| < 2^40 the quotient of an exact multiple of 90 is exact, 90*n is exactly representable and
// a - 90*n is exact (Sterbenz), so r (and exactness at multiples of 90 degrees) is the same as reducing
// mod 360 first -- and f64 fmod is by far the slowest thing in these kernels' scalar prologues.
// No division either (an f64 division is a ~15-instruction dependent sequence, and this runs twice in the latency chain of every
// update-step): n is first taken from a * (1/90), which is within one ulp of a / 90 and can therefore only be off by one
// where a / 90 is within an ulp of k + 1/2, i.e. where |r| comes out at or just beyond 45; r = a - 90 n is exact for either
// candidate, so ONE correction step restores exactly the n that round(a / 90) gives, ties (|r| == 45) included.
EKF_MHD void reduce90(double a, double &r, int &quad) {
    if (!(fabs(a) < 1099511627776.0)) a = fmod(a, 360.0);        // rare; Inf / NaN come out as NaN and stay NaN below
    const double q = a * (1.0 / 90.0);
    double n = copysign(floor(fabs(q) + 0.5), q);
    r = fma(-90.0, n, a);
    // round-half-away-from-zero of the TRUE quotient: r must lie in [-45, 45], and a tie goes to the larger |n|
    const bool up = (r > 45.0) | ((r == 45.0) & (a > 0.0)), down = (r < -45.0) | ((r == -45.0) & (a < 0.0));
    const double adj = EKF_SEL(up) ? 1.0 : (EKF_SEL(down) ? -1.0 : 0.0);
    n += adj;
    r = fma(-90.0, adj, r);                                       // exact (r -+ 90, or r itself)
    // n mod 4 without a 64-bit integer conversion: n - 4 floor(n / 4) is an exact double in {0, 1, 2, 3}
    quad = (int)(n - 4.0 * floor(n * 0.25));
}

// sin / cos on [-pi/4, pi/4] and atan on [0, inf): plain polynomial kernels instead of the libm routines.
// Why: every update-step carries two sincos and one atan2 on ONE lane in its dependent prologue; the device libm versions
// (general argument reduction, ~150 dependent instructions each) cost 1.3 us + 0.7 us of the ~7 us gather.  The argument
// is already reduced here (reduce90), so an odd / even minimax polynomial is all that is needed.  The same code runs on the
// host (ekf_motion_model), which therefore agrees with the kernels bit for bit.  Accuracy: < 1 ulp (sin, cos), < 2 ulp (atan)
// against glibc over 10^7 points (tests/test_abi_symbols.py::test_device_math_matches_libm runs the host build).
// Coefficients: the classic fdlibm minimax sets for sin / cos on [-pi/4, pi/4].
EKF_MHD double sin_pio4(double x) {
    const double z = x * x;
    double r = 1.58969099521155010221e-10;
    r = fma(r, z, -2.50507602534068634195e-08);
    r = fma(r, z, 2.75573137070700676789e-06);
    r = fma(r, z, -1.98412698298579493134e-04);
    r = fma(r, z, 8.33333333332248946124e-03);
    r = fma(r, z, -1.66666666666666324348e-01);
    return fma(x * z, r, x);
}

EKF_MHD double cos_pio4(double x) {
    const double z = x * x;
    double r = -1.13596475577881948265e-11;
    r = fma(r, z, 2.08757232129817482790e-09);
    r = fma(r, z, -2.75573143513906633035e-07);
    r = fma(r, z, 2.48015872894767294178e-05);
    r = fma(r, z, -1.38888888888741095749e-03);
    r = fma(r, z, 4.16666666666666019037e-02);
    // 1 - z/2 + z^2 r, with the rounding error of 1 - z/2 carried along (fdlibm's k_cos arrangement)
    const double hz = 0.5 * z, w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + z * (z * r));
}

// atan(t), t >= 0 finite or +inf.  t > 1 -> pi/2 - atan(1/t); then the nearest of the angles 0, pi/8, pi/4 is split off,
// atan(t) = k pi/8 + atan((t - c_k) / (1 + t c_k)), c_k = tan(k pi/8), leaving |u| <= tan(pi/16) = 0.199 for the odd
// Taylor series (12 terms: 0.199^25 / 25 < 2^-60).
EKF_MHD double atan_pos(double t) {
    const bool inv = t > 1.0;
    if (inv) t = 1.0 / t;                                   // +inf -> 0
    constexpr double kT8 = 0.41421356237309504880;          // tan(pi/8) = sqrt(2) - 1
    constexpr double kT16 = 0.19891236737965800691;         // tan(pi/16)
    constexpr double kT316 = 0.66817863791929891999;        // tan(3 pi/16)
    double base, u;
    if (t <= kT16) { base = 0.0; u = t; }
    else if (t <= kT316) { base = 0.39269908169872415481; u = (t - kT8) / fma(t, kT8, 1.0); }      // pi/8
    else { base = 0.78539816339744830962; u = (t - 1.0) / (t + 1.0); }                              // pi/4
    const double z = u * u;
    double r = 1.0 / 25.0;
    r = fma(r, z, -1.0 / 23.0);
    r = fma(r, z, 1.0 / 21.0);
    r = fma(r, z, -1.0 / 19.0);
    r = fma(r, z, 1.0 / 17.0);
    r = fma(r, z, -1.0 / 15.0);
    r = fma(r, z, 1.0 / 13.0);
    r = fma(r, z, -1.0 / 11.0);
    r = fma(r, z, 1.0 / 9.0);
    r = fma(r, z, -1.0 / 7.0);
    r = fma(r, z, 1.0 / 5.0);
    r = fma(r, z, -1.0 / 3.0);
    const double a = base + fma(u * z, r, u);
    return inv ? 1.57079632679489661923 - a : a;
}

// atan2 with the IEEE / MATLAB conventions for zeros, infinities and NaN
EKF_MHD double atan2_poly(double y, double x) {
    if (isnan(x) || isnan(y)) return NAN;
    const double ax = fabs(x), ay = fabs(y);
    double a;
    if (ay == 0.0) a = 0.0;                                  // atan2(+-0, x): 0 or pi
    else if (isinf(ax) && isinf(ay)) a = 0.78539816339744830962;
    else a = atan_pos(ay / ax);                              // ax == 0 -> +inf -> pi/2;  ax == inf -> 0
    if (signbit(x)) a = 3.14159265358979323846 - a;
    return copysign(a, y);
}

// exact at multiples of 90 degrees
EKF_MHD double sind(double a) {
    if (!isfinite(a)) return NAN;
    double r; int quad;
    reduce90(a, r, quad);
    const double t = kD2R * r;
    return quad == 0 ? sin_pio4(t) : quad == 1 ? cos_pio4(t) : quad == 2 ? -sin_pio4(t) : -cos_pio4(t);
}

EKF_MHD double cosd(double a) {
    if (!isfinite(a)) return NAN;
    double r; int quad;
    reduce90(a, r, quad);
    const double t = kD2R * r;
    return quad == 0 ? cos_pio4(t) : quad == 1 ? -sin_pio4(t) : quad == 2 ? -cos_pio4(t) : sin_pio4(t);
}

// sind and cosd of the same angle with one reduction and one sin/cos pair (non-finite angles give NaN through reduce90)
EKF_MHD void sincosd(double a, double &sn, double &cs) {
    double r; int quad;
    reduce90(a, r, quad);
    const double t = kD2R * r;
    const double s0 = sin_pio4(t), c0 = cos_pio4(t);
    // quad 0: (s, c)   1: (c, -s)   2: (-s, -c)   3: (-c, s)
    const double sa = EKF_SEL(quad & 1) ? c0 : s0, ca = EKF_SEL(quad & 1) ? s0 : c0;
    sn = EKF_SEL(quad & 2) ? -sa : sa;
    cs = EKF_SEL((quad + 1) & 2) ? -ca : ca;
}

EKF_MHD double atan2d(double y, double x) { return atan2_poly(y, x) * kR2D; }

// mod(a,360) with positive multiples of 360 mapped to 360 (Mapping Toolbox wrapTo360)
EKF_MHD double wrapTo360(double a) {
    if (!isfinite(a)) return NAN;
    // fast exact path for the usual range (one or two turns), fmod otherwise
    double w;
    if (a >= 0.0 && a < 720.0) w = a < 360.0 ? a : a - 360.0;     // exact
    else if (a < 0.0 && a >= -360.0) w = a;                       // == fmod(a,360) for |a| < 360
    else w = fmod(a, 360.0);
    if (w < 0.0) w += 360.0;
    if (w == 0.0 && a > 0.0) w = 360.0;
    return w;
}

// inverse of a 2x2 (row-major) the way inv() does it: LU with partial pivoting, no symmetry assumed.
// A singular phi yields inf/nan exactly as MATLAB's inv would (with a warning there).
EKF_MHD void inv2(const double a[4], double o[4]) {
    double p = a[0], q = a[1], r = a[2], t = a[3];
    const bool swap = fabs(r) > fabs(p);
    if (swap) { const double tp = p, tq = q; p = r; q = t; r = tp; t = tq; }
    const double l = r / p, u22 = t - l * q;
    const double i11 = 1.0 / p, i12 = -q / (p * u22), i22 = 1.0 / u22;
    const double m11 = i11 + i12 * (-l), m12 = i12, m21 = i22 * (-l), m22 = i22;
    if (swap) { o[0] = m12; o[1] = m11; o[2] = m22; o[3] = m21; }
    else      { o[0] = m11; o[1] = m12; o[2] = m21; o[3] = m22; }
}

}  // namespace ekfm
