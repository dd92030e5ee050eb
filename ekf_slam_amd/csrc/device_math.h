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

namespace ekfm {

constexpr double kD2R = 0.017453292519943295;
constexpr double kR2D = 57.29577951308232;

// a (deg) = 90*n + r with r in [-45,45]; quad = n mod 4.  round() is half-away-from-zero.
// No fmod: for |a| < 2^40 the quotient of an exact multiple of 90 is exact, 90*n is exactly representable and
// a - 90*n is exact (Sterbenz), so r (and exactness at multiples of 90 degrees) is the same as reducing
// mod 360 first -- and f64 fmod is by far the slowest thing in these kernels' scalar prologues.
EKF_MHD void reduce90(double a, double &r, int &quad) {
    if (fabs(a) >= 1099511627776.0) a = fmod(a, 360.0);
    const double q = a / 90.0;
    const double n = copysign(floor(fabs(q) + 0.5), q);
    r = fma(-90.0, n, a);
    quad = (int)(((long long)n) & 3);
}

// exact at multiples of 90 degrees
EKF_MHD double sind(double a) {
    if (!isfinite(a)) return NAN;
    double r; int quad;
    reduce90(a, r, quad);
    const double t = kD2R * r;
    return quad == 0 ? sin(t) : quad == 1 ? cos(t) : quad == 2 ? -sin(t) : -cos(t);
}

EKF_MHD double cosd(double a) {
    if (!isfinite(a)) return NAN;
    double r; int quad;
    reduce90(a, r, quad);
    const double t = kD2R * r;
    return quad == 0 ? cos(t) : quad == 1 ? -sin(t) : quad == 2 ? -cos(t) : sin(t);
}

// sind and cosd of the same angle with one reduction and one sin/cos pair
EKF_MHD void sincosd(double a, double &sn, double &cs) {
    if (!isfinite(a)) { sn = NAN; cs = NAN; return; }
    double r; int quad;
    reduce90(a, r, quad);
    const double t = kD2R * r;
    const double s0 = sin(t), c0 = cos(t);
    sn = quad == 0 ? s0 : quad == 1 ? c0 : quad == 2 ? -s0 : -c0;
    cs = quad == 0 ? c0 : quad == 1 ? -s0 : quad == 2 ? -c0 : s0;
}

EKF_MHD double atan2d(double y, double x) { return atan2(y, x) * kR2D; }

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
