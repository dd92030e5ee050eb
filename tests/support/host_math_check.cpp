// Host build of ekf_slam_amd/csrc/device_math.h: the polynomial sin / cos / atan kernels against glibc (long double).
// Prints "max_ulp sin cos atan atan2 special_mismatches"; tests/test_device_math_cpu.py asserts the bounds.
#include "device_math.h"
#include <cmath>
#include <cstdio>
#include <initializer_list>
#include <random>
static double ulp_err(double got, long double ref) {
    if (ref == 0) return got == 0 ? 0 : 1e9;
    int e; frexp((double)ref, &e);
    return (double)(fabsl((long double)got - ref) / ldexpl(1.0L, e - 53));
}
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2000000;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> u(-0.7853981633974483, 0.7853981633974483), le(-30, 30), xy(-100, 100);
    double ms = 0, mc = 0, ma = 0, ma2 = 0;
    for (int i = 0; i < n; ++i) {
        const double x = u(rng), t = exp(le(rng)), y2 = xy(rng), x2 = xy(rng);
        ms = fmax(ms, ulp_err(ekfm::sin_pio4(x), sinl((long double)x)));
        mc = fmax(mc, ulp_err(ekfm::cos_pio4(x), cosl((long double)x)));
        ma = fmax(ma, ulp_err(ekfm::atan_pos(t), atanl((long double)t)));
        ma2 = fmax(ma2, ulp_err(ekfm::atan2_poly(y2, x2), atan2l((long double)y2, (long double)x2)));
    }
    const double v[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY};
    int bad = 0;
    for (double y : v) for (double x : v) {
        const double a = ekfm::atan2_poly(y, x), b = atan2(y, x);
        if (!(fabs(a - b) <= 4.5e-16 && signbit(a) == signbit(b))) ++bad;
    }
    if (!std::isnan(ekfm::atan2_poly(NAN, 1.0)) || !std::isnan(ekfm::sind(INFINITY))) ++bad;
    // exactness at multiples of 90 degrees (cosd/sind call sites: EKF_SLAM.m:42,58-59)
    if (ekfm::sind(90) != 1 || ekfm::cosd(90) != 0 || ekfm::sind(180) != 0 || ekfm::cosd(180) != -1 || ekfm::sind(-270) != 1 ||
        ekfm::cosd(360) != 1 || ekfm::atan2d(1, 1) != 45 || ekfm::atan2d(1, 0) != 90 || ekfm::atan2d(0, -1) != 180) ++bad;
    // reduce90 (division-free) must give exactly n = round-half-away-from-zero(a / 90) of the TRUE quotient (80-bit arithmetic
    // here; the earlier floor(a / 90.0 + 0.5) form mis-rounded angles one ulp below 45 + 90 k to |r| > 45) and the exact
    // remainder: random angles, every tie a = 90 k + 45 and its neighbours, multiples of 90, tiny and large magnitudes
    auto ref_reduce = [](double a, double &r, int &quad) {
        const long double n = roundl((long double)a / 90.0L);          // halves away from zero
        r = (double)((long double)a - 90.0L * n);                       // exact: |a| < 2^40, 64-bit significand
        quad = (int)(((long long)n) & 3);
    };
    auto same_reduce = [&](double a) {
        double r1, r2; int q1, q2;
        ekfm::reduce90(a, r1, q1); ref_reduce(a, r2, q2);
        return r1 == r2 && q1 == q2 && fabs(r1) <= 45.0;
    };
    std::uniform_real_distribution<double> ang(-1.0e5, 1.0e5), small(-400, 400);
    for (int i = 0; i < n; ++i) if (!same_reduce(ang(rng)) || !same_reduce(small(rng))) ++bad;
    for (int k = -4000; k <= 4000; ++k)
        for (double base : { 90.0 * k + 45.0, 90.0 * k, 90.0 * k - 45.0 }) {
            double a = base;
            for (int s = 0; s < 3; ++s) { if (!same_reduce(a)) ++bad; a = nextafter(a, INFINITY); }
            a = base;
            for (int s = 0; s < 3; ++s) { if (!same_reduce(a)) ++bad; a = nextafter(a, -INFINITY); }
        }
    for (double a : { 1e-300, -1e-300, 4.9e-324, 44.99999999999999, 45.00000000000001, 1e9 + 45.0, -1e9 - 45.0, 1.0e12 }) if (!same_reduce(a)) ++bad;
    // sincosd == (sind, cosd), bit for bit, every quadrant and sign, and NaN for non-finite angles
    for (int i = 0; i < n / 4; ++i) {
        const double a = ang(rng);
        double sn, cs;
        ekfm::sincosd(a, sn, cs);
        if (sn != ekfm::sind(a) || cs != ekfm::cosd(a) || std::signbit(sn) != std::signbit(ekfm::sind(a)) || std::signbit(cs) != std::signbit(ekfm::cosd(a))) ++bad;
    }
    for (int k = -16; k <= 16; ++k) {
        double sn, cs;
        ekfm::sincosd(45.0 * k, sn, cs);
        if (sn != ekfm::sind(45.0 * k) || cs != ekfm::cosd(45.0 * k)) ++bad;
    }
    { double sn, cs; ekfm::sincosd(INFINITY, sn, cs); if (!std::isnan(sn) || !std::isnan(cs)) ++bad;
      ekfm::sincosd(NAN, sn, cs); if (!std::isnan(sn) || !std::isnan(cs)) ++bad;
      ekfm::sincosd(3.0e12 + 90.0, sn, cs); if (sn != ekfm::sind(3.0e12 + 90.0) || cs != ekfm::cosd(3.0e12 + 90.0)) ++bad; }
    printf("%.4f %.4f %.4f %.4f %d\n", ms, mc, ma, ma2, bad);
    return 0;
}
