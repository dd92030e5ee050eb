// Host build of ekf_slam_amd/csrc/device_math.h: the polynomial sin / cos / atan kernels against glibc (long double).
// Prints "max_ulp sin cos atan atan2 special_mismatches"; tests/test_device_math_cpu.py asserts the bounds.
#include "device_math.h"
#include <cmath>
#include <cstdio>
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
    printf("%.4f %.4f %.4f %.4f %d\n", ms, mc, ma, ma2, bad);
    return 0;
}
