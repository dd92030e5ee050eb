// Probe: sustained v_fma_f64 rate alone, and together with v_mfma_f64_16x16x4_f64 issued from the same wavefronts.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int kV, int kM>   // per loop iteration: kV*8 VALU FMAs and kM MFMAs
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    double v[16];
    d4 acc[8];
    for (int q = 0; q < 16; ++q) v[q] = q + threadIdx.x;
    for (int q = 0; q < 8; ++q) acc[q] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < kM) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < kV; ++q) v[(r * kV + q) & 15] = __builtin_fma(a, b, v[(r * kV + q) & 15]);
        }
    }
    double s = 0;
    for (int q = 0; q < 16; ++q) s += v[q];
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int kV, int kM>
void run(double* out, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 4, iters = 4000;
    k<kV, kM><<<grid, 256>>>(out, 100, 1.0, 1e-3); hipDeviceSynchronize();
    hipEventRecord(e0); k<kV, kM><<<grid, 256>>>(out, iters, 1.0, 1e-3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * 4;
    const double fv = waves * iters * 8.0 * kV * 64 * 2, fm = waves * iters * (double)kM * 2048;
    printf("%-28s %.3f ms  VALU %.1f TF  MFMA %.1f TF  total %.1f TF\n", name, ms, fv / ms * 1e-9, fm / ms * 1e-9, (fv + fm) / ms * 1e-9);
}
int main() {
    double* out; hipMalloc(&out, 8 * 256 * 256 * 16);
    run<8, 0>(out, "VALU only (4 waves/SIMD)");
    run<0, 8>(out, "MFMA only");
    run<4, 8>(out, "MFMA + 4 FMA per MFMA");
    run<8, 8>(out, "MFMA + 8 FMA per MFMA");
    run<16, 8>(out, "MFMA + 16 FMA per MFMA");
    run<24, 8>(out, "MFMA + 24 FMA per MFMA");
    return 0;
}
