// Probe: sustained rate of v_mfma_f64_16x16x4_f64 (8 independent accumulators per wave, W waves per SIMD on every CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    d4 acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    double s = 0;
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    double* out; hipMalloc(&out, 8 * 256 * 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpe = 1; wpe <= 4; wpe *= 2) {
        const int grid = 256 * wpe, iters = 20000;
        k<<<grid, 256>>>(out, 100, 1.0, 1e-3); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<grid, 256>>>(out, iters, 1.0, 1e-3); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)grid * 4 * iters * 8, flop = mfma * 2048;
        printf("waves/SIMD %d: %.3f ms, %.1f TFLOP/s f64, %.1f ns per MFMA per SIMD\n", wpe, ms, flop / ms * 1e-9,
               ms * 1e6 / ((double)iters * 8 * wpe));
    }
    return 0;
}
