// Probe: sustained rate of v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32 (independent accumulators, W waves per SIMD on every CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f4 acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = f4{0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
    }
    float s = 0;
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    }
    float s = 0;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 8 * 256 * 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpe = 1; wpe <= 4; wpe *= 2) {
        const int grid = 256 * wpe, iters = 20000;
        float ms;
        k16<<<grid, 256>>>(out, 100, 1.0f, 1e-3f); hipDeviceSynchronize();
        hipEventRecord(e0); k16<<<grid, 256>>>(out, iters, 1.0f, 1e-3f); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("16x16x4 f32, waves/SIMD %d: %.3f ms, %.1f TFLOP/s\n", wpe, ms, (double)grid * 4 * iters * 8 * 2048 / ms * 1e-9);
        k32<<<grid, 256>>>(out, 100, 1.0f, 1e-3f); hipDeviceSynchronize();
        hipEventRecord(e0); k32<<<grid, 256>>>(out, iters, 1.0f, 1e-3f); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("32x32x2 f32, waves/SIMD %d: %.3f ms, %.1f TFLOP/s\n", wpe, ms, (double)grid * 4 * iters * 4 * 4096 / ms * 1e-9);
    }
    return 0;
}
