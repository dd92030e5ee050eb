// Probe: how fast can v_mfma_f32_16x16x4_f32 issue?  One or two wavefronts per SIMD, eight independent accumulators, operands in
// registers only.  Prints s_memtime ticks per MFMA and the tick rate (ticks / event time).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/mfma_issue_rate.hip -o scripts/probes/mfma_issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>      // 0: builtin; 1: asm in place; 2: asm in place with a ds_read pair per 8 MFMAs
__global__ void k(float *out, unsigned long long *ticks, int iters) {
    __shared__ f4 lds[1024];
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{ 0, 0, 0, 0 };
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    lds[threadIdx.x] = f4{ a, b, a, b };
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (MODE == 3) {           // out of place: D != C, ping-pong between two accumulator sets
        f4 acc2[8];
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=&v"(acc2[i]) : "v"(a), "v"(b), "v"(acc[i]));
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=&v"(acc[i]) : "v"(a), "v"(b), "v"(acc2[i]));
        }
    } else if (MODE == 4) {    // in place, fragment reads one iteration ahead (two register sets)
        f4 x0 = lds[threadIdx.x], y0 = lds[(threadIdx.x + 64) & 1023], x1, y1;
        for (int it = 0; it < iters; it += 2) {
            x1 = lds[(threadIdx.x + it + 1) & 1023]; y1 = lds[(threadIdx.x + it + 65) & 1023];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, i < 4 ? x0[i & 3] : y0[i & 3], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            x0 = lds[(threadIdx.x + it + 2) & 1023]; y0 = lds[(threadIdx.x + it + 66) & 1023];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, i < 4 ? x1[i & 3] : y1[i & 3], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        } else {
            f4 x = f4{ b, b, b, b }, y = x;
            if (MODE == 2) { x = lds[(threadIdx.x + it) & 1023]; y = lds[(threadIdx.x + it + 64) & 1023]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(i < 4 ? x[i & 3] : y[i & 3]));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    f4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE> int run(int threads, int grid, const char *name) {
    const int iters = 20000;
    float *out; unsigned long long *tk;
    CHK(hipMalloc(&out, (size_t)grid * threads * 4)); CHK(hipMalloc(&tk, (size_t)grid * (threads / 64) * 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, out, tk, iters);
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, out, tk, iters);
    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * (threads / 64));
    CHK(hipMemcpy(h.data(), tk, h.size() * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += (double)v; s /= h.size();
    const double waves_per_simd = threads / 256.0;
    printf("%-34s %d thr x %d wg: %.2f ticks per MFMA and wavefront, %.2f per MFMA and SIMD; kernel %.3f ms -> %.2f GHz tick rate, %.1f TFLOP/s\n", name, threads, grid,
           s / (iters * 8.0), s / (iters * 8.0) / waves_per_simd, ms, s / (ms * 1e-3) / 1e9, 2048.0 * iters * 8 * (threads / 64) * grid / (ms * 1e-3) / 1e12);
    return 0;
}
int main() {
    for (int grid : { 1, 256 }) {
        run<3>(256, grid, "asm out of place, 1 wave/SIMD");
        run<4>(256, grid, "builtin + prefetched ds_read, 1 w/S");
        run<4>(512, grid, "builtin + prefetched ds_read, 2 w/S");
        run<0>(256, grid, "builtin, 1 wave/SIMD");
        run<1>(256, grid, "asm in place, 1 wave/SIMD");
        run<2>(256, grid, "asm + 2 ds_read, 1 wave/SIMD");
        run<0>(512, grid, "builtin, 2 waves/SIMD");
        run<1>(512, grid, "asm in place, 2 waves/SIMD");
        run<2>(512, grid, "asm + 2 ds_read, 2 waves/SIMD");
        run<1>(768, grid, "asm in place, 3 waves/SIMD");
    }
    return 0;
}
