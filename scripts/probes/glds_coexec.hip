// One loader wavefront's LDS-DMA issue rate BESIDE eight other wavefronts of its workgroup that run (a) MFMA loops, (b) ds_read_b128
// loops, (c) both, (d) nothing -- who slows the loader down?   build: hipcc --offload-arch=gfx950 -O3 scripts/probes/glds_coexec.hip -o ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f4_t __attribute__((ext_vector_type(4)));

// NLD loader wavefronts (waves 8 ..); PATH 0: LDS-DMA, 1: global_load_dwordx4 to registers + ds_write_b128 (4 pieces in flight per loader)
template <int WORK, int PRIO, int NLD = 1, int PATH = 0>
__global__ __launch_bounds__(512 + 64 * NLD) void k(const float *src, int n, float *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    __shared__ int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave >= 8) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        const unsigned lds0 = (unsigned)(unsigned long long)smem + 65536 + (wave - 8) * 8192;
        const unsigned voff = lane * 16;
        const char *base = (const char *)src + (size_t)(blockIdx.x % 8) * (1 << 20);
        unsigned keep;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (PATH == 0) {
            for (int i = wave - 8; i < n; i += NLD) {
                const char *b = base + (size_t)((i * 37) & 1023) * 1024;
                const unsigned d = lds0 + (unsigned)((i / NLD) & 7) * 1024;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(b), "s"(d) : "memory");
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f4_t r[4];
            char *wr = smem + 65536 + (wave - 8) * 8192 + lane * 16;
            for (int i = wave - 8; i < n; i += 4 * NLD) {
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = *(const f4_t *)(base + (size_t)(((i + q * NLD) * 37) & 1023) * 1024 + voff);
#pragma unroll
                for (int q = 0; q < 4; ++q) *(f4_t *)(wr + q * 1024) = r[q];
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { cyc[blockIdx.x * 8 + (wave - 8)] = t1 - t0; if (wave == 8) __hip_atomic_store(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        return;
    }
    f4_t acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = f4_t{ 0, 0, 0, 0 };
    float a = lane * 0.001f;
    f4_t b = { 1.f, 2.f, 3.f, 4.f }, b2 = b;
    const char *rd = smem + (wave * 64 + lane) * 16;
    for (int it = 0; it < 400000; ++it) {
        if (WORK & 2) { b = *(volatile f4_t *)(rd + (it & 3) * 8192); b2 = *(volatile f4_t *)(rd + 32768 + (it & 3) * 8192); a = *(volatile float *)(rd + 4 * (it & 7)); }
        if (WORK & 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[q], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[4 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b2[q], acc[4 + q], 0, 0, 0);
        }
        if (!(WORK & 1) && !(WORK & 2)) __builtin_amdgcn_s_sleep(8);
        if ((it & 15) == 0 && __hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
    }
    float s = a + b[0] + b2[1];
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int WORK, int PRIO, int NLD = 1, int PATH = 0> void run(const float *src, int n, float *out, unsigned long long *cyc, const char *what) {
    CHK(hipFuncSetAttribute((const void *)k<WORK, PRIO, NLD, PATH>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<WORK, PRIO, NLD, PATH>), dim3(256), dim3(512 + 64 * NLD), 131072, 0, src, n, out, cyc);
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<WORK, PRIO, NLD, PATH>), dim3(256), dim3(512 + 64 * NLD), 131072, 0, src, n, out, cyc);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%d loader(s), %s, beside %-26s prio %d: %7.1f ns per piece per CU (wall) = %5.1f GB/s per CU\n", NLD, PATH ? "registers + ds_write" : "LDS-DMA            ", what, PRIO,
           ms * 1e6 / n, 1024.0 / (ms * 1e6 / n));
}

int main() {
    const int n = 1 << 14;
    float *src, *out; unsigned long long *cyc;
    CHK(hipMalloc(&src, 8 << 20)); CHK(hipMemset(src, 0, 8 << 20)); CHK(hipMalloc(&out, 4096)); CHK(hipMalloc(&cyc, 256 * 64));
    run<0, 0>(src, n, out, cyc, "idle wavefronts (s_sleep)");
    run<1, 0>(src, n, out, cyc, "MFMA loops");
    run<2, 0>(src, n, out, cyc, "ds_read loops");
    run<3, 0>(src, n, out, cyc, "ds_read + MFMA loops");
    run<3, 1>(src, n, out, cyc, "ds_read + MFMA loops");
    run<0, 0, 4>(src, n, out, cyc, "idle wavefronts (s_sleep)");
    run<3, 0, 4>(src, n, out, cyc, "ds_read + MFMA loops");
    run<3, 0, 8>(src, n, out, cyc, "ds_read + MFMA loops");
    run<0, 0, 1, 1>(src, n, out, cyc, "idle wavefronts (s_sleep)");
    run<3, 0, 1, 1>(src, n, out, cyc, "ds_read + MFMA loops");
    run<0, 0, 4, 1>(src, n, out, cyc, "idle wavefronts (s_sleep)");
    run<3, 0, 4, 1>(src, n, out, cyc, "ds_read + MFMA loops");
    run<1, 0, 4, 1>(src, n, out, cyc, "MFMA loops");
    run<1, 0, 4, 0>(src, n, out, cyc, "MFMA loops");
    return 0;
}
