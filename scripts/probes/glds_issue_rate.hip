// What one wavefront's LDS-DMA (global_load_lds_dwordx4) stream costs, by how M0 (the LDS destination base) is handled.
//   build: hipcc --offload-arch=gfx950 -O3 scripts/probes/glds_issue_rate.hip -o scripts/probes/glds_issue_rate
// One 64-thread workgroup per CU (256), each issuing `n` 1 KiB pieces from an L2-resident region into a ring of LDS slots, at most
// `depth` outstanding (counted vmcnt).  Modes:
//   0  M0 saved, set, restored around EVERY piece (the recipe of flush32_pipe.h::glds16)
//   1  M0 set before every piece, not restored
//   2  M0 set once per 4 pieces, the other three addressed by the instruction's offset (it moves the LDS AND the global address)
//   3  M0 set once for the whole stream (every piece lands on the same KiB): the bare issue rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE, int DEPTH>
__global__ __launch_bounds__(64) void k(const float *src, int n, int region_kib, float *out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)smem;
    const unsigned voff = threadIdx.x * 16;
    const char *base = (const char *)src + (size_t)(blockIdx.x % 8) * region_kib * 1024;     // one region per XCD label
    unsigned keep;
    for (int i = 0; i < n; i += 4) {
        const char *b = base + (size_t)((i * 37) % region_kib) * 1024;
        const unsigned d = lds0 + (unsigned)(i & 28) * 1024;
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(b + q * 1024), "s"(d + q * 1024) : "memory");
        } else if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(b + q * 1024), "s"(d + q * 1024) : "memory");
        } else if (MODE == 2) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(voff), "s"(b), "s"(d) : "memory");
        } else {
            if (i == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds0) : "memory");
            asm volatile("global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(voff), "s"(b) : "memory");
        }
        if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (DEPTH == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (out) out[blockIdx.x * 64 + threadIdx.x] = *(float *)(smem + threadIdx.x * 4);
}

template <int MODE, int DEPTH> void run(const float *src, int n, int region_kib, float *out, const char *what) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(64), 32768, 0, src, n, region_kib, out);
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(64), 32768, 0, src, n, region_kib, out);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d depth %2d region %5d KiB/XCD: %7.1f ns per piece per wavefront = %6.1f GB/s per CU   (%s)\n", MODE, DEPTH, region_kib, ms * 1e6 / n,
           1024.0 / (ms * 1e6 / n), what);
}

int main() {
    const int n = 1 << 16;
    float *src, *out;
    const size_t bytes = (size_t)8 * 65536 * 1024;          // 512 MiB: 8 regions of up to 64 MiB
    CHK(hipMalloc(&src, bytes)); CHK(hipMemset(src, 0, bytes)); CHK(hipMalloc(&out, 256 * 64 * 4));
    for (int region : { 1024, 65536 }) {
        run<0, 32>(src, n, region, out, "M0 saved / set / restored per piece");
        run<1, 32>(src, n, region, out, "M0 set per piece");
        run<2, 32>(src, n, region, out, "M0 set per 4 pieces, offset: for the rest");
        run<3, 32>(src, n, region, out, "M0 set once");
        run<2, 8>(src, n, region, out, "M0 per 4 pieces, 8-12 outstanding");
        run<2, 60>(src, n, region, out, "M0 per 4 pieces, 60 outstanding");
        run<0, 60>(src, n, region, out, "M0 saved / set / restored, 60 outstanding");
    }
    return 0;
}
