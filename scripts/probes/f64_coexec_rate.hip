// Probe: do f64 MFMA and f64 VALU FMA execute CONCURRENTLY on gfx950, or do they share one f64 datapath?
// One workgroup per CU; its wavefronts land round-robin on the 4 SIMDs.  `wm` wavefronts per SIMD run a v_mfma_f64_16x16x4_f64 loop
// (8 independent accumulators), `wv` per SIMD a v_fma_f64 loop (32 independent accumulators); the iteration counts are balanced
// so that either kind alone takes about the same time; then both kinds run in ONE workgroup (each SIMD hosts both).
// Separate pipes: together ~ max(t_mfma, t_valu).  Shared f64 datapath: together ~ t_mfma + t_valu.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k(double* out, int iters_m, int iters_v, int mfma_waves, double a0, double b0) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave < mfma_waves) {
        d4 acc[8];
        for (int q = 0; q < 8; ++q) acc[q] = d4{0, 0, 0, 0};
        double a = a0 + threadIdx.x * 1e-9, b = b0;
        for (int i = 0; i < iters_m; ++i) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    } else {
        double acc[32];
        for (int q = 0; q < 32; ++q) acc[q] = q * 1e-3;
        const double a = a0 + threadIdx.x * 1e-9, b = b0;
        for (int i = 0; i < iters_v; ++i) {
#pragma unroll
            for (int q = 0; q < 32; ++q) acc[q] = fma(a, acc[q], b);
        }
        for (int q = 0; q < 32; ++q) s += acc[q];
    }
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}
static float run(double* out, int threads, int im, int iv, int mfma_waves) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, threads>>>(out, 50, 50, mfma_waves, 1.0, 1e-3); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<256, threads>>>(out, im, iv, mfma_waves, 0.999, 1e-3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    double* out; if (hipMalloc(&out, 256 * 1024 * 8) != hipSuccess) return 1;
    const int im = 10000;
    for (int w = 1; w <= 2; ++w) {           // w wavefronts per SIMD of EACH kind
        const int nw = 4 * w;                // wavefronts of one kind per workgroup
        const float tm = run(out, 64 * nw, im, 1, nw);                   // MFMA wavefronts only
        float tv1 = run(out, 64 * nw, 1, 20000, 0);                      // VALU wavefronts only, calibration
        const int iv = (int)(20000.0 * tm / tv1);                        // ... balanced against the MFMA time
        const float tv = run(out, 64 * nw, 1, iv, 0);
        const float tb = run(out, 128 * nw, im, iv, nw);                 // both kinds in one workgroup: every SIMD hosts w of each
        printf("%d wave(s)/SIMD of each kind: MFMA alone %.3f ms (%.1f TFLOP/s), VALU alone %.3f ms (%.1f TFLOP/s), together %.3f ms "
               "(sum %.3f, max %.3f) -> %s\n", w, tm, 256.0 * nw * im * 8 * 2048 / tm * 1e-9, tv,
               256.0 * nw * 64 * iv * 32 * 2 / tv * 1e-9, tb, tm + tv, tm > tv ? tm : tv,
               tb < 0.75 * (tm + tv) ? "CONCURRENT (separate pipes)" : "SERIAL (shared f64 datapath)");
    }
    return 0;
}
