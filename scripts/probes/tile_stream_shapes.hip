// Probe (gfx950): how fast does a read-modify-write pass stream the F64 tile store when each lane's elements are laid out as the F64
// matrix-core flush needs them (k_flush_mfma: a wavefront owns 16 rows x 128 columns, lane (lr, lc) register (bp, r) = row lr + 4r,
// 16 bytes at column 32 bp + 2 lc), as a function of the work item's size and of how loads and stores are ordered in time?
// The flush with no matrix work and no operand staging takes the same 0.553 ms at 10 000 landmarks as the full kernel at 20 pairs
// (profiles/round4_tuning.md 58) while the one-pair VALU pass (one tile row per wavefront, 4 KiB workgroups) streams the same
// bytes in 0.511: the matrix work is hidden, the item shape is what costs.  Variants (all add 1.0 to every element, in place):
//   0  row-per-wavefront, 4 KiB workgroups (k_downdate_w's shape)
//   1  64 x 128 items, 256 threads, 4 wavefronts / SIMD, one item per workgroup (the production flush's shape)
//   2  64 x 64 items, 5 wavefronts / SIMD (the production shape up to 12 pairs)
//   3  64 x 64 items, persistent workgroups, the NEXT item's tile requested before this item's stores (wpe from argv)
//   4  64 x 128 items, persistent, next item's tile requested before this item's stores (128 data registers: wpe <= 3)
//   5  64 x 128 items, persistent, in two column halves: half B of this item / half A of the next in flight behind each half's stores
//  10  64 x 128 items, one per workgroup, the four 32-column groups rolling (group g+1 requested before group g is stored)
//   6  as 1, 16 x 128 per WAVEFRONT with 64-thread workgroups (no workgroup-level granularity at all)
// Usage: tile_stream_shapes [landmarks=10000] [reps=20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int T = 128;
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ inline d2 ldnt(const double *p) { return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(p)); }
__device__ inline void stnt(double *p, d2 v) { __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(p)); }

template <int kWpe>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kWpe, kWpe)))
void k_rows_w(double *t, int64_t n16) {                                    // variant 0 at a given occupancy
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    d2 v = ldnt(t + 2 * i);
    v += 1.0;
    stnt(t + 2 * i, v);
}

__global__ __launch_bounds__(256) void k_rows(double *t, int64_t n16) {   // variant 0
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    d2 v = ldnt(t + 2 * i);
    v += 1.0;
    stnt(t + 2 * i, v);
}

// element offset of (item, wave, lane, bp, r) for kCols-column items of 64 rows
template <int kCols>
__device__ inline int64_t item_base(int64_t item, int wave, int lane) {
    constexpr int kColParts = T / kCols, kSubs = (T / 64) * kColParts;
    const int64_t tile = item / kSubs;
    const int sub = (int)(item - tile * kSubs);
    const int slab = sub / kColParts, cpart = sub - slab * kColParts;
    const int lr = lane >> 4, lc = lane & 15;
    return tile * (int64_t)(T * T) + (int64_t)(slab * 64 + wave * 16 + lr) * T + cpart * kCols + 2 * lc;
}

template <int kCols, int kWpe>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kWpe, kWpe)))
void k_item(double *t, int64_t nitems) {                                  // variants 1, 2
    constexpr int kBP = kCols / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        double *p = t + item_base<kCols>(it, wave, lane);
        d2 a[kBP][4];
#pragma unroll
        for (int bp = 0; bp < kBP; ++bp)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[bp][r] = ldnt(p + (4 * r) * T + 32 * bp);
#pragma unroll
        for (int bp = 0; bp < kBP; ++bp)
#pragma unroll
            for (int r = 0; r < 4; ++r) stnt(p + (4 * r) * T + 32 * bp, a[bp][r] + 1.0);
    }
}

template <int kCols, int kWpe>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kWpe, kWpe)))
void k_item_pipe(double *t, int64_t nitems) {                             // variants 3, 4
    constexpr int kBP = kCols / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t it = blockIdx.x;
    if (it >= nitems) return;
    d2 nx[kBP][4];
    double *p = t + item_base<kCols>(it, wave, lane);
#pragma unroll
    for (int bp = 0; bp < kBP; ++bp)
#pragma unroll
        for (int r = 0; r < 4; ++r) nx[bp][r] = ldnt(p + (4 * r) * T + 32 * bp);
    for (; it < nitems; it += gridDim.x) {
        d2 a[kBP][4];
#pragma unroll
        for (int bp = 0; bp < kBP; ++bp)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[bp][r] = nx[bp][r];
        const int64_t itn = it + gridDim.x;
        if (itn < nitems) {
            const double *pn = t + item_base<kCols>(itn, wave, lane);
#pragma unroll
            for (int bp = 0; bp < kBP; ++bp)
#pragma unroll
                for (int r = 0; r < 4; ++r) nx[bp][r] = ldnt(pn + (4 * r) * T + 32 * bp);
        }
#pragma unroll
        for (int bp = 0; bp < kBP; ++bp)
#pragma unroll
            for (int r = 0; r < 4; ++r) stnt(p + (4 * r) * T + 32 * bp, a[bp][r] + 1.0);
        p = t + item_base<kCols>(itn < nitems ? itn : it, wave, lane);
    }
}

template <int kWpe>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kWpe, kWpe)))
void k_item_halves(double *t, int64_t nitems) {                           // variant 5: 64 x 128 items as two 64-column halves, rolling
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t it = blockIdx.x;
    if (it >= nitems) return;
    d2 ha[2][4], hb[2][4];
    double *p = t + item_base<128>(it, wave, lane);
#pragma unroll
    for (int bp = 0; bp < 2; ++bp)
#pragma unroll
        for (int r = 0; r < 4; ++r) ha[bp][r] = ldnt(p + (4 * r) * T + 32 * bp);
#pragma unroll
    for (int bp = 0; bp < 2; ++bp)
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[bp][r] = ldnt(p + (4 * r) * T + 32 * (bp + 2));
    for (; it < nitems; it += gridDim.x) {
        const int64_t itn = it + gridDim.x;
        double *pn = t + item_base<128>(itn < nitems ? itn : it, wave, lane);
        d2 o[2][4];
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[bp][r] = ha[bp][r] + 1.0;
        if (itn < nitems) {
#pragma unroll
            for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                for (int r = 0; r < 4; ++r) ha[bp][r] = ldnt(pn + (4 * r) * T + 32 * bp);
        }
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int r = 0; r < 4; ++r) stnt(p + (4 * r) * T + 32 * bp, o[bp][r]);
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[bp][r] = hb[bp][r] + 1.0;
        if (itn < nitems) {
#pragma unroll
            for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                for (int r = 0; r < 4; ++r) hb[bp][r] = ldnt(pn + (4 * r) * T + 32 * (bp + 2));
        }
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int r = 0; r < 4; ++r) stnt(p + (4 * r) * T + 32 * (bp + 2), o[bp][r]);
        p = pn;
    }
}

__global__ __launch_bounds__(64) void k_wave_item(double *t, int64_t nwitems) {   // variant 6: 16 x 128 per single-wavefront workgroup
    const int lane = threadIdx.x;
    const int64_t wi = blockIdx.x;
    if (wi >= nwitems) return;
    double *p = t + item_base<128>(wi >> 2, (int)(wi & 3), lane);
    d2 a[4][4];
#pragma unroll
    for (int bp = 0; bp < 4; ++bp)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[bp][r] = ldnt(p + (4 * r) * T + 32 * bp);
#pragma unroll
    for (int bp = 0; bp < 4; ++bp)
#pragma unroll
        for (int r = 0; r < 4; ++r) stnt(p + (4 * r) * T + 32 * bp, a[bp][r] + 1.0);
}

// variant 7: one wavefront per workgroup, 16 rows x 128 columns, every instruction one CONTIGUOUS 1 KiB tile row (not the matrix-core layout): the
// burst depth of variant 6 with the access pattern of variant 0
__global__ __launch_bounds__(64) void k_wave_rows(double *t, int64_t nwitems) {
    const int lane = threadIdx.x;
    const int64_t wi = blockIdx.x;
    if (wi >= nwitems) return;
    double *p = t + wi * (16 * T) + 2 * lane;
    d2 a[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = ldnt(p + r * T);
#pragma unroll
    for (int r = 0; r < 16; ++r) stnt(p + r * T, a[r] + 1.0);
}

// variant 8: one wavefront per workgroup, matrix-core layout, but only ONE 32-column group (four instructions): kDepth-fold smaller bursts
__global__ __launch_bounds__(64) void k_wave_quarter(double *t, int64_t nq) {
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    if (q >= nq) return;
    double *p = t + item_base<128>(q >> 4, (int)((q >> 2) & 3), lane) + 32 * (int)(q & 3);
    d2 a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = ldnt(p + (4 * r) * T);
#pragma unroll
    for (int r = 0; r < 4; ++r) stnt(p + (4 * r) * T, a[r] + 1.0);
}

// variant 9: 64 x 128 items, persistent, rolling by 32-column QUARTERS: quarter q+kAhead requested before quarter q is stored
template <int kWpe, int kAhead>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kWpe, kWpe)))
void k_item_quarters(double *t, int64_t nitems) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nq = ((nitems - blockIdx.x + gridDim.x - 1) / gridDim.x) * 4;      // quarters this workgroup walks
    if ((int64_t)blockIdx.x >= nitems) return;
    auto qaddr = [&](int64_t q) { return t + item_base<128>(blockIdx.x + (q >> 2) * gridDim.x, wave, lane) + 32 * (int)(q & 3); };
    d2 buf[kAhead + 1][4];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
        if (j < nq) { const double *p = qaddr(j);
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[j][r] = ldnt(p + (4 * r) * T); }
    }
    for (int64_t q0 = 0; q0 < nq; q0 += kAhead + 1) {
#pragma unroll
        for (int j = 0; j <= kAhead; ++j) {
            const int64_t q = q0 + j;
            if (q < nq) {
                if (q + kAhead < nq) { const double *pn = qaddr(q + kAhead);
#pragma unroll
                    for (int r = 0; r < 4; ++r) buf[(j + kAhead) % (kAhead + 1)][r] = ldnt(pn + (4 * r) * T); }
                double *p = qaddr(q);
#pragma unroll
                for (int r = 0; r < 4; ++r) stnt(p + (4 * r) * T, buf[j][r] + 1.0);
            }
        }
    }
}

// variant 10: 64 x 128 items, one item per workgroup (not persistent), the item's four 32-column groups rolling: group g+1 requested before group g
// is stored -- what a pass with ALL pairs of an item staged in LDS (column-group-outer matrix loop) would stream like
template <int kWpe>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kWpe, kWpe)))
void k_item_rollq(double *t, int64_t nitems) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t it = blockIdx.x;
    if (it >= nitems) return;
    double *p = t + item_base<128>(it, wave, lane);
    d2 a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = ldnt(p + (4 * r) * T);
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = ldnt(p + (4 * r) * T + 32);
#pragma unroll
    for (int r = 0; r < 4; ++r) stnt(p + (4 * r) * T, a[r] + 1.0);
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = ldnt(p + (4 * r) * T + 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) stnt(p + (4 * r) * T + 32, b[r] + 1.0);
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = ldnt(p + (4 * r) * T + 96);
#pragma unroll
    for (int r = 0; r < 4; ++r) stnt(p + (4 * r) * T + 64, a[r] + 1.0);
#pragma unroll
    for (int r = 0; r < 4; ++r) stnt(p + (4 * r) * T + 96, b[r] + 1.0);
}

__global__ void k_check(const double *t, int64_t n, double want, unsigned long long *bad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && t[i] != want) atomicAdd(bad, 1ull);
}

int main(int argc, char **argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 10000;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int64_t nt1 = (2 * N + T - 1) / T, ntiles = nt1 * (nt1 + 1) / 2;
    const int64_t elems = ntiles * T * T;
    const double bytes = 2.0 * 8.0 * (double)elems;                       // read + write of what is stored (algorithmic: 8 n (n + 1), ~1.6 % less)
    double *t;
    CHECK(hipMalloc(&t, elems * 8));
    CHECK(hipMemset(t, 0, elems * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int ncu = 256;
    int64_t launches = 0;
    const int cold_ms = getenv("COLD") ? atoi(getenv("COLD")) : 0;     // idle time in front of every timed block (and no warm-up launches then)
    const int gap_us = getenv("GAP") ? atoi(getenv("GAP")) : -1;       // >= 0: launches timed one by one with that idle time in front of each
    auto timeit = [&](const char *name, auto launch) {
        launches += (cold_ms ? 0 : 3) + reps;
        if (!cold_ms) for (int i = 0; i < 3; ++i) launch();
        CHECK(hipDeviceSynchronize());
        if (cold_ms) usleep(1000 * cold_ms);
        float ms = 0.f;
        if (gap_us >= 0) {                                              // every launch timed on its own, the chip idle for gap_us in front of it
            for (int i = 0; i < reps; ++i) {
                CHECK(hipDeviceSynchronize());
                if (gap_us) usleep(gap_us);
                CHECK(hipEventRecord(e0));
                launch();
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float m1; CHECK(hipEventElapsedTime(&m1, e0, e1));
                ms += m1;
            }
        } else {
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        CHECK(hipGetLastError());
        ms /= reps;
        printf("%-44s %.4f ms  %.3f TB/s stored  (of 8 TB/s: %.3f)\n", name, ms, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0);
        fflush(stdout);
    };
    const int64_t n16 = elems / 2, it128 = ntiles * 2, it64 = ntiles * 4;
    for (int round = 0; round < 2; ++round) {
        timeit("0 rows, 4 KiB workgroups", [&] { hipLaunchKernelGGL(k_rows, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, t, n16); });
        timeit("0 rows, wpe 2", [&] { hipLaunchKernelGGL((k_rows_w<2>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, t, n16); });
        timeit("0 rows, wpe 3", [&] { hipLaunchKernelGGL((k_rows_w<3>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, t, n16); });
        timeit("0 rows, wpe 4", [&] { hipLaunchKernelGGL((k_rows_w<4>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, t, n16); });
        timeit("0 rows, wpe 6", [&] { hipLaunchKernelGGL((k_rows_w<6>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, t, n16); });
        timeit("1 64x128 items, wpe 4", [&] { hipLaunchKernelGGL((k_item<128, 4>), dim3((unsigned)it128), dim3(256), 0, 0, t, it128); });
        timeit("2 64x64 items, wpe 5", [&] { hipLaunchKernelGGL((k_item<64, 5>), dim3((unsigned)it64), dim3(256), 0, 0, t, it64); });
        timeit("2 64x64 items, wpe 2", [&] { hipLaunchKernelGGL((k_item<64, 2>), dim3((unsigned)it64), dim3(256), 0, 0, t, it64); });
        timeit("2 64x64 items, wpe 3", [&] { hipLaunchKernelGGL((k_item<64, 3>), dim3((unsigned)it64), dim3(256), 0, 0, t, it64); });
        timeit("2 64x64 items, wpe 4", [&] { hipLaunchKernelGGL((k_item<64, 4>), dim3((unsigned)it64), dim3(256), 0, 0, t, it64); });
        timeit("1 64x128 items, wpe 1", [&] { hipLaunchKernelGGL((k_item<128, 1>), dim3((unsigned)it128), dim3(256), 0, 0, t, it128); });
        timeit("1 64x128 items, wpe 2", [&] { hipLaunchKernelGGL((k_item<128, 2>), dim3((unsigned)it128), dim3(256), 0, 0, t, it128); });
        timeit("1 64x128 items, wpe 3", [&] { hipLaunchKernelGGL((k_item<128, 3>), dim3((unsigned)it128), dim3(256), 0, 0, t, it128); });
        timeit("2b 64x64 items, wpe 8", [&] { hipLaunchKernelGGL((k_item<64, 8>), dim3((unsigned)it64), dim3(256), 0, 0, t, it64); });
        timeit("3 64x64 persistent + prefetch, wpe 4", [&] { hipLaunchKernelGGL((k_item_pipe<64, 4>), dim3(ncu * 4), dim3(256), 0, 0, t, it64); });
        timeit("3 64x64 persistent + prefetch, wpe 5", [&] { hipLaunchKernelGGL((k_item_pipe<64, 5>), dim3(ncu * 5), dim3(256), 0, 0, t, it64); });
        timeit("3 64x64 persistent + prefetch, wpe 6", [&] { hipLaunchKernelGGL((k_item_pipe<64, 6>), dim3(ncu * 6), dim3(256), 0, 0, t, it64); });
        timeit("4 64x128 persistent + prefetch, wpe 2", [&] { hipLaunchKernelGGL((k_item_pipe<128, 2>), dim3(ncu * 2), dim3(256), 0, 0, t, it128); });
        timeit("4 64x128 persistent + prefetch, wpe 3", [&] { hipLaunchKernelGGL((k_item_pipe<128, 3>), dim3(ncu * 3), dim3(256), 0, 0, t, it128); });
        timeit("5 64x128 persistent, rolling halves, wpe 3", [&] { hipLaunchKernelGGL((k_item_halves<3>), dim3(ncu * 3), dim3(256), 0, 0, t, it128); });
        timeit("1p 64x128 items, persistent no prefetch, wpe 4", [&] { hipLaunchKernelGGL((k_item<128, 4>), dim3(ncu * 4), dim3(256), 0, 0, t, it128); });
        timeit("7 16 contiguous rows per 64-thread workgroup", [&] { hipLaunchKernelGGL(k_wave_rows, dim3((unsigned)(it128 * 4)), dim3(64), 0, 0, t, it128 * 4); });
        timeit("8 16x32 (one column group) per 64-thread wg", [&] { hipLaunchKernelGGL(k_wave_quarter, dim3((unsigned)(it128 * 16)), dim3(64), 0, 0, t, it128 * 16); });
        timeit("9 64x128 persistent, rolling quarters +1, wpe 4", [&] { hipLaunchKernelGGL((k_item_quarters<4, 1>), dim3(ncu * 4), dim3(256), 0, 0, t, it128); });
        timeit("9 64x128 persistent, rolling quarters +2, wpe 4", [&] { hipLaunchKernelGGL((k_item_quarters<4, 2>), dim3(ncu * 4), dim3(256), 0, 0, t, it128); });
        timeit("9 64x128 persistent, rolling quarters +3, wpe 4", [&] { hipLaunchKernelGGL((k_item_quarters<4, 3>), dim3(ncu * 4), dim3(256), 0, 0, t, it128); });
        timeit("9 64x128 persistent, rolling quarters +3, wpe 8", [&] { hipLaunchKernelGGL((k_item_quarters<8, 3>), dim3(ncu * 8), dim3(256), 0, 0, t, it128); });
        timeit("9 64x128 persistent, rolling quarters +7, wpe 4", [&] { hipLaunchKernelGGL((k_item_quarters<4, 7>), dim3(ncu * 4), dim3(256), 0, 0, t, it128); });
        timeit("10 64x128 items, rolling column groups, wpe 2", [&] { hipLaunchKernelGGL((k_item_rollq<2>), dim3((unsigned)it128), dim3(256), 0, 0, t, it128); });
        timeit("10 64x128 items, rolling column groups, wpe 3", [&] { hipLaunchKernelGGL((k_item_rollq<3>), dim3((unsigned)it128), dim3(256), 0, 0, t, it128); });
        timeit("10 64x128 items, rolling column groups, wpe 4", [&] { hipLaunchKernelGGL((k_item_rollq<4>), dim3((unsigned)it128), dim3(256), 0, 0, t, it128); });
        timeit("10 64x128 items, rolling column groups, wpe 8", [&] { hipLaunchKernelGGL((k_item_rollq<8>), dim3((unsigned)it128), dim3(256), 0, 0, t, it128); });
        timeit("6 16x128 per 64-thread workgroup", [&] { hipLaunchKernelGGL(k_wave_item, dim3((unsigned)(it128 * 4)), dim3(64), 0, 0, t, it128 * 4); });
    }
    // every variant touches every element exactly once per launch: all elements must hold the same count
    unsigned long long *bad, hbad = 0;
    CHECK(hipMalloc(&bad, 8));
    CHECK(hipMemset(bad, 0, 8));
    const double want = (double)launches;
    hipLaunchKernelGGL(k_check, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, 0, t, elems, want, bad);
    CHECK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost));
    printf("check: %llu of %lld elements differ from %.0f\n", hbad, (long long)elems, want);
    return hbad == 0 ? 0 : 2;
}
