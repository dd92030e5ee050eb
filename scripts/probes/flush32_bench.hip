// Probe for the F32-arithmetic pass over float tiles at 57-64 pairs: k_flush_strip32 (flush32_pipe.h: row strips, -K resident in LDS, loader
// wavefronts, persistent) against k_flush_mfma32 (flush32_mfma.h, one work item per workgroup) and against a scalar fmaf reference, in ONE
// process on one device.
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ekf_slam_amd/csrc scripts/probes/flush32_bench.hip -o scripts/probes/flush32_bench
//   run:    flush32_bench <landmarks> <npairs> [rounds=5] [check=1] [pstart=0] [reverse=0] [grid=256]      (STAMP=2: where a stage's cycles go)
// check: every output entry of every kernel bit-equal to  tile + fmaf-chain(from zero, ring order)  (out of place, all tiles).
// timing: in place, interleaved rounds, HIP events, median / min per kernel; TFLOP/s = 2 m n(n+1) / t, TB/s = 4 n(n+1) / t.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "flush32_mfma.h"
#include "flush32_pipe.h"
#include "flush32_split.h"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ uint32_t mix(uint64_t i, uint32_t seed) {
    uint64_t z = i * 0x9E3779B97F4A7C15ull + seed;
    z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29; z *= 0x94D049BB133111EBull; z ^= z >> 32;
    return (uint32_t)z;
}
__global__ void k_fill(float *p, int64_t n, uint32_t seed, float scale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = ((int32_t)mix((uint64_t)i, seed) * (1.0f / 2147483648.0f)) * scale;
}
// planar / negated copies from the interleaved pairs: Kil[s][e][xy] -> Kn[s][xy][e] = -K,  Gil -> Gpl
__global__ void k_planar(const float *il, float *pl, int64_t ldm, int64_t pair_stride, int pcap, float sign) {
    const int64_t n = (int64_t)pcap * ldm;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / ldm, e = i - s * ldm;
        pl[s * pair_stride + e] = sign * il[s * pair_stride + 2 * e];
        pl[s * pair_stride + ldm + e] = sign * il[s * pair_stride + 2 * e + 1];
    }
}
// the reference: one thread per entry, fmaf chain from zero in ring order, tile added once
__global__ void k_ref(const float *tiles, float *out, const int2 *work, int64_t nwork, const float *Kil, const float *Gil, int64_t pair_stride,
                      int pstart, int pcap, int npairs, TileMap tm) {
    constexpr int T = 256;
    const int2 ij = work[blockIdx.x / (T * T / 256)];
    const int sub = blockIdx.x % (T * T / 256);
    const int r = sub, c = threadIdx.x;              // 256 threads = one tile row
    const int64_t off = tm.tile_offset(ij.x, ij.y) + (int64_t)r * T + c;
    float acc = 0.0f;
    for (int p = 0; p < npairs; ++p) {
        const int64_t so = (int64_t)ring_slot(pstart, p, pcap) * pair_stride;
        const float kx = Kil[so + 2 * ((int64_t)ij.x * T + r)], ky = Kil[so + 2 * ((int64_t)ij.x * T + r) + 1];
        const float gx = Gil[so + 2 * ((int64_t)ij.y * T + c)], gy = Gil[so + 2 * ((int64_t)ij.y * T + c) + 1];
        acc = fmaf(-kx, gx, acc);
        acc = fmaf(-ky, gy, acc);
    }
    out[off] = tiles[off] + acc;
}
__global__ void k_diff(const float *a, const float *b, int64_t n, unsigned long long *bad, unsigned long long *first) {
    unsigned long long mine = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (__float_as_uint(a[i]) != __float_as_uint(b[i])) { ++mine; atomicMin(first, (unsigned long long)i); }
    if (mine) atomicAdd(bad, mine);
}

// ACC=1: the error of a pass's UPDATE against an F64 sum, entry by entry: err = |out - (tile + sum64)|, S = sum |k||g|; max and mean of err / S
// (with zero tiles, ZERO_TILES=1, `out` IS the float sum and err its accumulation error alone)
__global__ void k_acc(const float *tiles, const float *out, const int2 *work, const float *Kil, const float *Gil, int64_t pair_stride, int pstart, int pcap,
                      int npairs, TileMap tm, unsigned long long *maxbits, double *sum, unsigned long long *cnt, double *ssum) {
    constexpr int T = 256;
    const int2 ij = work[blockIdx.x / (T * T / 256)];
    const int r = blockIdx.x % (T * T / 256), c = threadIdx.x;
    const int64_t off = tm.tile_offset(ij.x, ij.y) + (int64_t)r * T + c;
    double acc = 0.0, S = 0.0;
    for (int p = 0; p < npairs; ++p) {
        const int64_t so = (int64_t)ring_slot(pstart, p, pcap) * pair_stride;
        const double kx = Kil[so + 2 * ((int64_t)ij.x * T + r)], ky = Kil[so + 2 * ((int64_t)ij.x * T + r) + 1];
        const double gx = Gil[so + 2 * ((int64_t)ij.y * T + c)], gy = Gil[so + 2 * ((int64_t)ij.y * T + c) + 1];
        acc += -kx * gx; acc += -ky * gy;
        S += fabs(kx * gx) + fabs(ky * gy);
    }
    const double err = fabs((double)out[off] - ((double)tiles[off] + acc)) / (S > 0 ? S : 1.0);
    atomicMax(maxbits, (unsigned long long)__double_as_longlong(err));
    // (one atomic per wavefront would do; this is a probe)
    double e = err, sg = ((double)out[off] - ((double)tiles[off] + acc)) / (S > 0 ? S : 1.0);
    for (int o = 32; o > 0; o >>= 1) { e += __shfl_down(e, o); sg += __shfl_down(sg, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(sum, e); atomicAdd(cnt, 64ull); atomicAdd(ssum, sg); }
}

// PREWARM_US: a register-only MFMA loop on every CU for about that long, launched right in front of each timed launch after the idle gap
__global__ void k_prewarm(float *out, long long cycles) {
    typedef float f4w __attribute__((ext_vector_type(4)));
    f4w acc[4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
    const float a = threadIdx.x * 1e-3f, b = 1.0f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < (1 << 20); ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        if ((it & 63) == 63 && (long long)__builtin_readcyclecounter() - t0 > cycles) break;
    }
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.0f) out[threadIdx.x] = acc[0][1];
}

// PREWARM_MB: stream that many megabytes of the tile store through every CU right in front of each timed launch
__global__ void k_prewarm_mem(const float4 *src, long long n, float *out) {
    float4 a = { 0, 0, 0, 0 };
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { const float4 v = src[i]; a.x += v.x; a.y += v.w; }
    if (a.x + a.y == 12345.678f) out[threadIdx.x] = a.x;
}

int main(int argc, char **argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 4000;
    const int npairs = argc > 2 ? atoi(argv[2]) : 64;
    const int rounds = argc > 3 ? atoi(argv[3]) : 5;
    const int check = argc > 4 ? atoi(argv[4]) : 1;
    const int pstart = argc > 5 ? atoi(argv[5]) : 0;
    const int reverse = argc > 6 ? atoi(argv[6]) : 0;
    const int grid = argc > 7 ? atoi(argv[7]) : 256;
    constexpr int T = 256;
    const int64_t n_mm = 2 * N, nt = (n_mm + T - 1) / T, ldm = nt * T;
    const int pcap = std::max(64, npairs);
    TileMap tm = ekf_make_tilemap(T, 1, 0);
    tm.reverse = reverse;
    // work list as abi.hip::refresh_work builds it: 8 x 8 super-tiles in row-major order, flattened, cut into 8 equal runs
    std::vector<int2> flat;
    const int SS = 8;
    const int64_t ns = (nt + SS - 1) / SS;
    for (int64_t si = 0; si < ns; ++si)
        for (int64_t sj = 0; sj <= si; ++sj)
            for (int64_t I = si * SS; I < nt && I < (si + 1) * SS; ++I)
                for (int64_t J = sj * SS; J <= I && J < (sj + 1) * SS; ++J) flat.push_back(make_int2((int)I, (int)J));
    const size_t tot = flat.size();
    size_t len = 0;
    std::vector<int2> st[8];
    for (int x = 0; x < 8; ++x) { st[x].assign(flat.begin() + (tot * x) / 8, flat.begin() + (tot * (x + 1)) / 8); len = std::max(len, st[x].size()); }
    std::vector<int2> wx(8 * len, make_int2(-1, -1));
    for (int x = 0; x < 8; ++x) std::copy(st[x].begin(), st[x].end(), wx.begin() + x * len);
    int2 *d_work = nullptr, *d_flat = nullptr;
    CHK(hipMalloc(&d_work, wx.size() * sizeof(int2)));
    CHK(hipMemcpy(d_work, wx.data(), wx.size() * sizeof(int2), hipMemcpyHostToDevice));
    CHK(hipMalloc(&d_flat, flat.size() * sizeof(int2)));
    CHK(hipMemcpy(d_flat, flat.data(), flat.size() * sizeof(int2), hipMemcpyHostToDevice));

    // strip segments (k_flush_strip32): column ranges of kSeg 128-column items, within a range the 128-row slabs from the diagonal down;
    // flattened in that order, cut into 8 equal runs (one per XCD stream), interleaved segment by segment, padded to a multiple of 8
    std::vector<int4> segflat;
    int64_t nsegs = 0;
    {
        const int LS = ekf_pipe32::kSeg;
        std::vector<std::vector<int4>> sl;
        const int64_t ncj = 2 * nt;
        for (int64_t c0 = 0; c0 < ncj; c0 += LS)
            for (int64_t rs = 0; rs < 2 * nt; ++rs) {
                const int64_t I = rs >> 1, cmax = 2 * I + 1;
                if (cmax < c0) continue;
                std::vector<int4> sgm;
                for (int64_t cj = c0; cj < c0 + LS && cj <= cmax; ++cj) sgm.push_back(ekf_pipe32::strip_entry(tm, (int)(rs >> 1), (int)(cj >> 1), (int)(rs & 1), (int)(cj & 1)));
                // ROT=1: each row slab starts its walk along the column range at a different item, so that the CUs of an XCD -- all on
                // the same column range -- do not ask for the same G lines at the same moment (every one of them would wait out the miss)
                if (getenv("ROT") && atoi(getenv("ROT")) && sgm.size() > 1) std::rotate(sgm.begin(), sgm.begin() + (rs * 5) % sgm.size(), sgm.end());
                sl.push_back(sgm);
            }
        const size_t ns = sl.size(), per = (ns + 7) / 8;
        nsegs = (int64_t)per * 8;
        segflat.assign((size_t)nsegs * LS, make_int4(0, 0, -1, -1));
        for (int x = 0; x < 8; ++x) {
            const size_t lo = ns * x / 8, hi = ns * (x + 1) / 8;
            for (size_t q = lo; q < hi; ++q) std::copy(sl[q].begin(), sl[q].end(), segflat.begin() + ((q - lo) * 8 + x) * LS);
        }
    }
    int4 *d_segs = nullptr;
    CHK(hipMalloc(&d_segs, segflat.size() * sizeof(int4)));
    CHK(hipMemcpy(d_segs, segflat.data(), segflat.size() * sizeof(int4), hipMemcpyHostToDevice));
    const int64_t telems = (int64_t)tot * T * T, pair_stride = 2 * ldm;
    float *tiles, *out_a = nullptr, *out_b = nullptr, *Kil, *Gil, *Kn, *Gpl;
    CHK(hipMalloc(&tiles, telems * 4));
    CHK(hipMalloc(&Kil, pair_stride * pcap * 4)); CHK(hipMalloc(&Gil, pair_stride * pcap * 4));
    CHK(hipMalloc(&Kn, pair_stride * pcap * 4)); CHK(hipMalloc(&Gpl, pair_stride * pcap * 4));
    k_fill<<<2048, 256>>>(tiles, telems, 1u, (getenv("ZERO_TILES") && atoi(getenv("ZERO_TILES"))) ? 0.0f : 10.0f);
    k_fill<<<1024, 256>>>(Kil, pair_stride * pcap, 2u, 0.05f);
    k_fill<<<1024, 256>>>(Gil, pair_stride * pcap, 3u, 0.05f);
    k_planar<<<1024, 256>>>(Kil, Kn, ldm, pair_stride, pcap, -1.0f);
    k_planar<<<1024, 256>>>(Gil, Gpl, ldm, pair_stride, pcap, 1.0f);
    CHK(hipDeviceSynchronize());

    auto launch_old = [&](float *dstp) {
        const int64_t g32 = 8 * (int64_t)len * (T / 128) * (T / 128);
        hipLaunchKernelGGL((k_flush_mfma32<T, 4, 2, 3, true>), dim3((unsigned)g32), dim3(256), 0, 0, (const float *)tiles, dstp, d_work, (int64_t)len,
                           (const float *)Kn, (const float *)Gpl, pair_stride, pstart, pcap, npairs, tm);
    };
    typedef void (*fns_t)(const float *, float *, const int4 *, int64_t, const float *, const float *, int64_t, int64_t, int, int, int, TileMap, float *, unsigned long long *);
    float *dump;
    CHK(hipMalloc(&dump, (size_t)grid * 2 * 128 * 256 * 4));
    struct VarS { const char *name; fns_t fn; int lds, threads, D; };
    const VarS vs[] = {
        { "k_flush_strip32<8>", ekf_pipe32::k_flush_strip32<8>, ekf_pipe32::lds_bytes_strip<8>(), 512, 2 },
        { "k_flush_strip32<8,.,4> (tile loads from k-step 4)", ekf_pipe32::k_flush_strip32<8, false, 4>, ekf_pipe32::lds_bytes_strip<8>(), 512, 2 },
        { "k_flush_strip32<8,.,8>", ekf_pipe32::k_flush_strip32<8, false, 8>, ekf_pipe32::lds_bytes_strip<8>(), 512, 2 },
        { "k_flush_strip32<8,.,12>", ekf_pipe32::k_flush_strip32<8, false, 12>, ekf_pipe32::lds_bytes_strip<8>(), 512, 2 },
        { "k_flush_strip32<8,.,16>", ekf_pipe32::k_flush_strip32<8, false, 16>, ekf_pipe32::lds_bytes_strip<8>(), 512, 2 },
    };
    constexpr int nvs = sizeof(vs) / sizeof(vs[0]);
    for (int v = 0; v < nvs; ++v) CHK(hipFuncSetAttribute((const void *)vs[v].fn, hipFuncAttributeMaxDynamicSharedMemorySize, vs[v].lds));
    auto launch_strip = [&](float *dstp, int v) {
        hipLaunchKernelGGL(vs[v].fn, dim3(grid), dim3(vs[v].threads), vs[v].lds, 0, (const float *)tiles, dstp, d_segs, nsegs, (const float *)Kn,
                           (const float *)Gpl, pair_stride, ldm, pstart, pcap, npairs, tm, dump, (unsigned long long *)nullptr);
    };

    // split arithmetic (flush32_split.h): the operand planes are cut from the planar float copies in front of every pass
    uint16_t *Kb3, *Gb3;
    CHK(hipMalloc(&Kb3, ekf_pipe32::split_plane_elems(ldm) * 2)); CHK(hipMalloc(&Gb3, ekf_pipe32::split_plane_elems(ldm) * 2));
    CHK(hipMemset(Kb3, 0, ekf_pipe32::split_plane_elems(ldm) * 2)); CHK(hipMemset(Gb3, 0, ekf_pipe32::split_plane_elems(ldm) * 2));
    typedef void (*fsp_t)(const float *, float *, const int4 *, int64_t, const uint16_t *, const uint16_t *, int64_t, TileMap, float *);
    // ABL=1: the ablations beside the kernel (diagnostic instances: results are wrong by construction)
    struct VarP { const char *name; fsp_t fn; int waves; };
    const VarP vp[] = {
        { "k_flush_split3<2>", ekf_pipe32::k_flush_split3<2, 0>, 8 },
        { "k_flush_split3<2,0,4>: two workgroups of four wavefronts per CU", ekf_pipe32::k_flush_split3<2, 0, 4>, 4 },
        { "k_flush_split3<1>: one chunk per item (up to 32 pairs)", ekf_pipe32::k_flush_split3<1, 0>, 8 },
        { "  abl: no tile stores", ekf_pipe32::k_flush_split3<2, 1> , 8 },
        { "  abl: no tile loads", ekf_pipe32::k_flush_split3<2, 2> , 8 },
        { "  abl: no tile traffic", ekf_pipe32::k_flush_split3<2, 3> , 8 },
        { "  abl: no G loads", ekf_pipe32::k_flush_split3<2, 4> , 8 },
        { "  abl: no traffic at all", ekf_pipe32::k_flush_split3<2, 7> , 8 },
        { "  <2,.,4> abl: no tile stores", ekf_pipe32::k_flush_split3<2, 1, 4>, 4 },
        { "  <2,.,4> abl: no tile traffic", ekf_pipe32::k_flush_split3<2, 3, 4>, 4 },
        { "  <2,.,4> abl: no traffic at all", ekf_pipe32::k_flush_split3<2, 7, 4>, 4 },
        { "  var: plain tile stores", ekf_pipe32::k_flush_split3<2, 8> , 8 },
        { "  var: plain tile loads", ekf_pipe32::k_flush_split3<2, 16> , 8 },
        { "  var: plain stores and loads", ekf_pipe32::k_flush_split3<2, 24> , 8 },
        { "  var: tile loads from group 0 (beside the stores)", ekf_pipe32::k_flush_split3<2, 0, 8, 0> , 8 },
        { "  var: tile loads from group 2", ekf_pipe32::k_flush_split3<2, 0, 8, 2> , 8 },
        { "  var: tile loads from group 3", ekf_pipe32::k_flush_split3<2, 0, 8, 3> , 8 },
        { "  var: tile loads from group 5", ekf_pipe32::k_flush_split3<2, 0, 8, 5> , 8 },
        { "  var: tile loads from group 8", ekf_pipe32::k_flush_split3<2, 0, 8, 8> , 8 },
    };
    const int nvp = (getenv("ABL") && atoi(getenv("ABL"))) ? (int)(sizeof(vp) / sizeof(vp[0])) : 3;
    for (int v = 0; v < nvp; ++v) CHK(hipFuncSetAttribute((const void *)vp[v].fn, hipFuncAttributeMaxDynamicSharedMemorySize, ekf_pipe32::lds_bytes_split(vp[v].waves)));
    auto launch_split = [&](float *dstp, bool cut, int v = 0) {
        if (cut) hipLaunchKernelGGL(ekf_pipe32::k_split_pairs, dim3((unsigned)(ldm / 256), ekf_pipe32::kKB, 2), dim3(256), 0, 0, (const float *)Kn, (const float *)Gpl, Kb3, Gb3,
                                    pair_stride, ldm, ldm, pstart, pcap, npairs);
        hipLaunchKernelGGL(vp[v].fn, dim3(grid * 8 / vp[v].waves), dim3(64 * vp[v].waves), ekf_pipe32::lds_bytes_split(vp[v].waves), 0, (const float *)tiles, dstp, d_segs, nsegs,
                           (const uint16_t *)Kb3, (const uint16_t *)Gb3, ldm, tm, dump);
    };
    auto strip_ok = [&](int v) { return (npairs + 7) / 8 == 8; };      // (instantiated for eight stages: 57-64 pairs)
    if (getenv("STAMP") && atoi(getenv("STAMP")) == 2) {
        constexpr int NLs = 0;
        CHK(hipFuncSetAttribute((const void *)ekf_pipe32::k_flush_strip32<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ekf_pipe32::lds_bytes_strip<8>()));
        unsigned long long *d_st;
        CHK(hipMalloc(&d_st, (size_t)grid * (8 + NLs) * 64));
        CHK(hipMemset(d_st, 0, (size_t)grid * (8 + NLs) * 64));
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL((ekf_pipe32::k_flush_strip32<8, true>), dim3(grid), dim3(512 + 64 * NLs), ekf_pipe32::lds_bytes_strip<8>(), 0, (const float *)tiles, tiles, d_segs, nsegs, (const float *)Kn,
                               (const float *)Gpl, pair_stride, ldm, pstart, pcap, npairs, tm, dump, d_st);
        CHK(hipDeviceSynchronize());
        std::vector<unsigned long long> hs((size_t)grid * (8 + NLs) * 8);
        CHK(hipMemcpy(hs.data(), d_st, hs.size() * 8, hipMemcpyDeviceToHost));
        double cs[8] = { 0 }, ls[8] = { 0 };
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < 8 + NLs; ++w)
                for (int q = 0; q < 8; ++q) (w >= 8 ? ls : cs)[q] += (double)hs[((size_t)b * (8 + NLs) + w) * 8 + q];
        const double items = (double)tot * 4 / grid * 2;                // (two launches)
        printf("k_flush_strip32<8,stamp> landmarks %lld pairs %d: s_memtime ticks per item and wavefront (ideal: 16384 for the two MFMA streams of a SIMD)\n", (long long)N, npairs);
        printf("  consumer: k-steps up to the barrier %.1f | lgkmcnt+barrier %.1f | last two k-steps %.1f | -K + wait for the tile loads %.1f | adds %.1f | item switch %.1f\n", cs[0] / (8.0 * grid) / items,
               cs[1] / (8.0 * grid) / items, cs[2] / (8.0 * grid) / items, cs[5] / (8.0 * grid) / items, cs[3] / (8.0 * grid) / items, cs[4] / (8.0 * grid) / items);
        for (int w = 0; w < 8 + NLs; ++w) {
            double a[8] = { 0 };
            for (int b = 0; b < grid; ++b)
                for (int q = 0; q < 8; ++q) a[q] += (double)hs[((size_t)b * (8 + NLs) + w) * 8 + q] / grid / (items / 2);
            printf("  wave %2d:", w);
            for (int q = 0; q < 8; ++q) printf(" %8.1f", a[q]);
            printf("\n");
        }
        return 0;
    }

    if (getenv("ACC") && atoi(getenv("ACC"))) {
        float *o; CHK(hipMalloc(&o, telems * 4));
        unsigned long long *d_m; double *d_s;
        CHK(hipMalloc(&d_m, 32)); d_s = (double *)(d_m + 1);
        auto report = [&](const char *name) {
            CHK(hipMemset(d_m, 0, 32));
            k_acc<<<(unsigned)(tot * (T * T / 256)), 256>>>(tiles, o, d_flat, Kil, Gil, pair_stride, pstart, pcap, npairs, tm, d_m, d_s, d_m + 2, (double *)(d_m + 3));
            CHK(hipDeviceSynchronize());
            unsigned long long h[4]; CHK(hipMemcpy(h, d_m, 32, hipMemcpyDeviceToHost));
            double mx, sm, sg; memcpy(&mx, &h[0], 8); memcpy(&sm, &h[1], 8); memcpy(&sg, &h[3], 8);
            printf("accuracy %-30s pairs %d: err / sum|k g|  max %.3e (%.2f x 2^-24)  mean %.3e (%.3f x 2^-24)  signed mean %.4f x 2^-24\n", name, npairs, mx, mx * 16777216.0, sm / (double)h[2], sm / (double)h[2] * 16777216.0, sg / (double)h[2] * 16777216.0);
        };
        CHK(hipMemset(o, 0xee, telems * 4));
        k_ref<<<(unsigned)(tot * (T * T / 256)), 256>>>(tiles, o, d_flat, (int64_t)tot, Kil, Gil, pair_stride, pstart, pcap, npairs, tm);
        CHK(hipDeviceSynchronize()); report("fmaf chain (reference)");
        CHK(hipMemset(o, 0xee, telems * 4)); launch_old(o); CHK(hipDeviceSynchronize()); report("k_flush_mfma32");
        if (npairs > 32) { CHK(hipMemset(o, 0xee, telems * 4)); launch_split(o, true, 0); CHK(hipDeviceSynchronize()); report("k_flush_split3<2>"); }
        if (npairs > 32) { CHK(hipMemset(o, 0xee, telems * 4)); launch_split(o, true, 1); CHK(hipDeviceSynchronize()); report("k_flush_split3<2,0,4>"); }
        if (npairs <= 32) { CHK(hipMemset(o, 0xee, telems * 4)); launch_split(o, true, 2); CHK(hipDeviceSynchronize()); report("k_flush_split3<1>"); }
        if (npairs <= 32) { CHK(hipMemset(o, 0xee, telems * 4)); launch_split(o, true, 0); CHK(hipDeviceSynchronize()); report("k_flush_split3<2> (zero-padded)"); }
        CHK(hipFree(o));
    }
    if (getenv("STAMP") && atoi(getenv("STAMP")) == 4) {
        CHK(hipFuncSetAttribute((const void *)ekf_pipe32::k_flush_split3<2, 64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, ekf_pipe32::lds_bytes_split(4)));
        hipLaunchKernelGGL(ekf_pipe32::k_split_pairs, dim3((unsigned)(ldm / 256), ekf_pipe32::kKB, 2), dim3(256), 0, 0, (const float *)Kn, (const float *)Gpl, Kb3, Gb3,
                           pair_stride, ldm, ldm, pstart, pcap, npairs);
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL((ekf_pipe32::k_flush_split3<2, 64, 4>), dim3(2 * grid), dim3(256), ekf_pipe32::lds_bytes_split(4), 0, (const float *)tiles, tiles, d_segs, nsegs,
                               (const uint16_t *)Kb3, (const uint16_t *)Gb3, ldm, tm, dump);
        CHK(hipDeviceSynchronize());
        std::vector<unsigned long long> hs(32);
        double wv[4][6] = { { 0 } }, tot6[6] = { 0 };
        for (int b = 0; b < 2 * grid; ++b) {
            CHK(hipMemcpy(hs.data(), dump + (size_t)b * 128 * 256, 32 * 8, hipMemcpyDeviceToHost));
            for (int w = 0; w < 4; ++w)
                for (int q = 0; q < 6; ++q) { tot6[q] += (double)hs[w * 8 + q]; wv[w][q] += (double)hs[w * 8 + q]; }
        }
        printf("k_flush_split3<2,stamps,4> landmarks %lld pairs %d: ticks per (half-)item and wavefront: first chunk %.1f | WAIT %.1f | second chunk %.1f | epilogue %.1f | WAIT + first read %.1f   (sum %.1f)\n",
               (long long)N, npairs, tot6[0] / tot6[5], tot6[1] / tot6[5], tot6[2] / tot6[5], tot6[3] / tot6[5], tot6[4] / tot6[5], (tot6[0] + tot6[1] + tot6[2] + tot6[3] + tot6[4]) / tot6[5]);
        for (int w = 0; w < 4; ++w)
            printf("  wave %d: %8.1f %8.1f %8.1f %8.1f %8.1f\n", w, wv[w][0] / wv[w][5], wv[w][1] / wv[w][5], wv[w][2] / wv[w][5], wv[w][3] / wv[w][5], wv[w][4] / wv[w][5]);
        return 0;
    }
    if (getenv("STAMP") && atoi(getenv("STAMP")) == 3) {
        // where a wavefront of the split-arithmetic pass spends its cycles (k_flush_split3<2, 64>: stamps in the workgroup's dump area)
        CHK(hipFuncSetAttribute((const void *)ekf_pipe32::k_flush_split3<2, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, ekf_pipe32::lds_bytes_split()));
        hipLaunchKernelGGL(ekf_pipe32::k_split_pairs, dim3((unsigned)(ldm / 256), ekf_pipe32::kKB, 2), dim3(256), 0, 0, (const float *)Kn, (const float *)Gpl, Kb3, Gb3,
                           pair_stride, ldm, ldm, pstart, pcap, npairs);
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL((ekf_pipe32::k_flush_split3<2, 64>), dim3(grid), dim3(512), ekf_pipe32::lds_bytes_split(), 0, (const float *)tiles, tiles, d_segs, nsegs,
                               (const uint16_t *)Kb3, (const uint16_t *)Gb3, ldm, tm, dump);
        CHK(hipDeviceSynchronize());
        std::vector<unsigned long long> hs(64);
        double tot6[8] = { 0 };
        double wv[8][8] = { { 0 } };
        for (int b = 0; b < grid; ++b) {
            CHK(hipMemcpy(hs.data(), dump + (size_t)b * 128 * 256, 64 * 8, hipMemcpyDeviceToHost));
            for (int w = 0; w < 8; ++w)
                for (int q = 0; q < 8; ++q) { tot6[q] += (double)hs[w * 8 + q]; wv[w][q] += (double)hs[w * 8 + q]; }
        }
        const double items = tot6[5] / 8.0 / grid;
        printf("k_flush_split3<2,stamps> landmarks %lld pairs %d: %.0f items per workgroup; s_memtime ticks per item and wavefront:\n", (long long)N, npairs, items);
        printf("  first chunk %.1f | WAIT %.1f | second chunk %.1f | epilogue %.1f | WAIT + first read %.1f   (sum %.1f); inside the chunks: the two G blocks %.1f, the eight tile pieces %.1f\n", tot6[0] / tot6[5], tot6[1] / tot6[5], tot6[2] / tot6[5],
               tot6[3] / tot6[5], tot6[4] / tot6[5], (tot6[0] + tot6[1] + tot6[2] + tot6[3] + tot6[4]) / tot6[5], tot6[6] / tot6[5], tot6[7] / tot6[5]);
        for (int w = 0; w < 8; ++w)
            printf("  wave %d: %8.1f %8.1f %8.1f %8.1f %8.1f | %8.1f %8.1f\n", w, wv[w][0] / wv[w][5], wv[w][1] / wv[w][5], wv[w][2] / wv[w][5], wv[w][3] / wv[w][5], wv[w][4] / wv[w][5], wv[w][6] / wv[w][5], wv[w][7] / wv[w][5]);
        return 0;
    }
    int rc = 0;
    if (check) {
        CHK(hipMalloc(&out_a, telems * 4)); CHK(hipMalloc(&out_b, telems * 4));
        unsigned long long *d_bad;
        CHK(hipMalloc(&d_bad, 16));
        auto compare = [&](const char *name) {
            unsigned long long init[2] = { 0, ~0ull }, res[2];
            CHK(hipMemcpy(d_bad, init, 16, hipMemcpyHostToDevice));
            k_diff<<<2048, 256>>>(out_a, out_b, telems, d_bad, d_bad + 1);
            CHK(hipDeviceSynchronize());
            CHK(hipMemcpy(res, d_bad, 16, hipMemcpyDeviceToHost));
            printf("check %-28s: %llu of %lld entries differ from the fmaf reference%s\n", name, res[0], (long long)telems, res[0] ? "  <-- WRONG" : "");
            if (res[0]) { printf("   first at element %llu (tile slot %llu, row %llu, col %llu)\n", res[1], res[1] / (T * T), (res[1] / T) % T, res[1] % T); rc = 1; }
        };
        CHK(hipMemset(out_a, 0xff, telems * 4));
        k_ref<<<(unsigned)(tot * (T * T / 256)), 256>>>(tiles, out_a, d_flat, (int64_t)tot, Kil, Gil, pair_stride, pstart, pcap, npairs, tm);
        CHK(hipDeviceSynchronize());
        CHK(hipMemset(out_b, 0xee, telems * 4)); launch_old(out_b); CHK(hipDeviceSynchronize()); compare("k_flush_mfma32<256,4,2,3,e>");
        for (int v = 0; v < nvs; ++v)
            if (strip_ok(v)) { CHK(hipMemset(out_b, 0xee, telems * 4)); launch_strip(out_b, v); CHK(hipDeviceSynchronize()); compare(vs[v].name); }
        CHK(hipFree(out_a)); CHK(hipFree(out_b));
    }
    if (rounds > 0) {
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        std::vector<std::string> names = { "k_flush_mfma32<256,4,2,3,early>" };
        for (int v = 0; v < nvs; ++v) names.push_back(vs[v].name);
        names.push_back("k_split_pairs + k_flush_split3<2>");
        for (int v = 0; v < nvp; ++v) names.push_back(std::string(vp[v].name) + (v ? "" : " alone"));
        const int nk = (int)names.size();
        std::vector<std::vector<float>> ms(nk);
        for (int r = 0; r < rounds + 1; ++r)
            for (int k = 0; k < nk; ++k) {
                if (k >= 1 && k <= nvs && !strip_ok(k - 1)) continue;
                if (k > nvs + 1 && k - nvs - 2 == 2 && npairs > 32) continue;      // <1> holds 32 pairs
                const int gap_ms = getenv("GAP_MS") ? atoi(getenv("GAP_MS")) : 0;      // GAP_MS: an idle device between single timed launches (the engine's duty cycle)
                if (gap_ms > 0) {
                    for (int rep = 0; rep < 3; ++rep) {
                        usleep(1000 * gap_ms);
                        if (getenv("PREWARM_US")) hipLaunchKernelGGL(k_prewarm, dim3(grid * 2), dim3(256), 0, 0, dump, (long long)atoi(getenv("PREWARM_US")) * 2100);      // (readcyclecounter: shader cycles, ~2.1 GHz)
                        if (getenv("PREWARM_MB")) hipLaunchKernelGGL(k_prewarm_mem, dim3(2048), dim3(256), 0, 0, (const float4 *)tiles, (long long)atoi(getenv("PREWARM_MB")) * 65536ll, dump);
                        CHK(hipEventRecord(e0, 0));
                        if (k == 0) launch_old(tiles); else if (k <= nvs) launch_strip(tiles, k - 1); else launch_split(tiles, k == nvs + 1, k > nvs + 1 ? k - nvs - 2 : 0);
                        tm.reverse ^= (reverse == 2);
                        CHK(hipEventRecord(e1, 0));
                        CHK(hipEventSynchronize(e1));
                        float t; CHK(hipEventElapsedTime(&t, e0, e1));
                        if (r > 0) ms[k].push_back(t);
                    }
                    continue;
                }
                CHK(hipEventRecord(e0, 0));
                for (int rep = 0; rep < 3; ++rep) {
                    if (k == 0) launch_old(tiles); else if (k <= nvs) launch_strip(tiles, k - 1); else launch_split(tiles, k == nvs + 1, k > nvs + 1 ? k - nvs - 2 : 0);
                    tm.reverse ^= (reverse == 2);
                }
                CHK(hipEventRecord(e1, 0));
                CHK(hipEventSynchronize(e1));
                float t; CHK(hipEventElapsedTime(&t, e0, e1));
                if (r > 0) ms[k].push_back(t / 3);
            }
        const double n = 3.0 + (double)n_mm;
        for (int k = 0; k < nk; ++k) {
            if (ms[k].empty()) continue;
            std::sort(ms[k].begin(), ms[k].end());
            const double med = ms[k][ms[k].size() / 2], mn = ms[k][0];
            printf("%-34s landmarks %lld pairs %d: median %.4f ms  min %.4f ms   %.1f TFLOP/s  %.2f TB/s (B_alg)\n", names[k].c_str(), (long long)N,
                   npairs, med, mn, 2.0 * 2 * npairs * n * (n + 1) / 2 / (med * 1e-3) / 1e12, 4.0 * n * (n + 1) / (med * 1e-3) / 1e12);
        }
    }
    return rc;
}
