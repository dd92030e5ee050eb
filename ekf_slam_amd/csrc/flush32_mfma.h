// The pass over float tiles with the rank-2m product on the f32 matrix pipe, one work item per workgroup (3 to 28 pending pairs;
// from 29 on: flush32_pipe.h).  Included by kernels.hip (inside its anonymous namespace's scope of helpers) and by
// scripts/probes/flush32_bench.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

#ifndef EKF_PASS_COMMON
#define EKF_PASS_COMMON
constexpr int kBlock = 256;
__device__ __forceinline__ int ring_slot(int pstart, int i, int pcap) { const int s = pstart + i; return s >= pcap ? s - pcap : s; }
#endif

// "F32 mixed precision with F64 innovation solve" (BASELINE.json configs[4]; cfg.pass_arith = EKF_ARITH_F32): the same pass over float tiles
// with the rank-2m product ON THE F32 MATRIX PIPE -- v_mfma_f32_16x16x4_f32, measured at three times the f64 instruction's rate on this chip
// (scripts/probes/mfma_f32_rate.hip).  The operands are the float copies of the pending pairs that the gather writes beside the F64 ones
// (DevState::Gp32 / Kp32); the accumulators hold only the pass's update -sum_i K_i G_i, summed in float from zero, and the float tile value is
// added to it ONCE (see below): one rounding at the entry's magnitude per pass, as the F64-arithmetic pass has.  Everything that DECIDES
// anything -- innovation, S, its inverse, K, the state, the robot block, the strip, the landmarks' diagonal blocks (DevState::diag) -- stays in
// F64 in the gather kernel (tolerance: DESIGN.md 5).
// Geometry: a workgroup = 64 kRG rows x 128 columns, a wavefront kRG x 16 rows x 128 columns = 8 kRG accumulator blocks; the f32 instruction's
// result layout differs from the f64 one's: lane (lr, lc) register r is row 4 lr + r (f64: lr + 4 r) of column block lc.
typedef float f4_t __attribute__((ext_vector_type(4)));

template <int T, int kChunk, int kRG, int kWpe, bool kEarly = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kWpe, kWpe)))
void k_flush_mfma32(const float *__restrict__ tiles, float *__restrict__ dst, const int2 *__restrict__ work, int64_t nwork,
                    const float *__restrict__ Kp, const float *__restrict__ Gp, int64_t pair_stride, int pstart, int pcap,
                    int npairs, TileMap tm) {
    // kRG: 16-row groups per wavefront -- a workgroup owns 64 kRG rows x 128 columns (kRG = 2: twice the bytes in flight per workgroup and
    // one read of G from LDS for 16 instead of 8 MFMAs).  Production: kChunk = 4, kRG = 2, four wavefronts per SIMD (profiles/round3_tuning.md 36).
    constexpr int kRows = 64 * kRG, kCols = 128, kKPad = kRows + 16;
    constexpr int kColParts = T / kCols, kSubsPerTile = (T / kRows) * kColParts;
    static_assert(T % kCols == 0 && T % kRows == 0, "a wavefront owns 16 kRG rows x 128 columns");
    static_assert(kChunk % 2 == 0 && (kChunk * kCols) % kBlock == 0 && (kChunk * kRows) % kBlock == 0, "bad chunk");
    __shared__ __attribute__((aligned(16))) float Gs[2 * kChunk][kCols];
    __shared__ float Ks[2 * kChunk][kKPad];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane >> 4, lc = lane & 15;
    const int64_t ldm = pair_stride >> 1;                             // a pair slot = two planes (x, y) of ldm floats
    const int64_t nitems = 8 * nwork * kSubsPerTile;
    for (int64_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int64_t vi = tm.reverse ? nwork * kSubsPerTile - 1 - (it >> 3) : (it >> 3);
        const int64_t w = vi / kSubsPerTile;
        const int sub = (int)(vi - w * kSubsPerTile);
        const int2 ij = work[(it & 7) * nwork + w];
        if (ij.x < 0) continue;
        const int slab = sub / kColParts, cpart = sub - slab * kColParts;
        const int row0 = slab * kRows + wave * 16;                    // row group rg: + 64 rg
        const int64_t toff = tm.tile_offset(ij.x, ij.y) + (int64_t)(row0 + 4 * lr) * T + cpart * kCols + 4 * lc;
        const float *__restrict__ tp = tiles + toff;
        float *__restrict__ td = dst + toff;
        // The accumulators start at ZERO and hold only the pass's update -sum_i K_i G_i; the tile value is added ONCE at the end (one float rounding
        // per entry and pass, as the F64-arithmetic pass has).  Accumulating onto the tile value itself (the first version: tile loaded into the
        // accumulators) rounds at the ENTRY's ulp after every rank-4 step: on the large entries (cross-covariances of appended landmarks, ~10) every
        // small decrement was lost -- 7e-7 on sampled blocks, 4e-6 on the digests after configs[4]'s 10 000 update-steps (tests/test_full_size_gpu.py).
        // The tile is requested after the last chunk, a row group's eight 16-byte pieces at a time (holding it in registers from the start spills
        // at four wavefronts per SIMD; an early "touch" load + the late one moved the bytes twice: profiles/round3_tuning.md 36).
        f4_t acc[kRG][2][4];                                          // [row group][16-byte group bp][column e in it][row r -> row0 + 64 rg + 4 lr + r]
#pragma unroll
        for (int rg = 0; rg < kRG; ++rg)
#pragma unroll
            for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[rg][bp][e] = f4_t{ 0.0f, 0.0f, 0.0f, 0.0f };
        const int64_t gcol0 = (int64_t)ij.y * T + cpart * kCols;
        const int64_t krow0 = (int64_t)ij.x * T + slab * kRows;
        constexpr int kPerG = kChunk * kCols / kBlock, kPerK = kChunk * kRows / kBlock;
        float2 tg[kPerG], tk[kPerK];                                  // (the pairs' float copies, written by the gather: DevState::Gp32 / Kp32)
        auto fetch = [&](int c0, int cn) {
#pragma unroll
            for (int q = 0; q < kPerG; ++q) {
                const int e = tid + q * kBlock, col = e & (kCols - 1);
                const int i = (e >> 7) < cn ? (e >> 7) : cn - 1;
                const float *gs = Gp + (int64_t)ring_slot(pstart, c0 + i, pcap) * pair_stride + gcol0 + col;      // planar copies: plane x, then plane y
                tg[q] = make_float2(gs[0], gs[ldm]);
            }
#pragma unroll
            for (int q = 0; q < kPerK; ++q) {
                const int e = tid + q * kBlock, row = e & (kRows - 1);
                const int i = (e / kRows) < cn ? (e / kRows) : cn - 1;
                const float *ks = Kp + (int64_t)ring_slot(pstart, c0 + i, pcap) * pair_stride + krow0 + row;      // (stored negated by the gather)
                tk[q] = make_float2(ks[0], ks[ldm]);
            }
        };
        auto stage = [&](int cn) {                           // the fetched chunk de-interleaved to [k][col] / [k][row]
#pragma unroll
            for (int q = 0; q < kPerG; ++q) {
                const int e = tid + q * kBlock, i = e >> 7, col = e & (kCols - 1);
                if (i < cn) { Gs[2 * i][col] = tg[q].x; Gs[2 * i + 1][col] = tg[q].y; }
                else if (i == cn) { Gs[2 * i][col] = 0.0f; Gs[2 * i + 1][col] = 0.0f; }       // pad of an odd count
            }
#pragma unroll
            for (int q = 0; q < kPerK; ++q) {
                const int e = tid + q * kBlock, i = e / kRows, row = e & (kRows - 1);
                if (i < cn) { Ks[2 * i][row] = tk[q].x; Ks[2 * i + 1][row] = tk[q].y; }
                else if (i == cn) { Ks[2 * i][row] = -0.0f; Ks[2 * i + 1][row] = -0.0f; }
            }
        };
        auto apply = [&](int cn) {
            const int ksteps = (cn + 1) >> 1;                         // two pairs = four rank-1 terms per MFMA
#pragma unroll 2
            for (int ks = 0; ks < ksteps; ++ks) {
                float a[kRG];
#pragma unroll
                for (int rg = 0; rg < kRG; ++rg) a[rg] = Ks[4 * ks + lr][64 * rg + wave * 16 + lc];
                f4_t b[2];
#pragma unroll
                for (int bp = 0; bp < 2; ++bp) b[bp] = *reinterpret_cast<const f4_t *>(&Gs[4 * ks + lr][64 * bp + 4 * lc]);
#pragma unroll
                for (int rg = 0; rg < kRG; ++rg)
#pragma unroll
                    for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[rg][bp][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rg], b[bp][e], acc[rg][bp][e], 0, 0, 0);
            }
        };
        auto count = [&](int c0) { return npairs - c0 < kChunk ? npairs - c0 : kChunk; };
        fetch(0, count(0));
        int c0 = 0;
        for (; c0 + kChunk < npairs; c0 += kChunk) {
            __syncthreads();                                          // everyone is done with the previous chunk
            stage(count(c0));
            __syncthreads();
            fetch(c0 + kChunk, count(c0 + kChunk));
            apply(count(c0));
        }
        __syncthreads();                                              // the last chunk, peeled: nothing is fetched behind it ...
        stage(count(c0));
        __syncthreads();
        f4_t te[kEarly ? kRG : 1][2][4];                              // ... kEarly: the tile is requested HERE, in front of the last chunk's matrix work
        if constexpr (kEarly) {
#pragma unroll
            for (int rg = 0; rg < kRG; ++rg)
#pragma unroll
                for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        te[rg][bp][r] = __builtin_nontemporal_load(reinterpret_cast<const f4_t *>(tp + (int64_t)(64 * rg + r) * T + 64 * bp));
        }
        apply(count(c0));
#pragma unroll
        for (int rg = 0; rg < kRG; ++rg) {
            f4_t tl[2][4];
#pragma unroll
            for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (kEarly) tl[bp][r] = te[rg][bp][r];
                    else tl[bp][r] = __builtin_nontemporal_load(reinterpret_cast<const f4_t *>(tp + (int64_t)(64 * rg + r) * T + 64 * bp));
                }
#pragma unroll
            for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f4_t o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[rg][bp][e][r];
                    o += tl[bp][r];
                    __builtin_nontemporal_store(o, reinterpret_cast<f4_t *>(td + (int64_t)(64 * rg + r) * T + 64 * bp));
                }
        }
    }
}

