// Fragment of kernels.hip (included there, at file scope): the batched pass on the f64 matrix cores, k_flush_mfma (EKF_SLAM.m:145, m corrections at once).
#pragma once

// Batched flush on the matrix cores (f64 tiles, T = 128) -- the production flush for two or more pending pairs.
// P_tile -= K_rows (64 x 2m) * G_cols (2m x 128) is a rank-2m update; v_mfma_f64_16x16x4_f64 applies four of its 2m
// rank-1 terms per instruction.  The instruction is a k-ordered chain of correctly rounded FMAs,
//     D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C))))
// (scripts/probes/mfma_f64_order.{hip,py}: 4096/4096 elements bit-equal to that chain and to no other order), so with
// A = -K (negation is exact) and the k index running (pair 0: x, y), (pair 1: x, y), ... in ring order the result is
// bit-identical to rank2_apply() applied pair after pair -- i.e. to the immediate (batch = 1) downdate.
// Mapping: a workgroup owns 64 rows x 128 columns of a tile, a wavefront 16 rows x 128 columns = 8 accumulator blocks
// (32 f64 per lane).  The MFMA "column" lane&15 of block (bp, e) is the PHYSICAL column 32*bp + 2*(lane&15) + e, so
// every lane still loads / stores 16 contiguous bytes of a tile row.  -K and G of a chunk of kChunk pairs are staged
// through LDS once per workgroup, de-interleaved to [k][row] / [k][col]; the per-k-step cost is one ds_read_b64 (A)
// and four ds_read_b128 (B) per 8 MFMAs.  An odd pair count is padded with A = -0.0, B = +0.0 (x + (-0) == x
// for every x, signed zeros included).
typedef double d4_t __attribute__((ext_vector_type(4)));

// Work items: 64 rows x kCols columns -- kCols = 64 (4 accumulator blocks per wavefront, five wavefronts per SIMD) up to 12 pairs, 128 beyond
// (launch_flush_mfma).  Two chunk sizes for the 128-column items: chunks of 4 pairs fit 4 wavefronts per SIMD (122 VGPRs) and win up to ~30 pairs (28 pairs: 0.603 vs 0.634 ms; 32: 0.651 vs 0.627), where the pass is
// HBM-bound and occupancy hides the tile latency; chunks of 8 pairs (3 wavefronts per SIMD, half the barriers) win beyond,
// where the f64 MFMA rate (measured 44-48 TFLOP/s, scripts/probes/mfma_f64_rate.hip) is the limit.
// Storage: f64 tiles with T = 128 (a work item = 64 rows x the 128 columns of a tile) and f32 tiles with T = 256 (a work item
// = 64 rows x one 128-column half; a lane's 16 bytes are 4 columns, widened to f64 on load and rounded once on store).
// kWaves: wavefronts per workgroup = 16-row groups per work item (4: the production shape; 8: 128-row items, G staged once per 128 rows -- measured,
// round4_tuning.md 58); kAbl (tuning builds only): 1 = no matrix work, 2 = no operand staging either (the item shape as a plain copy).
template <typename TS, int T, int kChunk, int kCols = 128, int kWpe = (kChunk <= 4 ? 4 : 3), int kWaves = 4, int kAbl = 0>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(kWpe, kWpe)))
void k_flush_mfma(const TS *__restrict__ tiles, TS *__restrict__ dst, const int2 *__restrict__ work, int64_t nwork,
                  const double *__restrict__ Kp, const double *__restrict__ Gp, int64_t pair_stride, int pstart, int pcap,
                  int npairs, TileMap tm) {
    constexpr int kBlock = 64 * kWaves;                               // (shadows the file-scope constant: this kernel's own workgroup size)
    constexpr int kRows = 16 * kWaves, kKPad = kRows + 16;
    constexpr int kE = 16 / (int)sizeof(TS);                          // columns in a lane's 16 bytes: 2 (f64) or 4 (f32)
    constexpr int kBP = kCols / (16 * kE);                            // 16-byte column groups per lane and row: 4 or 2
    constexpr int kColParts = T / kCols, kSubsPerTile = (T / kRows) * kColParts;
    static_assert((kBP * kE == 8 || kBP * kE == 4) && T % kCols == 0 && T % kRows == 0, "a wavefront owns 16 rows x 128 (64) columns = 8 (4) MFMA blocks");
    static_assert(kChunk % 2 == 0 && (kChunk * kCols) % kBlock == 0 && (kChunk * kRows) % kBlock == 0, "bad chunk");
    __shared__ __attribute__((aligned(16))) double Gs[2 * kChunk][kCols];
    __shared__ double Ks[2 * kChunk][kKPad];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane >> 4, lc = lane & 15;                        // MFMA k / row-group index, MFMA row / column index
    const int64_t nitems = 8 * nwork * kSubsPerTile;                  // 8 per-XCD streams (see k_downdate_w)
    for (int64_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int64_t vi = tm.reverse ? nwork * kSubsPerTile - 1 - (it >> 3) : (it >> 3);
        const int64_t w = vi / kSubsPerTile;
        const int sub = (int)(vi - w * kSubsPerTile);
        const int2 ij = work[(it & 7) * nwork + w];
        if (ij.x < 0) continue;                                       // padding of a shorter stream (uniform per workgroup)
        const int slab = sub / kColParts, cpart = sub - slab * kColParts;
        const int row0 = slab * kRows + wave * 16;
        const int64_t toff = tm.tile_offset(ij.x, ij.y) + (int64_t)(row0 + lr) * T + cpart * kCols + kE * lc;
        const TS *__restrict__ tp = tiles + toff;
        TS *__restrict__ td = dst + toff;
        d4_t acc[kBP][kE];                                            // [16-byte group bp][column e in it][row r -> row0 + lr + 4r]
#pragma unroll
        for (int bp = 0; bp < kBP; ++bp)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int e = 0; e < kE; ++e)                          // adjacent scalars: one 16-byte nontemporal load
                    acc[bp][e][r] = (double)__builtin_nontemporal_load(tp + (int64_t)(4 * r) * T + 16 * kE * bp + e);
        const int64_t gcol0 = (int64_t)ij.y * T + cpart * kCols;
        const int64_t krow0 = (int64_t)ij.x * T + slab * kRows;
        constexpr int kPerG = kChunk * kCols / kBlock, kPerK = kChunk * kRows / kBlock;
        double2 tg[kPerG], tk[kPerK];                                 // the NEXT chunk's operands, in flight while this one is applied
        auto fetch = [&](int c0, int cn) {
#pragma unroll
            for (int q = 0; q < kPerG; ++q) {
                const int e = tid + q * kBlock, col = e & (kCols - 1);
                const int i = (e / kCols) < cn ? (e / kCols) : cn - 1;      // clamp: always a valid pair, used only if in range
                tg[q] = reinterpret_cast<const double2 *>(Gp + (int64_t)ring_slot(pstart, c0 + i, pcap) * pair_stride)[gcol0 + col];
            }
#pragma unroll
            for (int q = 0; q < kPerK; ++q) {
                const int e = tid + q * kBlock, row = e & (kRows - 1);
                const int i = (e / kRows) < cn ? (e / kRows) : cn - 1;
                tk[q] = reinterpret_cast<const double2 *>(Kp + (int64_t)ring_slot(pstart, c0 + i, pcap) * pair_stride)[krow0 + row];
            }
        };
        if constexpr (kAbl < 2) fetch(0, npairs < kChunk ? npairs : kChunk);
        for (int c0 = 0; kAbl < 2 && c0 < npairs; c0 += kChunk) {
            const int cn = npairs - c0 < kChunk ? npairs - c0 : kChunk;
            __syncthreads();                                          // everyone is done with the previous chunk
#pragma unroll
            for (int q = 0; q < kPerG; ++q) {
                const int e = tid + q * kBlock, i = e / kCols, col = e & (kCols - 1);
                if (i < cn) { Gs[2 * i][col] = tg[q].x; Gs[2 * i + 1][col] = tg[q].y; }
                else if (i == cn) { Gs[2 * i][col] = 0.0; Gs[2 * i + 1][col] = 0.0; }       // pad of an odd count
            }
#pragma unroll
            for (int q = 0; q < kPerK; ++q) {
                const int e = tid + q * kBlock, i = e / kRows, row = e & (kRows - 1);
                if (i < cn) { Ks[2 * i][row] = -tk[q].x; Ks[2 * i + 1][row] = -tk[q].y; }
                else if (i == cn) { Ks[2 * i][row] = -0.0; Ks[2 * i + 1][row] = -0.0; }
            }
            __syncthreads();
            if (c0 + kChunk < npairs) fetch(c0 + kChunk, npairs - c0 - kChunk < kChunk ? npairs - c0 - kChunk : kChunk);
            const int ksteps = kAbl ? 0 : (cn + 1) >> 1;              // two pairs = four rank-1 terms per MFMA
#pragma unroll 2
            for (int ks = 0; ks < ksteps; ++ks) {
                const double a = Ks[4 * ks + lr][wave * 16 + lc];
                double2 b[kBP][kE / 2];
#pragma unroll
                for (int bp = 0; bp < kBP; ++bp)
#pragma unroll
                    for (int h = 0; h < kE / 2; ++h)
                        b[bp][h] = *reinterpret_cast<const double2 *>(&Gs[4 * ks + lr][16 * kE * bp + kE * lc + 2 * h]);
#pragma unroll
                for (int bp = 0; bp < kBP; ++bp)
#pragma unroll
                    for (int h = 0; h < kE / 2; ++h) {
                        acc[bp][2 * h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[bp][h].x, acc[bp][2 * h], 0, 0, 0);
                        acc[bp][2 * h + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[bp][h].y, acc[bp][2 * h + 1], 0, 0, 0);
                    }
            }
        }
        typedef TS store16_t __attribute__((ext_vector_type(kE)));                  // the lane's 16 bytes of a row: ONE store instruction
#pragma unroll
        for (int bp = 0; bp < kBP; ++bp)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                store16_t o;
#pragma unroll
                for (int e = 0; e < kE; ++e) o[e] = (TS)acc[bp][e][r];
                __builtin_nontemporal_store(o, reinterpret_cast<store16_t *>(td + (int64_t)(4 * r) * T + 16 * kE * bp));
            }
    }
}
