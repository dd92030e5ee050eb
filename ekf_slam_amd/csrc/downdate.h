// Fragment of kernels.hip (included there, inside its anonymous namespace, after kernels.h / device_math.h): the one-pair pass over P: k_downdate, k_downdate_w (EKF_SLAM.m:145).
#pragma once

// ---------------------------------------------------------------------------------------------------
// rank-2 downdate of the landmark block:  tile(I,J)[r][c] -= K(I*T+r,:) . G(:,J*T+c)
//
// The HBM-bound kernel.  Every unique entry of P is read once and written once (w*n*(n+1) bytes per
// launch); K and G (2 x n each) stay in L2.  A work item is a (tile, slab) pair: kSlab rows of one owned
// tile.  Inside it each lane owns one 16-byte column pair, so every wavefront load/store instruction moves
// 1 KiB of contiguous tile memory; the lane's four G values and the row's two K values come from L2.
// ---------------------------------------------------------------------------------------------------
template <typename TS, int T, int kSlab>
__global__ __launch_bounds__(kBlock) void k_downdate(const TS *__restrict__ tiles, TS *__restrict__ dst,
                                                     const int2 *__restrict__ work, int64_t nwork,
                                                     const double *__restrict__ Kp, const double *__restrict__ Gp,
                                                     int64_t pair_stride, int pstart, int pcap, int npairs, TileMap tm) {
    // kSlab = rows of a tile one workgroup handles (T = whole tile); a work item is (tile, slab).
    // npairs pending (K_i, G_i) pairs are applied, in slot order, to registers between ONE load and ONE
    // store of every element: one pass over P for npairs update-steps.
    using V2 = typename Vec2<TS>::type;
    constexpr int kPairsPerRow = T / 2;
    constexpr int kRowsPerPass = kBlock / kPairsPerRow;
    constexpr int kSlabsPerTile = T / kSlab;
    constexpr int kPasses = (kSlab + kRowsPerPass - 1) / kRowsPerPass;
    constexpr bool kExact = kRowsPerPass * kPasses == kSlab;
    static_assert(kPasses <= 8, "slab too tall for the register tile");
    const int tid = threadIdx.x;
    const int cp = tid % kPairsPerRow;       // column pair inside the tile
    const int r0 = tid / kPairsPerRow;       // first row of this lane inside the slab
    const int64_t nitems = nwork * kSlabsPerTile;
    for (int64_t it0 = blockIdx.x; it0 < nitems; it0 += gridDim.x) {
        const int64_t it = tm.reverse ? nitems - 1 - it0 : it0;       // alternate passes walk backwards (see k_downdate_w)
        const int64_t w = it / kSlabsPerTile;
        const int slab = (int)(it - w * kSlabsPerTile);
        const int2 ij = work[w];
        const int64_t toff = tm.tile_offset(ij.x, ij.y) + (int64_t)slab * kSlab * T;
        const TS *__restrict__ tp = tiles + toff;
        TS *__restrict__ td = dst + toff;
        double2 v[kPasses];
#pragma unroll
        for (int p = 0; p < kPasses; ++p) {
            const int r = r0 + p * kRowsPerPass;
            if (kExact || r < kSlab) {
                const V2 t = *reinterpret_cast<const V2 *>(tp + r * T + 2 * cp);
                v[p] = make_double2((double)t.x, (double)t.y);
            }
        }
        const int64_t gcol = (int64_t)ij.y * T + 2 * cp;
        const int64_t krow = (int64_t)ij.x * T + slab * kSlab;
        for (int i = 0; i < npairs; ++i) {
            const int64_t so = (int64_t)ring_slot(pstart, i, pcap) * pair_stride;
            const double2 *__restrict__ g2 = reinterpret_cast<const double2 *>(Gp + so) + gcol;
            const double2 *__restrict__ k2 = reinterpret_cast<const double2 *>(Kp + so) + krow;
            const double2 ga = g2[0], gb = g2[1];                  // (G1,G2) at columns 2cp and 2cp+1
#pragma unroll
            for (int p = 0; p < kPasses; ++p) {
                const int r = r0 + p * kRowsPerPass;
                if (kExact || r < kSlab) {
                    const double2 k = k2[r];
                    v[p].x = rank2_apply(v[p].x, k, ga);
                    v[p].y = rank2_apply(v[p].y, k, gb);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < kPasses; ++p) {
            const int r = r0 + p * kRowsPerPass;
            if (kExact || r < kSlab) {
                V2 o;
                o.x = (TS)v[p].x; o.y = (TS)v[p].y;
                *reinterpret_cast<V2 *>(td + r * T + 2 * cp) = o;
            }
        }
    }
}

// Wave-row variant for T = 64 / 128 (the production tile sizes).  A wavefront owns kSlab/4 CONSECUTIVE rows
// of the slab, so the K values it needs for one pair are one contiguous, wave-uniform run: they are fetched
// with scalar loads (no vector-memory or LDS traffic) and feed v_fma_f64 as SGPR operands.  Per pending pair a
// lane issues two 16-byte G loads (L1/L2 hits) and 4 FMAs per row pass; the tile data are loaded once and
// stored once whatever the number of pairs.  With one pair and kSlab = rows of one pass this is the plain
// streaming kernel; with m pairs and a taller slab it is one pass over P for m update-steps.
template <typename TS> struct Lane16;                       // 16 bytes of one tile row per lane
template <> struct Lane16<double> { using type = double2; static constexpr int kCols = 2; };
template <> struct Lane16<float>  { using type = float4;  static constexpr int kCols = 4; };
__device__ __forceinline__ void lane16_unpack(const double2 &t, double *v) { v[0] = t.x; v[1] = t.y; }
__device__ __forceinline__ void lane16_unpack(const float4 &t, double *v) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
__device__ __forceinline__ void lane16_pack(const double *v, double2 &t) { t.x = v[0]; t.y = v[1]; }
__device__ __forceinline__ void lane16_pack(const double *v, float4 &t) {
    t.x = (float)v[0]; t.y = (float)v[1]; t.z = (float)v[2]; t.w = (float)v[3];
}
__device__ __forceinline__ double lane16_get(const double2 &t, int q) { return q == 0 ? t.x : t.y; }
__device__ __forceinline__ double lane16_get(const float4 &t, int q) { return (double)(q == 0 ? t.x : q == 1 ? t.y : q == 2 ? t.z : t.w); }

// Out of line on purpose: the pass is HBM-bound to the last per cent, and with this code inlined its main body was scheduled 8 %
// slower (551 vs 510 us at 10 k landmarks) although only two tile lines in ~150 ever come here.
// M = P(j:j+1, :) as k_rowpanel lays it out: chunk k of T columns comes from tile (I_j, k) for k <= I_j (rows j, j+1 of the tile:
// M(a, c) = P(j+a, c)) and from tile (k, I_j) beyond (columns j, j+1 of the tile: M(a, c) = P(c, j+a)); on the diagonal tile the
// lower triangle is canonical: M(1, j+1) = P(j+1, j).  Local chunk kl = (k - k0) / world.
__device__ __attribute__((noinline)) void extract_next_row(double *__restrict__ send, int jm, int Ij, int tI, int tJ, int r, int c0, int ncols,
                                                           int T, int world, int rank, double v0, double v1, double v2, double v3) {
    const uint32_t wd = (uint32_t)world;
    const int k0 = (int)(((uint32_t)rank + wd - (uint32_t)Ij % wd) % wd);
    const bool rowtile = tI == Ij, coltile = tJ == Ij, diag = rowtile && coltile;
    for (int q = 0; q < ncols; ++q) {
        const int cc = c0 + q;                                      // tile column
        const double val = q == 0 ? v0 : q == 1 ? v1 : q == 2 ? v2 : v3;
        if (rowtile && (r == jm || r == jm + 1)) {
            const int a = r - jm;
            if (!diag || cc <= jm) {                                // c <= j: M(a, c) = P(j + a, c)
                const int64_t e = (int64_t)((tJ - k0) / world) * T + cc;
                send[2 * e + a] = val;
            }
            if (diag && a == 1 && (cc == jm || cc == jm + 1)) {     // canonical (j+1, j) = M(1, j+1); (j+1, j+1) = M(2, j+1)
                const int64_t e = (int64_t)((Ij - k0) / world) * T + jm + 1;
                send[2 * e + (cc - jm)] = val;
            }
        }
        if (coltile && (cc == jm || cc == jm + 1) && (tI > Ij || r > jm + 1)) {     // c = I T + r >= j + 2: M(a, c) = P(c, j + a)
            const int64_t e = (int64_t)((tI - k0) / world) * T + r;
            send[2 * e + (cc - jm)] = val;
        }
    }
}

// kNext (sharded handles, one pair per launch, the NEXT correction's landmark announced: ekf_hint_next): the pass also EXTRACTS the
// row-panel P(j:j+1, :) of that landmark into the exchange slab while the updated entries are in registers -- the workgroups
// that own rows j, j+1 of tile row I_j write the row part, those of tile column I_j the column part; what they write is what
// k_rowpanel would read back from the tiles a launch later (canonical lower-triangle entries, after the rounding to TS).  The
// next update-step then starts with its all-gather: one launch (~5 us of a shard's fixed cost) less.
struct NoNextRow {};
template <bool kNext> struct NextRowParam { using type = NoNextRow; };
template <> struct NextRowParam<true> { using type = NextRow; };

template <typename TS, int T, int kSlab, bool kXcd, bool kNext = false>
__global__ __launch_bounds__(kBlock) void k_downdate_w(const TS *__restrict__ tiles, TS *__restrict__ dst,
                                                       const int2 *__restrict__ work, int64_t nwork,
                                                       const double *__restrict__ Kp, const double *__restrict__ Gp,
                                                       int64_t pair_stride, int pstart, int pcap, int npairs, TileMap tm,
                                                       typename NextRowParam<kNext>::type nx) {
    using VL = typename Lane16<TS>::type;
    constexpr int kCols = Lane16<TS>::kCols;              // columns per lane: 2 (f64 tiles) or 4 (f32 tiles)
    constexpr int kLanesPerRow = T / kCols;               // 64: one row per wave instruction; 32: two rows
    constexpr int kRowsPerInstr = 64 / kLanesPerRow;
    constexpr int kRowsPerWave = kSlab / 4;               // consecutive rows owned by a wavefront
    constexpr int kPasses = kRowsPerWave / kRowsPerInstr;
    constexpr int kSlabsPerTile = T / kSlab;
    static_assert(kLanesPerRow == 64 || kLanesPerRow == 32, "tile edge / storage type combination not supported");
    static_assert(kPasses >= 1 && kPasses <= 8 && kPasses * kRowsPerInstr * 4 == kSlab, "bad slab");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane / kLanesPerRow;                  // which of the kRowsPerInstr rows this lane is on
    const int cl = lane % kLanesPerRow;                   // 16-byte column group inside the tile row
    // kXcd: `work` holds 8 streams of `nwork` tiles each (padded with (-1,-1)); workgroups b and b+8 run on the same
    // XCD (round-robin dispatch -- a speed assumption only), so workgroup b walks stream b % 8 and the XCD's
    // resident workgroups stay inside one or two super-tiles whose K/G slices fit its L2.
    // tm.reverse: every other pass walks the work list backwards, so that the tiles one pass wrote LAST are the ones the next
    // pass reads FIRST -- while they are still in the 256 MiB Infinity Cache (it keeps a line while the bytes touched between
    // two uses of it fit; a store larger than the cache that is always walked in the same direction never meets that).
    // Results do not depend on the order: every element is updated independently.
    const int64_t nitems = (kXcd ? 8 : 1) * nwork * kSlabsPerTile;
    const int64_t nv = nwork * kSlabsPerTile;
    for (int64_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int64_t vf = kXcd ? (it >> 3) : it;
        const int64_t vi = tm.reverse ? nv - 1 - vf : vf;
        const int64_t w = vi / kSlabsPerTile;
        const int slab = (int)(vi - w * kSlabsPerTile);
        const int2 ij = work[kXcd ? (it & 7) * nwork + w : w];
        if (kXcd && ij.x < 0) continue;
        const int row0 = slab * kSlab + wave * kRowsPerWave;            // first tile row of this wavefront
        const int64_t toff = tm.tile_offset(ij.x, ij.y) + (int64_t)(row0 + sub) * T + kCols * cl;
        const TS *__restrict__ tp = tiles + toff;
        TS *__restrict__ td = dst + toff;
        double v[kPasses][kCols];
#pragma unroll
        for (int p = 0; p < kPasses; ++p) {
            const VL t = *reinterpret_cast<const VL *>(tp + (int64_t)p * kRowsPerInstr * T);
            lane16_unpack(t, v[p]);
        }
        const int64_t gcol = (int64_t)ij.y * T + kCols * cl;
        const int64_t krow = (int64_t)ij.x * T + row0;                  // wave-uniform
        for (int i = 0; i < npairs; ++i) {
            const int64_t so = (int64_t)ring_slot(pstart, i, pcap) * pair_stride;
            const double2 *__restrict__ g2 = reinterpret_cast<const double2 *>(Gp + so) + gcol;
            const double2 *__restrict__ k2 = reinterpret_cast<const double2 *>(Kp + so) + krow;
            double2 g[kCols];                                           // (G1,G2) at this lane's columns
#pragma unroll
            for (int q = 0; q < kCols; ++q) g[q] = g2[q];
#pragma unroll
            for (int p = 0; p < kPasses; ++p) {
                double2 k = k2[p * kRowsPerInstr];                      // uniform address: scalar load when one row per instruction
                if (kRowsPerInstr == 2) { const double2 k1 = k2[p * 2 + 1]; if (sub) k = k1; }
#pragma unroll
                for (int q = 0; q < kCols; ++q) v[p][q] = rank2_apply(v[p][q], k, g[q]);
            }
        }
        VL stored[kPasses];
#pragma unroll
        for (int p = 0; p < kPasses; ++p) {
            lane16_pack(v[p], stored[p]);
            *reinterpret_cast<VL *>(td + (int64_t)p * kRowsPerInstr * T) = stored[p];
        }
        if constexpr (kNext) {
            const int Ij = (int)(nx.j >> tm.shift);
            if (__builtin_expect(ij.x == Ij || ij.y == Ij, 0)) {        // uniform per workgroup; two tile lines out of nt
#pragma unroll
                for (int p = 0; p < kPasses; ++p) {
                    double vv[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                    for (int q = 0; q < kCols; ++q) vv[q] = lane16_get(stored[p], q);      // what the tile now holds
                    extract_next_row(nx.send, (int)(nx.j & (T - 1)), Ij, ij.x, ij.y, row0 + sub + p * kRowsPerInstr, kCols * cl, kCols, T,
                                     tm.world, tm.rank, vv[0], vv[1], vv[2], vv[3]);
                }
            }
        }
    }
}
