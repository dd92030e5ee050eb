// HBM layout of the covariance P (n = 3 + 2N) -- shared by host and device code.
//
//   P = [ Prr  Prm ]   Prr : 3 x 3 robot block              9 doubles, row-major
//       [ Pmr  Pmm ]   Prm : 3 x 2N robot/landmark strip    3 rows of ldm doubles ("3 x 2 cross blocks")
//                      Pmm : 2N x 2N landmark block, symmetric: only the lower block triangle is stored,
//                            cut into T x T tiles (T even, so a landmark's 2 x 2 block never straddles a
//                            tile edge).  Tile (I,J), I >= J, is T*T contiguous elements, row-major.
//                            Diagonal tiles are stored (and updated) whole.
//
// Tile order is tile-row major: row I holds tiles J = 0..I, so growing the map appends tile rows at the
// end and never moves data (streaming append).  With the matrix split over `world` shards, tile (I,J)
// belongs to shard (I + J) mod world: within a tile row the column panels are dealt out cyclically,
// starting one shard later on each row.  Every shard then holds the same share of every tile row and
// of every tile column for any N (no re-balancing as the map grows), and the T-wide chunk k of a
// landmark row-panel P(j:j+1, :) always comes from shard (tile_row(j) + k) mod world -- a regular cyclic
// pattern a single equal-count all-gather can carry.  Local slot of an owned tile: row_base(I) + J / world.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define EKF_HD __host__ __device__ __forceinline__
#else
#define EKF_HD inline
#endif

struct TileMap {
    int32_t T;       // tile edge (elements), a power of two
    int32_t shift;   // log2(T)
    int32_t world;   // shards
    int32_t rank;    // this shard
    int32_t reverse = 0;   // pass kernels only: walk the work list backwards (set on alternate passes, abi.hip::next_pass_direction)

    // Tile indices are small non-negative numbers: all index arithmetic is 32-bit unsigned (a 64-bit integer division is
    // a ~100-instruction sequence on the GPU and these run in the latency-critical prologue of every gather), and a single
    // shard takes the division-free path.
    // number of local slots in tile rows [0, I)
    EKF_HD int64_t row_base(int64_t I) const {
        if (world == 1) return (I * (I + 1)) >> 1;
        const uint32_t a = (uint32_t)I / (uint32_t)world, b = (uint32_t)I - a * (uint32_t)world;
        return (((int64_t)world * a * (a + 1)) >> 1) + (int64_t)b * (a + 1);
    }
    EKF_HD int32_t owner(int64_t I, int64_t J) const { return world == 1 ? 0 : (int32_t)((uint32_t)(I + J) % (uint32_t)world); }
    EKF_HD bool mine(int64_t I, int64_t J) const { return owner(I, J) == rank; }
    // local slot of tile (I,J), I >= J, valid only if mine(I,J)
    EKF_HD int64_t slot(int64_t I, int64_t J) const { return row_base(I) + (world == 1 ? J : (int64_t)((uint32_t)J / (uint32_t)world)); }
    // tile rows covering m landmark-block rows; the same rounded up to whole tiles
    EKF_HD int64_t tiles_for(int64_t mm_rows) const { return (mm_rows + T - 1) >> shift; }
    EKF_HD int64_t padded(int64_t mm_rows) const { return tiles_for(mm_rows) << shift; }
    // element offset of tile (I,J) in the local tile store
    EKF_HD int64_t tile_offset(int64_t I, int64_t J) const { return slot(I, J) * (int64_t)T * T; }
    // local slots needed for nt tile rows
    EKF_HD int64_t slots_for_rows(int64_t nt) const { return row_base(nt); }
};

EKF_HD TileMap ekf_make_tilemap(int32_t T, int32_t world, int32_t rank) {
    TileMap tm;
    tm.T = T; tm.world = world; tm.rank = rank; tm.shift = 0;
    while ((1 << tm.shift) < T) ++tm.shift;
    return tm;
}

// tile rows covering m landmark-block rows
EKF_HD int64_t ekf_tiles_for(int64_t mm_rows, int32_t T) { return (mm_rows + T - 1) / T; }
