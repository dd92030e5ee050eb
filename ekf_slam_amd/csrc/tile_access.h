// Fragment of kernels.hip (included there, inside its anonymous namespace, after kernels.h / device_math.h): canonical element access of the tiled landmark block, the two-FMA downdate primitive.
#pragma once

// ---------------------------------------------------------------------------------------------------
// element access
// ---------------------------------------------------------------------------------------------------
// canonical (lower-triangle) element (r,c) of the landmark block; r,c are landmark-block indices
template <typename TS>
__device__ __forceinline__ double pmm_low(const TS *__restrict__ tiles, const TileMap &tm, int64_t r, int64_t c) {
    if (r < c) { const int64_t t = r; r = c; c = t; }
    const int64_t I = r >> tm.shift, J = c >> tm.shift;
    const int64_t m = tm.T - 1;
    return (double)tiles[tm.tile_offset(I, J) + ((r & m) << tm.shift) + (c & m)];
}

template <typename TS> struct Vec2;
template <> struct Vec2<double> { using type = double2; };
template <> struct Vec2<float> { using type = float2; };

// canonical elements (r,c) and (r,c+1) for r > c + 1 and even c: adjacent in one tile row -> one 16- / 8-byte load
template <typename TS>
__device__ __forceinline__ void pmm_low_pair(const TS *__restrict__ tiles, const TileMap &tm, int64_t r, int64_t c, double &v0,
                                             double &v1) {
    const int64_t I = r >> tm.shift, J = c >> tm.shift;
    const int64_t m = tm.T - 1;
    const typename Vec2<TS>::type t =
        *reinterpret_cast<const typename Vec2<TS>::type *>(tiles + tm.tile_offset(I, J) + ((r & m) << tm.shift) + (c & m));
    v0 = (double)t.x; v1 = (double)t.y;
}

template <typename TS>
__device__ __forceinline__ void pmm_low_store(TS *__restrict__ tiles, const TileMap &tm, int64_t r, int64_t c, double v) {
    if (r < c) { const int64_t t = r; r = c; c = t; }
    const int64_t I = r >> tm.shift, J = c >> tm.shift;
    const int64_t m = tm.T - 1;
    tiles[tm.tile_offset(I, J) + ((r & m) << tm.shift) + (c & m)] = (TS)v;
}

// THE rank-2 element update.  One definition, no FP contraction left to the compiler, so that the deferred
// path (rows patched on the fly from pending pairs) and the flush (pairs applied to the tiles) produce
// bit-identical values, and sharded == unsharded.
__device__ __forceinline__ double rank2_apply(double v, double2 k, double2 g) {
    return fma(-k.y, g.y, fma(-k.x, g.x, v));       // v - K(r,1) G(1,c) - K(r,2) G(2,c): two FMAs, fixed order
}


// full-state element P(r,c), r,c in [0, 3+n_mm)
template <typename TS>
__device__ __forceinline__ double p_at(const DevState &st, int cur, int64_t r, int64_t c) {
    if (r < 3 && c < 3) return st.prr[cur][3 * r + c];
    if (r < 3) return st.strip[cur][r * st.ldm + (c - 3)];
    if (c < 3) return st.strip[cur][c * st.ldm + (r - 3)];
    int64_t rm = r - 3, cm = c - 3;
    if (rm < cm) { const int64_t t = rm; rm = cm; cm = t; }
    if ((rm >> 1) == (cm >> 1)) return st.diag[st.dcur][3 * (rm >> 1) + (rm & 1) + (cm & 1)];     // a landmark's own 2x2 block: the live F64 copy
    if (!st.tm.mine(rm >> st.tm.shift, cm >> st.tm.shift)) return NAN;     // held by another shard
    return pmm_low<TS>((const TS *)st.tiles, st.tm, rm, cm);
}
