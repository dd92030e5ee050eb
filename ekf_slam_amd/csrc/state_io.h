// Fragment of kernels.hip (included there, inside its anonymous namespace, after kernels.h / device_math.h): dense <-> tiled, block reads, low-rank bulk load, digests.
#pragma once

// ---------------------------------------------------------------------------------------------------
// dense <-> tiled, block reads, low-rank bulk load, digests
// ---------------------------------------------------------------------------------------------------
template <typename TS>
__global__ __launch_bounds__(kBlock) void k_unpack_dense(DevState st, int cur, int64_t n, double *__restrict__ dense) {
    // column-major output; consecutive threads walk a column (consecutive rows)
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n * n) return;
    const int64_t c = e / n, r = e - c * n;
    dense[e] = p_at<TS>(st, cur, r, c);
}

template <typename TS>
__global__ __launch_bounds__(kBlock) void k_pack_dense(DevState st, int cur, int64_t n, const double *__restrict__ dense) {
    // one thread per element of the lower triangle (r >= c) of the column-major input
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n * n) return;
    const int64_t c = e / n, r = e - c * n;
    if (r < c) return;
    const double v = dense[e];
    if (r < 3) { st.prr[cur][3 * r + c] = v; st.prr[cur][3 * c + r] = v; return; }
    if (c < 3) { st.strip[cur][c * st.ldm + (r - 3)] = v; return; }
    const int64_t rm = r - 3, cm = c - 3;
    if ((rm >> 1) == (cm >> 1)) st.diag[st.dcur][3 * (rm >> 1) + (rm & 1) + (cm & 1)] = v;        // (every shard: the diagonal blocks are replicated)
    if (st.tm.mine(rm >> st.tm.shift, cm >> st.tm.shift)) pmm_low_store<TS>((TS *)st.tiles, st.tm, rm, cm, v);
}

template <typename TS>
__global__ __launch_bounds__(kBlock) void k_get_block(DevState st, int cur, int64_t r0, int64_t c0, int64_t nr, int64_t nc,
                                                      double *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= nr * nc) return;
    const int64_t c = e / nr, r = e - c * nr;
    out[e] = p_at<TS>(st, cur, r0 + r, c0 + c);
}

// what plot() reads (EKF_SLAM.m:180,205): P(1:2,1:2) and every landmark's 2x2 diagonal block, 4 doubles each, column-major
template <typename TS>
__global__ __launch_bounds__(kBlock) void k_get_diag_blocks(DevState st, int cur, int64_t N, double *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= 4 * (N + 1)) return;
    const int64_t b = e >> 2;
    const int r = (int)(e & 1), c = (int)((e >> 1) & 1);
    const int64_t j = b == 0 ? 0 : 3 + 2 * (b - 1);
    out[e] = p_at<TS>(st, cur, j + r, j + c);
}

// P = diag(d) + U U'.  Grid: x over (row, column-chunk) of the lower triangle in tile units is not needed
// here (one-off bulk load): one thread per lower-triangle element of the padded tile grid.
template <typename TS>
__global__ __launch_bounds__(kBlock) void k_lowrank_tiles(DevState st, int64_t n_mm, const int2 *__restrict__ work,
                                                          int64_t nwork, const double *__restrict__ d,
                                                          const double *__restrict__ U, int64_t k) {
    const int T = st.tm.T;
    const int64_t n = n_mm + 3;
    TS *__restrict__ tiles = (TS *)st.tiles;
    for (int64_t w = blockIdx.x; w < nwork; w += gridDim.x) {
        const int2 ij = work[w];
        TS *__restrict__ tp = tiles + st.tm.tile_offset(ij.x, ij.y);
        for (int e = threadIdx.x; e < T * T; e += kBlock) {
            const int rr = e >> st.tm.shift, cc = e & (T - 1);
            const int64_t r = (int64_t)ij.x * T + rr, c = (int64_t)ij.y * T + cc;
            double v = 0.0;
            if (r < n_mm && c < n_mm) {
                for (int64_t q = 0; q < k; ++q) v += U[q * n + 3 + r] * U[q * n + 3 + c];
                if (r == c) v += d[3 + r];
            }
            tp[e] = (TS)v;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_lowrank_robot(DevState st, int cur, int64_t n_mm, const double *__restrict__ d,
                                                          const double *__restrict__ U, int64_t k) {
    const int64_t n = n_mm + 3;
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c < n_mm) {
        for (int r = 0; r < 3; ++r) {
            double v = 0.0;
            for (int64_t q = 0; q < k; ++q) v += U[q * n + r] * U[q * n + 3 + c];
            st.strip[cur][r * st.ldm + c] = v;
        }
        // the landmark's own diagonal-block entries: the arithmetic of k_lowrank_tiles, in F64
        double *__restrict__ dg = st.diag[st.dcur] + 3 * (c >> 1);
        double vcc = 0.0, vlo = 0.0;
        for (int64_t q = 0; q < k; ++q) { vcc += U[q * n + 3 + c] * U[q * n + 3 + c]; if (c & 1) vlo += U[q * n + 3 + c] * U[q * n + 3 + c - 1]; }
        vcc += d[3 + c];
        if (c & 1) { dg[1] = vlo; dg[2] = vcc; } else dg[0] = vcc;
    }
    if (c == 0) {
        for (int r = 0; r < 3; ++r) for (int b = 0; b < 3; ++b) {
            double v = 0.0;
            for (int64_t q = 0; q < k; ++q) v += U[q * n + r] * U[q * n + b];
            if (r == b) v += d[r];
            st.prr[cur][3 * r + b] = v;
        }
    }
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) { if (tid < s) sh[tid] += sh[tid + s]; __syncthreads(); }
    const double r = sh[0];
    __syncthreads();
    return r;
}

template <typename TS>
__global__ __launch_bounds__(kBlock) void k_digest(DevState st, int cur, int64_t n_mm, const int2 *__restrict__ work,
                                                   int64_t nwork, double *__restrict__ out) {
    __shared__ double sh[kBlock];
    const int T = st.tm.T;
    const TS *__restrict__ tiles = (const TS *)st.tiles;
    double tr = 0.0, sm = 0.0, sq = 0.0;
    for (int64_t w = blockIdx.x; w < nwork; w += gridDim.x) {
        const int2 ij = work[w];
        const TS *__restrict__ tp = tiles + st.tm.tile_offset(ij.x, ij.y);
        for (int e = threadIdx.x; e < T * T; e += kBlock) {
            const int rr = e >> st.tm.shift, cc = e & (T - 1);
            const int64_t r = (int64_t)ij.x * T + rr, c = (int64_t)ij.y * T + cc;
            if (r < n_mm && c <= r) {
                const double v = ((r >> 1) == (c >> 1)) ? st.diag[st.dcur][3 * (r >> 1) + (r & 1) + (c & 1)] : (double)tp[e];
                sm += v; sq += v * v;
                if (r == c) tr += v;
            }
        }
    }
    if (blockIdx.x == 0 && st.tm.rank == 0) {
        // robot block (lower triangle) and strip are replicated: counted once, by shard 0
        for (int64_t c = threadIdx.x; c < n_mm; c += kBlock)
            for (int r = 0; r < 3; ++r) { const double v = st.strip[cur][r * st.ldm + c]; sm += v; sq += v * v; }
        if (threadIdx.x == 0)
            for (int r = 0; r < 3; ++r) for (int b = 0; b <= r; ++b) {
                const double v = st.prr[cur][3 * r + b];
                sm += v; sq += v * v;
                if (r == b) tr += v;
            }
    }
    tr = block_sum(tr, sh); sm = block_sum(sm, sh); sq = block_sum(sq, sh);
    // Deterministic across runs: every workgroup leaves its partial sums in its own slot; the workgroup that takes the last
    // ticket adds the slots in a fixed order (no floating-point atomics, so equal states give equal digests bit for bit).
    double *part = out + 4;
    int *ticket = (int *)(out + 3);
    __shared__ int last;
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x + 0] = tr; part[3 * blockIdx.x + 1] = sm; part[3 * blockIdx.x + 2] = sq;
        __threadfence();
        last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    double a = 0.0, b = 0.0, c = 0.0;
    for (int g = threadIdx.x; g < (int)gridDim.x; g += kBlock) {
        a += __builtin_nontemporal_load(part + 3 * g); b += __builtin_nontemporal_load(part + 3 * g + 1);
        c += __builtin_nontemporal_load(part + 3 * g + 2);
    }
    a = block_sum(a, sh); b = block_sum(b, sh); c = block_sum(c, sh);
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; out[2] = c; }
}

// cfg.async_flush: landmarks appended WHILE a pass is in flight were written (k_append) to the store that pass reads; the pass's own image of
// those rows in the store it writes is stale (it may have read a tile before the append reached it) -- and is the identity anyway, every pending
// pair of that pass having K = 0 on rows that did not exist when it was recorded.  When the pass retires, rows [lrow0, lrow0 + nrows) of every
// local tile of one tile row (slots slot0 .. slot0 + nslots - 1) are copied from the old store to the new one: 16 bytes per lane.
template <typename TS>
__global__ __launch_bounds__(kBlock) void k_copy_tile_rows(const TS *__restrict__ src, TS *__restrict__ dst, int64_t slot0, int64_t nslots,
                                                         int lrow0, int nrows, int T) {
    constexpr int kE = 16 / (int)sizeof(TS);
    const int lanes = T / kE;                                           // 16-byte pieces of a tile row (T >= 16: at least 2)
    const int per_wg = kBlock / lanes > 0 ? kBlock / lanes : 1;         // tile rows one workgroup copies
    const int sub = (int)threadIdx.x / lanes, piece = (int)threadIdx.x - sub * lanes;
    const int64_t item = (int64_t)blockIdx.x * per_wg + sub;            // (slot, row)
    if (sub >= per_wg || item >= nslots * nrows) return;
    const int64_t slot = slot0 + item / nrows;
    const int row = lrow0 + (int)(item % nrows);
    const int64_t off = slot * (int64_t)T * T + (int64_t)row * T + (int64_t)piece * kE;
    typedef TS v16_t __attribute__((ext_vector_type(kE)));
    *reinterpret_cast<v16_t *>(dst + off) = *reinterpret_cast<const v16_t *>(src + off);
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
