// Fragment of kernels.hip (included there, inside its anonymous namespace, after kernels.h / device_math.h): sharded handles: a landmark's row-panel into / out of the exchange area (PanelView, k_rowpanel*).
#pragma once

// Sharded source of the landmark row-panel M = P(j:j+1, landmark columns): after the all-gather every
// shard holds `world` slabs of `slab` doubles; the T-wide chunk k of M sits in the slab of shard
// (tile_row(j) + k) mod world at local chunk k / world, interleaved (M(1,c), M(2,c)).
struct PanelView {
    const double *recv;
    int64_t slab;       // doubles between two shards' contributions (rank stride)
    int64_t offset;     // doubles to this row-panel inside a shard's contribution (prefetched batches hold several)
    int64_t Ij;         // tile row of j
    int32_t patched;    // 1: the pending pairs are already applied (k_rowpanel did it); 0: base values, patch here
    __device__ __forceinline__ double2 at(const TileMap &tm, int64_t c) const {
        const int64_t k = c >> tm.shift;
        const uint32_t wd = (uint32_t)tm.world;                      // 32-bit unsigned: see layout.h
        const int64_t o = (uint32_t)(Ij + k) % wd;
        const int64_t e = o * slab + offset + ((((int64_t)((uint32_t)k / wd)) << tm.shift) + (c & (tm.T - 1))) * 2;
        return make_double2(recv[e], recv[e + 1]);
    }
};

// chunks (of T columns) of landmark row j's panel that this shard owns: chunk k comes from tile (I_j, k) or (k, I_j), owner (I_j + k) mod world
__host__ __device__ inline int64_t rowpanel_local_chunks(const TileMap &tm, int64_t j, int64_t n_mm) {
    const int64_t nt = (n_mm + tm.T - 1) >> tm.shift;
    const int64_t Ij = j >> tm.shift;
    const int64_t k0 = ((tm.rank - Ij) % tm.world + tm.world) % tm.world;
    return k0 >= nt ? 0 : (nt - k0 + tm.world - 1) / tm.world;
}

constexpr int kMaxPending = 128;       // 2 * max cfg.batch: pairs of an in-flight flush + pairs recorded since (LDS staging bound)

// One landmark's row-panel: this shard's chunks of M = P(j:j+1, :) (canonical lower-triangle entries, patched with the npend pending
// pairs in slot order exactly like the unsharded gather does) into `send`.  `upatch` (LDS) is staged by the caller's whole workgroup.
// kAsStored: the values as the tiles will hold them AFTER the pass that applies these pairs -- rounded through the storage type (the
// F64-arithmetic passes over float tiles round once, at the store; F64 tiles: no-op).
template <typename TS, bool kAsStored>
__device__ __forceinline__ void rowpanel_row(const DevState &st, int64_t j, int64_t n_mm, int pstart, int npend, double *__restrict__ send,
                                             int64_t nchunks_local, double2 *upatch) {
    const TileMap &tm = st.tm;
    const TS *__restrict__ tiles = (const TS *)st.tiles;
    for (int e = threadIdx.x; e < 4 * npend; e += kBlock) {
        const int i = e >> 2, which = e & 3;
        const double *base = (which < 2 ? st.Kp : st.Gp) + (int64_t)ring_slot(pstart, i, st.pcap) * st.pair_stride;
        upatch[e] = reinterpret_cast<const double2 *>(base)[j + (which & 1)];
    }
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;      // local column: kl * T + cc
    if (e >= (nchunks_local << tm.shift)) return;
    const int64_t Ij = j >> tm.shift;
    const uint32_t wd = (uint32_t)tm.world;
    const int64_t k0 = ((uint32_t)tm.rank + wd - (uint32_t)Ij % wd) % wd;   // first chunk owned by this shard
    const int64_t kl = e >> tm.shift, cc = e & (tm.T - 1);
    const int64_t c = ((k0 + kl * tm.world) << tm.shift) + cc;
    double m0 = 0.0, m1 = 0.0;
    const int64_t ps2 = st.pair_stride / 2;
    if (c < n_mm) {
        if (c <= j) {
            m0 = pmm_low<TS>(tiles, tm, j, c);
            m1 = pmm_low<TS>(tiles, tm, j + 1, c);
            // pending pairs 8 at a time: the 8 (independent) loads are in flight together, then applied in slot order
            const double2 *__restrict__ gp = reinterpret_cast<const double2 *>(st.Gp) + c;
            for (int i0 = 0; i0 < npend; i0 += 8) {
                double2 g[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) g[q] = gp[(int64_t)ring_slot(pstart, i0 + q < npend ? i0 + q : npend - 1, st.pcap) * ps2];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (i0 + q < npend) {
                        m0 = rank2_apply(m0, upatch[4 * (i0 + q) + 0], g[q]);
                        m1 = rank2_apply(m1, upatch[4 * (i0 + q) + 1], g[q]);
                    }
            }
        } else if (c >= j + 2) {
            pmm_low_pair<TS>(tiles, tm, c, j, m0, m1);
            const double2 *__restrict__ kp = reinterpret_cast<const double2 *>(st.Kp) + c;
            for (int i0 = 0; i0 < npend; i0 += 8) {
                double2 k[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) k[q] = kp[(int64_t)ring_slot(pstart, i0 + q < npend ? i0 + q : npend - 1, st.pcap) * ps2];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (i0 + q < npend) {
                        m0 = rank2_apply(m0, k[q], upatch[4 * (i0 + q) + 2]);
                        m1 = rank2_apply(m1, k[q], upatch[4 * (i0 + q) + 3]);
                    }
            }
        } else {                                                   // c == j + 1: canonical (j+1,j) and (j+1,j+1)
            m0 = pmm_low<TS>(tiles, tm, j + 1, j);
            m1 = pmm_low<TS>(tiles, tm, j + 1, j + 1);
            for (int i = 0; i < npend; ++i) {
                m0 = rank2_apply(m0, upatch[4 * i + 1], upatch[4 * i + 2]);
                m1 = rank2_apply(m1, upatch[4 * i + 1], upatch[4 * i + 3]);
            }
        }
    }
    if (kAsStored) { m0 = (double)(TS)m0; m1 = (double)(TS)m1; }
    reinterpret_cast<double2 *>(send)[e] = make_double2(m0, m1);
}

// Each shard copies the chunks of M it owns into its send slab.
// kDev (device-resident measure loop on a shard): the landmark is the arg-min over the association's per-workgroup winners
// (dl.parts_in), reduced by every wavefront itself exactly as k_gather<.., kDev> does a launch later -- j only when the winners name
// nothing inside the state; the number of chunks this shard owns follows from the landmark's tile row and is recomputed here (the
// launcher sized the grid for the most any tile row gives).
template <typename TS, bool kDev = false>
__global__ __launch_bounds__(kBlock) void k_rowpanel(DevState st, int64_t j, int64_t n_mm, int pstart, int npend,
                                                     double *__restrict__ send, int64_t nchunks_local,
                                                     typename DevLoopParam<kDev>::type dl) {
    __shared__ double2 upatch[kMaxPending * 4];     // per pending pair: K_i(j,:), K_i(j+1,:), G_i(:,j), G_i(:,j+1)
    if constexpr (kDev) {
        double dll;
        int dix;
        reduce_partials_wave(dl.parts_in, dl.nblk_in, dl.seq_in, threadIdx.x & 63, dll, dix);
        dix = __builtin_amdgcn_readfirstlane(dix);
        if (dix >= 0 && 2 * (int64_t)dix < n_mm) j = 2 * (int64_t)dix;
        nchunks_local = rowpanel_local_chunks(st.tm, j, n_mm);
    }
    rowpanel_row<TS, false>(st, j, n_mm, pstart, npend, send, nchunks_local, upatch);
}

// The row-panels of up to 64 landmarks AS THEY WILL BE AFTER THE PASS that applies the npend pending pairs (ekf_prefetch_next: the next
// batch's prefetch, extracted in front of this batch's pass so that its all-gather runs beside the pass): blockIdx.y picks the landmark.
template <typename TS>
__global__ __launch_bounds__(kBlock) void k_rowpanel_next(DevState st, RowList rows, int64_t n_mm, int pstart, int npend,
                                                          double *__restrict__ send, int64_t slab) {
    __shared__ double2 upatch[kMaxPending * 4];
    const int q = blockIdx.y;
    const int64_t j = rows.j[q];
    rowpanel_row<TS, true>(st, j, n_mm, pstart, npend, send + (int64_t)q * slab, rowpanel_local_chunks(st.tm, j, n_mm), upatch);
}


// The BASE row-panels (no pending pairs applied) of up to 64 landmarks in ONE launch: blockIdx.y picks the landmark, the rest
// is k_rowpanel with npend = 0.  A prefetch (ekf_prefetch_rows) used to launch k_rowpanel once per landmark: 32 launches of
// ~3 us in front of every batch's all-gather.
template <typename TS>
__global__ __launch_bounds__(kBlock) void k_rowpanel_base(DevState st, RowList rows, int64_t n_mm, double *__restrict__ send,
                                                          int64_t slab) {
    const TileMap &tm = st.tm;
    const TS *__restrict__ tiles = (const TS *)st.tiles;
    const int q = blockIdx.y;
    const int64_t j = rows.j[q];
    const uint32_t wd = (uint32_t)tm.world;
    const uint32_t Ij = (uint32_t)(j >> tm.shift), nt = (uint32_t)tm.tiles_for(n_mm);
    const uint32_t k0 = ((uint32_t)tm.rank + wd - Ij % wd) % wd;       // first chunk owned by this shard
    const uint32_t nloc = k0 >= nt ? 0u : (nt - k0 + wd - 1) / wd;
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;      // local column: kl * T + cc
    if (e >= ((int64_t)nloc << tm.shift)) return;
    const int64_t kl = e >> tm.shift, cc = e & (tm.T - 1);
    const int64_t c = (((int64_t)k0 + kl * tm.world) << tm.shift) + cc;
    double m0 = 0.0, m1 = 0.0;
    if (c < n_mm) {
        if (c <= j) { m0 = pmm_low<TS>(tiles, tm, j, c); m1 = pmm_low<TS>(tiles, tm, j + 1, c); }
        else if (c >= j + 2) pmm_low_pair<TS>(tiles, tm, c, j, m0, m1);
        else pmm_low_pair<TS>(tiles, tm, j + 1, j, m0, m1);           // c == j + 1: canonical (j+1,j) and (j+1,j+1)
    }
    reinterpret_cast<double2 *>(send + (int64_t)q * slab)[e] = make_double2(m0, m1);
}
