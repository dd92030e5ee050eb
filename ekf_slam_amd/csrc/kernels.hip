// gfx950 (MI355X / CDNA4) kernels of the EKF-SLAM update engine.
//
// The path as the reference writes it is HBM- or latency-bound (AI of the rank-2 downdate is 0.25 flop/B in f64), so the
// rules that matter are: 16-byte-per-lane coalesced accesses on whole 64-lane wavefronts, many independent loads in
// flight per lane, >> 256 workgroups per launch, nothing re-read from HBM that can be kept in registers, no host
// synchronisation between launches.  Two pieces ARE GEMM-shaped and run on the f64 matrix cores, bit-identical to
// their scalar fma formulation: the deferred rank-2m flush (k_flush_mfma) and the predict panel (k_predict_mfma).
//
// Reference expressions realised (file:line in the reference tree):
//   k_predict    P = F*P*F' + Q, x = f(x,u), wrapTo360          EKF_SLAM.m:40-51,56-65
//   k_append     state/covariance growth                         EKF_SLAM.m:67-98 (append.m:1-27)
//   k_gather     z_k, H_k, phi_k, K, x += K nu                   EKF_SLAM.m:125-144
//   k_downdate*  P = (I - K H_k) P  ==  P - K (H_k P)            EKF_SLAM.m:145
//   k_flush_*    the same for m deferred corrections in one pass  EKF_SLAM.m:145 (x m)
//   k_associate  per-landmark phi_k, Mahalanobis + signature     Correspondence.m:49-87
#include "kernels.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "device_math.h"
#include "flush32_mfma.h"     // kBlock, ring_slot, k_flush_mfma32
#include "flush32_pipe.h"     // k_flush_strip32
#include "flush32_split.h"    // k_split_pairs, k_flush_split3

// The kernels, by family (each header is a fragment of THIS translation unit, not a stand-alone interface):
namespace {

#include "tile_access.h"      // canonical element access, rank2_apply
#include "predict.h"          // k_predict, k_predict_mfma and the shared 3x3 part
#include "assoc_winners.h"    // association order, the self-validating winner entries
#include "append.h"           // k_append
#include "solve_small.h"      // the 5x5 solve, entry by entry
#include "rowpanel.h"         // PanelView, k_rowpanel, k_rowpanel_next, k_rowpanel_base
#include "gather.h"           // k_gather
#include "downdate.h"         // k_downdate, k_downdate_w
#include "associate.h"        // k_associate, k_assoc_merge
#include "state_io.h"         // dense <-> tiled, block reads, low-rank load, digest

}  // namespace

int gather_fuse_max_rows() { return kFuseMaxRows; }
size_t pass_split_plane_elems(int64_t ldm) { return ekf_pipe32::split_plane_elems(ldm); }

// ---------------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------------
#define EKF_STORAGE_DISPATCH(storage, EXPR_F64, EXPR_F32) \
    do { if ((storage) == 0) { EXPR_F64; } else { EXPR_F32; } } while (0)

hipError_t launch_predict(const DevState &st, const PredictArgs &a, int, hipStream_t s) {
    // MFMA panel product at large landmark counts (EKF_PREDICT_MFMA=0/1 forces the VALU / MFMA kernel)
    static const int force = ekf_tune_int("EKF_PREDICT_MFMA", -1);
    const bool mfma = force >= 0 ? force != 0 : a.n_mm >= 2048;
    if (mfma) {
        const int64_t nslices = (a.n_mm + 15) / 16;
        int64_t grid = cdiv(nslices > 0 ? nslices : 1, 4);
        if (grid > 1024) grid = 1024;
        hipLaunchKernelGGL(k_predict_mfma, dim3((unsigned)grid), dim3(kBlock), 0, s, st, a);
        return hipGetLastError();
    }
    const int64_t grid = cdiv(a.n_mm > 0 ? a.n_mm : 1, kBlock);
    hipLaunchKernelGGL(k_predict, dim3((unsigned)grid), dim3(kBlock), 0, s, st, a);
    return hipGetLastError();
}

hipError_t launch_append(const DevState &st, const AppendArgs &a, int storage, hipStream_t s, const DevLoopArgs *dlp,
                         const PredictArgs *fused_predict) {
    const int64_t n_mm = 2 * a.N;
    const int64_t grid = cdiv(n_mm > 0 ? n_mm : 1, kBlock);
    DevLoopArgs dl = {};
    if (dlp) dl = *dlp;
    PredictArgs pa = {};
    if (fused_predict) {
        pa = *fused_predict;
        EKF_STORAGE_DISPATCH(storage,
            hipLaunchKernelGGL((k_append<double, true>), dim3((unsigned)grid), dim3(kBlock), 0, s, st, a, dl, pa),
            hipLaunchKernelGGL((k_append<float, true>), dim3((unsigned)grid), dim3(kBlock), 0, s, st, a, dl, pa));
    } else {
        EKF_STORAGE_DISPATCH(storage,
            hipLaunchKernelGGL((k_append<double, false>), dim3((unsigned)grid), dim3(kBlock), 0, s, st, a, dl, pa),
            hipLaunchKernelGGL((k_append<float, false>), dim3((unsigned)grid), dim3(kBlock), 0, s, st, a, dl, pa));
    }
    return hipGetLastError();
}

hipError_t launch_gather(const DevState &st, const CorrectArgs &a, const PredictArgs *fused_predict, int storage,
                         hipStream_t s, bool fuse_downdate) {
    const int64_t cols = ekf_tiles_for(a.n_mm, st.tm.T) * st.tm.T;
    const int64_t grid = cdiv(cols, kGatherCols);
    PanelView pv;
    pv.recv = nullptr; pv.slab = 0; pv.offset = 0; pv.Ij = 0; pv.patched = 0;
    PredictArgs pa = {};
    if (fused_predict) pa = *fused_predict;
#define EKF_G(TS_, PRED_, FUSE_) hipLaunchKernelGGL((k_gather<TS_, false, PRED_, FUSE_>), dim3((unsigned)grid), dim3(kGatherBlock), 0, s, st, a, pv, pa, NoDevLoop{})
    if (fuse_downdate) {
        if (a.n_mm > kFuseMaxRows || grid != 1) return hipErrorInvalidValue;
        if (storage == 0) { if (fused_predict) EKF_G(double, true, true); else EKF_G(double, false, true); }
        else              { if (fused_predict) EKF_G(float, true, true); else EKF_G(float, false, true); }
    } else {
        if (storage == 0) { if (fused_predict) EKF_G(double, true, false); else EKF_G(double, false, false); }
        else              { if (fused_predict) EKF_G(float, true, false); else EKF_G(float, false, false); }
    }
#undef EKF_G
    return hipGetLastError();
}

int64_t gather_workgroups(const DevState &st, int64_t n_mm) { return cdiv(ekf_tiles_for(n_mm, st.tm.T) * st.tm.T, kGatherCols); }

hipError_t launch_gather_devloop(const DevState &st, const CorrectArgs &a, const PredictArgs *fused_predict, const DevLoopArgs &dl,
                                 int storage, hipStream_t s) {
    if (!dl.parts_in || !dl.rec || dl.nblk_in < 1 || a.n_mm < 2) return hipErrorInvalidValue;
    const int64_t grid = gather_workgroups(st, a.n_mm);
    PanelView pv;
    pv.recv = nullptr; pv.slab = 0; pv.offset = 0; pv.Ij = 0; pv.patched = 0;
    PredictArgs pa = {};
    if (fused_predict) pa = *fused_predict;
#define EKF_G(TS_, PRED_) hipLaunchKernelGGL((k_gather<TS_, false, PRED_, false, true>), dim3((unsigned)grid), dim3(kGatherBlock), 0, s, st, a, pv, pa, dl)
    if (storage == 0) { if (fused_predict) EKF_G(double, true); else EKF_G(double, false); }
    else              { if (fused_predict) EKF_G(float, true); else EKF_G(float, false); }
#undef EKF_G
    return hipGetLastError();
}

hipError_t launch_rowpanel(const DevState &st, int64_t j, int64_t n_mm, int pstart, int npend, double *send, int storage,
                           hipStream_t s) {
    const int64_t nloc = rowpanel_local_chunks(st.tm, j, n_mm);
    if (nloc == 0) return hipSuccess;
    const int64_t grid = cdiv(nloc * st.tm.T, kBlock);
    EKF_STORAGE_DISPATCH(storage,
        hipLaunchKernelGGL(k_rowpanel<double>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, j, n_mm, pstart, npend, send, nloc, NoDevLoop{}),
        hipLaunchKernelGGL(k_rowpanel<float>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, j, n_mm, pstart, npend, send, nloc, NoDevLoop{}));
    return hipGetLastError();
}

hipError_t launch_rowpanel_dev(const DevState &st, int64_t j, int64_t n_mm, int pstart, int npend, double *send, int storage,
                               hipStream_t s, const DevLoopArgs &dl) {
    if (!dl.parts_in || dl.nblk_in < 1) return hipErrorInvalidValue;
    const int64_t nt = ekf_tiles_for(n_mm, st.tm.T);
    const int64_t most = (nt + st.tm.world - 1) / st.tm.world;           // what the tile row with this shard's first chunk at k = 0 gives
    const int64_t grid = cdiv(most * st.tm.T, kBlock);
    EKF_STORAGE_DISPATCH(storage,
        hipLaunchKernelGGL((k_rowpanel<double, true>), dim3((unsigned)grid), dim3(kBlock), 0, s, st, j, n_mm, pstart, npend, send, (int64_t)0, dl),
        hipLaunchKernelGGL((k_rowpanel<float, true>), dim3((unsigned)grid), dim3(kBlock), 0, s, st, j, n_mm, pstart, npend, send, (int64_t)0, dl));
    return hipGetLastError();
}

hipError_t launch_rowpanel_next(const DevState &st, const int64_t *idx, int m, int64_t n_mm, int pstart, int npend, double *send,
                                int64_t slab, int storage, hipStream_t s) {
    if (m <= 0) return hipSuccess;
    if (m > 64 || npend < 0 || npend > kMaxPending) return hipErrorInvalidValue;
    RowList rows;
    rows.m = m;
    for (int q = 0; q < m; ++q) rows.j[q] = (int32_t)(2 * idx[q]);
    const int64_t nt = st.tm.tiles_for(n_mm);
    const int64_t max_chunks = (nt + st.tm.world - 1) / st.tm.world;
    if (max_chunks == 0) return hipSuccess;
    const int64_t grid = cdiv(max_chunks * st.tm.T, kBlock);
    EKF_STORAGE_DISPATCH(storage,
        hipLaunchKernelGGL(k_rowpanel_next<double>, dim3((unsigned)grid, (unsigned)m), dim3(kBlock), 0, s, st, rows, n_mm, pstart, npend, send, slab),
        hipLaunchKernelGGL(k_rowpanel_next<float>, dim3((unsigned)grid, (unsigned)m), dim3(kBlock), 0, s, st, rows, n_mm, pstart, npend, send, slab));
    return hipGetLastError();
}

hipError_t launch_rowpanel_base(const DevState &st, const int64_t *idx, int m, int64_t n_mm, double *send, int64_t slab,
                                int storage, hipStream_t s) {
    if (m <= 0) return hipSuccess;
    if (m > 64) return hipErrorInvalidValue;
    RowList rows;
    rows.m = m;
    for (int q = 0; q < m; ++q) rows.j[q] = (int32_t)(2 * idx[q]);
    const int64_t nt = st.tm.tiles_for(n_mm);
    const int64_t max_chunks = (nt + st.tm.world - 1) / st.tm.world;
    if (max_chunks == 0) return hipSuccess;
    const int64_t grid = cdiv(max_chunks * st.tm.T, kBlock);
    EKF_STORAGE_DISPATCH(storage,
        hipLaunchKernelGGL(k_rowpanel_base<double>, dim3((unsigned)grid, (unsigned)m), dim3(kBlock), 0, s, st, rows, n_mm, send, slab),
        hipLaunchKernelGGL(k_rowpanel_base<float>, dim3((unsigned)grid, (unsigned)m), dim3(kBlock), 0, s, st, rows, n_mm, send, slab));
    return hipGetLastError();
}

hipError_t launch_gather_sharded(const DevState &st, const CorrectArgs &a, const PredictArgs *fused_predict, const double *recv,
                                 int64_t rank_stride, int64_t offset, bool patched, int storage, hipStream_t s, const DevLoopArgs *dl) {
    const int64_t cols = ekf_tiles_for(a.n_mm, st.tm.T) * st.tm.T;
    const int64_t grid = cdiv(cols, kGatherCols);
    PanelView pv;
    pv.recv = recv; pv.slab = rank_stride; pv.offset = offset; pv.Ij = a.j >> st.tm.shift; pv.patched = patched ? 1 : 0;
    PredictArgs pa = {};
    if (fused_predict) pa = *fused_predict;
    if (dl) {
        if (!dl->parts_in || !dl->rec || dl->nblk_in < 1 || a.n_mm < 2) return hipErrorInvalidValue;
#define EKF_GD(TS_, PRED_) hipLaunchKernelGGL((k_gather<TS_, true, PRED_, false, true>), dim3((unsigned)grid), dim3(kGatherBlock), 0, s, st, a, pv, pa, *dl)
        if (storage == 0) { if (fused_predict) EKF_GD(double, true); else EKF_GD(double, false); }
        else              { if (fused_predict) EKF_GD(float, true); else EKF_GD(float, false); }
#undef EKF_GD
        return hipGetLastError();
    }
#define EKF_G(TS_, PRED_) hipLaunchKernelGGL((k_gather<TS_, true, PRED_>), dim3((unsigned)grid), dim3(kGatherBlock), 0, s, st, a, pv, pa, NoDevLoop{})
    if (storage == 0) { if (fused_predict) EKF_G(double, true); else EKF_G(double, false); }
    else              { if (fused_predict) EKF_G(float, true); else EKF_G(float, false); }
#undef EKF_G
    return hipGetLastError();
}

#include "flush64_mfma.h"

// what was launched, for the measurement hooks (ekf_downdate_kernel_name): "k_xxx<double,128,4,false>"
static void name_kernel(char *out, const char *base, size_t elt, int T, int p3, int xcd) {
    if (!out) return;
    if (xcd < 0) snprintf(out, 64, "%s<%s,%d,%d>", base, elt == 8 ? "double" : "float", T, p3);
    else snprintf(out, 64, "%s<%s,%d,%d,%s>", base, elt == 8 ? "double" : "float", T, p3, xcd ? "true" : "false");
}

// Dynamic LDS beyond 64 KiB has to be allowed per kernel AND per device (a ShardGroup drives several devices from one process): set once for
// the device that is current at the launch, remembered in a bit mask.
static hipError_t allow_dynamic_lds(const void *fn, size_t bytes, std::atomic<uint64_t> &done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// the MFMA flush for the (storage type, tile edge) pairs it exists for; false: not applicable, use the VALU kernels
template <typename TS, int T>
static bool launch_flush_mfma(const DevState &st, void *dstv, const int2 *work_xcd, int64_t xcd_len, int pstart, int npairs,
                              int grid_cap, hipStream_t s, char *kname, int arith, const PassAux *aux) {
    constexpr bool kHave = (sizeof(TS) == 8 && T == 128) || (sizeof(TS) == 4 && T == 256);
    if constexpr (kHave) {
        static const bool use_mfma = ekf_tune_int("EKF_FLUSH_MFMA", 1) != 0;
        static const int chunk_switch = ekf_tune_int("EKF_FLUSH_MFMA_SWITCH", 30);
        // F32 tiles: also for a single pair -- the 64 x 128 work items stream the float tiles faster than the one-pair VALU kernel
        // (40 k landmarks: 4.4 ms vs 4.9 ms per pass); F64 tiles: the one-pair VALU kernel is the faster one (0.53 vs 0.56 ms)
        constexpr int kMinPairs = sizeof(TS) == 4 ? 1 : 2;
        if (!use_mfma || npairs < kMinPairs || !work_xcd || xcd_len <= 0) return false;
        constexpr int kSubs = (T / 64) * (T / 128);
        int64_t grid = 8 * xcd_len * kSubs;
        if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
        if constexpr (sizeof(TS) == 4) {
            static const bool use_strip = ekf_tune_int("EKF_PASS_STRIP", 1) != 0;
            if (arith == 2 && npairs >= 28 && npairs <= 64 && aux && aux->segs && aux->nsegs > 0 && aux->Kb3 && aux->Gb3) {
                // cfg.pass_arith = EKF_ARITH_SPLIT3, 28-64 pairs (below, the F32 kernels are the faster ones: 20 pairs 4.27 against 4.75 ms at 40 000 landmarks, 32
                // pairs 5.08 against 4.74 -- round4_tuning.md 57): the float copies cut into three bf16 planes (logical pair order, zeros beyond
                // npairs), then the strip form of the pass on the bf16 matrix pipe (flush32_split.h) -- bound by HBM, not by the matrix pipe
                static std::atomic<uint64_t> lds_ok{0};
                const hipError_t attr = allow_dynamic_lds((const void *)ekf_pipe32::k_flush_split3<2>, ekf_pipe32::lds_bytes_split(), lds_ok);
                if (attr == hipSuccess) {
                    hipLaunchKernelGGL(ekf_pipe32::k_split_pairs, dim3((unsigned)(aux->cols / 256), ekf_pipe32::kKB, 2), dim3(256), 0, s, (const float *)st.Kp32,
                                       (const float *)st.Gp32, aux->Kb3, aux->Gb3, st.pair_stride, st.ldm, aux->cols, pstart, st.pcap, npairs);
                    hipLaunchKernelGGL((ekf_pipe32::k_flush_split3<2>), dim3((unsigned)aux->grid), dim3(512), ekf_pipe32::lds_bytes_split(), s,
                                       (const float *)st.tiles, (float *)dstv, aux->segs, aux->nsegs, (const uint16_t *)aux->Kb3, (const uint16_t *)aux->Gb3, st.ldm,
                                       st.tm, aux->dump);
                    if (kname) snprintf(kname, 64, "k_flush_split3<2>");
                    return true;
                }
            }
            if (use_strip && arith == 1 && npairs > 56 && npairs <= 64 && aux && aux->segs && aux->nsegs > 0) {
                // 57-64 pairs (eight stages of eight): the strip form -- one persistent workgroup per CU walks row strips with -K in its
                // wavefronts' registers and a whole item's G double-buffered in LDS (flush32_pipe.h): 7.3 ms against 8.1-8.3 at 40 000
                // landmarks and 64 pairs, same bits
                static std::atomic<uint64_t> lds_ok{0};
                const hipError_t attr = allow_dynamic_lds((const void *)ekf_pipe32::k_flush_strip32<8>, ekf_pipe32::lds_bytes_strip<8>(), lds_ok);
                if (attr == hipSuccess) {
                    hipLaunchKernelGGL((ekf_pipe32::k_flush_strip32<8>), dim3((unsigned)aux->grid), dim3(512), ekf_pipe32::lds_bytes_strip<8>(), s,
                                       (const float *)st.tiles, (float *)dstv, aux->segs, aux->nsegs, (const float *)st.Kp32, (const float *)st.Gp32, st.pair_stride,
                                       st.ldm, pstart, st.pcap, npairs, st.tm, aux->dump, (unsigned long long *)nullptr);
                    if (kname) snprintf(kname, 64, "k_flush_strip32<8>");
                    return true;
                }
            }
            if (arith >= 1 && npairs > 2) {                           // cfg.pass_arith = EKF_ARITH_F32 (and EKF_ARITH_SPLIT3 up to 27 pairs): the f32 matrix pipe (one or two pairs: the pass is
                                                                      // purely HBM-bound and the F64-arithmetic kernel below streams it 5 % faster, 4.0 against 4.3 ms at 40 k)
#define EKF_M32(CH, RG, WPE) do { int64_t g32 = 8 * xcd_len * (T / (64 * RG)) * (T / 128); if (grid_cap > 0 && g32 > grid_cap) g32 = grid_cap; \
                                  hipLaunchKernelGGL((k_flush_mfma32<T, CH, RG, WPE>), dim3((unsigned)g32), dim3(kBlock), 0, s, (const float *)st.tiles, \
                                           (float *)dstv, work_xcd, xcd_len, st.Kp32, st.Gp32, st.pair_stride, pstart, st.pcap, npairs, st.tm); \
                                  if (kname) snprintf(kname, 64, "k_flush_mfma32<%d,%d,%d,%d>", T, CH, RG, WPE); } while (0)
#define EKF_M32E(CH, RG, WPE) do { int64_t g32 = 8 * xcd_len * (T / (64 * RG)) * (T / 128); if (grid_cap > 0 && g32 > grid_cap) g32 = grid_cap; \
                                  hipLaunchKernelGGL((k_flush_mfma32<T, CH, RG, WPE, true>), dim3((unsigned)g32), dim3(kBlock), 0, s, (const float *)st.tiles, \
                                           (float *)dstv, work_xcd, xcd_len, st.Kp32, st.Gp32, st.pair_stride, pstart, st.pcap, npairs, st.tm); \
                                  if (kname) snprintf(kname, 64, "k_flush_mfma32<%d,%d,%d,%d,early>", T, CH, RG, WPE); } while (0)
#ifdef EKF_TUNING
                const int v = (ekf_tune_int("EKF_MFMA32_EARLY", 0) ? 1 : 0) + (ekf_tune_int("EKF_MFMA32_EARLY", 0) ? 10 : 1) *
                              (100 * ekf_tune_int("EKF_MFMA32_RG", 0) + 10 * ekf_tune_int("EKF_MFMA32_CHUNK", 4) + ekf_tune_int("EKF_MFMA32_WPE", 4));
                switch (v) {
                    case 144: EKF_M32(4, 1, 4); return true;
                    case 146: EKF_M32(4, 1, 6); return true;
                    case 184: EKF_M32(8, 1, 4); return true;
                    case 243: EKF_M32(4, 2, 3); return true;
                    case 244: EKF_M32(4, 2, 4); return true;
                    case 2831: EKF_M32E(8, 2, 3); return true;
                    case 2821: EKF_M32E(8, 2, 2); return true;
                    case 2431: EKF_M32E(4, 2, 3); return true;       // + early tile request (EKF_MFMA32_RG=24 encodes "2, early")
                    case 2441: EKF_M32E(4, 2, 4); return true;
                    case 2421: EKF_M32E(4, 2, 2); return true;
                    default: break;
                }
#endif
                // three wavefronts per SIMD; from five pairs on the tile is requested in front of the LAST chunk's matrix work instead of after it
                // (its 64 registers are free once nothing is fetched any more): 5.37 against 5.63-5.67 ms at 32 pairs, 4.07 against 4.11-4.20 at 12,
                // equal at 64 (profiles/round3_tuning.md 40)
                if (npairs <= 4) EKF_M32(4, 2, 3); else EKF_M32E(4, 2, 3);
#undef EKF_M32E
#undef EKF_M32
                return true;
            }
        }
        if constexpr (sizeof(TS) == 8) {
            // Up to 12 pairs the pass is HBM-bound and gains from finer work items: 64 rows x 64 columns (32 KiB, 4 accumulator blocks per wavefront,
            // five wavefronts per SIMD) -- 0.543 against 0.566 ms at 2-8 pairs, 10 000 landmarks; from ~16 pairs on the 128-column items win (G is read
            // from L2 once per 128 instead of 64 columns: 20 pairs 0.554 against 0.564).  profiles/round3_tuning.md 37.
            static const int half_max = ekf_tune_int("EKF_FLUSH_HALF_MAX", 12);
            if (npairs <= half_max) {
                int64_t g2 = 8 * xcd_len * (T / 64) * (T / 64);
                if (grid_cap > 0 && g2 > grid_cap) g2 = grid_cap;
#ifdef EKF_TUNING
#define EKF_F64H(WPE) do { hipLaunchKernelGGL((k_flush_mfma<TS, T, 4, 64, WPE>), dim3((unsigned)g2), dim3(kBlock), 0, s, (const TS *)st.tiles, (TS *)dstv, \
                                   work_xcd, xcd_len, st.Kp, st.Gp, st.pair_stride, pstart, st.pcap, npairs, st.tm); \
                if (kname) snprintf(kname, 64, "k_flush_mfma<double,%d,4,64,wpe%d>", T, WPE); return true; } while (0)
                switch (ekf_tune_int("EKF_FLUSH_HALF_WPE", 5)) {      // sweep 58: fewer wavefronts per SIMD = a narrower window of addresses in flight
                    case 2: EKF_F64H(2);
                    case 3: EKF_F64H(3);
                    case 4: EKF_F64H(4);
                    default: break;
                }
#undef EKF_F64H
#endif
                hipLaunchKernelGGL((k_flush_mfma<TS, T, 4, 64, 5>), dim3((unsigned)g2), dim3(kBlock), 0, s, (const TS *)st.tiles, (TS *)dstv,
                                   work_xcd, xcd_len, st.Kp, st.Gp, st.pair_stride, pstart, st.pcap, npairs, st.tm);
                if (kname) snprintf(kname, 64, "k_flush_mfma<double,%d,4,64>", T);
                return true;
            }
        }
#ifdef EKF_TUNING
        if constexpr (sizeof(TS) == 8) {                              // sweep 58: 128-row items (eight wavefronts), ablations of the production shape
            const int v = 100 * ekf_tune_int("EKF_FLUSH_WAVES", 4) + 10 * ekf_tune_int("EKF_FLUSH_CHUNK", npairs <= chunk_switch ? 4 : 8) + ekf_tune_int("EKF_FLUSH_ABL", 0);
#define EKF_F64V(CH, WPE, WAVES, ABL) do { int64_t gv = 8 * xcd_len * (T / (16 * WAVES)); if (grid_cap > 0 && gv > grid_cap) gv = grid_cap; \
                hipLaunchKernelGGL((k_flush_mfma<TS, T, CH, 128, WPE, WAVES, ABL>), dim3((unsigned)gv), dim3(64 * WAVES), 0, s, (const TS *)st.tiles, (TS *)dstv, \
                                   work_xcd, xcd_len, st.Kp, st.Gp, st.pair_stride, pstart, st.pcap, npairs, st.tm); \
                if (kname) snprintf(kname, 64, "k_flush_mfma<double,%d,%d,w%d,abl%d>", T, CH, WAVES, ABL); return true; } while (0)
            switch (v) {
                case 340: EKF_F64V(4, 3, 4, 0);                       // (the hundreds digit 3: the production shape at three wavefronts per SIMD)
                case 441: EKF_F64V(4, 4, 4, 1);
                case 442: EKF_F64V(4, 4, 4, 2);
                case 840: EKF_F64V(4, 4, 8, 0);
                case 880: EKF_F64V(8, 4, 8, 0);
                case 841: EKF_F64V(4, 4, 8, 1);
                case 842: EKF_F64V(4, 4, 8, 2);
                default: break;
            }
#undef EKF_F64V
        }
#endif
        if (npairs <= chunk_switch)
            hipLaunchKernelGGL((k_flush_mfma<TS, T, 4>), dim3((unsigned)grid), dim3(kBlock), 0, s, (const TS *)st.tiles, (TS *)dstv,
                               work_xcd, xcd_len, st.Kp, st.Gp, st.pair_stride, pstart, st.pcap, npairs, st.tm);
        else
            hipLaunchKernelGGL((k_flush_mfma<TS, T, 8>), dim3((unsigned)grid), dim3(kBlock), 0, s, (const TS *)st.tiles, (TS *)dstv,
                               work_xcd, xcd_len, st.Kp, st.Gp, st.pair_stride, pstart, st.pcap, npairs, st.tm);
        name_kernel(kname, "k_flush_mfma", sizeof(TS), T, npairs <= chunk_switch ? 4 : 8, -1);
        return true;
    } else {
        return false;
    }
}

template <typename TS, int T, int kSlab>
static hipError_t launch_downdate_ts(const DevState &st, void *dstv, const int2 *work, int64_t nwork, const int2 *work_xcd,
                                     int64_t xcd_len, int pstart, int npairs, int grid_cap, hipStream_t s, char *kname,
                                     const NextRow *nx, bool *extracted, int arith, const PassAux *aux) {
    constexpr int kLanes = T / Lane16<TS>::kCols;
    static const bool use_xcd = ekf_tune_int("EKF_FLUSH_XCD", 1) != 0;
    if (use_xcd && launch_flush_mfma<TS, T>(st, dstv, work_xcd, xcd_len, pstart, npairs, grid_cap, s, kname, arith, aux)) return hipGetLastError();
    if constexpr (kLanes == 64 || kLanes == 32) {
        if (npairs > 1 && use_xcd && work_xcd && xcd_len > 0) {
            int64_t grid = 8 * xcd_len * (T / kSlab);
            if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
            hipLaunchKernelGGL((k_downdate_w<TS, T, kSlab, true>), dim3((unsigned)grid), dim3(kBlock), 0, s, (const TS *)st.tiles,
                               (TS *)dstv, work_xcd, xcd_len, st.Kp, st.Gp, st.pair_stride, pstart, st.pcap, npairs, st.tm, NoNextRow{});
            name_kernel(kname, "k_downdate_w", sizeof(TS), T, kSlab, 1);
        } else if (nx && nx->j >= 0 && npairs == 1) {
            int64_t grid = nwork * (T / kSlab);
            if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
            hipLaunchKernelGGL((k_downdate_w<TS, T, kSlab, false, true>), dim3((unsigned)grid), dim3(kBlock), 0, s, (const TS *)st.tiles,
                               (TS *)dstv, work, nwork, st.Kp, st.Gp, st.pair_stride, pstart, st.pcap, npairs, st.tm, *nx);
            if (kname) snprintf(kname, 64, "k_downdate_w<%s,%d,%d,false,+rowpanel>", sizeof(TS) == 8 ? "double" : "float", T, kSlab);
            if (extracted) *extracted = true;
        } else {
            int64_t grid = nwork * (T / kSlab);
            if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
            hipLaunchKernelGGL((k_downdate_w<TS, T, kSlab, false>), dim3((unsigned)grid), dim3(kBlock), 0, s, (const TS *)st.tiles,
                               (TS *)dstv, work, nwork, st.Kp, st.Gp, st.pair_stride, pstart, st.pcap, npairs, st.tm, NoNextRow{});
            name_kernel(kname, "k_downdate_w", sizeof(TS), T, kSlab, 0);
        }
    } else {
        int64_t grid = nwork * (T / kSlab);
        if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
        hipLaunchKernelGGL((k_downdate<TS, T, kSlab>), dim3((unsigned)grid), dim3(kBlock), 0, s, (const TS *)st.tiles, (TS *)dstv,
                           work, nwork, st.Kp, st.Gp, st.pair_stride, pstart, st.pcap, npairs, st.tm);
        name_kernel(kname, "k_downdate", sizeof(TS), T, kSlab, -1);
    }
    return hipGetLastError();
}

// Granularity.  One pair: ONE pass per workgroup (each lane loads, updates and stores exactly one 16-byte
// piece of a tile row; a workgroup covers 4 KiB of a tile): 6.06 TB/s at 10k landmarks vs 5.49 TB/s for a whole
// 64x64 tile per workgroup (profiles/round1_tuning.md).  Several pairs: the G vectors are re-read from L2 once
// per workgroup and pair, so a taller slab amortises them.  EKF_DOWNDATE_SLAB / EKF_DOWNDATE_SLAB_BATCH (rows
// per workgroup for 1 pair / several pairs) and EKF_DOWNDATE_GRID (grid cap) are tuning hooks.
// Production tiles: T = 128 for f64 storage, T = 256 for f32 storage (one 1 KiB tile row per wave instruction,
// K wave-uniform); T = 16 / 32 (generic kernel) and T = 64 exist for small maps and tests.
template <typename TS>
static hipError_t launch_downdate_t(const DevState &st, void *dstv, const int2 *work, int64_t nwork, const int2 *work_xcd,
                                    int64_t xcd_len, int pstart, int npairs, int grid_cap, int slab, hipStream_t s, char *kname,
                                    const NextRow *nx, bool *extracted, int arith, const PassAux *aux) {
    if (nwork <= 0 || npairs <= 0) return hipSuccess;
    constexpr bool kF32 = sizeof(TS) == 4;
#define EKF_DD(TT, SS) return launch_downdate_ts<TS, TT, SS>(st, dstv, work, nwork, work_xcd, xcd_len, pstart, npairs, grid_cap, s, kname, nx, extracted, arith, aux)
    if constexpr (kF32) {
        switch (st.tm.T) {
            case 16: EKF_DD(16, 16);
            case 32: EKF_DD(32, 32);
            case 64: EKF_DD(64, 64);                 // 16 lanes per row -> generic kernel
            case 128: if (npairs > 1) EKF_DD(128, 64); EKF_DD(128, 8);      // two rows per wave instruction
            case 256: if (slab == 32) EKF_DD(256, 32); if (slab == 16) EKF_DD(256, 16); if (slab == 8) EKF_DD(256, 8);
                      if (slab == 4) EKF_DD(256, 4); if (npairs > 1) EKF_DD(256, 32); EKF_DD(256, 4);
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (st.tm.T) {
            case 16: EKF_DD(16, 16);
            case 32: if (slab == 32) EKF_DD(32, 32); EKF_DD(32, 16);
            case 64: if (slab == 64) EKF_DD(64, 64); if (slab == 32) EKF_DD(64, 32); if (slab == 16) EKF_DD(64, 16);
                     if (slab == 8) EKF_DD(64, 8); if (npairs > 1) EKF_DD(64, 64); EKF_DD(64, 8);
            case 128: if (slab == 32) EKF_DD(128, 32); if (slab == 16) EKF_DD(128, 16); if (slab == 8) EKF_DD(128, 8);
                      if (slab == 4) EKF_DD(128, 4); if (npairs > 1) EKF_DD(128, 32); EKF_DD(128, 4);
            default: return hipErrorInvalidValue;
        }
    }
#undef EKF_DD
}

hipError_t launch_downdate(const DevState &st, void *dst, const int2 *work, int64_t nwork, const int2 *work_xcd, int64_t xcd_len,
                           int pstart, int npairs, int storage, int grid_cap, hipStream_t s, char *kname, const NextRow *nx,
                           bool *extracted, int arith, const PassAux *aux) {
    if (extracted) *extracted = false;
    static const int slab1 = ekf_tune_int("EKF_DOWNDATE_SLAB", 0);
    static const int slabm = ekf_tune_int("EKF_DOWNDATE_SLAB_BATCH", 0);
    const int slab = npairs > 1 ? slabm : slab1;
    return storage == 0 ? launch_downdate_t<double>(st, dst, work, nwork, work_xcd, xcd_len, pstart, npairs, grid_cap, slab, s, kname, nx, extracted, arith, aux)
                        : launch_downdate_t<float>(st, dst, work, nwork, work_xcd, xcd_len, pstart, npairs, grid_cap, slab, s, kname, nx, extracted, arith, aux);
}

// Strip work list (flush32_pipe.h): column ranges of kSeg consecutive OWNED 128-column items; within a range the 128-row slabs from the
// diagonal down, each a segment of up to kSeg items; the segments, in that order, cut into 8 equal contiguous runs (the CUs of an XCD then
// work on the same column range -- the same G -- at the same time) and interleaved run by run; padded with empty segments to a multiple of 8.
int64_t build_strip_segments(const TileMap &tm, int64_t nt, std::vector<int4> &out) {
    constexpr int L = ekf_pipe32::kSeg;
    std::vector<std::vector<int4>> segs;
    const int64_t ncj = 2 * nt, step = (int64_t)L * tm.world;            // a row owns every world-th tile of a range: ~L owned items per range
    for (int64_t c0 = 0; c0 < ncj; c0 += step)
        for (int64_t rs = 0; rs < 2 * nt; ++rs) {
            const int64_t I = rs >> 1, cmax = 2 * I + 1;
            if (cmax < c0) continue;
            std::vector<int4> sg;
            for (int64_t cj = c0; cj < c0 + step && cj <= cmax; ++cj) {
                if (!tm.mine(I, cj >> 1)) continue;
                sg.push_back(ekf_pipe32::strip_entry(tm, (int)I, (int)(cj >> 1), (int)(rs & 1), (int)(cj & 1)));
                if ((int)sg.size() == L) { segs.push_back(sg); sg.clear(); }
            }
            if (!sg.empty()) segs.push_back(sg);
        }
    const size_t ns = segs.size(), per = (ns + 7) / 8;
    out.assign(per * 8 * L, make_int4(0, 0, -1, -1));
    for (int x = 0; x < 8; ++x) {
        const size_t lo = ns * x / 8, hi = ns * (x + 1) / 8;
        for (size_t q = lo; q < hi; ++q) std::copy(segs[q].begin(), segs[q].end(), out.begin() + ((q - lo) * 8 + x) * L);
    }
    return (int64_t)(per * 8);
}

hipError_t launch_associate(const DevState &st, const AssocArgs &a, double *pos_cost, double *sig_cost,
                            AssocDecision *partial, int *ticket, AssocDecision *decision, AssocHostPartial *host_partials, int seq,
                            double *cand, int storage, hipStream_t s, const PredictArgs *fused_predict) {
    const int64_t grid = cdiv(a.N > 0 ? a.N : 1, kAssocBlock);
    PredictArgs pa = {};
    if (fused_predict) pa = *fused_predict;
#define EKF_A(TS_, PRED_) hipLaunchKernelGGL((k_associate<TS_, PRED_>), dim3((unsigned)grid), dim3(kAssocBlock), 0, s, st, a, pos_cost, sig_cost, \
                                             partial, ticket, decision, host_partials, seq, cand, pa)
    if (storage == 0) { if (fused_predict) EKF_A(double, true); else EKF_A(double, false); }
    else              { if (fused_predict) EKF_A(float, true); else EKF_A(float, false); }
#undef EKF_A
    return hipGetLastError();
}

hipError_t launch_assoc_merge(const DevState &st, const double *recv, int world, int64_t count, int64_t N, bool want_costs,
                              double *pos_cost, AssocDecision *decision, AssocDecision *host_decision, int seq, hipStream_t s) {
    int64_t grid = want_costs ? cdiv(N > 0 ? N : 1, kBlock) : 1;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_assoc_merge, dim3((unsigned)grid), dim3(kBlock), 0, s, st.tm, recv, world, count, N, want_costs ? 1 : 0,
                       pos_cost, decision, host_decision, seq);
    return hipGetLastError();
}

hipError_t launch_copy_rows(const TileMap &tm, const void *src, void *dst, int64_t r0, int64_t r1, int storage, hipStream_t s) {
    if (r1 <= r0) return hipSuccess;
    const int T = tm.T;
    if (T > kBlock * (storage == 0 ? 2 : 4)) return hipErrorInvalidValue;                    // a tile row must fit one workgroup's lanes
    for (int64_t I = r0 >> tm.shift; I <= (r1 - 1) >> tm.shift; ++I) {
        const int64_t lo = std::max<int64_t>(r0, I * T) - I * T, hi = std::min<int64_t>(r1, (I + 1) * T) - I * T;
        const int64_t slot0 = tm.row_base(I), nslots = tm.row_base(I + 1) - slot0;
        if (nslots <= 0 || hi <= lo) continue;
        const int lanes = T / (storage == 0 ? 2 : 4), per_wg = kBlock / lanes > 0 ? kBlock / lanes : 1;
        const int64_t grid = cdiv(nslots * (hi - lo), per_wg);
        EKF_STORAGE_DISPATCH(storage,
            hipLaunchKernelGGL(k_copy_tile_rows<double>, dim3((unsigned)grid), dim3(kBlock), 0, s, (const double *)src, (double *)dst, slot0, nslots, (int)lo, (int)(hi - lo), T),
            hipLaunchKernelGGL(k_copy_tile_rows<float>, dim3((unsigned)grid), dim3(kBlock), 0, s, (const float *)src, (float *)dst, slot0, nslots, (int)lo, (int)(hi - lo), T));
    }
    return hipGetLastError();
}

hipError_t launch_unpack_dense(const DevState &st, int cur, int64_t n_mm, double *dense, int storage, hipStream_t s) {
    const int64_t n = n_mm + 3;
    const int64_t grid = cdiv(n * n, kBlock);
    EKF_STORAGE_DISPATCH(storage,
        hipLaunchKernelGGL(k_unpack_dense<double>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, cur, n, dense),
        hipLaunchKernelGGL(k_unpack_dense<float>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, cur, n, dense));
    return hipGetLastError();
}

hipError_t launch_pack_dense(const DevState &st, int cur, int64_t n_mm, const double *dense, int storage, hipStream_t s) {
    const int64_t n = n_mm + 3;
    const int64_t grid = cdiv(n * n, kBlock);
    EKF_STORAGE_DISPATCH(storage,
        hipLaunchKernelGGL(k_pack_dense<double>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, cur, n, dense),
        hipLaunchKernelGGL(k_pack_dense<float>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, cur, n, dense));
    return hipGetLastError();
}

hipError_t launch_get_block(const DevState &st, int cur, int64_t r0, int64_t c0, int64_t nr, int64_t nc, double *out,
                            int storage, hipStream_t s) {
    const int64_t grid = cdiv(nr * nc, kBlock);
    EKF_STORAGE_DISPATCH(storage,
        hipLaunchKernelGGL(k_get_block<double>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, cur, r0, c0, nr, nc, out),
        hipLaunchKernelGGL(k_get_block<float>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, cur, r0, c0, nr, nc, out));
    return hipGetLastError();
}

hipError_t launch_get_diag_blocks(const DevState &st, int cur, int64_t N, double *out, int storage, hipStream_t s) {
    const int64_t grid = cdiv(4 * (N + 1), kBlock);
    EKF_STORAGE_DISPATCH(storage,
        hipLaunchKernelGGL(k_get_diag_blocks<double>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, cur, N, out),
        hipLaunchKernelGGL(k_get_diag_blocks<float>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, cur, N, out));
    return hipGetLastError();
}

hipError_t launch_lowrank(const DevState &st, int cur, int64_t n_mm, const int2 *work, int64_t nwork, const double *d,
                          const double *U, int64_t k, int storage, hipStream_t s) {
    if (nwork > 0) {
        const int64_t grid = nwork < 65536 ? nwork : 65536;
        EKF_STORAGE_DISPATCH(storage,
            hipLaunchKernelGGL(k_lowrank_tiles<double>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, n_mm, work, nwork, d, U, k),
            hipLaunchKernelGGL(k_lowrank_tiles<float>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, n_mm, work, nwork, d, U, k));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const int64_t grid = cdiv(n_mm > 0 ? n_mm : 1, kBlock);
    hipLaunchKernelGGL(k_lowrank_robot, dim3((unsigned)grid), dim3(kBlock), 0, s, st, cur, n_mm, d, U, k);
    return hipGetLastError();
}

hipError_t launch_digest(const DevState &st, int cur, int64_t n_mm, const int2 *work, int64_t nwork, double *out,
                         int storage, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out, 0, 4 * sizeof(double), s);      // sums + the ticket; partial slots follow (kDigestSlots)
    if (e != hipSuccess) return e;
    int64_t grid = nwork < kDigestGrid ? nwork : kDigestGrid;
    if (grid < 1) grid = 1;
    EKF_STORAGE_DISPATCH(storage,
        hipLaunchKernelGGL(k_digest<double>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, cur, n_mm, work, nwork, out),
        hipLaunchKernelGGL(k_digest<float>, dim3((unsigned)grid), dim3(kBlock), 0, s, st, cur, n_mm, work, nwork, out));
    return hipGetLastError();
}

