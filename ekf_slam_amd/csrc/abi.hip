// C ABI of libekfslam (include/ekfslam.h): handle, HBM buffers, launch sequencing, measure() dispatch.
// No torch types, no exceptions across the boundary.  There is NO CPU fallback: without a HIP device
// ekf_create fails with EKF_ERR_NO_DEVICE.
#include "../../include/ekfslam.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "device_math.h"
#include "flush32_pipe.h"
#include "kernels.h"
#include "layout.h"

struct KernelTimer {
    bool enabled = false;
    std::vector<hipEvent_t> ev;   // start/stop pairs
    size_t used = 0;              // events used since the last read
};

// RCCL is bound at run time (dlopen) so that single-GPU users never load it.
struct RcclApi {
    void *dl = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, ekf_comm_id, int) = nullptr;      // ncclUniqueId is 128 opaque bytes by value
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

static RcclApi g_rccl;

static bool rccl_load(std::string &err) {
    if (g_rccl.dl) return true;
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void *dl = nullptr;
    for (const char *n : names) { dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (dl) break; }
    if (!dl) { err = std::string("dlopen(librccl): ") + dlerror(); return false; }
    RcclApi a;
    a.dl = dl;
    a.GetUniqueId = (int (*)(void *))dlsym(dl, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void **, int, ekf_comm_id, int))dlsym(dl, "ncclCommInitRank");
    a.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(dl, "ncclAllGather");
    a.CommDestroy = (int (*)(void *))dlsym(dl, "ncclCommDestroy");
    a.GetErrorString = (const char *(*)(int))dlsym(dl, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy || !a.GetErrorString) {
        err = "librccl lacks a required symbol";
        return false;
    }
    g_rccl = a;
    return true;
}

struct ekf_handle {
    ekf_config cfg;
    int64_t N = 0;         // landmarks in the state (host mirror; appends are host-initiated)
    int64_t cap = 0;
    int32_t T = 64;
    int32_t storage = 0;
    int32_t cur = 0;       // which of the double buffers holds the live x / Prr / strip
    int32_t batch = 1;     // corrections per pass over P
    int32_t npend = 0;     // pending pairs a reader must apply (tiles hold P_base; live P = P_base - sum of pending K_i G_i)
    int32_t pstart = 0;    // ring slot of the oldest pending pair
    // Asynchronous flush (cfg.batch > 1, f64/f32 alike): the pass over P runs on a second stream from the current
    // tile store into the OTHER one while the next corrections keep reading the current store plus all pending
    // pairs (those being flushed, `nfrozen`, and the ones recorded since).  At the next batch boundary the stores
    // swap.  Readers of P, appends and state loads first retire the in-flight flush.
    bool async_flush = false;
    void *tilebuf[2] = { nullptr, nullptr };
    int32_t base = 0;          // tilebuf[base] == st.tiles: the store kernels read
    int32_t nfrozen = 0;       // pending pairs that belong to the in-flight flush (the oldest ones)
    bool inflight = false;
    hipStream_t flush_stream = nullptr;
    hipEvent_t ev_pairs = nullptr, ev_flushed = nullptr, ev_rows = nullptr;
    hipEvent_t ev_xchg = nullptr;    // ekf_exchange_local: this shard's copies of one exchange are done
    // lazy predict: ekf_predict only records u; the next correction folds it into its gather kernel (one launch
    // instead of two, identical arithmetic); any other consumer of x / P launches k_predict first
    bool have_pp = false;
    PredictArgs pp;
    std::vector<double> s_host;   // host mirror of the signatures (they only change through host calls)
    // (signature, landmark) sorted by signature: the signature-only decision of a large map looks at the few landmarks whose
    // signature lies within the threshold of z(3) instead of all N (the mirror's O(N) scan per observation would pace the host
    // at ~10 us per row from 10 k landmarks on).  Rebuilt lazily after bulk changes, kept up to date by appends.
    // Appends go to an unsorted TAIL that every query scans linearly and that is merged into the sorted part once it holds
    // kSortedTail entries (an insertion into the sorted vector moved ~0.8 MB per append at 50 k landmarks, on the host's
    // critical path of a streaming-append step).
    mutable std::vector<std::pair<double, int64_t>> s_sorted, s_tail;
    mutable bool s_sorted_ok = false;
    static constexpr size_t kSortedTail = 2048;
    // run-ahead throttle: the host may queue at most ~2*kThrottle update-steps ahead of the device.  Measured: the
    // first time ~150-190 launches are outstanding on a stream, one launch call blocks for 35-45 ms (the runtime
    // grows a per-queue pool); with the run-ahead bounded below that the stall never happens.
    hipEvent_t throttle_ev[2] = { nullptr, nullptr };
    bool throttle_set[2] = { false, false };
    int throttle_k = 0, since_mark = 0;
    DevState st;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // work list of owned lower-triangle tiles for the active tile rows
    int2 *d_work = nullptr;
    int64_t nwork = 0, work_rows = -1, work_cap = 0;
    // pinned staging of the work lists (refresh_work): uploads are queued on the stream with no host wait -- a stream
    // synchronisation here drains a queue that may hold a whole batch and its pass (configs[4]: a 2.4 ms bubble per new tile row)
    char *wl_stage = nullptr;
    size_t wl_stage_bytes = 0;
    hipEvent_t ev_wl = nullptr;
    bool wl_busy = false;
    // the same tiles arranged as 8 per-XCD streams of super-tiles (batched flush: keeps each XCD's K/G working set
    // inside its own 4 MiB L2); stream x is work_xcd[x * xcd_len .. ), padded with (-1,-1)
    int2 *d_work_xcd = nullptr;
    int64_t xcd_len = 0;
    // cfg.pass_arith = EKF_ARITH_F32: the strip form of the pass (flush32_pipe.h) -- its work list, the dump area
    PassAux aux = { nullptr, 0, nullptr, 0, 0, nullptr, nullptr };
    int4 *d_segs = nullptr;
    int64_t segs_cap = 0;
    AssocDecision *d_partial = nullptr, *d_decision = nullptr, *h_decision = nullptr;
    AssocDecision *h_decision_dev = nullptr;   // device-side address of the mapped h_decision (k_assoc_merge, the sharded path, writes it)
    int *d_ticket = nullptr;                   // k_associate's last-workgroup ticket (device-side consumers only)
    int32_t assoc_seq = 0;
    // k_associate's workgroups store their winners into MAPPED host memory and the host takes the arg-min: kSpecRing + 1 sets of
    // parts_stride entries (one per workgroup at capacity).  Sets 0..kSpecRing-1 form the ring of cfg.device_assoc == 2 (measure()
    // dispatches on the host mirror's decision while k_associate runs for every observation in the stream; the device's decisions
    // are VERIFIED against the host's before measure() returns); set kSpecRing serves the calls that wait for their decision.
    static constexpr int kSpecRing = 64;
    AssocHostPartial *h_parts = nullptr, *h_parts_dev = nullptr;
    int64_t parts_stride = 0;
    bool assoc_poll = true;                    // false (tuning builds, EKF_ASSOC_POLL=0): wait by stream synchronisation instead of polling the mapped entries
    struct Spec { int32_t seq, is_new, nblk; int64_t idx, idx_N; };   // idx_N: landmarks at launch (the default index of a new one)
    std::vector<Spec> spec;
    // Device-resident measure loop (cfg.device_assoc == 3, the default of EKF_MODE_UC): an observation's association decision is
    // produced AND consumed on the device (kernels.h: DevLoopArgs); the host queues the launches from its mirror's prediction of
    // the control flow (append or correct: a function of z(3) and s alone when w_pos == 0) and reads what the device decided
    // afterwards, from a ring of records in mapped memory -- verified lazily (the next ekf_measure sweeps what has landed;
    // every call that synchronises or reads state checks the rest first).
    static constexpr int kLoopRing = 256;
    AssocHostPartial *d_lparts = nullptr;      // DEVICE: 2 sets of lparts_stride per-workgroup winners
    int64_t lparts_stride = 0;
    int32_t loop_set = 0;                      // set written last
    AssocHostPartial *h_lrec = nullptr, *h_lrec_dev = nullptr;     // MAPPED: kLoopRing decision records
    struct LoopSpec { int32_t seq, is_new; int64_t idx; };
    std::vector<LoopSpec> lspec;               // predictions of records lrec_tail .. lrec_head-1 (ring positions mod kLoopRing)
    uint64_t lrec_head = 0, lrec_tail = 0;
    double *d_pos_cost = nullptr, *d_sig_cost = nullptr, *d_digest = nullptr;
    double *h_small = nullptr;   // pinned 32 doubles
    // sharded correction: exchange slabs (own allocations, or caller-provided device buffers)
    bool sharded = false;          // world > 1, or cfg.force_sharded (the sharded code path with one rank, on one GPU)
    double *own_send = nullptr, *own_recv = nullptr, *send = nullptr, *recv = nullptr;
    int64_t slab_cap = 0;          // doubles per shard slab at capacity
    int64_t xchg_cap = 0;          // doubles of the send area (the receive area holds world times as many)
    int64_t slab = 0;              // doubles per shard slab of the pending correction
    bool pending = false;          // an exchange is between begin and finish ...
    bool assoc_costs = false;      // the pending association's exchange carries the position costs too
    int pending_kind = 0;          // ... 1: one correction's row-panel, 2: a prefetch of several base row-panels, 3: association candidates
    int64_t x_count = 0;           // doubles per shard of the pending exchange
    CorrectArgs pending_args;
    // prefetched BASE row-panels (ekf_prefetch_rows): valid until the tiles change (flush) or the map grows
    bool pf_valid = false;
    int32_t pf_m = 0;
    int64_t pf_slab = 0, pf_N = 0;
    std::vector<int64_t> pf_idx;
    double *pf_store = nullptr;    // world x batch x slab_cap
    // the row-panel of landmark nx_idx, extracted by the last pass over P itself (ekf_hint_next + k_downdate_w<.., kNext>): valid while
    // the tiles, the map size and the send area stay as that pass left them and nothing is pending
    int64_t hint_idx = -1;         // ekf_hint_next: the landmark the NEXT ekf_correct will name
    int64_t inflight_N = 0;      // cfg.async_flush: landmarks when the in-flight pass was launched (it writes rows < 2 * inflight_N) ...
    bool appended_inflight = false;   // ... and whether landmarks were appended since (their rows are copied to the new store when it retires)
    int flush_cus = 0;           // cfg.async_flush with a CU-masked pass stream: the CUs that stream may use (0: the whole device)
    bool nx_valid = false;
    int64_t nx_idx = -1, nx_N = 0;
    void *comm = nullptr;          // ncclComm_t
    // ekf_prefetch_next: the landmarks of the batch AFTER the current one.  When the current batch completes, their row-panels are
    // extracted as the pass will leave them (k_rowpanel_next) in front of the pass, and the all-gather runs on xchg_stream beside it.
    std::vector<int64_t> pn_idx;
    int64_t pn_N = -1;
    hipStream_t xchg_stream = nullptr;
    hipEvent_t ev_pn_ready = nullptr, ev_pn_done = nullptr;
    int32_t (*xhook)(void *) = nullptr;   // ekf_exchange_set_hook: the caller's all-gather, called where the library-owned one would run
    void *xhook_ctx = nullptr;
    KernelTimer timers[EKF_KERNEL_COUNT];
    std::vector<void *> allocs;
    int64_t bytes = 0;
    int grid_cap = 0;
    char dd_kernel[64] = "";       // kernel instance of the last downdate / flush launch (ekf_downdate_kernel_name)
    int32_t dd_pairs = 0;          // pairs it applied
    std::string err;
};

namespace {

int32_t fail(ekf_handle *h, int32_t status, const char *what, hipError_t e = hipSuccess) {
    if (h) {
        h->err = what ? what : "";
        if (e != hipSuccess) { h->err += ": "; h->err += hipGetErrorString(e); }
    }
    return status;
}

#define HIPCHK(h, call)                                                      \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return fail((h), EKF_ERR_HIP, #call, e_);      \
    } while (0)

#define REQUIRE(h, cond, status, msg)                                        \
    do { if (!(cond)) return fail((h), (status), (msg)); } while (0)

template <typename Tp>
hipError_t dalloc(ekf_handle *h, Tp **p, size_t count) {
    void *q = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(Tp);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return e;
    e = hipMemset(q, 0, bytes);
    if (e != hipSuccess) return e;
    h->allocs.push_back(q);
    h->bytes += (int64_t)bytes;
    *p = (Tp *)q;
    return hipSuccess;
}

inline int64_t n_mm(const ekf_handle *h) { return 2 * h->N; }

// Pair slots must read as zero beyond the active columns (the pass kernels read whole tile-wide slices of K and G): whenever the
// map shrinks or the state is replaced, every ring is cleared -- the F64 pairs AND their float copies (cfg.pass_arith = EKF_ARITH_F32).
hipError_t clear_pairs(ekf_handle *h) {
    const size_t elems = (size_t)h->st.pair_stride * h->st.pcap * 2;       // G ring, then K ring: one allocation each
    hipError_t e = hipMemsetAsync(h->st.Gp, 0, elems * 8, h->stream);
    if (e == hipSuccess && h->st.Gp32) e = hipMemsetAsync(h->st.Gp32, 0, elems * 4, h->stream);
    return e;
}
inline size_t elt_size(const ekf_handle *h) { return h->storage == EKF_STORE_F64 ? 8 : 4; }

int32_t use_device(ekf_handle *h) {
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return EKF_OK;
}

// (re)build the list of owned tiles for the active tile rows.  The lists are built in pinned memory and uploaded by asynchronous
// copies in stream order (the kernels that read them follow on the same stream); the staging area is reused only after the event
// behind the previous upload has passed.
int32_t refresh_work(ekf_handle *h) {
    const int64_t nt = ekf_tiles_for(n_mm(h), h->T);
    if (nt == h->work_rows) return EKF_OK;
    const size_t b_work = (size_t)h->work_cap * sizeof(int2), b_xcd = 8 * b_work, b_segs = (size_t)h->segs_cap * sizeof(int4);
    if (!h->wl_stage) {
        h->wl_stage_bytes = b_work + b_xcd + b_segs;
        HIPCHK(h, hipHostMalloc((void **)&h->wl_stage, h->wl_stage_bytes ? h->wl_stage_bytes : 16, hipHostMallocDefault));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_wl, hipEventDisableTiming));
    }
    if (h->wl_busy) { HIPCHK(h, hipEventSynchronize(h->ev_wl)); h->wl_busy = false; }
    int2 *w = reinterpret_cast<int2 *>(h->wl_stage);
    int2 *flat = reinterpret_cast<int2 *>(h->wl_stage + b_work);
    int4 *segs = reinterpret_cast<int4 *>(h->wl_stage + b_work + b_xcd);
    size_t nw = 0;
    REQUIRE(h, h->st.tm.slots_for_rows(nt) <= h->work_cap, EKF_ERR_STATE, "work list overflow");
    for (int64_t I = 0; I < nt; ++I)
        for (int64_t J = 0; J <= I; ++J)
            if (h->st.tm.mine(I, J)) w[nw++] = make_int2((int)I, (int)J);
    if (nw) HIPCHK(h, hipMemcpyAsync(h->d_work, w, nw * sizeof(int2), hipMemcpyHostToDevice, h->stream));
    h->nwork = (int64_t)nw;
    h->work_rows = nt;

    // per-XCD streams: super-tiles of S x S tiles, largest first onto the least loaded stream
    static const int S = std::max(1, ekf_tune_int("EKF_SUPERTILE", 8));
    struct Super { int64_t si, sj; std::vector<int2> tiles; };
    std::vector<Super> supers;
    const int64_t ns = (nt + S - 1) / S;
    for (int64_t si = 0; si < ns; ++si)
        for (int64_t sj = 0; sj <= si; ++sj) {
            Super sp; sp.si = si; sp.sj = sj;
            for (int64_t I = si * S; I < nt && I < (si + 1) * S; ++I)
                for (int64_t J = sj * S; J <= I && J < (sj + 1) * S; ++J)
                    if (h->st.tm.mine(I, J)) sp.tiles.push_back(make_int2((int)I, (int)J));
            if (!sp.tiles.empty()) supers.push_back(std::move(sp));
        }
    // Order of the streams.  1 (default): the super-tiles in row-major order (si, then sj), flattened tile by tile and cut into 8
    // equal contiguous runs -- an XCD walks along a band of S tile rows, so the band's K slice (S x 64 KiB at 32 pairs) stays in
    // its L2 for the whole band and only the G slice changes from one super-tile to the next; runs are equal to within one tile.
    // 0: round 1's schedule (largest super-tile first onto the least loaded stream): every super-tile fetched both slices anew
    // and the streams differed by up to a super-tile (profiles/round2_tuning.md).
    static const int order = ekf_tune_int("EKF_XCD_ORDER", 1);
    std::vector<int2> stream[8];
    if (order == 0) {
        std::stable_sort(supers.begin(), supers.end(), [](const Super &a, const Super &b) { return a.tiles.size() > b.tiles.size(); });
        for (const Super &sp : supers) {
            int best = 0;
            for (int x = 1; x < 8; ++x) if (stream[x].size() < stream[best].size()) best = x;
            stream[best].insert(stream[best].end(), sp.tiles.begin(), sp.tiles.end());
        }
    } else {
        std::vector<int2> flat_order;
        flat_order.reserve(nw);
        for (const Super &sp : supers) flat_order.insert(flat_order.end(), sp.tiles.begin(), sp.tiles.end());
        const size_t tot = flat_order.size();
        for (int x = 0; x < 8; ++x)
            stream[x].assign(flat_order.begin() + (tot * x) / 8, flat_order.begin() + (tot * (x + 1)) / 8);
    }
    size_t len = 0;
    for (int x = 0; x < 8; ++x) len = std::max(len, stream[x].size());
    REQUIRE(h, (int64_t)(8 * len) <= 8 * h->work_cap, EKF_ERR_STATE, "XCD work list overflow");
    std::fill(flat, flat + 8 * len, make_int2(-1, -1));
    for (int x = 0; x < 8; ++x) std::copy(stream[x].begin(), stream[x].end(), flat + x * len);
    if (len) HIPCHK(h, hipMemcpyAsync(h->d_work_xcd, flat, 8 * len * sizeof(int2), hipMemcpyHostToDevice, h->stream));
    h->xcd_len = (int64_t)len;
    if (h->d_segs) {                                  // the strip work list of the same tiles
        std::vector<int4> sg;
        const int64_t nsegs = build_strip_segments(h->st.tm, nt, sg);
        REQUIRE(h, (int64_t)sg.size() <= h->segs_cap, EKF_ERR_STATE, "strip work list overflow");
        if (!sg.empty()) {
            std::copy(sg.begin(), sg.end(), segs);
            HIPCHK(h, hipMemcpyAsync(h->d_segs, segs, sg.size() * sizeof(int4), hipMemcpyHostToDevice, h->stream));
        }
        h->aux.segs = h->d_segs;
        h->aux.nsegs = nsegs;
        h->aux.cols = nt * h->T;
    }
    HIPCHK(h, hipEventRecord(h->ev_wl, h->stream));
    h->wl_busy = true;
    return EKF_OK;
}

struct TimedLaunch {
    ekf_handle *h;
    KernelTimer *t;
    hipEvent_t stop = nullptr;
    TimedLaunch(ekf_handle *h_, int which) : h(h_), t(&h_->timers[which]) {
        if (!t->enabled) { t = nullptr; return; }
        if (t->used + 2 > t->ev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { t = nullptr; return; }
            t->ev.push_back(a); t->ev.push_back(b);
        }
        hipEventRecord(t->ev[t->used], h->stream);
        stop = t->ev[t->used + 1];
        t->used += 2;
    }
    ~TimedLaunch() { if (t) hipEventRecord(stop, h->stream); }
};

void colmajor2(const double R[4], double &r00, double &r01, double &r10, double &r11) {
    r00 = R[0]; r10 = R[1]; r01 = R[2]; r11 = R[3];
}

// The in-flight asynchronous flush becomes visible: later kernels on the main stream wait for it, the stores swap,
// its pairs leave the pending list.
int32_t retire_inflight(ekf_handle *h) {
    if (!h->inflight) return EKF_OK;
    if (h->appended_inflight) {
        // Landmarks appended beside the pass (do_append) sit in the old store only: their rows go to the new one behind the pass, ON THE PASS'S
        // STREAM -- the next pass follows in that stream's order (it does not wait for the main stream beyond ev_pairs) and must find them.
        // The copy waits for the appends (main stream, all issued by now); the main stream then waits for the copy instead of the pass.
        if (!h->ev_rows) HIPCHK(h, hipEventCreateWithFlags(&h->ev_rows, hipEventDisableTiming));
        HIPCHK(h, hipEventRecord(h->ev_rows, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->flush_stream, h->ev_rows, 0));
        HIPCHK(h, launch_copy_rows(h->st.tm, h->tilebuf[h->base], h->tilebuf[h->base ^ 1], 2 * h->inflight_N, 2 * h->N, h->storage, h->flush_stream));
        HIPCHK(h, hipEventRecord(h->ev_flushed, h->flush_stream));
        h->appended_inflight = false;
    }
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_flushed, 0));
    h->base ^= 1;
    h->st.tiles = h->tilebuf[h->base];
    h->pstart = (h->pstart + h->nfrozen) % h->st.pcap;
    h->npend -= h->nfrozen;
    h->nfrozen = 0;
    h->inflight = false;
    h->pf_valid = false;       // prefetched row-panels were base values of the old store
    h->nx_valid = false;
    return EKF_OK;
}

// Every other pass over P walks its work list backwards (TileMap::reverse, read by the pass kernels only): what one pass wrote
// last the next one reads first, out of the Infinity Cache -- 4 % off the pass at 10 k landmarks (1.6 GB of tiles), 10 % at
// 5 k (400 MB).  A store that fits the cache whole is resident either way and measured 1.5 % faster walked forwards, so the
// direction only alternates above kCacheBytes.  cfg.pass_direction = 1 / 2 forces never / always.
void next_pass_direction(ekf_handle *h) {
    const int force = h->cfg.pass_direction == 1 ? 0 : h->cfg.pass_direction == 2 ? 1 : -1;
    constexpr int64_t kCacheBytes = 256ll << 20;
    const int64_t nt = ekf_tiles_for(n_mm(h), h->T);
    const int64_t store = nt * (nt + 1) / 2 / std::max(1, h->cfg.world) * (int64_t)h->T * h->T * (h->storage == EKF_STORE_F64 ? 8 : 4);
    const bool alternate = force >= 0 ? force != 0 : store > kCacheBytes;
    h->st.tm.reverse = alternate ? (h->st.tm.reverse ^ 1) : 0;
}

int64_t slab_for(const ekf_handle *h, int64_t mm_rows) {
    const int64_t nt = ekf_tiles_for(mm_rows, h->T);
    const int64_t cmax = (nt + h->cfg.world - 1) / h->cfg.world;
    return cmax * h->T * 2;
}

// Where a correction's row-panel is extracted to.  With the library's own communicator and its own buffers: straight into this
// rank's segment of the receive area -- the all-gather is then IN PLACE (sendbuff == recvbuff + rank * count): no local copy inside
// the collective, and with one rank nothing at all.  Caller-provided buffers / a host-run exchange keep the separate send area.
double *corr_send(const ekf_handle *h, int64_t slab) {
    return (h->comm && h->send == h->own_send && h->recv == h->own_recv) ? h->recv + (size_t)h->cfg.rank * (size_t)slab : h->send;
}

// apply ALL pending pairs to the tiles now, in place on the main stream: ONE pass over P for npend update-steps
int32_t exchange_rccl(ekf_handle *h);
int32_t flush_pending(ekf_handle *h, bool batch_done = false) {
    int32_t rc = retire_inflight(h);
    if (rc) return rc;
    if (h->npend == 0) return EKF_OK;
    rc = refresh_work(h);
    if (rc) return rc;
    next_pass_direction(h);
    bool extracted = false;
    const int64_t hint = h->hint_idx;
    // ekf_prefetch_next: the next batch's row-panels, as THIS pass will leave them, are extracted now; their all-gather runs beside the pass
    bool pn = false, pn_side = false;
    if (batch_done && !h->pn_idx.empty()) {
        if (h->sharded && !h->pending && h->pn_N == h->N && (h->comm || h->xhook)) {
            const int32_t m = (int32_t)h->pn_idx.size();
            const int64_t slab = slab_for(h, n_mm(h));
            {
                TimedLaunch tl(h, EKF_KERNEL_ROWPANEL);
                HIPCHK(h, launch_rowpanel_next(h->st, h->pn_idx.data(), m, n_mm(h), h->pstart, h->npend, h->send, slab, h->storage, h->stream));
            }
            h->pf_valid = false; h->nx_valid = false;
            h->pf_idx = h->pn_idx; h->pf_m = m; h->pf_slab = slab; h->pf_N = h->N;
            h->x_count = (int64_t)m * slab;
            // Where the all-gather runs.  On the handle's stream, in front of the pass: what ships.  On a stream of its own BESIDE the
            // pass (tuning builds, EKF_PN_SIDE_STREAM=1): built, bit-identical, and on one GPU twice as slow per update-step -- with a
            // second stream in use every dispatch of the main stream costs ~50 us more on this runtime (the same finding as
            // cfg.async_flush, profiles/round2_tuning.md 21-22; round4_tuning.md 49).  To be measured again where the all-gather
            // crosses xGMI and is long enough to be worth hiding.
            static const int pn_side_stream = ekf_tune_int("EKF_PN_SIDE_STREAM", 0);
            if (h->comm && !pn_side_stream) {
                const int r = g_rccl.AllGather(h->send, h->recv, (size_t)h->x_count, /*ncclDouble*/ 8, h->comm, h->stream);
                if (r != 0) return fail(h, EKF_ERR_COMM, g_rccl.GetErrorString(r));
                HIPCHK(h, hipMemcpyAsync(h->pf_store, h->recv, (size_t)h->x_count * h->cfg.world * sizeof(double), hipMemcpyDeviceToDevice,
                                         h->stream));
            } else if (h->comm) {
                if (!h->xchg_stream) {
                    HIPCHK(h, hipStreamCreateWithFlags(&h->xchg_stream, hipStreamNonBlocking));
                    HIPCHK(h, hipEventCreateWithFlags(&h->ev_pn_ready, hipEventDisableTiming));
                    HIPCHK(h, hipEventCreateWithFlags(&h->ev_pn_done, hipEventDisableTiming));
                }
                HIPCHK(h, hipEventRecord(h->ev_pn_ready, h->stream));
                HIPCHK(h, hipStreamWaitEvent(h->xchg_stream, h->ev_pn_ready, 0));
                const int r = g_rccl.AllGather(h->send, h->recv, (size_t)h->x_count, /*ncclDouble*/ 8, h->comm, h->xchg_stream);
                if (r != 0) return fail(h, EKF_ERR_COMM, g_rccl.GetErrorString(r));
                HIPCHK(h, hipMemcpyAsync(h->pf_store, h->recv, (size_t)h->x_count * h->cfg.world * sizeof(double), hipMemcpyDeviceToDevice,
                                         h->xchg_stream));
                HIPCHK(h, hipEventRecord(h->ev_pn_done, h->xchg_stream));
                pn_side = true;
            } else {
                // transport (d): the caller's all-gather runs on the host's schedule, i.e. in front of the pass
                h->pending = true; h->pending_kind = 2;
                rc = exchange_rccl(h);
                h->pending = false; h->pending_kind = 0;
                if (rc) return rc;
                HIPCHK(h, hipMemcpyAsync(h->pf_store, h->recv, (size_t)h->x_count * h->cfg.world * sizeof(double), hipMemcpyDeviceToDevice,
                                         h->stream));
            }
            pn = true;
        }
        h->pn_idx.clear();
    }
    {
        // a sharded handle that was told which landmark the next correction names lets this pass extract that row-panel
        NextRow nx = { -1, nullptr };
        if (h->sharded && h->npend == 1 && !h->pending && hint >= 0 && hint < h->N) {
            nx.j = 2 * hint;
            nx.send = corr_send(h, slab_for(h, n_mm(h)));
        }
        TimedLaunch tl(h, EKF_KERNEL_DOWNDATE);
        HIPCHK(h, launch_downdate(h->st, h->st.tiles, h->d_work, h->nwork, h->d_work_xcd, h->xcd_len, h->pstart, h->npend,
                                  h->storage, h->grid_cap, h->stream, h->dd_kernel, nx.j >= 0 ? &nx : nullptr, &extracted, h->cfg.pass_arith,
                                  h->d_segs ? &h->aux : nullptr));
        h->dd_pairs = h->npend;
    }
    h->npend = 0;
    h->pstart = 0;
    h->pf_valid = pn;          // prefetched row-panels were base values of the old tiles -- unless they were extracted as this pass leaves them
    if (pn_side) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_pn_done, 0));      // whatever follows the pass may read them (and reuse the exchange areas)
    h->nx_valid = extracted && !pn;
    if (extracted) { h->nx_idx = hint; h->nx_N = h->N; }
    return EKF_OK;
}

// a batch is complete: start its pass over P.  Synchronous engines do it in place; asynchronous ones launch it on the
// flush stream into the other tile store and keep going.
int32_t batch_complete(ekf_handle *h) {
    if (!h->async_flush) return flush_pending(h, /*batch_done*/ true);
    // Recorded BEFORE the main stream is made to wait for the previous pass (retire_inflight): every pair of this batch has been
    // written and every reader of the store this pass overwrites is queued in front of it -- that is all the new pass depends on
    // (the previous pass precedes it in the flush stream's own order).  Recording it after that wait would chain the passes
    // through two cross-stream hand-overs per batch (previous pass -> main stream -> this pass): ~30 us per update-step at batch 1.
    HIPCHK(h, hipEventRecord(h->ev_pairs, h->stream));
    int32_t rc = retire_inflight(h);              // at most one flush in flight; later main-stream kernels read its output
    if (rc) return rc;
    rc = refresh_work(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamWaitEvent(h->flush_stream, h->ev_pairs, 0));
    {
        KernelTimer *t = &h->timers[EKF_KERNEL_DOWNDATE];
        hipEvent_t stop = nullptr;
        if (t->enabled) {
            if (t->used + 2 > t->ev.size()) {
                hipEvent_t a, b;
                HIPCHK(h, hipEventCreate(&a)); HIPCHK(h, hipEventCreate(&b));
                t->ev.push_back(a); t->ev.push_back(b);
            }
            HIPCHK(h, hipEventRecord(t->ev[t->used], h->flush_stream));
            stop = t->ev[t->used + 1];
            t->used += 2;
        }
        next_pass_direction(h);
        HIPCHK(h, launch_downdate(h->st, h->tilebuf[h->base ^ 1], h->d_work, h->nwork, h->d_work_xcd, h->xcd_len, h->pstart,
                                  h->npend, h->storage, h->grid_cap, h->flush_stream, h->dd_kernel, nullptr, nullptr, h->cfg.pass_arith,
                                  h->d_segs ? &h->aux : nullptr));
        h->dd_pairs = h->npend;
        if (stop) HIPCHK(h, hipEventRecord(stop, h->flush_stream));
    }
    HIPCHK(h, hipEventRecord(h->ev_flushed, h->flush_stream));
    h->nfrozen = h->npend;
    h->inflight = true;
    h->inflight_N = h->N;
    h->appended_inflight = false;
    return EKF_OK;
}

// one snapshot of a self-validating 16-byte entry (kernels.h: AssocHostPartial): payload, and the launch number its tag stands
// for GIVEN that payload
struct PartView { double ll; int32_t index; int32_t seq; };
inline PartView read_part(const volatile AssocHostPartial *e) {
    const volatile uint64_t *w = reinterpret_cast<const volatile uint64_t *>(e);
    const uint64_t lo = w[0], hi = w[1];
    PartView v;
    memcpy(&v.ll, &lo, 8);
    v.index = (int32_t)(uint32_t)(hi & 0xffffffffu);
    v.seq = (int32_t)((uint32_t)(hi >> 32) - assoc_part_mix((uint32_t)(lo & 0xffffffffu), (uint32_t)(lo >> 32), (uint32_t)v.index));
    return v;
}

// Device-resident measure loop: compare what the device decided (records in mapped memory) with what the host predicted when it
// queued the launches.  block == false: only the records that have landed; block == true: all of them (the stream is synchronised
// if the newest has not landed within the polling bound).
int32_t verify_loop(ekf_handle *h, bool block) {
    if (h->lrec_tail == h->lrec_head) return EKF_OK;
    if (block) {
        const volatile AssocHostPartial *newest = h->h_lrec + (h->lrec_head - 1) % ekf_handle::kLoopRing;
        const int32_t want = h->lspec.back().seq;
        bool landed = false;
        for (int spin = 0; spin < 200000 && !landed; ++spin) { landed = read_part(newest).seq == want; if (!landed) __builtin_ia32_pause(); }
        if (!landed) HIPCHK(h, hipStreamSynchronize(h->stream));
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    size_t done = 0;
    int32_t rc = EKF_OK;
    for (; h->lrec_tail < h->lrec_head; ++h->lrec_tail, ++done) {
        const ekf_handle::LoopSpec &sp = h->lspec[done];
        const PartView v = read_part(h->h_lrec + h->lrec_tail % ekf_handle::kLoopRing);
        if (v.seq != sp.seq) {
            if (!block) break;                                          // not there yet (records land in stream order)
            rc = fail(h, EKF_ERR_STATE, "measure: a decision record of the device-resident loop is missing");
            continue;
        }
        const bool same = sp.is_new ? v.index == -1 : (int64_t)v.index == sp.idx;
        if (!same) {
            char buf[200];
            snprintf(buf, sizeof buf, "measure: the device association decided %s %d where the host mirror of the signatures "
                     "predicted %s %lld; the state is no longer the reference's", v.index == -1 ? "new landmark" : v.index == -2 ?
                     "(stale winner entries)" : "landmark", (int)v.index, sp.is_new ? "new landmark" : "landmark", (long long)sp.idx);
            rc = fail(h, EKF_ERR_STATE, buf);
        }
    }
    h->lspec.erase(h->lspec.begin(), h->lspec.begin() + (ptrdiff_t)done);
    return rc;
}

int32_t materialize_predict(ekf_handle *h) {
    if (!h->have_pp) return EKF_OK;
    h->have_pp = false;
    PredictArgs a = h->pp;
    a.n_mm = n_mm(h); a.cur = h->cur;
    {
        TimedLaunch tl(h, EKF_KERNEL_PREDICT);
        HIPCHK(h, launch_predict(h->st, a, h->storage, h->stream));
    }
    h->cur ^= 1;
    return EKF_OK;
}

int32_t do_predict(ekf_handle *h, const double u[2]) {
    int32_t rc = materialize_predict(h);           // an earlier predict that nothing consumed yet
    if (rc) return rc;
    h->pp.u0 = u[0]; h->pp.u1 = u[1]; h->pp.C = h->cfg.C; h->pp.n_mm = 0; h->pp.cur = 0;
    h->have_pp = true;
    static const bool lazy = ekf_tune_int("EKF_LAZY_PREDICT", 1) != 0;
    return lazy ? EKF_OK : materialize_predict(h);
}

// device + every deferred host-side decision that the caller's next read depends on
int32_t enter(ekf_handle *h) {
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int32_t rc = verify_loop(h, /*block*/ true);
    if (rc) return rc;
    return materialize_predict(h);
}

int32_t do_append(ekf_handle *h, const double u[2], const double R[4], const double pos[2], double signature,
                  const DevLoopArgs *dl = nullptr) {
    REQUIRE(h, h->N < h->cap, EKF_ERR_CAPACITY, "append: capacity_landmarks exhausted");
    // A pass in flight (cfg.async_flush) is not waited for: the new rows are written to the store the main stream reads (the one the pass
    // reads too) and copied to the pass's output when it retires -- the pass leaves rows that did not exist at its launch as they are (its
    // pairs have K = 0 there), and k_append writes nothing but the new landmark's own two rows of the tiles.
    if (h->inflight) h->appended_inflight = true;
    AppendArgs a;
    a.u0 = u[0]; a.u1 = u[1];
    colmajor2(R, a.R00, a.R01, a.R10, a.R11);
    a.pos0 = pos[0]; a.pos1 = pos[1]; a.signature = signature; a.N = h->N; a.cur = h->cur;
    {
        // a recorded predict is carried out by the append launch itself (k_append<.., kPredict>): the state moves to the other buffer
        PredictArgs pa = h->pp;
        pa.n_mm = n_mm(h); pa.cur = h->cur;
        const PredictArgs *fuse = h->have_pp ? &pa : nullptr;
        TimedLaunch tl(h, EKF_KERNEL_APPEND);
        HIPCHK(h, launch_append(h->st, a, h->storage, h->stream, dl, fuse));
        if (fuse) { h->have_pp = false; h->cur ^= 1; }
    }
    if ((int64_t)h->s_host.size() > h->N) { h->s_host.resize((size_t)h->N); h->s_sorted_ok = false; }
    h->s_host.push_back(signature);
    if (h->s_sorted_ok && signature == signature)         // the index learns of it through its unsorted tail; NaN never matches anything
        h->s_tail.emplace_back(signature, h->N);
    h->N += 1;
    h->pf_valid = false;
    h->nx_valid = false;
    h->pn_idx.clear();         // (an announced prefetch spoke of the map before it grew)
    return EKF_OK;
}

// the buffer the PENDING exchange's contribution sits in (what a caller-run all-gather must send)
double *pending_send(const ekf_handle *h) {
    return (h->pending && h->pending_kind == 1) ? corr_send(h, h->x_count) : h->send;
}

constexpr int kThrottle = 48;

// run-ahead throttle (see ekf_handle::throttle_ev): called once per update-step
int32_t throttle_step(ekf_handle *h) {
    if (++h->since_mark >= kThrottle) {
        h->since_mark = 0;
        const int k = h->throttle_k;
        if (!h->throttle_ev[k]) HIPCHK(h, hipEventCreateWithFlags(&h->throttle_ev[k], hipEventDisableTiming));
        HIPCHK(h, hipEventRecord(h->throttle_ev[k], h->stream));
        h->throttle_set[k] = true;
        if (h->throttle_set[k ^ 1]) HIPCHK(h, hipEventSynchronize(h->throttle_ev[k ^ 1]));   // the mark before this one
        h->throttle_k = k ^ 1;
    }
    return EKF_OK;
}

int32_t finish_step(ekf_handle *h) {
    h->cur ^= 1;
    h->st.dcur ^= 1;           // the gather wrote the diagonal blocks' live copies, with its pair applied, to the other buffer
    h->npend += 1;
    const int32_t rc = (h->npend - h->nfrozen) >= h->batch ? batch_complete(h) : EKF_OK;
    h->hint_idx = -1;          // a hint speaks of the correction that follows THIS one only
    if (rc) return rc;
    return throttle_step(h);
}

void fill_correct_args(ekf_handle *h, CorrectArgs &a, const double z[2], const double R[4], int64_t idx) {
    a.z0 = z[0]; a.z1 = z[1];
    colmajor2(R, a.R00, a.R01, a.R10, a.R11);
    a.j = 2 * idx; a.n_mm = n_mm(h); a.cur = h->cur; a.npend = h->npend; a.pstart = h->pstart;
}

// slot of landmark idx among the prefetched base row-panels, or -1
int prefetch_slot(const ekf_handle *h, int64_t idx) {
    if (!h->pf_valid || h->pf_N != h->N) return -1;
    for (int q = 0; q < h->pf_m; ++q) if (h->pf_idx[(size_t)q] == idx) return q;
    return -1;
}

int32_t correct_begin(ekf_handle *h, const double z[2], const double R[4], int64_t idx) {
    REQUIRE(h, idx >= 0 && idx < h->N, EKF_ERR_INDEX, "correct: landmark index outside the state");
    REQUIRE(h, !h->pending, EKF_ERR_STATE, "correct_begin: an exchange is already pending");
    int32_t rc = refresh_work(h);
    if (rc) return rc;
    fill_correct_args(h, h->pending_args, z, R, idx);
    h->slab = slab_for(h, h->pending_args.n_mm);
    if (h->nx_valid && h->nx_idx == idx && h->nx_N == h->N && h->npend == 0) {
        // the last pass over P left this row-panel in the send area (ekf_hint_next): nothing to extract
    } else {
        TimedLaunch tl(h, EKF_KERNEL_ROWPANEL);
        HIPCHK(h, launch_rowpanel(h->st, h->pending_args.j, h->pending_args.n_mm, h->pstart, h->npend, corr_send(h, h->slab), h->storage,
                                  h->stream));
    }
    h->nx_valid = false;
    h->pending = true; h->pending_kind = 1; h->x_count = h->slab;
    return EKF_OK;
}

int32_t correct_finish(ekf_handle *h) {
    REQUIRE(h, h->pending && h->pending_kind == 1, EKF_ERR_STATE, "correct_finish: no correction pending");
    h->pending = false; h->pending_kind = 0;
    {
        TimedLaunch tl(h, EKF_KERNEL_GATHER);
        const PredictArgs *fuse = h->have_pp ? &h->pp : nullptr;
        HIPCHK(h, launch_gather_sharded(h->st, h->pending_args, fuse, h->recv, h->slab, 0, /*patched*/ true, h->storage,
                                        h->stream));
        h->have_pp = false;
    }
    return finish_step(h);
}

// copy the BASE row-panels (no pending pairs applied: they are applied at correction time) of m landmarks into
// the send buffer; after the all-gather the corrections on these landmarks need no exchange of their own
int32_t prefetch_begin(ekf_handle *h, const int64_t *idx, int32_t m) {
    REQUIRE(h, !h->pending, EKF_ERR_STATE, "prefetch_begin: an exchange is already pending");
    REQUIRE(h, m >= 1 && m <= h->batch, EKF_ERR_INVALID_ARG, "prefetch: between 1 and cfg.batch landmarks");
    for (int32_t q = 0; q < m; ++q)
        REQUIRE(h, idx[q] >= 0 && idx[q] < h->N, EKF_ERR_INDEX, "prefetch: landmark index outside the state");
    const int64_t slab = slab_for(h, n_mm(h));
    {
        TimedLaunch tl(h, EKF_KERNEL_ROWPANEL);
        HIPCHK(h, launch_rowpanel_base(h->st, idx, m, n_mm(h), h->send, slab, h->storage, h->stream));
    }
    h->pf_valid = false;
    h->nx_valid = false;
    h->nx_valid = false;       // (a prefetch's all-gather overwrites the receive area the extracted panel sits in)
    h->pf_idx.assign(idx, idx + m);
    h->pf_m = m; h->pf_slab = slab; h->pf_N = h->N;
    h->pending = true; h->pending_kind = 2; h->x_count = (int64_t)m * slab;
    return EKF_OK;
}

int32_t prefetch_finish(ekf_handle *h) {
    REQUIRE(h, h->pending && h->pending_kind == 2, EKF_ERR_STATE, "prefetch_finish: no prefetch pending");
    h->pending = false; h->pending_kind = 0;
    HIPCHK(h, hipMemcpyAsync(h->pf_store, h->recv, (size_t)h->x_count * h->cfg.world * sizeof(double),
                             hipMemcpyDeviceToDevice, h->stream));
    h->pf_valid = true;
    return EKF_OK;
}

int32_t exchange_rccl(ekf_handle *h) {
    if (h->comm == nullptr && h->xhook != nullptr) {
        // transport (d): the caller moves the pending contribution (ekf_exchange_info), between begin and finish as for (b) / (c)
        const int32_t rc = h->xhook(h->xhook_ctx);
        return rc == EKF_OK ? EKF_OK : fail(h, EKF_ERR_COMM, "the exchange hook reported a failure");
    }
    REQUIRE(h, h->comm != nullptr, EKF_ERR_STATE,
            "sharded handle without a communicator: call ekf_comm_init, or drive the begin / your own all-gather / "
            "finish calls");
    const double *src = pending_send(h);
    TimedLaunch tl(h, EKF_KERNEL_EXCHANGE);
    const int r = g_rccl.AllGather(src, h->recv, (size_t)h->x_count, /*ncclDouble*/ 8, h->comm, h->stream);
    if (r != 0) { h->pending = false; return fail(h, EKF_ERR_COMM, g_rccl.GetErrorString(r)); }
    return EKF_OK;
}

int32_t do_correct(ekf_handle *h, const double z[2], const double R[4], int64_t idx) {
    if (h->sharded) {
        REQUIRE(h, idx >= 0 && idx < h->N, EKF_ERR_INDEX, "correct: landmark index outside the state");
        const int q = prefetch_slot(h, idx);
        if (q >= 0 && slab_for(h, n_mm(h)) == h->pf_slab) {
            // base row-panel already on every shard: no exchange, pending pairs applied inside the gather
            REQUIRE(h, !h->pending, EKF_ERR_STATE, "correct: an exchange is pending");
            CorrectArgs a;
            fill_correct_args(h, a, z, R, idx);
            {
                TimedLaunch tl(h, EKF_KERNEL_GATHER);
                const PredictArgs *fuse = h->have_pp ? &h->pp : nullptr;
                HIPCHK(h, launch_gather_sharded(h->st, a, fuse, h->pf_store, (int64_t)h->pf_m * h->pf_slab, (int64_t)q * h->pf_slab,
                                                /*patched*/ false, h->storage, h->stream));
                h->have_pp = false;
            }
            return finish_step(h);
        }
        int32_t rc = correct_begin(h, z, R, idx);
        if (rc) return rc;
        rc = exchange_rccl(h);
        if (rc) { h->pending = false; return rc; }
        return correct_finish(h);
    }
    REQUIRE(h, idx >= 0 && idx < h->N, EKF_ERR_INDEX, "correct: landmark index outside the state");
    int32_t rc = refresh_work(h);
    if (rc) return rc;
    CorrectArgs a;
    fill_correct_args(h, a, z, R, idx);
    // small maps (one workgroup covers every column), every correction rewriting P at once: the downdate runs inside the gather
    // kernel -- one launch per update-step instead of two
    static const bool fuse_small = ekf_tune_int("EKF_FUSE_SMALL", 1) != 0;
    const bool fused = fuse_small && h->batch == 1 && !h->async_flush && h->npend == 0 && a.n_mm <= gather_fuse_max_rows() &&
                       ekf_tiles_for(a.n_mm, h->T) * h->T <= 256;
    {
        TimedLaunch tl(h, EKF_KERNEL_GATHER);
        const PredictArgs *fuse = h->have_pp ? &h->pp : nullptr;
        HIPCHK(h, launch_gather(h->st, a, fuse, h->storage, h->stream, fused));
        h->have_pp = false;
    }
    if (fused) {                     // the pair never became pending: nothing to flush, only the double buffers flip
        h->cur ^= 1;
        h->st.dcur ^= 1;
        snprintf(h->dd_kernel, sizeof h->dd_kernel, "k_gather<%s,fused downdate>", h->storage == EKF_STORE_F64 ? "double" : "float");
        h->dd_pairs = 1;
        return throttle_step(h);
    }
    return finish_step(h);
}

// device-resident measure loop: the correction of the landmark the DEVICE's association names (dl.parts_in); idx, the host
// mirror's prediction, only keeps a launch whose winners name nothing inside the state.  Never the small-map fused form.
int32_t do_correct_dev(ekf_handle *h, const double z[2], const double R[4], int64_t idx, const DevLoopArgs &dl) {
    REQUIRE(h, idx >= 0 && idx < h->N, EKF_ERR_INDEX, "correct: landmark index outside the state");
    int32_t rc = refresh_work(h);
    if (rc) return rc;
    CorrectArgs a;
    fill_correct_args(h, a, z, R, idx);
    if (h->sharded) {
        // a shard: extraction of the row-panel of the landmark the DEVICE names (every shard holds the same winners: the association
        // runs on replicated data -- x, s, the strip, the live diagonal blocks), the all-gather, the gather on the exchanged panel
        REQUIRE(h, !h->pending, EKF_ERR_STATE, "correct: an exchange is pending");
        h->slab = slab_for(h, a.n_mm);
        {
            TimedLaunch tl(h, EKF_KERNEL_ROWPANEL);
            HIPCHK(h, launch_rowpanel_dev(h->st, a.j, a.n_mm, h->pstart, h->npend, corr_send(h, h->slab), h->storage, h->stream, dl));
        }
        h->nx_valid = false;
        h->pending = true; h->pending_kind = 1; h->x_count = h->slab;
        rc = exchange_rccl(h);
        h->pending = false; h->pending_kind = 0;
        if (rc) return rc;
        {
            TimedLaunch tl(h, EKF_KERNEL_GATHER);
            const PredictArgs *fuse = h->have_pp ? &h->pp : nullptr;
            HIPCHK(h, launch_gather_sharded(h->st, a, fuse, h->recv, h->slab, 0, /*patched*/ true, h->storage, h->stream, &dl));
            h->have_pp = false;
        }
        return finish_step(h);
    }
    {
        TimedLaunch tl(h, EKF_KERNEL_GATHER);
        const PredictArgs *fuse = h->have_pp ? &h->pp : nullptr;
        HIPCHK(h, launch_gather_devloop(h->st, a, fuse, dl, h->storage, h->stream));
        h->have_pp = false;
    }
    return finish_step(h);
}

// queue k_associate for observation z on the handle's stream; the decision goes to the device copy and, if host_slot != nullptr,
// to that mapped host slot (sequence number `seq` written last)
inline int32_t next_assoc_seq(ekf_handle *h) { return ++h->assoc_seq == 0 ? ++h->assoc_seq : h->assoc_seq; }   // never 0: a slot's initial value

// wait (bounded poll, then stream synchronisation) until the mapped slot carries sequence number seq
bool wait_mapped_seq(volatile AssocDecision *slot, int32_t seq) {
    for (int spin = 0; spin < 2000000; ++spin) {
        if (slot->seq == seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return true; }
        __builtin_ia32_pause();
    }
    return false;
}

inline int32_t assoc_blocks(int64_t N) { return (int32_t)((N + kAssocBlock - 1) / kAssocBlock); }

// all nblk workgroups of launch `seq` have stored their winner (self-validating entries: kernels.h)
bool wait_parts(volatile AssocHostPartial *set, int32_t nblk, int32_t seq) {
    int32_t b = 0;
    for (int spin = 0; spin < 2000000; ++spin) {
        while (b < nblk && read_part(set + b).seq == seq) ++b;
        if (b == nblk) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return true; }
        __builtin_ia32_pause();
    }
    return false;
}

// Correspondence.m:78-85 over the workgroups' winners: lowest likelihood, lowest index on ties (the order of the kernel's own
// reductions); nothing below the threshold anywhere -> new landmark, index N (0-based)
int32_t reduce_parts(ekf_handle *h, volatile AssocHostPartial *set, int32_t nblk, int32_t seq, int64_t N, int32_t *is_new, int64_t *idx) {
    double best = INFINITY;
    int64_t at = -1;
    for (int32_t b = 0; b < nblk; ++b) {
        const PartView v = read_part(set + b);
        REQUIRE(h, v.seq == seq, EKF_ERR_STATE, "associate: a workgroup's result is missing from the mapped buffer");
        const double ll = v.ll;
        const int64_t ix = v.index;
        if (ix >= 0 && (at < 0 || ll < best || (ll == best && ix < at))) { best = ll; at = ix; }
    }
    *is_new = at < 0 ? 1 : 0;
    *idx = at < 0 ? N : at;
    return EKF_OK;
}

// the decision of launch `seq` (nblk workgroups, entries in `set`): poll, or synchronise the stream, then reduce
int32_t collect_decision(ekf_handle *h, AssocHostPartial *set, int32_t nblk, int32_t seq, int64_t N, bool may_poll, int32_t *is_new,
                         int64_t *idx) {
    if (!(may_poll && h->assoc_poll && wait_parts(set, nblk, seq))) {
        HIPCHK(h, hipStreamSynchronize(h->stream));            // the kernel has retired: its stores to mapped memory are complete
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    return reduce_parts(h, set, nblk, seq, N, is_new, idx);
}

// exchange == false: the decision of this launch is final (unsharded, or sharded with the signature-only likelihood, which every
// shard evaluates identically from replicated data); exchange == true (sharded): this shard nominates among the landmarks whose
// diagonal block it holds and leaves its candidate -- and, want_costs, their position costs -- in the send area
// fold_predict (device-resident measure loop): a recorded predict(u) is carried out BY the association launch (it is k_predict and
// k_associate in one), so the scan's first row costs no k_predict launch and its correction folds nothing
int32_t launch_assoc(ekf_handle *h, const double z[3], const double R[4], AssocHostPartial *host_set_dev, int32_t seq,
                     bool exchange = false, bool want_costs = false, bool fold_predict = false) {
    REQUIRE(h, h->N >= 1, EKF_ERR_STATE, "associate: the state holds no landmark (Correspondence.m:29)");
    if (!fold_predict) {
        const int32_t rcp = materialize_predict(h);
        if (rcp) return rcp;
    }
    AssocArgs a;
    a.z0 = z[0]; a.z1 = z[1]; a.z2 = z[2];
    colmajor2(R, a.R00, a.R01, a.R10, a.R11);
    a.s_cost = h->cfg.s_cost; a.s_thresh = h->cfg.s_thresh; a.w_pos = h->cfg.w_pos;
    a.N = h->N; a.cur = h->cur; a.npend = h->npend; a.pstart = h->pstart;
    a.own_only = exchange ? 1 : 0;
    TimedLaunch tl(h, EKF_KERNEL_ASSOCIATE);
    HIPCHK(h, launch_associate(h->st, a, exchange ? (want_costs ? h->send + 4 : nullptr) : h->d_pos_cost, h->d_sig_cost,
                               h->d_partial, h->d_ticket, h->d_decision, exchange ? nullptr : host_set_dev, seq,
                               exchange ? h->send : nullptr, h->storage, h->stream, (fold_predict && h->have_pp) ? &h->pp : nullptr));
    if (fold_predict && h->have_pp) { h->have_pp = false; h->cur ^= 1; }       // the launch wrote the predicted state to the other buffer
    return EKF_OK;
}

// sharded association, first half: candidates (+ position costs) into the send area; the exchange moves x_count doubles
int32_t assoc_begin(ekf_handle *h, const double z[3], const double R[4], bool want_costs) {
    REQUIRE(h, !h->pending, EKF_ERR_STATE, "associate_begin: an exchange is already pending");
    const int32_t rc = launch_assoc(h, z, R, nullptr, 0, /*exchange*/ true, want_costs);
    if (rc) return rc;
    h->pending = true; h->pending_kind = 3; h->x_count = 4 + (want_costs ? h->N : 0);
    h->nx_valid = false;       // (the candidates' all-gather overwrites the receive area)
    h->assoc_costs = want_costs;
    return EKF_OK;
}

// second half: every shard takes the same arg-min over the gathered candidates; then as do_associate
int32_t assoc_finish(ekf_handle *h, int32_t *is_new, int64_t *idx, double *pos_cost, double *sig_cost) {
    REQUIRE(h, h->pending && h->pending_kind == 3, EKF_ERR_STATE, "associate_finish: no association pending");
    REQUIRE(h, !pos_cost || h->assoc_costs, EKF_ERR_STATE, "associate_finish: position costs were not requested at begin");
    h->pending = false; h->pending_kind = 0;
    const int32_t seq = next_assoc_seq(h);
    HIPCHK(h, launch_assoc_merge(h->st, h->recv, h->cfg.world, h->x_count, h->N, h->assoc_costs, h->d_pos_cost, h->d_decision,
                                 h->h_decision_dev, seq, h->stream));
    if (pos_cost) HIPCHK(h, hipMemcpyAsync(pos_cost, h->d_pos_cost, (size_t)h->N * 8, hipMemcpyDeviceToHost, h->stream));
    if (sig_cost) HIPCHK(h, hipMemcpyAsync(sig_cost, h->d_sig_cost, (size_t)h->N * 8, hipMemcpyDeviceToHost, h->stream));
    const bool have = h->h_decision_dev && !pos_cost && !sig_cost && wait_mapped_seq(h->h_decision, seq);
    if (!have) {
        HIPCHK(h, hipMemcpyAsync(h->h_decision, h->d_decision, sizeof(AssocDecision), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    *is_new = h->h_decision->is_new;
    *idx = h->h_decision->index;
    return EKF_OK;
}

// cfg.device_assoc == 2: every decision the device has produced since the last call must equal the host mirror's
int32_t verify_speculated(ekf_handle *h) {
    if (h->spec.empty()) return EKF_OK;
    const size_t n = h->spec.size();
    // wait for the NEWEST launch only: once its workgroups have reported, the launches queued before it on the same stream have
    // retired and their stores (posted in order) have landed
    {
        const ekf_handle::Spec &sp = h->spec[n - 1];
        AssocHostPartial *set = h->h_parts + (int64_t)((n - 1) % ekf_handle::kSpecRing) * h->parts_stride;
        if (!(h->assoc_poll && wait_parts(set, sp.nblk, sp.seq))) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
    }
    int32_t rc = EKF_OK;
    for (size_t q = 0; q < n && !rc; ++q) {
        const ekf_handle::Spec &sp = h->spec[q];
        int32_t is_new = 0; int64_t idx = 0;
        rc = reduce_parts(h, h->h_parts + (int64_t)(q % ekf_handle::kSpecRing) * h->parts_stride, sp.nblk, sp.seq, sp.idx_N, &is_new, &idx);
        if (!rc && (is_new != sp.is_new || idx != sp.idx))
            rc = fail(h, EKF_ERR_STATE, "measure: the device association disagrees with the host mirror of the signatures");
    }
    h->spec.clear();
    return rc;
}

int32_t do_associate(ekf_handle *h, const double z[3], const double R[4], int32_t *is_new, int64_t *idx,
                     double *pos_cost, double *sig_cost) {
    // Sharded handles need NO exchange here (they did until round 3): the position cost needs each landmark's own 2x2 block,
    // and those blocks are replicated, live, on every shard (DevState::diag) -- every shard evaluates every landmark and takes the same
    // decision from the same bits.  (ekf_associate_begin / _finish remain for hosts that were written around the exchange.)
    const int32_t seq = next_assoc_seq(h);
    const int64_t N = h->N;
    AssocHostPartial *set = h->h_parts + (int64_t)ekf_handle::kSpecRing * h->parts_stride;
    int32_t rc = launch_assoc(h, z, R, h->h_parts_dev + (int64_t)ekf_handle::kSpecRing * h->parts_stride, seq);
    if (rc) return rc;
    if (pos_cost) HIPCHK(h, hipMemcpyAsync(pos_cost, h->d_pos_cost, (size_t)N * 8, hipMemcpyDeviceToHost, h->stream));
    if (sig_cost) HIPCHK(h, hipMemcpyAsync(sig_cost, h->d_sig_cost, (size_t)N * 8, hipMemcpyDeviceToHost, h->stream));
    // measure()'s path: every workgroup stores its winner into mapped host memory (one 16-byte store: payload + sequence number)
    // and the host takes the arg-min as soon as all of them carry this launch's number -- ~2 us after the kernel's last store,
    // against ~15 us for a device->host copy + stream synchronisation.  Bounded: after ~2 ms the stream is synchronised instead.
    // With cost vectors asked for the copies above need the synchronisation anyway.
    return collect_decision(h, set, assoc_blocks(N), seq, N, !pos_cost && !sig_cost, is_new, idx);
}

// Correspondence.m:40-43,71,75,78-85 with the live likelihood (signature cost only): the landmark of lowest likelihood among those
// at or below the threshold, the lowest index on ties (strict '<' in index order, :81); nothing below the threshold -> (new, N).
void associate_signature_only(const ekf_handle *h, double z3, int32_t *is_new, int64_t *idx) {
    const int64_t N = h->N;
    *is_new = 1; *idx = N;
    double best = INFINITY;
    const double inv_cost = 1.0 / h->cfg.s_cost, thresh = h->cfg.s_thresh;
    // ll = (d c) d <= thresh only if |d| <= sqrt(thresh / c) (up to rounding: the window below is a strict superset); on a
    // large map only the landmarks inside that window of the sorted index are evaluated -- with the very same expression
    const double w = (inv_cost > 0.0 && thresh >= 0.0) ? sqrt(thresh / inv_cost) * (1.0 + 1e-9) + 1e-300 : INFINITY;
    if (N >= 256 && w < INFINITY && z3 == z3) {
        if (!h->s_sorted_ok || (int64_t)(h->s_sorted.size() + h->s_tail.size()) > N) {
            h->s_sorted.clear();
            h->s_tail.clear();
            h->s_sorted.reserve((size_t)N);
            for (int64_t k = 0; k < N; ++k) if (h->s_host[(size_t)k] == h->s_host[(size_t)k]) h->s_sorted.emplace_back(h->s_host[(size_t)k], k);
            std::sort(h->s_sorted.begin(), h->s_sorted.end());
            h->s_sorted_ok = true;
        } else if (h->s_tail.size() >= ekf_handle::kSortedTail) {
            const size_t mid = h->s_sorted.size();
            std::sort(h->s_tail.begin(), h->s_tail.end());
            h->s_sorted.insert(h->s_sorted.end(), h->s_tail.begin(), h->s_tail.end());
            std::inplace_merge(h->s_sorted.begin(), h->s_sorted.begin() + (ptrdiff_t)mid, h->s_sorted.end());
            h->s_tail.clear();
        }
        const auto lo = std::lower_bound(h->s_sorted.begin(), h->s_sorted.end(), std::pair<double, int64_t>(z3 - w, -1));
        const auto hi = std::upper_bound(lo, h->s_sorted.end(), std::pair<double, int64_t>(z3 + w, INT64_MAX));
        if (hi - lo < N / 2) {
            auto consider = [&](double sk, int64_t k) {
                const double d = z3 - sk;
                const double ll = d * inv_cost * d;
                if (ll <= thresh && (ll < best || (ll == best && k < *idx))) { *is_new = 0; best = ll; *idx = k; }
            };
            for (auto it = lo; it != hi; ++it) consider(it->first, it->second);
            for (const auto &e : h->s_tail) consider(e.first, e.second);     // landmarks appended since the last merge
            return;
        }
    }
    for (int64_t k = 0; k < N; ++k) {
        const double d = z3 - h->s_host[(size_t)k];
        const double ll = d * inv_cost * d;
        if (ll <= thresh && ll < best) { *is_new = 0; best = ll; *idx = k; }
    }
}

// landmark(find([landmark.index] == key)).loc  (key < 0: find([landmark.index]), i.e. all non-zero indices)
int32_t lookup_loc(ekf_handle *h, const double *lm_index, const double *lm_loc, int64_t L, bool any_nonzero, double key,
                   double loc[2]) {
    int64_t hits = 0, at = -1;
    for (int64_t i = 0; i < L; ++i) {
        const bool m = any_nonzero ? (lm_index[i] != 0.0) : (lm_index[i] == key);
        if (m) { ++hits; at = i; }
    }
    if (hits != 1) {
        char buf[160];
        snprintf(buf, sizeof buf, "measure: landmark lookup matched %lld entries (the reference's append() call is "
                 "only well-formed for exactly one)", (long long)hits);
        return fail(h, EKF_ERR_LOOKUP, buf);
    }
    loc[0] = lm_loc[at];
    loc[1] = lm_loc[L + at];
    return EKF_OK;
}

}  // namespace

// =====================================================================================================
extern "C" {

int32_t ekf_abi_version(void) { return EKF_ABI_VERSION; }

const char *ekf_status_string(int32_t s) {
    switch (s) {
        case EKF_OK: return "ok";
        case EKF_ERR_INVALID_ARG: return "invalid argument";
        case EKF_ERR_NO_DEVICE: return "no HIP device";
        case EKF_ERR_HIP: return "HIP runtime error";
        case EKF_ERR_CAPACITY: return "landmark capacity exhausted";
        case EKF_ERR_INDEX: return "landmark index outside the state";
        case EKF_ERR_LOOKUP: return "landmark table lookup did not match exactly one entry";
        case EKF_ERR_STATE: return "call not valid in the current state";
        case EKF_ERR_COMM: return "multi-GPU exchange failed";
        default: return "unknown status";
    }
}

int32_t ekf_config_default(ekf_config *cfg, int32_t mode) {
    if (!cfg || (mode != EKF_MODE_KNOWN && mode != EKF_MODE_UC)) return EKF_ERR_INVALID_ARG;
    memset(cfg, 0, sizeof *cfg);
    cfg->C = 0.2;                                     // EKF_SLAM.m:12
    cfg->Rc[0] = mode == EKF_MODE_KNOWN ? .01 : .1;   // EKF_SLAM.m:13 / EKF_SLAM_UC.m:13
    cfg->Rc[1] = 5;
    cfg->s_cost = .00000000001;                       // EKF_SLAM.m:14 / EKF_SLAM_UC.m:16
    cfg->s_thresh = 1000000000;                       // EKF_SLAM.m:16 / EKF_SLAM_UC.m:16
    cfg->w_pos = 0.0;                                 // Correspondence.m:75 is the live line
    cfg->capacity_landmarks = 1024;
    cfg->mode = mode;
    cfg->storage = EKF_STORE_F64;
    cfg->device = 0;
    cfg->tile = 0;
    cfg->rank = 0;
    cfg->world = 1;
    cfg->batch = 1;
    cfg->async_flush = 0;
    cfg->device_assoc = mode == EKF_MODE_UC ? 3 : 0;   // unknown correspondence: the association runs, and is consumed, on the device
    return EKF_OK;
}

int32_t ekf_create(const ekf_config *cfg, ekf_handle **out) {
    if (!cfg || !out) return EKF_ERR_INVALID_ARG;
    *out = nullptr;
    const int32_t T = cfg->tile == 0 ? (cfg->storage == EKF_STORE_F32 ? 256 : 128) : cfg->tile;
    const int32_t world = cfg->world <= 0 ? 1 : cfg->world;
    if (!(T == 16 || T == 32 || T == 64 || T == 128 || (T == 256 && cfg->storage == EKF_STORE_F32))) return EKF_ERR_INVALID_ARG;
    if (cfg->capacity_landmarks < 1 || cfg->rank < 0 || cfg->rank >= world) return EKF_ERR_INVALID_ARG;
    if (cfg->storage != EKF_STORE_F64 && cfg->storage != EKF_STORE_F32) return EKF_ERR_INVALID_ARG;
    if (cfg->mode != EKF_MODE_KNOWN && cfg->mode != EKF_MODE_UC) return EKF_ERR_INVALID_ARG;
    if (cfg->batch < 0 || cfg->batch > 64) return EKF_ERR_INVALID_ARG;
    if (cfg->pass_arith != EKF_ARITH_F64 &&
        !((cfg->pass_arith == EKF_ARITH_F32 || cfg->pass_arith == EKF_ARITH_SPLIT3) && cfg->storage == EKF_STORE_F32 && T == 256))
        return EKF_ERR_INVALID_ARG;                   // the f32-arithmetic passes exist for float tiles of edge 256 only
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
        return EKF_ERR_NO_DEVICE;

    ekf_handle *h = new (std::nothrow) ekf_handle();
    if (!h) return EKF_ERR_INVALID_ARG;
    h->cfg = *cfg;
    h->cfg.world = world;
    h->cfg.tile = T;
    h->T = T;
    h->cap = cfg->capacity_landmarks;
    h->storage = cfg->storage;
    *out = h;   // returned even on failure so the caller can read ekf_last_error, then ekf_destroy

    HIPCHK(h, hipSetDevice(cfg->device));
    {
        // the main stream carries the latency-bound step kernels: highest priority, so that their few workgroups are
        // dispatched ahead of the tens of thousands a concurrent flush (own stream, lowest priority) has queued
        int lo = 0, hi = 0;
        HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(h, hipStreamCreateWithPriority(&h->own_stream, hipStreamNonBlocking, hi));
    }
    h->stream = h->own_stream;

    const int64_t nt_cap = ekf_tiles_for(2 * h->cap, T);
    const int64_t ldm = nt_cap * T;
    h->st.ldm = ldm;
    h->st.tm = ekf_make_tilemap(T, world, cfg->rank);
    const int64_t slots = h->st.tm.slots_for_rows(nt_cap);
    h->work_cap = slots;
    for (int b = 0; b < 2; ++b) {
        HIPCHK(h, dalloc(h, &h->st.x[b], (size_t)(3 + ldm)));
        HIPCHK(h, dalloc(h, &h->st.prr[b], 16));
        HIPCHK(h, dalloc(h, &h->st.strip[b], (size_t)(3 * ldm)));
        HIPCHK(h, dalloc(h, &h->st.diag[b], (size_t)(3 * h->cap)));       // live F64 copies of the 2x2 diagonal blocks (kernels.h)
    }
    h->st.dcur = 0;
    {
        char *tiles = nullptr;
        HIPCHK(h, dalloc(h, &tiles, (size_t)slots * T * T * elt_size(h)));
        h->st.tiles = tiles;
        h->tilebuf[0] = tiles;
        h->async_flush = cfg->async_flush != 0;      // batch 1 too: the pass of update-step i then runs beside the gather of i + 1
        if (h->async_flush) {
            char *tiles2 = nullptr;
            HIPCHK(h, dalloc(h, &tiles2, (size_t)slots * T * T * elt_size(h)));
            h->tilebuf[1] = tiles2;
            {
                // The pass over P fills every CU (3 wavefronts x 146 VGPRs per SIMD); a gather launched meanwhile then waits for
                // workgroup slots -- measured 20 us per gather, stream priorities do not help (profiles/round1_tuning.md, sweep
                // 12).  So the flush stream is confined to a CU mask that leaves `reserve` CUs (default 32 = 4 per XCD) to the
                // gather chain.  Reserved set {32a + 8b + a}: 4 CUs on every XCD whether mask bits map to XCDs round-robin
                // (bit % 8) or in blocks of 32.  (Tuning builds: EKF_ASYNC_RESERVE_CUS=0 gives a plain lowest-priority stream.)
                // (Split arithmetic: 64 -- its pass is not bound by the matrix pipe and loses less to fewer CUs than the corrections gain from more:
                // configs[4] at 40 000 landmarks 9.6 k update-steps/s against 8.8 k with 32 and 9.0 k synchronous; F32 arithmetic: 7.3 k with 32, 6.6 k
                // with 64; counts that are not a multiple of 32 leave the persistent pass kernels two workgroups on some CU: round4_tuning.md 59.)
                int reserve = ekf_tune_int("EKF_ASYNC_RESERVE_CUS", cfg->pass_arith == EKF_ARITH_SPLIT3 ? 64 : 32);
                hipDeviceProp_t prop;
                HIPCHK(h, hipGetDeviceProperties(&prop, cfg->device));
                const int ncu = prop.multiProcessorCount;
                if (reserve > 0 && ncu == 256) {
                    if (reserve > 128) reserve = 128;
                    uint32_t mask[8];
                    for (int w = 0; w < 8; ++w) mask[w] = 0xffffffffu;
                    int taken = 0;
                    for (int b = 0; b < 16 && taken < reserve; ++b)           // b < 4: the balanced set above; then its shifts
                        for (int a = 0; a < 8 && taken < reserve; ++a) {
                            const int bit = 32 * a + 8 * (b & 3) + ((a + (b >> 2)) & 7);
                            if (mask[bit >> 5] & (1u << (bit & 31))) { mask[bit >> 5] &= ~(1u << (bit & 31)); ++taken; }
                        }
                    HIPCHK(h, hipExtStreamCreateWithCUMask(&h->flush_stream, 8, mask));
                    h->flush_cus = ncu - taken;                                // what a persistent pass kernel on that stream can occupy
                } else {
                    int lo = 0, hi = 0;
                    HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
                    HIPCHK(h, hipStreamCreateWithPriority(&h->flush_stream, hipStreamNonBlocking, lo));
                }
            }
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_pairs, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_flushed, hipEventDisableTiming));
        }
    }
    HIPCHK(h, dalloc(h, &h->st.s, (size_t)h->cap));
    h->batch = cfg->batch < 1 ? 1 : cfg->batch;
    h->cfg.batch = h->batch;
    h->st.pair_stride = 2 * ldm;
    h->st.pcap = h->async_flush ? 2 * h->batch : h->batch;      // in-flight batch + the batch being recorded
    // ONE allocation, G pairs then K pairs: k_gather addresses both from one uniform base with 32-bit lane offsets
    HIPCHK(h, dalloc(h, &h->st.Gp, (size_t)(2 * ldm) * h->st.pcap * 2));
    h->st.Kp = h->st.Gp + (size_t)(2 * ldm) * h->st.pcap;
    h->st.Gp32 = nullptr; h->st.Kp32 = nullptr;
    if (cfg->pass_arith != EKF_ARITH_F64) {
        HIPCHK(h, dalloc(h, &h->st.Gp32, (size_t)(2 * ldm) * h->st.pcap * 2));
        h->st.Kp32 = h->st.Gp32 + (size_t)(2 * ldm) * h->st.pcap;
        // the strip form of the pass: work list (every item once + one padded segment per 128-row slab and column range at most), dump
        hipDeviceProp_t prop;
        HIPCHK(h, hipGetDeviceProperties(&prop, cfg->device));
        h->aux.grid = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (h->flush_cus > 0) h->aux.grid = h->flush_cus;                 // cfg.async_flush: one persistent workgroup per CU of the pass stream's mask
        h->aux.grid -= h->aux.grid % 8;                                   // (block b walks XCD stream b & 7)
        if (h->aux.grid < 8) h->aux.grid = 8;
        const int64_t ranges = (2 * nt_cap + ekf_pipe32::kSeg * world - 1) / (ekf_pipe32::kSeg * world);
        h->segs_cap = 4 * slots + (2 * nt_cap * ranges + 8) * ekf_pipe32::kSeg;
        HIPCHK(h, dalloc(h, &h->d_segs, (size_t)h->segs_cap));
        float *dump = nullptr;
        HIPCHK(h, dalloc(h, &dump, (size_t)h->aux.grid * ekf_pipe32::kDumpFloats));
        h->aux.dump = dump;
        if (cfg->pass_arith == EKF_ARITH_SPLIT3) {
            // the bf16 planes of the pending pairs (flush32_split.h), cut from the float copies in front of every pass of 28-64 pairs
            HIPCHK(h, dalloc(h, &h->aux.Kb3, pass_split_plane_elems(ldm)));
            HIPCHK(h, dalloc(h, &h->aux.Gb3, pass_split_plane_elems(ldm)));
        }
    }
    HIPCHK(h, dalloc(h, &h->st.small, 32));
    HIPCHK(h, dalloc(h, &h->d_work, (size_t)slots));
    HIPCHK(h, dalloc(h, &h->d_work_xcd, (size_t)slots * 8));
    HIPCHK(h, dalloc(h, &h->d_partial, (size_t)((h->cap + kAssocBlock - 1) / kAssocBlock)));
    HIPCHK(h, dalloc(h, &h->d_decision, 1));
    HIPCHK(h, dalloc(h, &h->d_ticket, 1));
    HIPCHK(h, dalloc(h, &h->d_pos_cost, (size_t)h->cap));
    HIPCHK(h, dalloc(h, &h->d_sig_cost, (size_t)h->cap));
    HIPCHK(h, dalloc(h, &h->d_digest, kDigestDoubles));
    {
        h->sharded = world > 1 || cfg->force_sharded != 0;
        if (h->sharded) {
            h->slab_cap = slab_for(h, 2 * h->cap);
            const size_t rows = (size_t)(cfg->batch < 1 ? 1 : cfg->batch);     // a prefetch carries up to `batch` row-panels
            // ... and an association's exchange a candidate + one position cost per landmark
            h->xchg_cap = std::max<int64_t>(h->slab_cap * (int64_t)rows, 4 + h->cap);
            HIPCHK(h, dalloc(h, &h->own_send, (size_t)h->xchg_cap));
            HIPCHK(h, dalloc(h, &h->own_recv, (size_t)h->xchg_cap * world));
            HIPCHK(h, dalloc(h, &h->pf_store, (size_t)h->slab_cap * rows * world));
            h->send = h->own_send;
            h->recv = h->own_recv;
        }
    }
    HIPCHK(h, hipHostMalloc((void **)&h->h_decision, sizeof(AssocDecision), hipHostMallocMapped));
    memset(h->h_decision, 0, sizeof(AssocDecision));
    {
        static const bool poll = ekf_tune_int("EKF_ASSOC_POLL", 1) != 0;
        void *dp = nullptr;
        if (poll && hipHostGetDevicePointer(&dp, h->h_decision, 0) == hipSuccess) h->h_decision_dev = (AssocDecision *)dp;
    }
    {
        // k_associate's per-workgroup winners (see ekf_handle::h_parts): kSpecRing sets for cfg.device_assoc == 2, one more for
        // the calls that wait; 16 bytes per workgroup at capacity
        static const bool poll = ekf_tune_int("EKF_ASSOC_POLL", 1) != 0;
        h->assoc_poll = poll;
        h->parts_stride = assoc_blocks(h->cap > 0 ? h->cap : 1);
        const size_t bytes = sizeof(AssocHostPartial) * (size_t)h->parts_stride * (ekf_handle::kSpecRing + 1);
        HIPCHK(h, hipHostMalloc((void **)&h->h_parts, bytes, hipHostMallocMapped));
        memset(h->h_parts, 0, bytes);
        // sequence numbers start at 1: an entry that was never written must read as launch 0
        for (size_t e = 0; e < (size_t)h->parts_stride * (ekf_handle::kSpecRing + 1); ++e) h->h_parts[e].tag = (int32_t)assoc_part_mix(0, 0, 0);
        void *dp = nullptr;
        HIPCHK(h, hipHostGetDevicePointer(&dp, h->h_parts, 0));
        h->h_parts_dev = (AssocHostPartial *)dp;
    }
    {
        // device-resident measure loop: two sets of per-workgroup winners (a k_associate launch has ceil(N / 256) workgroups, a
        // k_gather launch one per 256 padded columns) and the ring of decision records
        h->lparts_stride = std::max<int64_t>(assoc_blocks(h->cap), gather_workgroups(h->st, 2 * h->cap));
        HIPCHK(h, dalloc(h, &h->d_lparts, (size_t)(2 * h->lparts_stride)));
        const size_t bytes = sizeof(AssocHostPartial) * ekf_handle::kLoopRing;
        HIPCHK(h, hipHostMalloc((void **)&h->h_lrec, bytes, hipHostMallocMapped));
        memset(h->h_lrec, 0, bytes);
        for (int e = 0; e < ekf_handle::kLoopRing; ++e) h->h_lrec[e].tag = (int32_t)assoc_part_mix(0, 0, 0);   // reads as launch 0
        void *dp = nullptr;
        HIPCHK(h, hipHostGetDevicePointer(&dp, h->h_lrec, 0));
        h->h_lrec_dev = (AssocHostPartial *)dp;
    }
    HIPCHK(h, hipHostMalloc((void **)&h->h_small, 32 * sizeof(double), hipHostMallocDefault));

    // x = [0 0 0]; P = 0.1*eye(3)   (EKF_SLAM.m:28-31, EKF_SLAM_UC.m:29-32)
    const double prr0[9] = { 0.1, 0, 0, 0, 0.1, 0, 0, 0, 0.1 };
    HIPCHK(h, hipMemcpy(h->st.prr[0], prr0, sizeof prr0, hipMemcpyHostToDevice));
    h->cur = 0;
    h->N = 0;
    h->grid_cap = ekf_tune_int("EKF_DOWNDATE_GRID", 0);
    HIPCHK(h, hipDeviceSynchronize());
    return EKF_OK;
}

int32_t ekf_destroy(ekf_handle *h) {
    if (!h) return EKF_OK;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    for (hipEvent_t e : h->throttle_ev) if (e) hipEventDestroy(e);
    if (h->flush_stream) { hipStreamSynchronize(h->flush_stream); hipStreamDestroy(h->flush_stream); }
    if (h->xchg_stream) { hipStreamSynchronize(h->xchg_stream); hipStreamDestroy(h->xchg_stream); hipEventDestroy(h->ev_pn_ready); hipEventDestroy(h->ev_pn_done); }
    if (h->ev_xchg) hipEventDestroy(h->ev_xchg);
    if (h->ev_pairs) hipEventDestroy(h->ev_pairs);
    if (h->ev_flushed) hipEventDestroy(h->ev_flushed);
    if (h->ev_rows) hipEventDestroy(h->ev_rows);
    for (auto &t : h->timers) for (hipEvent_t e : t.ev) hipEventDestroy(e);
    for (void *p : h->allocs) hipFree(p);
    if (h->h_decision) hipHostFree(h->h_decision);
    if (h->h_parts) hipHostFree(h->h_parts);
    if (h->h_lrec) hipHostFree(h->h_lrec);
    if (h->h_small) hipHostFree(h->h_small);
    if (h->wl_stage) { hipHostFree(h->wl_stage); hipEventDestroy(h->ev_wl); }
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
    return EKF_OK;
}

const char *ekf_last_error(const ekf_handle *h) { return h ? h->err.c_str() : "null handle"; }

int32_t ekf_set_stream(ekf_handle *h, void *hip_stream) {
    if (!h) return EKF_ERR_INVALID_ARG;
    int32_t rc = enter(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return EKF_OK;
}

int32_t ekf_sync(ekf_handle *h) {
    if (!h) return EKF_ERR_INVALID_ARG;
    int32_t rc = enter(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->flush_stream) HIPCHK(h, hipStreamSynchronize(h->flush_stream));
    return EKF_OK;
}

int32_t ekf_flush(ekf_handle *h) {
    if (!h) return EKF_ERR_INVALID_ARG;
    REQUIRE(h, !h->pending, EKF_ERR_STATE, "flush: a sharded correction is between begin and finish");
    int32_t rc = use_device(h);
    return rc ? rc : flush_pending(h);
}

int32_t ekf_pending(ekf_handle *h, int32_t *npending) {
    if (!h || !npending) return EKF_ERR_INVALID_ARG;
    *npending = h->npend;
    return EKF_OK;
}

int32_t ekf_set_params(ekf_handle *h, double C, const double Rc[2], double s_cost, double s_thresh, double w_pos) {
    if (!h || !Rc) return fail(h, EKF_ERR_INVALID_ARG, "set_params: null argument");
    h->cfg.C = C; h->cfg.Rc[0] = Rc[0]; h->cfg.Rc[1] = Rc[1];
    h->cfg.s_cost = s_cost; h->cfg.s_thresh = s_thresh; h->cfg.w_pos = w_pos;
    return EKF_OK;
}

int32_t ekf_predict(ekf_handle *h, const double u[2]) {
    if (!h || !u) return fail(h, EKF_ERR_INVALID_ARG, "predict: null argument");
    int32_t rc = use_device(h);
    return rc ? rc : do_predict(h, u);
}

int32_t ekf_motion_model(const double *x, int64_t n, const double u[2], double *x_new, double *F) {
    if (!x || !u || !x_new || n < 3) return EKF_ERR_INVALID_ARG;
    const double th = x[2];
    for (int64_t i = 0; i < n; ++i) x_new[i] = x[i];
    x_new[0] = x[0] + u[0] * ekfm::cosd(th + u[1]);           // EKF_SLAM.m:58-60
    x_new[1] = x[1] + u[0] * ekfm::sind(th + u[1]);
    x_new[2] = th + u[1];
    if (F) {
        for (int64_t i = 0; i < n * n; ++i) F[i] = 0.0;
        for (int64_t i = 0; i < n; ++i) F[i * n + i] = 1.0;  // eye(n)
        F[2 * n + 0] = -1 * u[0] * ekfm::sind(th);           // F(1,3), column-major
        F[2 * n + 1] = u[0] * ekfm::cosd(th);                // F(2,3)
    }
    return EKF_OK;
}

int32_t ekf_append(ekf_handle *h, const double u[2], const double R[4], const double pos[2], double signature) {
    if (!h || !u || !R || !pos) return fail(h, EKF_ERR_INVALID_ARG, "append: null argument");
    int32_t rc = use_device(h);
    return rc ? rc : do_append(h, u, R, pos, signature);
}

int32_t ekf_correct(ekf_handle *h, const double z[2], const double R[4], int64_t idx) {
    if (!h || !z || !R) return fail(h, EKF_ERR_INVALID_ARG, "correct: null argument");
    int32_t rc = use_device(h);
    return rc ? rc : do_correct(h, z, R, idx);
}

int32_t ekf_associate(ekf_handle *h, const double z[3], const double R[4], int32_t *is_new, int64_t *idx,
                      double *pos_cost, double *sig_cost) {
    if (!h || !z || !R || !is_new || !idx) return fail(h, EKF_ERR_INVALID_ARG, "associate: null argument");
    int32_t rc = use_device(h);
    return rc ? rc : do_associate(h, z, R, is_new, idx, pos_cost, sig_cost);
}

int32_t ekf_measure(ekf_handle *h, const double *obs, int64_t m, const double u[2], const double *lm_index,
                    const double *lm_loc, int64_t L) {
    if (!h || !u || m < 0 || L < 0 || (m > 0 && !obs) || (L > 0 && (!lm_index || !lm_loc)))
        return fail(h, EKF_ERR_INVALID_ARG, "measure: bad argument");
    int32_t rc = use_device(h);
    if (rc) return rc;
    // The loop below decides row by row whether to append or correct, and a correction on a shard needs an exchange in the
    // middle of it: only the library-owned communicator can run that.  A host that runs the all-gather itself (transport (b)
    // / (c) of ekfslam.h) drives append / correct_begin / its exchange / correct_finish per row -- refused here, up front,
    // before any row has changed the state.
    REQUIRE(h, !(h->sharded && h->comm == nullptr && h->xhook == nullptr && m > 0), EKF_ERR_STATE,
            "measure: a sharded handle needs the library-owned communicator (ekf_comm_init) or an exchange hook "
            "(ekf_exchange_set_hook); with a host-run exchange call ekf_append / ekf_correct_begin / ekf_correct_finish per observation");
    const bool dev_loop = h->cfg.mode == EKF_MODE_UC && h->cfg.device_assoc == 3 && h->cfg.w_pos == 0.0;
    if (h->sharded && (h->comm || h->xhook) && h->batch > 1 && m > 1 && h->N > 0 && !dev_loop) {      // (the device loop names its landmarks on the device)
        // the scan's corrections are known before the loop runs: fetch their base row-panels in ONE exchange
        // (rows that turn out to append drop the prefetch again; the per-row exchange then takes over)
        std::vector<int64_t> want;
        for (int64_t ii = 0; ii < m && (int64_t)want.size() < h->batch; ++ii) {
            int64_t idx = -1;
            if (h->cfg.mode == EKF_MODE_KNOWN) { if (!(obs[2 * m + ii] > (double)h->N) && ii < h->N) idx = ii; }
            else if (h->cfg.w_pos == 0.0) { int32_t nw = 0; associate_signature_only(h, obs[2 * m + ii], &nw, &idx); if (nw) idx = -1; }
            if (idx >= 0 && std::find(want.begin(), want.end(), idx) == want.end()) want.push_back(idx);
        }
        if (want.size() > 1) {
            rc = prefetch_begin(h, want.data(), (int32_t)want.size());     // reads tiles only: a lazy predict stays lazy
            if (rc) return rc;
            rc = exchange_rccl(h);
            if (rc) { h->pending = false; return rc; }
            rc = prefetch_finish(h);
            if (rc) return rc;
        }
    }
    // cfg.device_assoc == 3 (the default of EKF_MODE_UC): the device-resident loop.  Per observation the host queues
    //   [k_associate, only if the previous launch did not already evaluate this observation's association]  ->
    //   k_gather (takes the landmark from the device's decision; its epilogue evaluates the NEXT observation's association)
    //   or k_append (checks the device found nothing below the threshold)  ->  the pass over P when a batch is complete
    // with no wait anywhere: which of the two it queues is the host mirror's prediction (exact when w_pos == 0: the reference's
    // live likelihood is a function of z(3) and s alone, Correspondence.m:71,75), what the device decided comes back in records
    // that are checked later (verify_loop).  With w_pos != 0 the host cannot predict the branch: the waited path below.
    // On a shard the same loop runs on every rank (the association reads replicated data only, so every rank's device names the same
    // landmark); a correction is k_rowpanel<kDev> (the panel of the landmark the device names) -> all-gather -> k_gather<sharded, kDev>.
    struct { bool have; int set; int32_t seq, nblk; } nxt = { false, 0, 0, 0 };     // winners of row ii's association already on the device
    if (dev_loop) {
        rc = verify_loop(h, /*block*/ false);                              // what earlier scans' launches have reported by now
        if (rc) return rc;
    }
    for (int64_t ii = 0; ii < m; ++ii) {                                   // EKF_SLAM.m:107
        const double z[3] = { obs[ii], obs[m + ii], obs[2 * m + ii] };
        const double R[4] = { z[0] * h->cfg.Rc[0], 0.0, 0.0, z[1] * h->cfg.Rc[1] };   // :108
        double loc[2];
        if (h->N == 0) {                                                   // :110-111  length(x) < 4
            rc = lookup_loc(h, lm_index, lm_loc, L, true, 0.0, loc);
            if (rc) return rc;
            rc = do_append(h, u, R, loc, 1.0);
        } else if (h->cfg.mode == EKF_MODE_KNOWN) {
            if (z[2] > (double)h->N) {                                     // :118-120
                rc = lookup_loc(h, lm_index, lm_loc, L, false, z[2], loc);
                if (rc) return rc;
                rc = do_append(h, u, R, loc, z[2]);
            } else {
                // a shard that rewrites P per correction lets this row's pass extract the next row's panel (ekf_hint_next)
                if (h->sharded && h->batch == 1 && ii + 1 < m && !(obs[2 * m + ii + 1] > (double)h->N) && ii + 1 < h->N) h->hint_idx = ii + 1;
                rc = do_correct(h, z, R, ii);                              // :123  idx = ii
            }
        } else if (dev_loop) {
            int32_t is_new = 0;
            int64_t idx = 0;
            associate_signature_only(h, z[2], &is_new, &idx);              // the prediction that shapes the queue
            DevLoopArgs dl = {};
            if (!nxt.have) {                                               // EKF_SLAM_UC.m:119, as a launch of its own
                nxt.set = h->loop_set ^ 1; nxt.seq = next_assoc_seq(h); nxt.nblk = assoc_blocks(h->N);
                rc = launch_assoc(h, z, R, h->d_lparts + (int64_t)nxt.set * h->lparts_stride, nxt.seq, false, false,
                                  /*fold_predict*/ !is_new);       // an append materialises the predict anyway
                if (rc) return rc;
                h->loop_set = nxt.set;
            }
            dl.parts_in = h->d_lparts + (int64_t)nxt.set * h->lparts_stride; dl.nblk_in = nxt.nblk; dl.seq_in = nxt.seq;
            if (h->lrec_head - h->lrec_tail >= (uint64_t)ekf_handle::kLoopRing) { rc = verify_loop(h, /*block*/ true); if (rc) return rc; }
            dl.rec = h->h_lrec_dev + h->lrec_head % ekf_handle::kLoopRing;
            dl.seq_rec = next_assoc_seq(h);
            const int set_in = nxt.set;
            nxt.have = false;
            if (is_new) {                                                  // EKF_SLAM_UC.m:121-123
                rc = lookup_loc(h, lm_index, lm_loc, L, false, (double)(idx + 1), loc);
                if (rc) return rc;
                rc = do_append(h, u, R, loc, (double)(idx + 1), &dl);
            } else {
                if (ii + 1 < m) {                                          // the next row's association rides in this correction
                    const double zn[3] = { obs[ii + 1], obs[m + ii + 1], obs[2 * m + ii + 1] };
                    dl.parts_out = h->d_lparts + (int64_t)(set_in ^ 1) * h->lparts_stride;
                    dl.seq_out = next_assoc_seq(h);
                    dl.z0 = zn[0]; dl.z1 = zn[1]; dl.z2 = zn[2];
                    dl.R00 = zn[0] * h->cfg.Rc[0]; dl.R01 = 0.0; dl.R10 = 0.0; dl.R11 = zn[1] * h->cfg.Rc[1];   // :108 / UC :110
                    dl.s_cost = h->cfg.s_cost; dl.s_thresh = h->cfg.s_thresh; dl.w_pos = h->cfg.w_pos;
                    nxt.have = true; nxt.set = set_in ^ 1; nxt.seq = dl.seq_out; nxt.nblk = (int32_t)gather_workgroups(h->st, n_mm(h));
                }
                rc = do_correct_dev(h, z, R, idx, dl);
                if (!rc && nxt.have) h->loop_set = nxt.set;
            }
            if (rc) return rc;
            h->lspec.push_back({ dl.seq_rec, is_new, idx });
            ++h->lrec_head;
        } else {
            int32_t is_new = 0;
            int64_t idx = 0;
            if (h->cfg.w_pos == 0.0 && h->cfg.device_assoc != 1) {
                // The reference's decision is a pure function of z(3) and s: the Mahalanobis position cost it also
                // evaluates is discarded (Correspondence.m:74-75).  With w_pos == 0 measure() therefore decides from
                // the host mirror of s -- same arithmetic as k_associate, no launch, no device->host sync.
                // ekf_associate() always runs the full device computation.
                associate_signature_only(h, z[2], &is_new, &idx);
                if (h->cfg.device_assoc == 2) {
                    // ... and with device_assoc == 2 the device evaluates the association all the same (per-landmark phi_k,
                    // Mahalanobis and signature cost, arg-min), queued behind the previous row's kernels; the host does not wait
                    // for it but checks every decision against its own before measure() returns
                    if ((int)h->spec.size() == ekf_handle::kSpecRing) { rc = verify_speculated(h); if (rc) return rc; }
                    const int32_t seq = next_assoc_seq(h);
                    rc = launch_assoc(h, z, R, h->h_parts_dev + (int64_t)(h->spec.size() % ekf_handle::kSpecRing) * h->parts_stride, seq);
                    if (rc) { verify_speculated(h); return rc; }
                    h->spec.push_back({ seq, is_new, assoc_blocks(h->N), idx, h->N });
                }
            } else {
                rc = do_associate(h, z, R, &is_new, &idx, nullptr, nullptr);   // EKF_SLAM_UC.m:119
                if (rc) return rc;
            }
            if (is_new) {                                                  // EKF_SLAM_UC.m:121-123
                rc = lookup_loc(h, lm_index, lm_loc, L, false, (double)(idx + 1), loc);
                if (rc) { verify_speculated(h); return rc; }
                rc = do_append(h, u, R, loc, (double)(idx + 1));
            } else {
                rc = do_correct(h, z, R, idx);
            }
        }
        if (rc) { verify_speculated(h); return rc; }
    }
    return verify_speculated(h);
}

int32_t ekf_hint_next(ekf_handle *h, int64_t idx) {
    if (!h) return EKF_ERR_INVALID_ARG;
    h->hint_idx = (idx >= 0 && idx < h->N) ? idx : -1;
    return EKF_OK;
}

int32_t ekf_correct_begin(ekf_handle *h, const double z[2], const double R[4], int64_t idx) {
    if (!h || !z || !R) return fail(h, EKF_ERR_INVALID_ARG, "correct_begin: null argument");
    REQUIRE(h, h->sharded, EKF_ERR_STATE, "correct_begin: handle is not sharded (use ekf_correct)");
    int32_t rc = use_device(h);
    return rc ? rc : correct_begin(h, z, R, idx);
}

int32_t ekf_correct_finish(ekf_handle *h) {
    if (!h) return EKF_ERR_INVALID_ARG;
    int32_t rc = use_device(h);
    return rc ? rc : correct_finish(h);
}

int32_t ekf_associate_begin(ekf_handle *h, const double z[3], const double R[4], int32_t want_costs) {
    if (!h || !z || !R) return fail(h, EKF_ERR_INVALID_ARG, "associate_begin: null argument");
    REQUIRE(h, h->sharded, EKF_ERR_STATE, "associate_begin: handle is not sharded (use ekf_associate)");
    int32_t rc = use_device(h);
    return rc ? rc : assoc_begin(h, z, R, want_costs != 0);
}

int32_t ekf_associate_finish(ekf_handle *h, int32_t *is_new, int64_t *idx, double *pos_cost, double *sig_cost) {
    if (!h || !is_new || !idx) return fail(h, EKF_ERR_INVALID_ARG, "associate_finish: null argument");
    int32_t rc = use_device(h);
    return rc ? rc : assoc_finish(h, is_new, idx, pos_cost, sig_cost);
}

int32_t ekf_prefetch_begin(ekf_handle *h, const int64_t *idx, int32_t m) {
    if (!h || !idx) return fail(h, EKF_ERR_INVALID_ARG, "prefetch_begin: null argument");
    REQUIRE(h, h->sharded, EKF_ERR_STATE, "prefetch_begin: handle is not sharded");
    int32_t rc = use_device(h);
    return rc ? rc : prefetch_begin(h, idx, m);
}

int32_t ekf_prefetch_finish(ekf_handle *h) {
    if (!h) return EKF_ERR_INVALID_ARG;
    int32_t rc = use_device(h);
    return rc ? rc : prefetch_finish(h);
}

int32_t ekf_prefetch_rows(ekf_handle *h, const int64_t *idx, int32_t m) {
    if (!h || !idx) return fail(h, EKF_ERR_INVALID_ARG, "prefetch_rows: null argument");
    if (!h->sharded) return EKF_OK;                  // nothing to exchange on an unsharded handle
    int32_t rc = use_device(h);
    if (rc) return rc;
    rc = prefetch_begin(h, idx, m);
    if (rc) return rc;
    rc = exchange_rccl(h);
    if (rc) { h->pending = false; return rc; }
    return prefetch_finish(h);
}

int32_t ekf_prefetch_next(ekf_handle *h, const int64_t *idx, int32_t m) {
    if (!h || (m > 0 && !idx) || m < 0) return fail(h, EKF_ERR_INVALID_ARG, "prefetch_next: bad argument");
    if (!h->sharded) return EKF_OK;                  // nothing to exchange on an unsharded handle
    h->pn_idx.clear();
    if (m == 0) return EKF_OK;
    REQUIRE(h, h->batch > 1 && !h->async_flush, EKF_ERR_STATE, "prefetch_next: needs cfg.batch > 1 and a synchronous flush");
    REQUIRE(h, h->cfg.pass_arith == EKF_ARITH_F64, EKF_ERR_STATE,
            "prefetch_next: with cfg.pass_arith = EKF_ARITH_F32 the pass's result is not what an extraction in front of it can compute");
    REQUIRE(h, h->comm != nullptr || h->xhook != nullptr, EKF_ERR_STATE,
            "prefetch_next: needs the library-owned communicator (ekf_comm_init) or an exchange hook (ekf_exchange_set_hook)");
    REQUIRE(h, m <= h->batch && m <= 64, EKF_ERR_INVALID_ARG, "prefetch_next: between 1 and min(cfg.batch, 64) landmarks");
    for (int32_t q = 0; q < m; ++q)
        REQUIRE(h, idx[q] >= 0 && idx[q] < h->N, EKF_ERR_INDEX, "prefetch_next: landmark index outside the state");
    h->pn_idx.assign(idx, idx + m);
    h->pn_N = h->N;
    return EKF_OK;
}

int32_t ekf_exchange_info(ekf_handle *h, void **send, void **recv, int64_t *count, int64_t *count_capacity) {
    if (!h) return EKF_ERR_INVALID_ARG;
    REQUIRE(h, h->sharded, EKF_ERR_STATE, "exchange_info: handle is not sharded");
    if (send) *send = pending_send(h);
    if (recv) *recv = h->recv;
    if (count) *count = h->pending ? h->x_count : h->slab;
    if (count_capacity) *count_capacity = h->xchg_cap;
    return EKF_OK;
}

int32_t ekf_exchange_set_buffers(ekf_handle *h, void *send, void *recv) {
    if (!h) return EKF_ERR_INVALID_ARG;
    REQUIRE(h, h->sharded && !h->pending, EKF_ERR_STATE, "exchange_set_buffers: not sharded, or a correction is pending");
    h->send = send ? (double *)send : h->own_send;
    h->recv = recv ? (double *)recv : h->own_recv;
    h->nx_valid = false;       // corr_send() may point elsewhere now: a hinted extraction sits in the old area
    return EKF_OK;
}

int32_t ekf_exchange_set_hook(ekf_handle *h, int32_t (*hook)(void *), void *ctx) {
    if (!h) return EKF_ERR_INVALID_ARG;
    REQUIRE(h, h->sharded && !h->pending, EKF_ERR_STATE, "exchange_set_hook: not sharded, or an exchange is pending");
    REQUIRE(h, h->comm == nullptr || hook == nullptr, EKF_ERR_STATE, "exchange_set_hook: the handle has a communicator of its own");
    h->xhook = hook;
    h->xhook_ctx = ctx;
    h->nx_valid = false;
    return EKF_OK;
}

int32_t ekf_exchange_local(ekf_handle **hs, int32_t world) {
    if (!hs || world < 1) return EKF_ERR_INVALID_ARG;
    for (int r = 0; r < world; ++r) {
        if (!hs[r]) return EKF_ERR_INVALID_ARG;
        REQUIRE(hs[r], hs[r]->sharded && hs[r]->cfg.world == world && hs[r]->cfg.rank == r && hs[r]->pending &&
                           hs[r]->x_count == hs[0]->x_count && hs[r]->pending_kind == hs[0]->pending_kind,
                EKF_ERR_STATE, "exchange_local: handles must be the shards 0..world-1 of one filter, each between the same begin and finish");
    }
    // producers first: every shard's send slab must be complete before anyone copies it
    for (int r = 0; r < world; ++r) {
        HIPCHK(hs[r], hipSetDevice(hs[r]->cfg.device));
        HIPCHK(hs[r], hipStreamSynchronize(hs[r]->stream));
    }
    const size_t bytes = (size_t)hs[0]->x_count * sizeof(double);
    for (int dst = 0; dst < world; ++dst) {
        ekf_handle *d = hs[dst];
        HIPCHK(d, hipSetDevice(d->cfg.device));
        for (int src = 0; src < world; ++src) {
            const double *from = pending_send(hs[src]);
            double *to = d->recv + (size_t)src * d->x_count;
            if (from == to) continue;                                  // already in place (own segment of the own receive area)
            HIPCHK(d, hipMemcpyPeerAsync(to, d->cfg.device, from, hs[src]->cfg.device, bytes, d->stream));
        }
        if (!d->ev_xchg) HIPCHK(d, hipEventCreateWithFlags(&d->ev_xchg, hipEventDisableTiming));
        HIPCHK(d, hipEventRecord(d->ev_xchg, d->stream));
    }
    // consumers before the next producers: a shard's stream may run ahead into its next extract (k_rowpanel overwrites its
    // send slab) while another shard's stream has not yet copied that slab -- every stream waits for every shard's copies.
    // (Found as an intermittent divergence of the replicated state across a 4-shard group on one GPU.)
    for (int r = 0; r < world; ++r) {
        HIPCHK(hs[r], hipSetDevice(hs[r]->cfg.device));
        for (int dst = 0; dst < world; ++dst)
            if (dst != r) HIPCHK(hs[r], hipStreamWaitEvent(hs[r]->stream, hs[dst]->ev_xchg, 0));
    }
    return EKF_OK;
}

int32_t ekf_comm_unique_id(ekf_comm_id *id) {
    if (!id) return EKF_ERR_INVALID_ARG;
    std::string err;
    if (!rccl_load(err)) return EKF_ERR_COMM;
    return g_rccl.GetUniqueId(id) == 0 ? EKF_OK : EKF_ERR_COMM;
}

int32_t ekf_comm_init(ekf_handle *h, const ekf_comm_id *id) {
    if (!h || !id) return fail(h, EKF_ERR_INVALID_ARG, "comm_init: null argument");
    REQUIRE(h, h->sharded, EKF_ERR_STATE, "comm_init: handle is not sharded");
    REQUIRE(h, h->comm == nullptr, EKF_ERR_STATE, "comm_init: communicator already attached");
    int32_t rc = use_device(h);
    if (rc) return rc;
    std::string err;
    if (!rccl_load(err)) return fail(h, EKF_ERR_COMM, err.c_str());
    void *comm = nullptr;
    const int r = g_rccl.CommInitRank(&comm, h->cfg.world, *id, h->cfg.rank);
    if (r != 0) return fail(h, EKF_ERR_COMM, g_rccl.GetErrorString(r));
    h->comm = comm;
    h->nx_valid = false;       // with a communicator the row-panel goes straight into the receive area (corr_send)
    return EKF_OK;
}

int32_t ekf_shard_owner(int32_t world, int64_t I, int64_t J) {
    if (world < 1 || I < 0 || J < 0) return -1;
    return ekf_make_tilemap(64, world, 0).owner(I, J);
}

int64_t ekf_shard_slot(int32_t world, int64_t I, int64_t J) {
    if (world < 1 || I < 0 || J < 0 || J > I) return -1;
    return ekf_make_tilemap(64, world, 0).slot(I, J);
}

int32_t ekf_shard_panel_source(int32_t world, int64_t tile_row_j, int64_t chunk, int32_t *owner, int64_t *local_chunk) {
    if (world < 1 || tile_row_j < 0 || chunk < 0 || !owner || !local_chunk) return EKF_ERR_INVALID_ARG;
    *owner = (int32_t)((tile_row_j + chunk) % world);
    *local_chunk = chunk / world;
    return EKF_OK;
}

int32_t ekf_num_landmarks(ekf_handle *h, int64_t *N) {
    if (!h || !N) return fail(h, EKF_ERR_INVALID_ARG, "num_landmarks: null argument");
    *N = h->N;
    return EKF_OK;
}

int32_t ekf_get_x(ekf_handle *h, double *x) {
    if (!h || !x) return fail(h, EKF_ERR_INVALID_ARG, "get_x: null argument");
    int32_t rc = enter(h);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(x, h->st.x[h->cur], (size_t)(3 + n_mm(h)) * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return EKF_OK;
}

int32_t ekf_set_x(ekf_handle *h, const double *x, int64_t n) {
    if (!h || !x) return fail(h, EKF_ERR_INVALID_ARG, "set_x: null argument");
    REQUIRE(h, n >= 3 && (n - 3) % 2 == 0 && (n - 3) / 2 <= h->cap, EKF_ERR_INVALID_ARG, "set_x: bad length");
    int32_t rc = enter(h);
    if (rc) return rc;
    rc = flush_pending(h);     // pending pairs belong to the old state
    if (rc) return rc;
    if ((n - 3) / 2 < h->N) HIPCHK(h, clear_pairs(h));      // shrinking the map
    h->N = (n - 3) / 2;
    h->pf_valid = false;       // a prefetch belongs to the state it was taken from
    h->nx_valid = false;
    h->s_host.resize((size_t)h->N, 0.0);
    h->s_sorted_ok = false;
    HIPCHK(h, hipMemcpyAsync(h->st.x[h->cur], x, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return EKF_OK;
}

int32_t ekf_get_s(ekf_handle *h, double *s) {
    if (!h || (!s && h->N > 0)) return fail(h, EKF_ERR_INVALID_ARG, "get_s: null argument");
    int32_t rc = enter(h);
    if (rc) return rc;
    if (h->N > 0) HIPCHK(h, hipMemcpyAsync(s, h->st.s, (size_t)h->N * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return EKF_OK;
}

int32_t ekf_set_s(ekf_handle *h, const double *s, int64_t N) {
    if (!h || (!s && N > 0)) return fail(h, EKF_ERR_INVALID_ARG, "set_s: null argument");
    REQUIRE(h, N == h->N, EKF_ERR_INVALID_ARG, "set_s: length must equal the number of landmarks (set x first)");
    int32_t rc = enter(h);
    if (rc) return rc;
    if (N > 0) HIPCHK(h, hipMemcpyAsync(h->st.s, s, (size_t)N * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->s_host.assign(s, s + N);
    h->s_sorted_ok = false;
    return EKF_OK;
}

int32_t ekf_diag_poke_device_signature(ekf_handle *h, int64_t idx, double value) {
    if (!h) return EKF_ERR_INVALID_ARG;
    REQUIRE(h, idx >= 0 && idx < h->N, EKF_ERR_INVALID_ARG, "diag_poke_device_signature: no such landmark");
    int32_t rc = enter(h);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->st.s + idx, &value, 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));      // (`value` is this call's stack)
    return EKF_OK;
}

int32_t ekf_get_P(ekf_handle *h, double *P) {
    if (!h || !P) return fail(h, EKF_ERR_INVALID_ARG, "get_P: null argument");
    int32_t rc = enter(h);
    if (rc) return rc;
    rc = flush_pending(h);
    if (rc) return rc;
    const int64_t n = 3 + n_mm(h);
    double *dense = nullptr;
    HIPCHK(h, hipMalloc((void **)&dense, (size_t)(n * n) * 8));
    hipError_t e = launch_unpack_dense(h->st, h->cur, n_mm(h), dense, h->storage, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(P, dense, (size_t)(n * n) * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(dense);
    if (e != hipSuccess) return fail(h, EKF_ERR_HIP, "get_P", e);
    return EKF_OK;
}

int32_t ekf_set_P(ekf_handle *h, const double *P, int64_t n) {
    if (!h || !P) return fail(h, EKF_ERR_INVALID_ARG, "set_P: null argument");
    REQUIRE(h, n == 3 + n_mm(h), EKF_ERR_INVALID_ARG, "set_P: n must equal length(x) (set x first)");
    int32_t rc = enter(h);
    if (rc) return rc;
    { const int32_t rcr = retire_inflight(h); if (rcr) return rcr; }
    h->npend = 0; h->pstart = 0;   // the whole covariance is replaced ...
    h->pf_valid = false;           // ... and with it every prefetched base row-panel
    h->nx_valid = false;
    double *dense = nullptr;
    HIPCHK(h, hipMalloc((void **)&dense, (size_t)(n * n) * 8));
    hipError_t e = hipMemcpyAsync(dense, P, (size_t)(n * n) * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = launch_pack_dense(h->st, h->cur, n_mm(h), dense, h->storage, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(dense);
    if (e != hipSuccess) return fail(h, EKF_ERR_HIP, "set_P", e);
    return EKF_OK;
}

int32_t ekf_get_P_block(ekf_handle *h, int64_t r0, int64_t c0, int64_t nr, int64_t nc, double *out) {
    if (!h || !out) return fail(h, EKF_ERR_INVALID_ARG, "get_P_block: null argument");
    const int64_t n = 3 + n_mm(h);
    REQUIRE(h, r0 >= 0 && c0 >= 0 && nr >= 1 && nc >= 1 && r0 + nr <= n && c0 + nc <= n, EKF_ERR_INVALID_ARG,
            "get_P_block: block outside P");
    int32_t rc = enter(h);
    if (rc) return rc;
    rc = flush_pending(h);
    if (rc) return rc;
    double *d = nullptr;
    HIPCHK(h, hipMalloc((void **)&d, (size_t)(nr * nc) * 8));
    hipError_t e = launch_get_block(h->st, h->cur, r0, c0, nr, nc, d, h->storage, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, (size_t)(nr * nc) * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(h, EKF_ERR_HIP, "get_P_block", e);
    return EKF_OK;
}

int32_t ekf_get_P_diag_blocks(ekf_handle *h, double *out) {
    if (!h || !out) return fail(h, EKF_ERR_INVALID_ARG, "get_P_diag_blocks: null argument");
    int32_t rc = enter(h);
    if (rc) return rc;
    // no pass over P: what plot() reads (EKF_SLAM.m:180,205) is the robot block and the landmarks' own 2x2 blocks, and both are live
    // (DevState::prr, DevState::diag carry every correction so far, pending or not)
    const size_t bytes = (size_t)(4 * (h->N + 1)) * 8;
    double *d = nullptr;
    HIPCHK(h, hipMalloc((void **)&d, bytes));
    hipError_t e = launch_get_diag_blocks(h->st, h->cur, h->N, d, h->storage, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(h, EKF_ERR_HIP, "get_P_diag_blocks", e);
    return EKF_OK;
}

int32_t ekf_get_Q(ekf_handle *h, double Q[9]) {
    if (!h || !Q) return fail(h, EKF_ERR_INVALID_ARG, "get_Q: null argument");
    int32_t rc = enter(h);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->h_small, h->st.small, 32 * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Q[c * 3 + r] = h->h_small[12 + 3 * r + c];
    return EKF_OK;
}

int32_t ekf_load_lowrank_state(ekf_handle *h, int64_t N, const double *x, const double *s, const double *d,
                               const double *U, int64_t k) {
    if (!h || !x || !s || !d || !U) return fail(h, EKF_ERR_INVALID_ARG, "load_lowrank_state: null argument");
    REQUIRE(h, N >= 0 && N <= h->cap && k >= 1, EKF_ERR_INVALID_ARG, "load_lowrank_state: bad N or k");
    int32_t rc = enter(h);
    if (rc) return rc;
    const int64_t n = 3 + 2 * N;
    if (N < h->N) HIPCHK(h, clear_pairs(h));
    h->N = N;
    h->s_host.assign(s, s + N);
    h->s_sorted_ok = false;
    { const int32_t rcr = retire_inflight(h); if (rcr) return rcr; }
    h->npend = 0; h->pstart = 0;   // the whole state is replaced ...
    h->pf_valid = false;           // ... and with it every prefetched base row-panel
    h->nx_valid = false;
    rc = refresh_work(h);
    if (rc) return rc;
    double *dd = nullptr, *dU = nullptr;
    HIPCHK(h, hipMalloc((void **)&dd, (size_t)n * 8));
    hipError_t e = hipMalloc((void **)&dU, (size_t)(n * k) * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(dd, d, (size_t)n * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dU, U, (size_t)(n * k) * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->st.x[h->cur], x, (size_t)n * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && N > 0) e = hipMemcpyAsync(h->st.s, s, (size_t)N * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = launch_lowrank(h->st, h->cur, 2 * N, h->d_work, h->nwork, dd, dU, k, h->storage, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(dd);
    if (dU) hipFree(dU);
    if (e != hipSuccess) return fail(h, EKF_ERR_HIP, "load_lowrank_state", e);
    return EKF_OK;
}

namespace {
struct CkptHeader {
    char magic[8];
    int64_t N;
    int32_t tile, storage, world, rank;
    int64_t tile_bytes;      // bytes of the tile section
    int64_t reserved[3];
};
static_assert(sizeof(CkptHeader) == 64, "checkpoint header is 64 bytes");

// device -> file / file -> device through a bounded pinned staging buffer
int32_t stream_out(ekf_handle *h, FILE *f, const void *dev, size_t bytes, void *stage, size_t stage_bytes) {
    const char *p = (const char *)dev;
    while (bytes) {
        const size_t n = bytes < stage_bytes ? bytes : stage_bytes;
        HIPCHK(h, hipMemcpyAsync(stage, p, n, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (fwrite(stage, 1, n, f) != n) return fail(h, EKF_ERR_STATE, "checkpoint: short write");
        p += n; bytes -= n;
    }
    return EKF_OK;
}
int32_t stream_in(ekf_handle *h, FILE *f, void *dev, size_t bytes, void *stage, size_t stage_bytes) {
    char *p = (char *)dev;
    while (bytes) {
        const size_t n = bytes < stage_bytes ? bytes : stage_bytes;
        if (fread(stage, 1, n, f) != n) return fail(h, EKF_ERR_STATE, "checkpoint: short read");
        HIPCHK(h, hipMemcpyAsync(p, stage, n, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        p += n; bytes -= n;
    }
    return EKF_OK;
}
}  // namespace

int32_t ekf_checkpoint_save(ekf_handle *h, const char *path) {
    if (!h || !path) return fail(h, EKF_ERR_INVALID_ARG, "checkpoint_save: null argument");
    int32_t rc = enter(h);
    if (rc) return rc;
    rc = flush_pending(h);
    if (rc) return rc;
    FILE *f = fopen(path, "wb");
    REQUIRE(h, f != nullptr, EKF_ERR_STATE, "checkpoint_save: cannot open the file for writing");
    const size_t stage_bytes = (size_t)32 << 20;
    void *stage = nullptr;
    if (hipHostMalloc(&stage, stage_bytes, hipHostMallocDefault) != hipSuccess) { fclose(f); return fail(h, EKF_ERR_HIP, "checkpoint: staging buffer"); }
    const int64_t nmm = n_mm(h), nt = ekf_tiles_for(nmm, h->T);
    CkptHeader hd;
    memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, "EKFSLAM2", 8);
    hd.N = h->N; hd.tile = h->T; hd.storage = h->storage; hd.world = h->cfg.world; hd.rank = h->cfg.rank;
    hd.tile_bytes = h->st.tm.slots_for_rows(nt) * (int64_t)h->T * h->T * (int64_t)elt_size(h);
    rc = fwrite(&hd, sizeof hd, 1, f) == 1 ? EKF_OK : fail(h, EKF_ERR_STATE, "checkpoint: short write");
    if (!rc) rc = stream_out(h, f, h->st.x[h->cur], (size_t)(3 + nmm) * 8, stage, stage_bytes);
    if (!rc && h->N > 0) rc = stream_out(h, f, h->st.s, (size_t)h->N * 8, stage, stage_bytes);
    if (!rc) rc = stream_out(h, f, h->st.prr[h->cur], 9 * 8, stage, stage_bytes);
    for (int r = 0; r < 3 && !rc && nmm > 0; ++r)
        rc = stream_out(h, f, h->st.strip[h->cur] + (size_t)r * h->st.ldm, (size_t)nmm * 8, stage, stage_bytes);
    if (!rc && h->N > 0) rc = stream_out(h, f, h->st.diag[h->st.dcur], (size_t)(3 * h->N) * 8, stage, stage_bytes);
    if (!rc && hd.tile_bytes > 0) rc = stream_out(h, f, h->st.tiles, (size_t)hd.tile_bytes, stage, stage_bytes);
    hipHostFree(stage);
    if (fclose(f) != 0 && !rc) rc = fail(h, EKF_ERR_STATE, "checkpoint: close failed");
    return rc;
}

int32_t ekf_checkpoint_load(ekf_handle *h, const char *path) {
    if (!h || !path) return fail(h, EKF_ERR_INVALID_ARG, "checkpoint_load: null argument");
    int32_t rc = enter(h);
    if (rc) return rc;
    FILE *f = fopen(path, "rb");
    REQUIRE(h, f != nullptr, EKF_ERR_STATE, "checkpoint_load: cannot open the file");
    void *stage = nullptr;
    // every exit below goes through here: the file is closed and the staging buffer released whatever happened
    auto done = [&](int32_t status) { if (stage) hipHostFree(stage); fclose(f); return status; };
    CkptHeader hd;
    if (fread(&hd, sizeof hd, 1, f) != 1) return done(fail(h, EKF_ERR_STATE, "checkpoint_load: not an EKFSLAM2 file"));
    if (memcmp(hd.magic, "EKFSLAM1", 8) == 0)
        return done(fail(h, EKF_ERR_STATE, "checkpoint_load: EKFSLAM1 file -- that format (no section for the landmarks' live diagonal "
                                           "blocks) is no longer read; re-save the state with this library (INTEGRATION.md, checkpoints)"));
    if (memcmp(hd.magic, "EKFSLAM2", 8) != 0) return done(fail(h, EKF_ERR_STATE, "checkpoint_load: not an EKFSLAM2 file"));
    if (hd.tile != h->T || hd.storage != h->storage || hd.world != h->cfg.world || hd.rank != h->cfg.rank || hd.N < 0 || hd.N > h->cap)
        return done(fail(h, EKF_ERR_STATE, "checkpoint_load: tile edge, storage, shard or capacity do not match this handle"));
    const int64_t nmm = 2 * hd.N, nt = ekf_tiles_for(nmm, h->T);
    if (hd.tile_bytes != h->st.tm.slots_for_rows(nt) * (int64_t)h->T * h->T * (int64_t)elt_size(h))
        return done(fail(h, EKF_ERR_STATE, "checkpoint_load: tile section size mismatch"));
    // the whole payload must be there BEFORE any device state is overwritten: a truncated file leaves the handle as it was
    const int64_t payload = (3 + nmm) * 8 + hd.N * 8 + 9 * 8 + 3 * nmm * 8 + 3 * hd.N * 8 + hd.tile_bytes;
    if (fseek(f, 0, SEEK_END) != 0) return done(fail(h, EKF_ERR_STATE, "checkpoint_load: cannot seek"));
    const long fsize = ftell(f);
    if (fsize < 0 || (int64_t)fsize != (int64_t)sizeof hd + payload)
        return done(fail(h, EKF_ERR_STATE, "checkpoint_load: file length does not match its header (truncated?)"));
    if (fseek(f, (long)sizeof hd, SEEK_SET) != 0) return done(fail(h, EKF_ERR_STATE, "checkpoint_load: cannot seek"));
    const size_t stage_bytes = (size_t)32 << 20;
    if (hipHostMalloc(&stage, stage_bytes, hipHostMallocDefault) != hipSuccess) { stage = nullptr; return done(fail(h, EKF_ERR_HIP, "checkpoint: staging buffer")); }
    rc = retire_inflight(h);
    if (rc) return done(rc);
    h->npend = 0; h->pstart = 0; h->pf_valid = false; h->nx_valid = false; h->have_pp = false;
    hipError_t e = clear_pairs(h);
    if (e != hipSuccess) return done(fail(h, EKF_ERR_HIP, "checkpoint_load: clearing the pending pairs", e));
    std::vector<double> shost((size_t)hd.N);
    rc = stream_in(h, f, h->st.x[h->cur], (size_t)(3 + nmm) * 8, stage, stage_bytes);
    if (!rc && hd.N > 0) {
        const long at = ftell(f);
        rc = stream_in(h, f, h->st.s, (size_t)hd.N * 8, stage, stage_bytes);
        if (!rc) { fseek(f, at, SEEK_SET); if (fread(shost.data(), 8, (size_t)hd.N, f) != (size_t)hd.N) rc = fail(h, EKF_ERR_STATE, "checkpoint: short read"); }
    }
    if (!rc) rc = stream_in(h, f, h->st.prr[h->cur], 9 * 8, stage, stage_bytes);
    for (int r = 0; r < 3 && !rc && nmm > 0; ++r)
        rc = stream_in(h, f, h->st.strip[h->cur] + (size_t)r * h->st.ldm, (size_t)nmm * 8, stage, stage_bytes);
    if (!rc && hd.N > 0) rc = stream_in(h, f, h->st.diag[h->st.dcur], (size_t)(3 * hd.N) * 8, stage, stage_bytes);
    if (!rc && hd.tile_bytes > 0) rc = stream_in(h, f, h->st.tiles, (size_t)hd.tile_bytes, stage, stage_bytes);
    // N follows x even when a later section failed (an I/O error mid-way): x and N must never disagree
    h->N = hd.N;
    h->s_host = shost;
    h->s_sorted_ok = false;
    h->work_rows = -1;
    return done(rc);
}

int32_t ekf_P_digest(ekf_handle *h, double out[3]) {
    if (!h || !out) return fail(h, EKF_ERR_INVALID_ARG, "P_digest: null argument");
    int32_t rc = enter(h);
    if (rc) return rc;
    rc = flush_pending(h);
    if (rc) return rc;
    rc = refresh_work(h);
    if (rc) return rc;
    HIPCHK(h, launch_digest(h->st, h->cur, n_mm(h), h->d_work, h->nwork, h->d_digest, h->storage, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_small, h->d_digest, 3 * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    out[0] = h->h_small[0]; out[1] = h->h_small[1]; out[2] = h->h_small[2];
    return EKF_OK;
}

int32_t ekf_device_bytes(ekf_handle *h, int64_t *bytes) {
    if (!h || !bytes) return fail(h, EKF_ERR_INVALID_ARG, "device_bytes: null argument");
    *bytes = h->bytes;
    return EKF_OK;
}

int32_t ekf_kernel_timing_enable(ekf_handle *h, int32_t which, int32_t on) {
    if (!h || which < 0 || which >= EKF_KERNEL_COUNT) return fail(h, EKF_ERR_INVALID_ARG, "kernel_timing_enable: bad kernel id");
    int32_t rc = use_device(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    KernelTimer &t = h->timers[which];
    if (on) {
        // create the event pool now: hipEventCreate inside a timed region costs tens of microseconds per launch
        const size_t reserve = on > 512 ? (size_t)on : 512;
        while (t.ev.size() < 2 * reserve) {
            hipEvent_t e;
            HIPCHK(h, hipEventCreate(&e));
            t.ev.push_back(e);
        }
    }
    t.enabled = on != 0;
    t.used = 0;
    return EKF_OK;
}

int32_t ekf_kernel_timing_read(ekf_handle *h, int32_t which, int64_t *launches, double *total_ms) {
    if (!h || which < 0 || which >= EKF_KERNEL_COUNT || !launches || !total_ms)
        return fail(h, EKF_ERR_INVALID_ARG, "kernel_timing_read: bad argument");
    int32_t rc = use_device(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->flush_stream) HIPCHK(h, hipStreamSynchronize(h->flush_stream));
    KernelTimer &t = h->timers[which];
    double tot = 0.0;
    for (size_t i = 0; i + 1 < t.used; i += 2) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]));
        tot += ms;
    }
    *launches = (int64_t)(t.used / 2);
    *total_ms = tot;
    t.used = 0;
    return EKF_OK;
}

const char *ekf_downdate_kernel_name(const ekf_handle *h, int32_t *pairs) {
    if (!h) return "";
    if (pairs) *pairs = h->dd_pairs;
    return h->dd_kernel;
}

int32_t ekf_downdate_algorithmic_bytes(ekf_handle *h, int64_t *bytes) {
    if (!h || !bytes) return fail(h, EKF_ERR_INVALID_ARG, "downdate_algorithmic_bytes: null argument");
    const int64_t n = 3 + n_mm(h);
    *bytes = (int64_t)elt_size(h) * n * (n + 1);
    return EKF_OK;
}

}  // extern "C"
