// Host-side launch interface of the gfx950 kernels (implemented in kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "layout.h"

// Tuning switches.  The PRODUCT library compiles every one of them to its default constant: no environment variable changes
// which kernel a handle runs.  Only a -DEKF_TUNING build (make -C ekf_slam_amd/csrc tuning -> libekfslam_tuning.so, used by
// scripts/ab_*.sh and scripts/tune_*.py through EKF_LIB_PATH) reads them from the environment.
#ifdef EKF_TUNING
#include <stdlib.h>
inline int ekf_tune_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
#else
constexpr int ekf_tune_int(const char *, int dflt) { return dflt; }
#endif

// Device-resident filter state.  Passed BY VALUE to every kernel.
//   x / prr / strip are double-buffered: every kernel reads buffer `cur` and writes a complete buffer
//   `cur ^ 1`, so no workgroup ever reads a value another workgroup of the same launch overwrites.
struct DevState {
    double *x[2];      // state vector, 3 + ldm
    double *prr[2];    // P(1:3,1:3), row-major 3x3
    double *strip[2];  // P(1:3, 4:end): 3 rows of ldm
    void   *tiles;     // local tile store of the landmark block (double or float)
    double *s;         // signatures, cap
    // Pending rank-2 pairs: slot i holds G_i = H_s P(S,:) over landmark columns, interleaved (G(1,c), G(2,c)),
    // and K_i over landmark rows, interleaved (K(r,1), K(r,2)); slots are pair_stride doubles apart.  The tiles
    // hold P_base; the live landmark block is P_base - sum_i K_i G_i (applied in slot order).
    // The slots form a ring of `pcap` entries: a kernel that is told (pstart, npend) sees the pairs in slots
    // (pstart + i) mod pcap, i = 0 .. npend-1, oldest first.
    double *Gp;           // pending G pairs, then (same allocation) ...
    double *Kp;           // ... the pending K pairs: Kp = Gp + pcap * pair_stride (k_gather relies on a 32-bit offset between them)
    float  *Gp32;         // cfg.pass_arith = EKF_ARITH_F32 only (nullptr otherwise): float copies of the same pairs in the same slots, written by the
    float  *Kp32;         // gather beside the F64 ones for the F32-arithmetic pass (half the operand bytes) -- PLANAR (slot s: plane x = ldm floats, then
                          // plane y) and, for K, NEGATED (-(float)K, exact): the pass's LDS-DMA pieces are plain copies (flush32_pipe.h)
    int64_t pair_stride;   // 2 * ldm
    int32_t pcap;          // slots in the ring
    double *small;     // 32: Gr[2][3] (0..5), Kr[3][2] (6..11), Q[9] (12..20)
    int64_t ldm;       // strip leading dimension = landmark-block capacity rounded up to T
    TileMap tm;
    // The 2x2 DIAGONAL blocks of the landmark block, kept LIVE in F64 beside the tiles: landmark k's (P(2k,2k), P(2k+1,2k), P(2k+1,2k+1))
    // at diag[dcur][3k .. 3k+2].  Every correction's gather kernel applies its own pair to them at once (each column lane holds K(c,:)
    // and G(:,c)) -- the chain base - sum_i K_i G_i in slot order that a reader of the tiles would have to re-run over the pending pairs,
    // so in F64 the values equal the patched tile entries bit for bit -- reading buffer dcur, writing dcur ^ 1 (flipped per correction,
    // independently of `cur`: a predict does not touch them).  Why: (1) whoever needs a diagonal block (the correction's own solve, the
    // association of every landmark) reads three doubles instead of patching a chain of pending pairs; (2) with F32 tiles these are the
    // LARGE entries of P (an appended, not yet re-observed landmark: ~17 against a bulk of 0.1) whose sub-ulp updates a float store loses
    // -- in F64 they lose nothing.  Replicated on every shard (the gather is).  The tiles' own copies of these entries keep being
    // maintained by the passes but are no longer read.
    double *diag[2];
    int32_t dcur;
};

struct CorrectArgs {
    double z0, z1;            // [range, bearing_deg]
    double R00, R01, R10, R11;
    int64_t j;                // landmark-block row of the corrected landmark (2*idx)
    int64_t n_mm;             // active landmark-block size (2N)
    int32_t cur;
    int32_t npend;            // pending pairs before this correction; its own pair goes to ring position npend
    int32_t pstart;           // ring slot of the oldest pending pair
};

struct PredictArgs {
    double u0, u1, C;
    int64_t n_mm;
    int32_t cur;
};

struct AppendArgs {
    double u0, u1;
    double R00, R01, R10, R11;
    double pos0, pos1, signature;
    int64_t N;                // landmarks before the append
    int32_t cur;
};

struct AssocArgs {
    double z0, z1, z2;
    double R00, R01, R10, R11;
    double s_cost, s_thresh, w_pos;
    int64_t N;
    int32_t cur;
    int32_t npend;
    int32_t pstart;
    int32_t own_only;         // sharded association with an exchange: nominate only landmarks whose diagonal block this shard holds
};

struct AssocDecision {        // written by the device, read back by the host
    int64_t index;            // 0-based; == N for a new landmark
    int32_t is_new;
    int32_t seq;              // launch sequence number, written LAST (the host may poll a mapped copy for it)
    double  min_ll;
};

// One workgroup's winner, as it lands in MAPPED HOST memory: 16 bytes written by ONE store instruction of one lane (payload and
// sequence number arrive together; the host reads `seq` first, then the payload).  index: 0-based, -1 = nothing passed the threshold.
struct alignas(16) AssocHostPartial {
    double  min_ll;
    int32_t index;
    int32_t tag;       // launch sequence number + assoc_part_mix(payload): see below
};
// The 16 bytes leave the GPU in one store instruction and have been observed to land whole, but that is not an architectural
// promise -- so the entry validates itself: tag = seq + mix(payload).  A reader that sees a torn entry (new tag, old payload or
// the reverse) computes a sequence number that is not the one it waits for (up to a 2^-32 coincidence) and simply polls again.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t assoc_part_mix(uint32_t ll_lo, uint32_t ll_hi, uint32_t index) {
    uint32_t m = ll_lo * 0x9E3779B1u ^ (ll_hi + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (index + 0x165667B1u) * 0xC2B2AE35u;
    m ^= m >> 15; m *= 0x2C1B3C6Du; m ^= m >> 13;
    return m;
}

// Device-resident measure() loop (EKF_SLAM_UC.m:107-151 without a host round trip per observation): the association decision of
// an observation is PRODUCED on the device (k_associate, or the epilogue of the previous observation's k_gather) as one winner per
// workgroup, and CONSUMED on the device by the next launch (k_gather takes its landmark from the arg-min over those winners;
// k_append checks that nothing passed the threshold).  The host only learns the decisions afterwards, from `rec`.
struct DevLoopArgs {
    const AssocHostPartial *parts_in;   // DEVICE: per-workgroup winners of THIS observation's association; nullptr: not in use
    AssocHostPartial *rec;              // MAPPED HOST: the decision this launch consumed, one self-validating 16-byte store
                                        //   (index: landmark 0-based, -1 = new landmark, -2 = a winner entry did not carry seq_in)
    AssocHostPartial *parts_out;        // DEVICE: winners of the NEXT observation's association, evaluated in k_gather's epilogue
                                        //   on the state this correction leaves (one entry per k_gather workgroup); nullptr: none
    int32_t nblk_in, seq_in;            // entries of parts_in and the launch number they must carry
    int32_t seq_rec, seq_out;           // launch numbers stamped on rec / parts_out
    double z0, z1, z2;                  // the next observation [range, bearing_deg, signature] and its R
    double R00, R01, R10, R11;
    double s_cost, s_thresh, w_pos;
};

constexpr int kAssocBlock = 256;       // 4 wavefronts = one per SIMD: the per-landmark solve is a dependent f64 chain (1024 measured slower: 16 wavefronts share one CU's f64 issue)

hipError_t launch_predict(const DevState &st, const PredictArgs &a, int storage, hipStream_t s);
// dl != nullptr (device-resident measure loop): the kernel also reduces dl->parts_in and records the decision in dl->rec
hipError_t launch_append(const DevState &st, const AppendArgs &a, int storage, hipStream_t s, const DevLoopArgs *dl = nullptr,
                         const PredictArgs *fused_predict = nullptr);
// fused_predict != nullptr folds predict(u) into the correction (one launch instead of two, identical arithmetic)
// fuse_downdate: the kernel also applies its pair to the landmark block (small maps: a.n_mm <= gather_fuse_max_rows(), one
// workgroup); the pair is then NOT written to the pending ring and no downdate launch must follow
hipError_t launch_gather(const DevState &st, const CorrectArgs &a, const PredictArgs *fused_predict, int storage,
                         hipStream_t s, bool fuse_downdate);
int gather_fuse_max_rows();
// device-resident measure loop: the corrected landmark is the arg-min over dl.parts_in (a.j is the fallback that keeps a launch
// whose winners name no landmark inside the state); dl.parts_out != nullptr adds the next observation's association
hipError_t launch_gather_devloop(const DevState &st, const CorrectArgs &a, const PredictArgs *fused_predict, const DevLoopArgs &dl,
                                 int storage, hipStream_t s);
int64_t gather_workgroups(const DevState &st, int64_t n_mm);
// sharded correction: (1) every shard copies the chunks of the landmark row-panel P(j:j+1,:) it owns into `send`
// (slab layout: local chunk kl of T columns, interleaved pairs), (2) the slabs are all-gathered into `recv`
// (world slabs of `slab` doubles), (3) the gather/solve kernel reads the panel from `recv` instead of the tiles.
struct RowList { int32_t m; int32_t j[64]; };      // landmark-block rows (2 * landmark index) of one prefetch

// base row-panels of m <= 64 landmarks (0-based indices idx) into send + q * slab, one launch
hipError_t launch_rowpanel_base(const DevState &st, const int64_t *idx, int m, int64_t n_mm, double *send, int64_t slab,
                                int storage, hipStream_t s);
// the row-panels of m <= 64 landmarks as the tiles will hold them after the pass that applies the npend pending pairs, into send + q * slab
hipError_t launch_rowpanel_next(const DevState &st, const int64_t *idx, int m, int64_t n_mm, int pstart, int npend, double *send,
                                int64_t slab, int storage, hipStream_t s);
hipError_t launch_rowpanel(const DevState &st, int64_t j, int64_t n_mm, int pstart, int npend, double *send, int storage,
                           hipStream_t s);
// device-resident measure loop on a shard: the row-panel of the landmark the DEVICE's association names (dl.parts_in); j, the host
// mirror's prediction, only when the winners name nothing inside the state
hipError_t launch_rowpanel_dev(const DevState &st, int64_t j, int64_t n_mm, int pstart, int npend, double *send, int storage,
                               hipStream_t s, const DevLoopArgs &dl);
// recv: `world` contributions `rank_stride` doubles apart; this correction's row-panel starts `offset` doubles into
// each; patched: the pending pairs are already applied to it (k_rowpanel) -- otherwise the gather applies them
// dl != nullptr (device-resident measure loop on a shard): as launch_gather_devloop, on the exchanged row-panel
hipError_t launch_gather_sharded(const DevState &st, const CorrectArgs &a, const PredictArgs *fused_predict, const double *recv,
                                 int64_t rank_stride, int64_t offset, bool patched, int storage, hipStream_t s,
                                 const DevLoopArgs *dl = nullptr);
// tiles -= sum_{i < npairs} K_i G_i (in slot order) over the work list (I,J pairs, device array) of `nwork` owned
// lower-triangle tiles: ONE pass over P for npairs update-steps
// work_xcd / xcd_len: the same tiles as 8 per-XCD streams (stream x = work_xcd[x*xcd_len ..), padded with (-1,-1)),
// used when several pairs are applied so that each XCD's K/G working set stays inside its own L2
// dst: tile store the result is written to (== st.tiles for an in-place flush; a second buffer for the asynchronous
// flush, which must not disturb kernels still reading st.tiles); pstart: ring slot of the first pair
// kname: nullptr, or 64 bytes that receive the name of the kernel instance that was launched ("k_flush_mfma<double,128,8>")
// nx (sharded handles; nullptr or j < 0: none): ALSO extract the row-panel of landmark-block rows nx->j, nx->j + 1 into nx->send (layout
// of launch_rowpanel) from the updated entries; *extracted tells whether the instance that was launched did it (one pair per
// launch, wavefront-per-row tile shapes only)
struct NextRow { int64_t j; double *send; };
// what the strip form of the F32-arithmetic pass needs besides the tiles and the pairs (flush32_pipe.h; nullptr: that form is not used)
struct PassAux {
    const int4 *segs;      // strip work list: nsegs segments of ekf_pipe32::kSeg entries (strip_entry), 8 interleaved per-XCD streams
    int64_t nsegs;         // a multiple of 8
    float *dump;           // kDumpFloats floats per workgroup of the pass's grid
    int grid;              // workgroups the dump area was sized for (one per CU)
    int64_t cols;          // rows / columns of the landmark block the active tile rows cover (a multiple of 256): what k_split_pairs cuts
    uint16_t *Kb3, *Gb3;   // cfg.pass_arith = EKF_ARITH_SPLIT3 only (nullptr otherwise): the bf16 planes of the pending pairs, cut in front
                           // of each pass (flush32_split.h: split_plane_elems(ldm) elements each)
};
size_t pass_split_plane_elems(int64_t ldm);     // uint16_t elements of PassAux::Kb3 (and of Gb3) for a landmark block of leading dimension ldm
hipError_t launch_downdate(const DevState &st, void *dst, const int2 *work, int64_t nwork, const int2 *work_xcd, int64_t xcd_len,
                           int pstart, int npairs, int storage, int grid_cap, hipStream_t s, char *kname, const NextRow *nx = nullptr,
                           bool *extracted = nullptr, int arith = 0, const PassAux *aux = nullptr);
// strip work list of the owned lower-triangle tiles of nt tile rows (host side; entries for flush32_pipe.h::k_flush_strip32): returns
// the segment count (a multiple of 8) and fills `out` with nsegs * kSeg entries
int64_t build_strip_segments(const TileMap &tm, int64_t nt, std::vector<int4> &out);
// pos_cost / sig_cost: device arrays of N or nullptr; partial: device scratch of >= ceil(N/kAssocBlock) entries; ticket: a
// device int, zero between launches (the last workgroup to finish reduces the partials and resets it: one launch, no
// finishing kernel); decision: device copy.  host_partials != nullptr (mapped host memory, one entry per workgroup): NO cross-workgroup
// step on the device at all -- every workgroup stores its winner there and the HOST takes the arg-min over the ceil(N / kAssocBlock)
// entries once each carries `seq` (the ticket + release / acquire hand-over it replaces was ~4 of the kernel's 9 us)
// fused_predict != nullptr: the launch is ALSO k_predict -- the recorded predict(u) is applied to what the kernel reads and the
// predicted pose / Prr / strip / Q are written to state buffer a.cur ^ 1 (the caller flips its buffer index afterwards)
hipError_t launch_associate(const DevState &st, const AssocArgs &a, double *pos_cost, double *sig_cost,
                            AssocDecision *partial, int *ticket, AssocDecision *decision, AssocHostPartial *host_partials, int seq,
                            double *cand, int storage, hipStream_t s, const PredictArgs *fused_predict = nullptr);
// cand (nullptr or 4 device doubles): this shard's candidate {likelihood, index or -1, 0, 0} for the all-gather of a sharded
// association; launch_assoc_merge takes the arg-min over the `world` gathered contributions of `count` doubles each (candidate,
// then -- want_costs -- N position costs) and writes the decision like launch_associate does
hipError_t launch_assoc_merge(const DevState &st, const double *recv, int world, int64_t count, int64_t N, bool want_costs,
                              double *pos_cost, AssocDecision *decision, AssocDecision *host_decision, int seq, hipStream_t s);
// dense (column-major, n x n, device) <-> tiled
// cfg.async_flush: rows [r0, r1) of the landmark block (every local tile of the tile rows they lie in) copied from one tile store to the other
hipError_t launch_copy_rows(const TileMap &tm, const void *src, void *dst, int64_t r0, int64_t r1, int storage, hipStream_t s);
hipError_t launch_unpack_dense(const DevState &st, int cur, int64_t n_mm, double *dense, int storage, hipStream_t s);
hipError_t launch_pack_dense(const DevState &st, int cur, int64_t n_mm, const double *dense, int storage, hipStream_t s);
hipError_t launch_get_block(const DevState &st, int cur, int64_t r0, int64_t c0, int64_t nr, int64_t nc,
                            double *out, int storage, hipStream_t s);
// out (device, 4 * (N+1) doubles): P(1:2,1:2), then the 2x2 diagonal block of every landmark, each column-major
hipError_t launch_get_diag_blocks(const DevState &st, int cur, int64_t N, double *out, int storage, hipStream_t s);
// P = diag(d) + U U' (d: n, U: n x k column-major, device)
hipError_t launch_lowrank(const DevState &st, int cur, int64_t n_mm, const int2 *work, int64_t nwork, const double *d,
                          const double *U, int64_t k, int storage, hipStream_t s);
// out (device, kDigestDoubles doubles; [0..2] = trace, sum and sum of squares over the lower triangle, then a ticket and one
// slot of partial sums per workgroup, added in a fixed order: equal states give bit-equal digests)
constexpr int kDigestGrid = 2048;
constexpr int kDigestDoubles = 4 + 3 * kDigestGrid;
hipError_t launch_digest(const DevState &st, int cur, int64_t n_mm, const int2 *work, int64_t nwork, double *out,
                         int storage, hipStream_t s);
