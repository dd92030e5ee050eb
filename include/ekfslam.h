/*
 * libekfslam -- C ABI of the MI355X-native EKF-SLAM update engine.
 *
 * The reference (SamShue/EKF_SLAM) is pure MATLAB and has NO plugin / operator / FFI interface; its
 * boundary is the class-method surface of EKF_SLAM.m / EKF_SLAM_UC.m / Correspondence.m / append.m.  Each
 * entry point below names the reference method it replaces (file:line in the reference tree).  A MEX
 * gateway (matlab/ekfslam_mex.c) and a ctypes binding (ekf_slam_amd/_lib.py) bind exactly these symbols.
 *
 * Conventions
 *   - every call returns an int32 status (EKF_OK == 0) and never throws across the ABI;
 *     ekf_last_error(h) gives the message of the last failure on that handle;
 *   - arrays are caller-owned HOST buffers of IEEE doubles; matrices are COLUMN-MAJOR (MATLAB native);
 *   - landmark indices are 0-BASED here (the MEX / Python layers convert from the reference's 1-based);
 *   - angles are degrees, exactly as in the reference;
 *   - a handle is not thread-safe; all work is queued on the handle's HIP stream and calls that return
 *     data synchronise that stream, the others are asynchronous;
 *   - state lives in HBM: x (3+2N), s (N) and P in a tiled symmetric block layout (3x3 robot block,
 *     3 x 2N robot/landmark strip, T x T tiles of the lower block triangle of the landmark block).
 */
#ifndef EKFSLAM_H
#define EKFSLAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EKF_ABI_VERSION 1

enum {
    EKF_OK = 0,
    EKF_ERR_INVALID_ARG = 1,
    EKF_ERR_NO_DEVICE = 2,   /* no HIP device / HIP runtime failure at create */
    EKF_ERR_HIP = 3,         /* a HIP call failed; see ekf_last_error         */
    EKF_ERR_CAPACITY = 4,    /* append beyond capacity_landmarks              */
    EKF_ERR_INDEX = 5,       /* landmark index outside the state              */
    EKF_ERR_LOOKUP = 6,      /* landmark-table lookup did not match exactly one entry (MATLAB would error) */
    EKF_ERR_STATE = 7,       /* call not valid in the handle's current state  */
    EKF_ERR_COMM = 8         /* multi-GPU exchange failed                     */
};

enum { EKF_MODE_KNOWN = 0,   /* EKF_SLAM.m    : known correspondence   */
       EKF_MODE_UC = 1 };    /* EKF_SLAM_UC.m : unknown correspondence */

enum { EKF_STORE_F64 = 0,    /* P tiles stored as double                               */
       EKF_STORE_F32 = 1 };  /* P tiles stored as float, every solve still in double   */
enum { EKF_ARITH_F64 = 0,    /* the pass over P forms P - sum K_i G_i in double (one rounding per pass when the tiles are float) */
       EKF_ARITH_F32 = 1,    /* F32 tiles with tile = 256 only: the pass runs on the f32 matrix pipe (cfg.pass_arith below)      */
       EKF_ARITH_SPLIT3 = 2 };/* as EKF_ARITH_F32, every float operand cut exactly into three bfloat16 pieces: bf16 matrix pipe  */

/* Hard-coded property defaults of the reference collected in one struct
 * (EKF_SLAM.m:12-16, EKF_SLAM_UC.m:13,16). */
typedef struct ekf_config {
    double  C;                   /* process-noise constant              EKF_SLAM.m:12               */
    double  Rc[2];               /* measurement-noise constants         EKF_SLAM.m:13 / _UC.m:13    */
    double  s_cost;              /* signature cost                      EKF_SLAM.m:14 / _UC.m:16    */
    double  s_thresh;            /* new-landmark threshold              EKF_SLAM.m:16 / _UC.m:16    */
    double  w_pos;               /* weight of the Mahalanobis position cost in the association
                                    likelihood; 0 reproduces the live line Correspondence.m:75,
                                    1 the commented-out line Correspondence.m:74                    */
    int64_t capacity_landmarks;  /* HBM is sized for this many landmarks (streaming append never reallocates) */
    int32_t mode;                /* EKF_MODE_*  (selects the ekf_measure dispatch)                  */
    int32_t storage;             /* EKF_STORE_*                                                     */
    int32_t device;              /* HIP device ordinal                                              */
    int32_t tile;                /* tile edge T in elements: 16, 32, 64, 128 (256 for F32 storage);
                                    0 = default (128 for F64 storage, 256 for F32 storage)           */
    int32_t rank;                /* shard rank  (0 when world == 1)                                 */
    int32_t world;               /* number of shards P is split over; 0 or 1 = unsharded            */
    int32_t batch;               /* deferred downdate: up to `batch` corrections are kept as pending rank-2
                                    pairs (the rows later corrections need are patched on the fly) and applied
                                    to P in ONE pass; with F64 tiles results are bit-identical to batch = 1 (F32 tiles round once
                                    per pass, so the batch moves the roundings: equal within the F32 tolerance).  0 or 1 = every
                                    correction rewrites P immediately (EKF_SLAM.m:145 as written); max 64  */
    int32_t async_flush;         /* run each pass over P on a second stream, from the current tile store into a
                                    second one (2x tile memory), while the next corrections go on reading the current
                                    store plus all pending pairs; stores swap at the next batch boundary.  With
                                    batch = 1 that is the as-written update-step software-pipelined: the pass of step i
                                    beside the gather (and, sharded, the exchange) of step i + 1.  Same bits as
                                    async_flush = 0 with F64 tiles (with F32 tiles the corrections beside a pass read its pairs
                                    unrounded where the synchronous engine reads the rounded tiles: equal within the F32 tolerance).  The second stream is confined to a CU mask that leaves 32 CUs
                                    (64 with EKF_ARITH_SPLIT3) to the main stream.  ekf_append beside a pass does not wait for it
                                    (the new rows reach the second store when the pass retires).  Pays when a batch's steps take
                                    about as long as its pass; with batch = 1 it has measured slower than the in-place
                                    pass at every map size on one GPU (two stores defeat the cache). */
    int32_t device_assoc;        /* EKF_MODE_UC, ekf_measure (EKF_SLAM_UC.m:107-151), when w_pos == 0 (the reference's live likelihood is
                                    signature-only, Correspondence.m:75, so WHETHER a row appends or corrects is a function of z(3) and
                                    s alone and the host's mirror of s can predict it):
                                    3 (default of EKF_MODE_UC): the device-resident loop.  Every observation's association runs on
                                       the device (per-landmark phi_k, Mahalanobis and signature cost, thresholded arg-min:
                                       Correspondence.m:49-87 as the reference evaluates it) and its decision is CONSUMED on the
                                       device: the correction's gather kernel takes its landmark from the association's winners, and
                                       evaluates the next observation's association in its own epilogue (one launch per
                                       observation); an append checks that nothing passed the threshold.  The host queues all m rows
                                       without a single wait -- which branch it queues is its mirror's prediction -- and the decisions
                                       the device took come back as records that are checked against the prediction later: the
                                       next ekf_measure checks what has landed, every call that synchronises or reads or loads state
                                       (ekf_sync, ekf_get_*, ekf_set_*, ...) checks the rest first and returns EKF_ERR_STATE on a
                                       mismatch (it cannot happen unless s was changed behind the library's back).  On a sharded
                                       handle the same loop runs on every shard (the association reads replicated data only): a
                                       correction extracts the row-panel of the landmark the device names, exchanges it (the library's
                                       communicator or the hook of transport (d)) and gathers on the exchanged panel.
                                    0: the decision is taken from the host mirror of s -- no association launch at all;
                                    1: the association kernel runs for every observation and the host WAITS for its decision;
                                    2: the kernel runs for every observation, the host dispatches on its mirror's decision (passed
                                       to the gather kernel as an argument) and VERIFIES every device decision before ekf_measure
                                       returns (EKF_ERR_STATE on a mismatch).
                                    With w_pos != 0 the branch cannot be predicted: the device decides and is waited for (as 1)
                                    whatever this says; ekf_associate() always runs on the device and waits. */
    int32_t pass_direction;      /* the pass over P: 0 (default) = every other pass walks its work list backwards when the shard's
                                    tile store exceeds the 256 MiB Infinity Cache (what one pass wrote last the next reads first,
                                    on-die), forwards otherwise; 1 = always forwards; 2 = always alternate.  Same bits. */
    int32_t force_sharded;       /* 1: run the sharded code path (row-panel extraction, exchange, sharded gather) although
                                    world == 1 -- how that path is exercised and timed on a single GPU */
    int32_t pass_arith;          /* EKF_ARITH_*: arithmetic of the pass over P.  EKF_ARITH_F32 ("F32 mixed precision with F64 innovation
                                    solve", BASELINE.json configs[4]) needs storage = EKF_STORE_F32 and tile = 256 (EKF_ERR_INVALID_ARG
                                    otherwise): the pass's update -sum K_i G_i is formed from float copies of the pending pairs and
                                    summed in float on the f32 matrix pipe (three times the f64 pipe's measured rate), then added to the
                                    float tile value ONCE -- an entry still sees one rounding at its own magnitude per pass, as with
                                    EKF_ARITH_F64.  Passes of one or two pairs are purely HBM-bound and keep the F64-arithmetic kernel.
                                    The innovation, S, K, x, the robot block, the strip and the landmarks' 2x2 diagonal blocks are F64 as
                                    always.  Costs pcap x 4 N floats for the copies.  Measured against the F64 engine: DESIGN.md section 5
                                    (the whole configs[4] workload, 40 000 -> 50 000 landmarks: 2e-8).
                                    EKF_ARITH_SPLIT3 (same preconditions): the same float copies, each cut EXACTLY into three bfloat16
                                    pieces (3 x 8 significant bits) in front of the pass; a product is the sum of the six partial
                                    products that matter (what is dropped is at most 2^-23 of it, 0.09 x 2^-24 in the root mean square),
                                    each exact in float, summed in float on the bf16 matrix pipe from zero, then added to the tile
                                    value once.  Same error class as EKF_ARITH_F32 (a float sum of 2m terms; measured against an F64
                                    sum beside the fmaf chain: DESIGN.md section 5), NOT the same bits; at 28-64 pending pairs the
                                    pass is then bound by HBM instead of the f32 matrix pipe.  Up to 27 pairs: EKF_ARITH_F32's kernels
                                    (faster there).
                                    Costs 2 x 768 bytes per row of P for the planes. */
    int32_t reserved[2];
} ekf_config;

typedef struct ekf_handle ekf_handle;

/* ---- library ---- */
int32_t     ekf_abi_version(void);
const char *ekf_status_string(int32_t status);
/* Fill *cfg with the reference's property defaults for `mode` (Rc = [.01,5] known, [.1,5] UC). */
int32_t     ekf_config_default(ekf_config *cfg, int32_t mode);

/* ---- lifecycle: EKF_SLAM() / EKF_SLAM_UC() constructors (EKF_SLAM.m:26-34, EKF_SLAM_UC.m:27-36):
 *      x = [0 0 0], P = 0.1*eye(3), s = [] ---- */
int32_t     ekf_create(const ekf_config *cfg, ekf_handle **out);
int32_t     ekf_destroy(ekf_handle *h);
const char *ekf_last_error(const ekf_handle *h);
/* Use an existing hipStream_t (e.g. torch's current stream) for all subsequent work; NULL restores the
 * handle's own stream. */
int32_t     ekf_set_stream(ekf_handle *h, void *hip_stream);
int32_t     ekf_sync(ekf_handle *h);
/* Apply all pending rank-2 pairs to P now (no-op when nothing is pending).  Every call that reads P
 * (ekf_get_P, ekf_get_P_block, ekf_P_digest) does this itself. */
int32_t     ekf_flush(ekf_handle *h);
int32_t     ekf_pending(ekf_handle *h, int32_t *npending);
/* The reference's tunables are public properties that may be reassigned at any time (EKF_SLAM.m:12-16):
 * update C, Rc, s_cost, s_thresh, w_pos of a live handle. */
int32_t     ekf_set_params(ekf_handle *h, double C, const double Rc[2], double s_cost, double s_thresh, double w_pos);

/* ---- hot path ---- */
/* predict(h,u)  EKF_SLAM.m:40-51 (EKF_SLAM_UC.m:42-53): u = [dD, dTheta_deg]. */
int32_t ekf_predict(ekf_handle *h, const double u[2]);

/* [x_new,F] = f(h,x,u)  EKF_SLAM.m:56-65: pure host function on caller arrays (public method of the
 * reference).  x, x_new: n doubles; F: n x n column-major or NULL. */
int32_t ekf_motion_model(const double *x, int64_t n, const double u[2], double *x_new, double *F);

/* append(h,u,R,landmarkPos,signature)  EKF_SLAM.m:67-98 (EKF_SLAM_UC.m:69-100).  R: 2x2 column-major. */
int32_t ekf_append(ekf_handle *h, const double u[2], const double R[4], const double pos[2], double signature);

/* Correction body of measure()  EKF_SLAM.m:124-145 (EKF_SLAM_UC.m:125-146) for landmark `idx` (0-based):
 * innovation, H_k, phi_k, K, x += K nu, P = (I - K H_k) P.  z = [range, bearing_deg]. */
int32_t ekf_correct(ekf_handle *h, const double z[2], const double R[4], int64_t idx);

/* [newLL,index] = estimateCorrespondence(h,z,R,x,P,s)  Correspondence.m:28-88 on the handle's x, P, s.
 * z = [range, bearing_deg, signature].  *idx is 0-based (== N for a new landmark).  pos_cost / sig_cost:
 * optional N-element outputs (Correspondence.m:69,71), may be NULL. */
int32_t ekf_associate(ekf_handle *h, const double z[3], const double R[4], int32_t *is_new, int64_t *idx,
                      double *pos_cost, double *sig_cost);
/* On a sharded handle (cfg.world > 1) ekf_associate / ekf_measure need NO exchange, whatever w_pos: signatures, x, the robot block and
 * the strip are replicated, and so are the landmarks' own 2x2 diagonal blocks (live F64 copies that every correction's gather kernel
 * updates on every shard) -- every shard scores every landmark and takes the same decision, the unsharded handle's bit for bit.
 * A second protocol exists for hosts that exchange candidates instead: every shard scores the landmarks whose diagonal TILE it
 * holds, the candidates {likelihood, index} -- and the position costs, if asked for -- travel in one all-gather of 4 (+ N) doubles per
 * shard run by the caller between _begin and _finish (ekf_exchange_info, ekf_exchange_local), and every shard takes the same strict
 * arg-min (lowest likelihood, lowest index on ties). */
int32_t ekf_associate_begin(ekf_handle *h, const double z[3], const double R[4], int32_t want_costs);
int32_t ekf_associate_finish(ekf_handle *h, int32_t *is_new, int64_t *idx, double *pos_cost, double *sig_cost);

/* measure(h,laserData,u,landmark_list) AFTER the landmark front-end has run, i.e. the loop
 * EKF_SLAM.m:105-150 / EKF_SLAM_UC.m:107-151 over observed_LL (m x 3 column-major [range, bearing_deg, index]).
 * The landmark struct array the loop looks `loc` up in (landmark_list.landmarkObj.landmark(k).index/.loc,
 * EKF_SLAM.m:111,120) is passed as lm_index (L) and lm_loc (L x 2 column-major).  Dispatch follows
 * cfg.mode, including the reference's quirks: the empty-map row only appends (signature 1), the
 * known-correspondence branch corrects landmark ii (the row number, EKF_SLAM.m:123), R = diag(z1*Rc1, z2*Rc2). */
int32_t ekf_measure(ekf_handle *h, const double *observed_LL, int64_t m, const double u[2],
                    const double *lm_index, const double *lm_loc, int64_t L);

/* Sharded handles, cfg.batch = 1 (every correction rewrites P at once): tell the handle which
 * landmark (0-based) the NEXT ekf_correct will name.  The pass over P that ends the current correction then also extracts that
 * landmark's row-panel into the exchange area (its entries are in registers anyway), so the next update-step starts with its
 * all-gather instead of an extraction launch (with the library's own communicator and buffers that all-gather runs in place).  A hint holds for one correction; a wrong or
 * missing one only costs the extraction launch back.  Results are bit-identical either way.  No-op on other handles. */
int32_t ekf_hint_next(ekf_handle *h, int64_t idx);

/* ---- multi-GPU: P split over `world` shards (cfg.rank / cfg.world), tile (I,J) on shard (I+J) mod world ----
 * x, s, the robot block and the robot/landmark strip are replicated; predict, append and associate need no
 * exchange.  A correction needs the 2 x 2N landmark row-panel P(j:j+1,:), whose T-wide chunk k lives on shard
 * (tile_row(j) + k) mod world: one equal-count all-gather per update-step.  Three ways to run it:
 *   (a) ekf_comm_init: the library owns an RCCL communicator (one process per GPU); ekf_correct / ekf_measure
 *       then run extract -> ncclAllGather -> solve -> downdate on the handle's stream;
 *   (b) ekf_correct_begin, the caller's own all-gather over the device buffers of ekf_exchange_info (e.g.
 *       torch.distributed on buffers given through ekf_exchange_set_buffers), ekf_correct_finish;
 *   (c) ekf_exchange_local: one host thread driving every shard of the filter in ONE process (the way a
 *       MATLAB host would): begin on all handles, ekf_exchange_local, finish on all handles;
 *   (d) ekf_exchange_set_hook: the caller's all-gather as a callback, run wherever (a) would run ncclAllGather -- inside
 *       ekf_correct, ekf_prefetch_rows and, which (b) and (c) cannot do, in the middle of ekf_measure's loop.  The
 *       hook finds the handle between begin and finish (ekf_exchange_info names buffers and count) and returns 0 or
 *       an error; with one host thread per shard it is a barrier, ekf_exchange_local on one thread, a barrier
 *       (ekf_slam_amd/sharding.py: ShardGroup.measure). */
typedef struct ekf_comm_id { char internal[128]; } ekf_comm_id;   /* == ncclUniqueId */
int32_t ekf_comm_unique_id(ekf_comm_id *id);                      /* rank 0 creates it, the host broadcasts it */
int32_t ekf_comm_init(ekf_handle *h, const ekf_comm_id *id);      /* collective over all shards */
int32_t ekf_exchange_set_hook(ekf_handle *h, int32_t (*hook)(void *ctx), void *ctx);   /* hook == NULL removes it */
int32_t ekf_correct_begin(ekf_handle *h, const double z[2], const double R[4], int64_t idx);
int32_t ekf_correct_finish(ekf_handle *h);
/* Latency hiding for hosts that know which landmarks the next corrections touch (a scan's observation list):
 * all-gather the BASE row-panels of up to cfg.batch landmarks in ONE exchange; later corrections on them run with no
 * exchange of their own (the pending pairs are applied inside the gather kernel).  The prefetch is dropped when P is
 * rewritten (a flush) or the map grows.  ekf_prefetch_rows = begin + ncclAllGather + finish (transport (a)); begin /
 * finish bracket the caller's all-gather for transports (b) and (c).  No-op on an unsharded handle. */
int32_t ekf_prefetch_rows(ekf_handle *h, const int64_t *idx, int32_t m);
/* The same for the batch AFTER the current one, announced while the current one is still being recorded: when the current batch
 * completes, the row-panels of these landmarks are extracted AS THE BATCH'S PASS WILL LEAVE THEM (pending pairs applied in slot
 * order, rounded through the storage type -- bit for bit what a prefetch after the pass would read) and exchanged in front of the
 * pass: the next batch starts with its prefetch in place, and the exchange no longer depends on the pass (a build that runs it on a
 * stream of its own beside the pass exists behind a tuning switch; on one GPU it is slower, abi.hip: flush_pending).  Dropped if the map grows before the batch completes.  Needs cfg.batch > 1, a synchronous
 * flush and cfg.pass_arith = EKF_ARITH_F64 (EKF_ERR_STATE otherwise); m = 0 withdraws an announcement; no-op on an unsharded handle. */
int32_t ekf_prefetch_next(ekf_handle *h, const int64_t *idx, int32_t m);
int32_t ekf_prefetch_begin(ekf_handle *h, const int64_t *idx, int32_t m);
int32_t ekf_prefetch_finish(ekf_handle *h);
/* Device pointers of the exchange: send area (*count doubles valid for the pending begin) and receive area (world
 * contributions of *count doubles, contribution r from shard r); *count_capacity = largest count at capacity
 * (cfg.batch row-panels, or an association's candidate + one position cost per landmark, whichever is larger). */
int32_t ekf_exchange_info(ekf_handle *h, void **send, void **recv, int64_t *count, int64_t *count_capacity);
/* Use caller-owned device buffers (>= count_capacity and world * count_capacity doubles); NULL restores the own ones. */
int32_t ekf_exchange_set_buffers(ekf_handle *h, void *send, void *recv);
int32_t ekf_exchange_local(ekf_handle **shards, int32_t world);
/* Host-only descriptions of the shard plan (no GPU needed): owner of tile (I,J); its slot in the owner's tile
 * store; which shard / local chunk supplies chunk `chunk` of the row-panel of a landmark in tile row tile_row_j. */
int32_t ekf_shard_owner(int32_t world, int64_t I, int64_t J);
int64_t ekf_shard_slot(int32_t world, int64_t I, int64_t J);
int32_t ekf_shard_panel_source(int32_t world, int64_t tile_row_j, int64_t chunk, int32_t *owner, int64_t *local_chunk);

/* ---- state access (the reference's public properties x, P, Q, s; EKF_SLAM.m:6-9) ---- */
int32_t ekf_num_landmarks(ekf_handle *h, int64_t *N);
int32_t ekf_get_x(ekf_handle *h, double *x /* 3+2N */);
int32_t ekf_set_x(ekf_handle *h, const double *x, int64_t n);
int32_t ekf_get_s(ekf_handle *h, double *s /* N */);
int32_t ekf_set_s(ekf_handle *h, const double *s, int64_t N);
/* Diagnostic -- a fault injector for tests of the device-resident measure loop's verification, of no use to a host: overwrites the DEVICE copy
 * of signature idx (0-based) and leaves the host mirror alone.  The next ekf_measure whose association involves that landmark then queues its
 * launches from a prediction the device contradicts; every launch stays inside the state (a correction falls back to the predicted landmark,
 * a predicted append appends), ekf_measure itself returns EKF_OK (it waits for nothing),
 * and the FIRST synchronising call afterwards (ekf_sync, any getter, ekf_flush ...) returns EKF_ERR_STATE once, with the decision and the
 * prediction in ekf_last_error: from there on the state is no longer the reference's -- reload it (ekf_set_x / _P / _s, a checkpoint). */
int32_t ekf_diag_poke_device_signature(ekf_handle *h, int64_t idx, double value);
/* Dense n x n column-major P.  set_P stores the lower triangle (P is a covariance: symmetric).  On a shard
 * (world > 1) get_P / get_P_block return NaN for landmark-block entries held by another shard. */
int32_t ekf_get_P(ekf_handle *h, double *P);
int32_t ekf_set_P(ekf_handle *h, const double *P, int64_t n);
/* P(r0:r0+nr-1, c0:c0+nc-1) into out (nr x nc column-major): what plot() reads (EKF_SLAM.m:180,205). */
int32_t ekf_get_P_block(ekf_handle *h, int64_t r0, int64_t c0, int64_t nr, int64_t nc, double *out);
/* Everything plot() reads of P in ONE call (EKF_SLAM.m:180 robotSigma, :205 landmarkSigma): out holds 4 * (N+1) doubles --
 * P(1:2,1:2), then the 2x2 diagonal block of landmark 1..N, each column-major.  NaN for blocks held by another shard. */
int32_t ekf_get_P_diag_blocks(ekf_handle *h, double *out /* 4*(N+1) */);
/* The 3x3 non-zero block of the last predict's Q (EKF_SLAM.m:43-44), column-major. */
int32_t ekf_get_Q(ekf_handle *h, double Q[9]);
/* Bulk state load used to start large benchmarks: sets N landmarks, x (3+2N), s (N) and
 * P = diag(d) + U U' with d (3+2N) > 0 and U ((3+2N) x k column-major), built tile by tile on the device. */
int32_t ekf_load_lowrank_state(ekf_handle *h, int64_t N, const double *x, const double *s,
                               const double *d, const double *U, int64_t k);
/* Order-independent digests of P computed on the device over the unique (lower-triangle) entries:
 * out[0] = trace, out[1] = sum of the lower triangle incl. diagonal, out[2] = sum of squares of it. */
int32_t ekf_P_digest(ekf_handle *h, double out[3]);
/* Bytes of HBM held by the handle (all buffers). */
int32_t ekf_device_bytes(ekf_handle *h, int64_t *bytes);

/* ---- checkpoint (the reference has none; its whole state is the four properties x, P, Q, s, EKF_SLAM.m:6-9) ----
 * Binary file: 64-byte header (magic "EKFSLAM2", N, tile, storage, world, rank), then x, s, the robot block, the
 * robot/landmark strip, the landmarks' live F64 diagonal blocks (3 doubles each) and this handle's tiles of the active
 * tile rows, bit for bit (pending pairs are flushed first).  Loading needs a handle with the same tile edge, storage type
 * and shard (rank/world) and capacity >= N; a sharded filter is one file per shard.  Files with the magic "EKFSLAM1"
 * (no diagonal-block section) are rejected with EKF_ERR_STATE and a message that says so: there is no migration. */
int32_t ekf_checkpoint_save(ekf_handle *h, const char *path);
int32_t ekf_checkpoint_load(ekf_handle *h, const char *path);

/* ---- measurement hooks ---- */
enum { EKF_KERNEL_DOWNDATE = 0, EKF_KERNEL_GATHER = 1, EKF_KERNEL_PREDICT = 2, EKF_KERNEL_ASSOCIATE = 3,
       EKF_KERNEL_APPEND = 4,
       EKF_KERNEL_ROWPANEL = 5,   /* sharded handles: the extraction of a correction's (or a prefetch's) row-panels into the send area */
       EKF_KERNEL_EXCHANGE = 6,   /* sharded handles with ekf_comm_init: the all-gather on the library's communicator */
       EKF_KERNEL_COUNT = 7 };
/* Bracket every launch of kernel `which` with HIP events on the handle's stream (on != 0; on > 512 also reserves
 * event pairs for that many launches between two reads, so that none is created inside a timed region) and read the
 * accumulated launch count and device time; reading synchronises the stream and resets the counters. */
int32_t ekf_kernel_timing_enable(ekf_handle *h, int32_t which, int32_t on);
int32_t ekf_kernel_timing_read(ekf_handle *h, int32_t which, int64_t *launches, double *total_ms);
/* Name of the kernel instance the LAST downdate / flush launch of this handle used, as the launcher chose it
 * (e.g. "k_downdate_w<double,128,4,false>", "k_flush_mfma<double,128,8>"), and the number of pending pairs it applied
 * (*pairs, may be NULL); "" before the first launch.  The string is owned by the handle. */
const char *ekf_downdate_kernel_name(const ekf_handle *h, int32_t *pairs);
/* Algorithmic bytes one launch of the downdate kernel moves at the current N: every unique entry of the
 * symmetric P read once and written once = w * n * (n+1), n = 3+2N. */
int32_t ekf_downdate_algorithmic_bytes(ekf_handle *h, int64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif /* EKFSLAM_H */
