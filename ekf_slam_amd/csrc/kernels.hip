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
#include <cstdio>
#include <cstdlib>

#include "device_math.h"
#include "flush32_mfma.h"     // kBlock, ring_slot, k_flush_mfma32
#include "flush32_pipe.h"     // k_flush_pipe32

namespace {


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

// ---------------------------------------------------------------------------------------------------
// predict: one thread per strip column; thread 0 also owns the pose, Prr and Q
// ---------------------------------------------------------------------------------------------------
// The pose / robot-block part of predict, shared by k_predict and the predict-fused gather so that both give
// bit-identical results.  in: pose[3], M = Prr (row-major); out: fa = F(1,3), fb = F(2,3), new pose, Prr' , Q.
struct PredictSmall { double fa, fb; double pose[3]; double prr[9]; double Q[9]; };

// The only pieces whose results depend on how the compiler expands them are the libm calls (inlined copies of
// sin/cos/atan2 gave different last bits in different kernels).  They live in noinline wrappers -- ONE machine-code
// body shared by every kernel -- so that the standalone predict, the predict folded into a correction and the
// association kernel cannot differ by a rounding.  Everything else is plain IEEE arithmetic (-ffp-contract=off).
// (results by VALUE: reference parameters of a noinline function live on the stack, i.e. in scratch memory -- a global-memory
//  round trip in the middle of the latency chain)
__device__ __attribute__((noinline)) double2 sincosd_ni(double a) { double sn, cs; ekfm::sincosd(a, sn, cs); return make_double2(sn, cs); }
__device__ __attribute__((noinline)) double bearing_ni(double d1, double d0, double th) {
    return ekfm::wrapTo360(ekfm::atan2d(d1, d0) - th);                                // EKF_SLAM.m:130
}

// Per-entry forms of predict: ONE definition of every expression, used by the serial composition below (k_predict, one lane)
// and by the lane-parallel one in k_gather (one entry per lane) -- so the two cannot differ by a rounding.
// F(1,3), F(2,3) use the PRE-motion heading, no pi/180 (EKF_SLAM.m:63-64); W = [u1 cosd th; u1 sind th; u2] (EKF_SLAM.m:42)
__device__ __forceinline__ void predict_common(double u0, double u1, double sn, double cs, double &fa, double &fb, double W[3]) {
    fa = -1 * u0 * sn;
    fb = u0 * cs;
    W[0] = u0 * cs; W[1] = u0 * sn; W[2] = u1;
}
// (F*P)(i,c) for the 3x3 robot block from column c of P = (x0, x1, x2): rows 1, 2 pick up F(.,3) * P(3,c).  Operands by VALUE so
// that a lane-parallel caller hands over values it selected, with no indexed access to a register array (scratch) or to LDS.
__device__ __forceinline__ double predict_fp(int i, double x0, double x1, double x2, double fa, double fb) {
    const double r0 = x0 + fa * x2, r1 = x1 + fb * x2;           // both rows formed, then SELECTED (EKF_SEL: v_cndmask, no branches)
    return EKF_SEL(i == 0) ? r0 : (EKF_SEL(i == 1) ? r1 : x2);
}
// entry (i,j) of F*Prr*F' + Q and of Q = (W*C)*W'  (EKF_SLAM.m:44,47); cj / c2 = columns j and 3 of Prr, wi / wj = W(i), W(j)
__device__ __forceinline__ void predict_prr_entry(int i, int j, const double cj[3], const double c2[3], double fa, double fb, double wi,
                                                  double wj, double C, double &out, double &q) {
    const double m1 = predict_fp(i, cj[0], cj[1], cj[2], fa, fb), p2 = predict_fp(i, c2[0], c2[1], c2[2], fa, fb);
    const double c0 = m1 + fa * p2, c1 = m1 + fb * p2;
    const double m2 = EKF_SEL(j == 0) ? c0 : (EKF_SEL(j == 1) ? c1 : m1);                   // (F*P)*F'
    q = (wi * C) * wj;
    out = m2 + q;
}
// new pose entry i  (EKF_SLAM.m:58-60,50)
__device__ __forceinline__ double predict_pose_entry(const double pose[3], int i, double u0, double u1, double sn2, double cs2) {
    return i == 0 ? pose[0] + u0 * cs2 : i == 1 ? pose[1] + u0 * sn2 : ekfm::wrapTo360(pose[2] + u1);
}

// predict, given sind/cosd of the pre-motion heading (sn, cs) and of heading + u2 (sn2, cs2): serial composition
__device__ __forceinline__ void predict_finish(const double pose[3], const double prr_in[9], double u0, double u1, double C,
                                               double sn, double cs, double sn2, double cs2, PredictSmall &o) {
    double W[3];
    predict_common(u0, u1, sn, cs, o.fa, o.fb, W);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // the robot block is kept EXACTLY symmetric: entry (i,j) and its mirror are both the lower-triangle entry's value (see the
            // correction's Prr update in k_gather for why)
            const int a = i > j ? i : j, b = i > j ? j : i;
            const double cj[3] = { prr_in[b], prr_in[3 + b], prr_in[6 + b] }, c2[3] = { prr_in[2], prr_in[5], prr_in[8] };
            predict_prr_entry(a, b, cj, c2, o.fa, o.fb, W[a], W[b], C, o.prr[3 * i + j], o.Q[3 * i + j]);
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) o.pose[i] = predict_pose_entry(pose, i, u0, u1, sn2, cs2);
}

__device__ __forceinline__ void predict_small(const double pose[3], const double prr_in[9], double u0, double u1, double C,
                                              PredictSmall &o) {
    double sn, cs, sn2, cs2;
    const double2 sc = sincosd_ni(pose[2]), sc2 = sincosd_ni(pose[2] + u1);
    sn = sc.x; cs = sc.y; sn2 = sc2.x; cs2 = sc2.y;
    predict_finish(pose, prr_in, u0, u1, C, sn, cs, sn2, cs2, o);
}

// strip column under F*P: (F*P)(1,:) = P(1,:) + F(1,3) P(3,:), (F*P)(2,:) = P(2,:) + F(2,3) P(3,:)
__device__ __forceinline__ void predict_strip(double &s0, double &s1, double s2, double fa, double fb) {
    s0 = fma(fa, s2, s0);
    s1 = fma(fb, s2, s1);
}

__global__ __launch_bounds__(kBlock) void k_predict(DevState st, PredictArgs a) {
    __shared__ PredictSmall ps;
    const int cur = a.cur, nxt = cur ^ 1;
    const double *__restrict__ x = st.x[cur];
    double *__restrict__ xn = st.x[nxt];
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (threadIdx.x == 0) {
        const double pose[3] = { x[0], x[1], x[2] };
        double prr[9];
        for (int i = 0; i < 9; ++i) prr[i] = st.prr[cur][i];
        predict_small(pose, prr, a.u0, a.u1, a.C, ps);
    }
    __syncthreads();
    if (c < a.n_mm) {
        const double *__restrict__ s = st.strip[cur];
        double *__restrict__ sn = st.strip[nxt];
        double s0 = s[c], s1 = s[st.ldm + c];
        const double s2 = s[2 * st.ldm + c];
        predict_strip(s0, s1, s2, ps.fa, ps.fb);
        sn[c] = s0;
        sn[st.ldm + c] = s1;
        sn[2 * st.ldm + c] = s2;
        xn[3 + c] = x[3 + c];
    }
    if (c == 0) {
        for (int i = 0; i < 9; ++i) { st.prr[nxt][i] = ps.prr[i]; st.small[12 + i] = ps.Q[i]; }
        for (int i = 0; i < 3; ++i) xn[i] = ps.pose[i];
    }
}

// The strip part of P <- F P F' as a panel product on the F64 matrix cores: (F P)(1:3, landmark columns) =
// F_rr (3x3) * strip (3 x 2N).  One v_mfma_f64_16x16x4_f64 per wavefront and 16 columns: A = F_rr zero-padded to
// 16x4 (lane l holds A[l&15][l>>4]), B = a 4x16 slice of the strip with a zero 4th row (lane l holds
// B[l>>4][l&15]), D row (l>>4) + 4*reg, column l&15 -> register 0 of lanes 0..47 is the new 3x16 slice.
// The f64 MFMA is a k-ordered chain of correctly rounded FMAs (scripts/probes/mfma_f64_order.*): with the unit / F(1:2,3)
// operands above it computes fma(fa, s2, fma(0, s1, fma(1, s0, 0))) = fma(fa, s2, s0), i.e. exactly predict_strip() -- the
// standalone and the fused predict agree bit for bit (tests/test_deferred_gpu.py).  The panel is 3 x 2N and costs < 1 % of
// an update-step, so this is about using the matrix unit for a GEMM-shaped piece of the path, not about speed.
typedef double mfma_f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kBlock) void k_predict_mfma(DevState st, PredictArgs a) {
    __shared__ PredictSmall ps;
    const int cur = a.cur, nxt = cur ^ 1;
    const double *__restrict__ x = st.x[cur];
    double *__restrict__ xn = st.x[nxt];
    if (threadIdx.x == 0) {
        const double pose[3] = { x[0], x[1], x[2] };
        double prr[9];
        for (int i = 0; i < 9; ++i) prr[i] = st.prr[cur][i];
        predict_small(pose, prr, a.u0, a.u1, a.C, ps);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, jj = lane & 15;                 // k index of the A/B operands, column within the slice
    // A[i][k], i = lane & 15 (rows 3..15 are zero padding), k = lane >> 4
    double av = 0.0;
    if (jj < 3) {
        if (kk == jj) av = 1.0;
        else if (kk == 2 && jj == 0) av = ps.fa;
        else if (kk == 2 && jj == 1) av = ps.fb;
    }
    const double *__restrict__ s = st.strip[cur];
    double *__restrict__ sn = st.strip[nxt];
    const int64_t nslices = (a.n_mm + 15) / 16;
    for (int64_t sl = (int64_t)blockIdx.x * 4 + wave; sl < nslices; sl += (int64_t)gridDim.x * 4) {
        const int64_t c = sl * 16 + jj;
        const bool live = c < a.n_mm;
        const double bv = (kk < 3 && live) ? s[kk * st.ldm + c] : 0.0;
        const mfma_f64x4 zero = { 0.0, 0.0, 0.0, 0.0 };
        const mfma_f64x4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, zero, 0, 0, 0);
        if (kk < 3 && live) sn[kk * st.ldm + c] = d[0];
        if (kk == 3 && live) xn[3 + c] = x[3 + c];            // the otherwise idle quarter copies the landmark states
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int i = 0; i < 9; ++i) { st.prr[nxt][i] = ps.prr[i]; st.small[12 + i] = ps.Q[i]; }
        for (int i = 0; i < 3; ++i) xn[i] = ps.pose[i];
    }
}

// ---------------------------------------------------------------------------------------------------
// association order and the per-workgroup winner entries (used by k_associate, and by the kernels of the device-resident
// measure loop that consume a decision: k_gather, k_append)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool assoc_better(double la, int64_t ia, double lb, int64_t ib) {
    // strict '<' on the likelihood, first (lowest) index wins ties (Correspondence.m:81)
    return la < lb || (la == lb && ia < ib);
}

// arg-min over a wavefront under assoc_better's order (a total order: every lane ends with the same winner)
__device__ __forceinline__ void wave_argmin(double &ll, int64_t &ix) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ol = __shfl_xor(ll, off);
        const int64_t oi = __shfl_xor((long long)ix, off);
        if (assoc_better(ol, oi, ll, ix)) { ll = ol; ix = oi; }
    }
}

// The same arg-min when few lanes hold a candidate at all (lanes without one carry ix == INT64_MAX).  With the reference's live
// likelihood (signature cost only, Correspondence.m:75) at most the landmarks whose signature lies within the threshold of z(3)
// are candidates -- normally ONE in the whole map -- so the butterfly (6 steps of four ds_bpermute each, ~900 clocks at the tail of
// a latency chain) is replaced by a ballot and, for a single candidate, two v_readlane.  Same result in every case.
__device__ __forceinline__ void wave_argmin_sparse(double &ll, int64_t &ix) {
    const unsigned long long m = __ballot(ix != INT64_MAX);
    if (m == 0ull) { ll = INFINITY; ix = INT64_MAX; return; }                      // (wave-uniform branches)
    if ((m & (m - 1ull)) == 0ull) {
        const int src = __ffsll((long long)m) - 1;
        const int lo = __builtin_amdgcn_readlane(__double2loint(ll), src), hi = __builtin_amdgcn_readlane(__double2hiint(ll), src);
        const int il = __builtin_amdgcn_readlane((int)(ix & 0xffffffffll), src), ih = __builtin_amdgcn_readlane((int)(ix >> 32), src);
        ll = __hiloint2double(hi, lo);
        ix = ((int64_t)ih << 32) | (int64_t)(uint32_t)il;
        return;
    }
    wave_argmin(ll, ix);
}

// One self-validating 16-byte entry (kernels.h: AssocHostPartial): payload and launch number in ONE store instruction.
__device__ __forceinline__ void store_partial(AssocHostPartial *dst, double ll, int index, int seq) {
    typedef int part_v4 __attribute__((ext_vector_type(4)));
    part_v4 v;
    const long long lb = __double_as_longlong(ll);
    v.x = (int)(lb & 0xffffffffll); v.y = (int)(lb >> 32);
    v.z = index;
    v.w = (int)((uint32_t)seq + assoc_part_mix((uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z));
    *reinterpret_cast<part_v4 *>(dst) = v;
}

// Correspondence.m:78-85 over the per-workgroup winners of one association launch (lowest likelihood, lowest index on ties --
// the order of the kernel's own reductions), by ONE wavefront; every lane returns the same (ll, ix):
// ix >= 0 the matched landmark, -1 nothing passed the threshold (new landmark), -2 an entry does not carry launch number `seq`.
__device__ __forceinline__ void reduce_partials_wave(const AssocHostPartial *__restrict__ parts, int nblk, int seq, int lane,
                                                     double &ll, int &ix) {
    typedef int part_v4 __attribute__((ext_vector_type(4)));
    double bl = INFINITY;
    int64_t bi = INT64_MAX;
    int bad = 0;
    for (int b = lane; b < nblk; b += 64) {
        const part_v4 v = *reinterpret_cast<const part_v4 *>(parts + b);
        const int got = (int)((uint32_t)v.w - assoc_part_mix((uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z));
        const double pl = __longlong_as_double(((long long)v.y << 32) | (long long)(uint32_t)v.x);
        if (got != seq) bad = 1;
        else if (v.z >= 0 && assoc_better(pl, (int64_t)v.z, bl, bi)) { bl = pl; bi = v.z; }
    }
    wave_argmin_sparse(bl, bi);
    bad = __any(bad);
    ll = bl;
    ix = bad ? -2 : (bi == INT64_MAX ? -1 : (int)bi);
}

struct NoDevLoop {};
template <bool kDev> struct DevLoopParam { using type = NoDevLoop; };
template <> struct DevLoopParam<true> { using type = DevLoopArgs; };

// ---------------------------------------------------------------------------------------------------
// append: in place on buffer `cur` (only new slots are written)
// ---------------------------------------------------------------------------------------------------
// kPredict: a recorded predict(u) (ekf_predict is lazy) is carried out by THIS launch -- every workgroup's first lane runs the small 3x3 part
// (as k_predict does), every column lane predicts its strip column (predict_strip: the same two FMAs as k_predict / k_predict_mfma) and copies
// its x entry, everything is written to the other state buffer (a.cur ^ 1), and the append itself reads the predicted values: predict -> append
// costs one launch instead of two, same bits (tests/test_deferred_gpu.py).
template <typename TS, bool kPredict = false>
__global__ __launch_bounds__(kBlock) void k_append(DevState st, AppendArgs a, DevLoopArgs dl, PredictArgs pa) {
    __shared__ PredictSmall aps;
    const int cur = a.cur;
    const int out = kPredict ? (cur ^ 1) : cur;                  // the buffer this launch leaves the state in
    if constexpr (kPredict) {
        if (threadIdx.x == 0) {
            const double pose[3] = { st.x[cur][0], st.x[cur][1], st.x[cur][2] };
            double prr_in[9];
            for (int i = 0; i < 9; ++i) prr_in[i] = st.prr[cur][i];
            predict_small(pose, prr_in, pa.u0, pa.u1, pa.C, aps);
        }
        __syncthreads();
    }
    if (dl.parts_in != nullptr && blockIdx.x == 0 && threadIdx.x < 64) {
        // device-resident measure loop: the association of this observation must have found nothing below the threshold
        // (EKF_SLAM_UC.m:121); what it did find goes to the host's record
        double dll;
        int dix;
        reduce_partials_wave(dl.parts_in, dl.nblk_in, dl.seq_in, (int)threadIdx.x, dll, dix);
        if (threadIdx.x == 0) store_partial(dl.rec, dll, dix, dl.seq_rec);
    }
    double *__restrict__ x = st.x[out];
    double *__restrict__ s = st.strip[out];
    const double *__restrict__ prr = kPredict ? aps.prr : st.prr[cur];
    TS *__restrict__ tiles = (TS *)st.tiles;
    const int64_t n_mm = 2 * a.N;          // old landmark-block size; new rows are n_mm, n_mm + 1
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const double th = kPredict ? aps.pose[2] : x[2];             // post-predict heading (EKF_SLAM.m:84-85)
    const double jxr[2][3] = { { 1, 0, -a.u0 * ekfm::sind(th) }, { 0, 1, a.u0 * ekfm::cosd(th) } };
    if (c < n_mm) {
        // F: P(new, lm) = jxr * P(lm, 1:3)'   (EKF_SLAM.m:95); the column strip equals the row strip here
        const double *__restrict__ sin_ = st.strip[cur];
        double s0 = sin_[c], s1 = sin_[st.ldm + c];
        const double s2 = sin_[2 * st.ldm + c];
        if constexpr (kPredict) {
            predict_strip(s0, s1, s2, aps.fa, aps.fb);
            s[c] = s0; s[st.ldm + c] = s1; s[2 * st.ldm + c] = s2;
            x[3 + c] = st.x[cur][3 + c];
        }
        for (int i = 0; i < 2; ++i) {
            const double v = jxr[i][0] * s0 + jxr[i][1] * s1 + jxr[i][2] * s2;
            if (st.tm.mine((n_mm + i) >> st.tm.shift, c >> st.tm.shift))
                pmm_low_store<TS>(tiles, st.tm, n_mm + i, c, v);
        }
    }
    if (c == 0) {
        if constexpr (kPredict) {
            for (int i = 0; i < 9; ++i) { st.prr[out][i] = aps.prr[i]; st.small[12 + i] = aps.Q[i]; }
            for (int i = 0; i < 3; ++i) x[i] = aps.pose[i];
        }
        x[3 + n_mm] = a.pos0;                                                         // EKF_SLAM.m:79
        x[3 + n_mm + 1] = a.pos1;
        st.s[a.N] = a.signature;                                                      // EKF_SLAM.m:70
        const double jz[2][2] = { { ekfm::cosd(a.u1), -a.u0 * ekfm::sind(a.u1) },
                                  { ekfm::sind(a.u1),  a.u0 * ekfm::cosd(a.u1) } };   // EKF_SLAM.m:87-88
        const double R[2][2] = { { a.R00, a.R01 }, { a.R10, a.R11 } };
        double t[2][3], c1[2][2], t2[2][2], c2[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) {
            double acc = 0; for (int k = 0; k < 3; ++k) acc += jxr[i][k] * prr[3 * k + j]; t[i][j] = acc; }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
            double acc = 0; for (int k = 0; k < 3; ++k) acc += t[i][k] * jxr[j][k]; c1[i][j] = acc; }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
            double acc = 0; for (int k = 0; k < 2; ++k) acc += jz[i][k] * R[k][j]; t2[i][j] = acc; }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
            double acc = 0; for (int k = 0; k < 2; ++k) acc += t2[i][k] * jz[j][k]; c2[i][j] = acc; }
        // C: jxr*Prr*jxr' + jz*R*jz' (EKF_SLAM.m:91); only the lower triangle of the 2x2 block is canonical
        {
            double *__restrict__ dg = st.diag[st.dcur] + 3 * a.N;        // the new landmark's diagonal block, live F64 copy (every shard)
            dg[0] = c1[0][0] + c2[0][0]; dg[1] = c1[1][0] + c2[1][0]; dg[2] = c1[1][1] + c2[1][1];
        }
        if (st.tm.mine(n_mm >> st.tm.shift, n_mm >> st.tm.shift)) {
            pmm_low_store<TS>(tiles, st.tm, n_mm, n_mm, c1[0][0] + c2[0][0]);
            pmm_low_store<TS>(tiles, st.tm, n_mm + 1, n_mm, c1[1][0] + c2[1][0]);
            pmm_low_store<TS>(tiles, st.tm, n_mm + 1, n_mm + 1, c1[1][1] + c2[1][1]);
        }
        // I: P(1:3,new) = Prr*jxr' (EKF_SLAM.m:92); H is its mirror and shares the strip storage
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) {
            double acc = 0; for (int k = 0; k < 3; ++k) acc += prr[3 * i + k] * jxr[j][k];
            s[i * st.ldm + n_mm + j] = acc; }
    }
}

// ---------------------------------------------------------------------------------------------------
// gather + solve.  One thread per landmark column c:
//     G(:,c) = H_s * P(S,c),  K(c,:) = G(:,c)' * inv(phi),  x(c) += K(c,:) nu,  strip(:,c) -= K_r G(:,c)
// The 5x5 sub-block P(S,S) that phi needs is fetched by every workgroup (19 doubles, L2-resident), so
// there is no inter-workgroup dependency and the whole correction is two launches.
// ---------------------------------------------------------------------------------------------------
struct SmallSolve {
    double Hs[2][5];
    double Phi[4];     // inv(phi), row-major
    double nu[2];
    double Kr[3][2];
    double Gr[2][3];
};

// pss: 0..8 Prr row-major; 9+2t+b = P(t, j+b), t<3, b<2; 15+2t+b = canonical P(j+t, j+b); 19..21 x_r; 22..23 x_j
// the measurement Jacobian block H_s from delta = landmark - robot  (EKF_SLAM.m:125-127,137-138)
__device__ __forceinline__ void solve_hs(double d0, double d1, double &sq, double Hs[2][5]) {
    const double q = d0 * d0 + d1 * d1;
    sq = sqrt(q);
    const double iq = 1 / q;
    const double e[2][5] = { { -sq * d0, -sq * d1, 0, sq * d0, sq * d1 }, { d1, -d0, -q, -d1, d0 } };
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 5; ++b) Hs[a][b] = iq * e[a][b];
}

// Per-entry forms of the small solve (same idea as predict_*_entry above).  h / g are ROWS of H_s / G(:,S), so that a lane-parallel
// caller can select its row without indexing a register array dynamically (that would put the array in scratch memory);
// `pss` may be a register array (serial callers, static b) or the LDS copy (lane-parallel caller, b = f(lane)).
// G(a, S(b)) = H_s(a,:) * P(S, S(b))  (EKF_SLAM.m:141, first product); b < 3: robot columns (P(j+t, b) is stored as
// strip(b, j+t)), b >= 3: columns j, j+1
__device__ __forceinline__ double solve_gs_entry(const double *pss, const double h[5], int b) {
    double acc = 0;
    if (b < 3) {
        for (int t = 0; t < 3; ++t) acc += h[t] * pss[3 * t + b];
        for (int t = 0; t < 2; ++t) acc += h[3 + t] * pss[9 + 2 * b + t];
    } else {
        for (int t = 0; t < 3; ++t) acc += h[t] * pss[9 + 2 * t + (b - 3)];
        for (int t = 0; t < 2; ++t) acc += h[3 + t] * pss[15 + 2 * t + (b - 3)];
    }
    return acc;
}
// phi(a,b) = G(a,S) * H_s(b,:)' + R(a,b)  (EKF_SLAM.m:141)
__device__ __forceinline__ double solve_phi_entry(const double g[5], const double h[5], double Rab) {
    double acc = 0;
    for (int t = 0; t < 5; ++t) acc += g[t] * h[t];
    return acc + Rab;
}
// K_r(b,cc) = G_r(:,b)' * inv(phi)(:,cc)
__device__ __forceinline__ double solve_kr_entry(double g0b, double g1b, double phi_c, double phi_2c) {
    return g0b * phi_c + g1b * phi_2c;
}

// everything after H_s and the predicted measurement (zhat0 = range, zhat1 = bearing): G(:,S), phi, inv(phi), nu, K_r
__device__ __forceinline__ void solve_rest(const double *pss, double zhat0, double zhat1, double z0, double z1, double R00,
                                           double R01, double R10, double R11, SmallSolve &o) {
    double GS[2][5];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 5; ++b) {
            GS[a][b] = solve_gs_entry(pss, o.Hs[a], b);
            if (b < 3) o.Gr[a][b] = GS[a][b];
        }
    const double R[2][2] = { { R00, R01 }, { R10, R11 } };
    double phi[4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) phi[2 * a + b] = solve_phi_entry(GS[a], o.Hs[b], R[a][b]);   // :141
    ekfm::inv2(phi, o.Phi);                                                            // :143 phi_k^-1
    o.nu[0] = z0 - zhat0;                                                              // :144 (bearing NOT wrapped)
    o.nu[1] = z1 - zhat1;
    for (int b = 0; b < 3; ++b) for (int cc = 0; cc < 2; ++cc) o.Kr[b][cc] = solve_kr_entry(o.Gr[0][b], o.Gr[1][b], o.Phi[cc], o.Phi[2 + cc]);
}

// serial composition (association kernel: one lane per landmark)
__device__ __forceinline__ void solve_small(const double *pss, double z0, double z1, double R00, double R01, double R10,
                                            double R11, SmallSolve &o) {
    const double d0 = pss[22] - pss[19], d1 = pss[23] - pss[20];                       // EKF_SLAM.m:125-126
    double sq;
    solve_hs(d0, d1, sq, o.Hs);
    const double bearing = bearing_ni(d1, d0, pss[21]);
    solve_rest(pss, sq, bearing, z0, z1, R00, R01, R10, R11, o);
}

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

// Workgroup layout of k_gather: 256 column lanes (wavefronts 0-3) + three helper wavefronts that own no column:
//   wavefront 4  CHAIN     the small solve, one matrix ENTRY per lane (predict's 3x3, H_s, the 2x5 / 2x2 products, K_r)
//   wavefront 5  DIAG      the pending pairs on the landmark's own 2x2 block (a chain in slot order: inherently serial)
//   wavefront 6  BEARING   sincos of the new heading, atan2, the innovation nu
// Round 1 ran the whole solve on ONE lane of wavefront 0 (~400 dependent f64 operations, 4 700 clocks) behind a barrier that
// also waited for every column lane's loads and patches.  Now the solve depends only on the 24 small operands; the helpers
// meet the column lanes at ONE barrier, when K_r, H_s, inv(phi) and nu are in LDS.
constexpr int kGatherCols = 256;
constexpr int kGatherBlock = kGatherCols + 3 * 64;

__device__ __forceinline__ double lane_bcast(double v, int src) {        // value of lane `src` (compile-time) on every lane
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_gather(double v, int src) {       // value of lane `src` (per-lane index): ds_bpermute x 2
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// value of the neighbouring lane (lane ^ 1): a DPP quad permutation [1,0,3,2] -- two VALU moves, no LDS crossbar round trip (what
// __shfl_xor's ds_bpermute costs at the tail of this kernel's latency chain)
__device__ __forceinline__ double lane_xor1(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// LDS traffic of ONE wavefront is ordered; this only keeps the compiler from moving accesses across it and drains the queue
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// kFused (small maps, unsharded, batch 1: the whole landmark block fits ONE workgroup's columns): the rank-2 downdate of
// EKF_SLAM.m:145 runs at the end of this kernel instead of in a launch of its own -- same arithmetic (rank2_apply per element), one
// launch per update-step instead of two; the pair is handed over in LDS and never goes to the pending ring.
constexpr int kFuseMaxRows = 48;                     // landmark-block rows (24 landmarks) up to which the fused form is used (beyond: slower than two launches)
constexpr int kFuseElems = kFuseMaxRows * kFuseMaxRows / 256;     // elements of the block per column lane, all in flight together

// kDev (device-resident measure loop): the corrected landmark is not a kernel argument but the arg-min over the
// per-workgroup winners of this observation's association (dl.parts_in), reduced redundantly by every wavefront; and the NEXT
// observation's association (Correspondence.m:49-87: per-landmark phi_k, Mahalanobis + signature cost, thresholded arg-min) is
// evaluated in the epilogue by the column lanes, from the values this correction has just produced -- x', strip', Prr', the
// landmark's own 2x2 block (its live F64 copy, to which the lanes have just applied this correction's pair) -- with the per-entry
// functions k_associate uses: one launch per observation instead of two.
template <typename TS, bool kSharded, bool kPredict, bool kFused = false, bool kDev = false>
__global__ __launch_bounds__(kGatherBlock) void k_gather(DevState st, CorrectArgs a, PanelView pv, PredictArgs pa,
                                                         typename DevLoopParam<kDev>::type dl) {
    static_assert(!kDev || !kFused, "the device loop never drives the small-map fused form");
    __shared__ double pss[24];
    __shared__ SmallSolve sol;
    __shared__ PredictSmall ps;
    __shared__ double pose_sh[3];                   // the pose the correction starts from (predicted when predict is folded in);
                                                    // pss[19..21] keep the BASE pose: the BEARING wavefront reads it concurrently
    __shared__ int diag_ready;                      // DIAG -> CHAIN: the patched 2x2 block is in pss[15..18]
    __shared__ int staged_cnt;                      // column wavefronts that have written their share of `upatch` (0..4)
    __shared__ double2 upatch[kMaxPending * 4];     // per pending pair: K_i(j,:), K_i(j+1,:), G_i(:,j), G_i(:,j+1)
    const int tid = threadIdx.x;
    const int role = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0-3 columns, 4 chain, 5 diag, 6 bearing
    const int lane = tid & 63;
    const int cur = a.cur;
    // double-buffered state: both pointers of a pair arrive with the first kernel-argument fetch and are SELECTED (indexing the
    // by-value struct with `cur` makes the compiler fetch the pointer with a second, dependent scalar load)
    const double *__restrict__ x = cur ? st.x[1] : st.x[0];
    const double *__restrict__ strip = cur ? st.strip[1] : st.strip[0];
    const double *__restrict__ prr_cur = cur ? st.prr[1] : st.prr[0];
    double *__restrict__ x_nxt = cur ? st.x[0] : st.x[1];
    double *__restrict__ strip_nxt = cur ? st.strip[0] : st.strip[1];
    double *__restrict__ prr_nxt = cur ? st.prr[0] : st.prr[1];
    const TS *__restrict__ tiles = (const TS *)st.tiles;
    int64_t j = a.j;
    const int64_t ldm = st.ldm;
    const int npend = a.npend, pstart = a.pstart;
    if constexpr (kDev) {
        // EKF_SLAM_UC.m:119-125: idx comes from the association, on the device.  Every wavefront reduces the winners itself
        // (<= 64 entries: one 16-byte load per lane + a butterfly) -- no LDS, no barrier in front of the j-dependent loads.
        double dll;
        int dix;
        reduce_partials_wave(dl.parts_in, dl.nblk_in, dl.seq_in, lane, dll, dix);
        dix = __builtin_amdgcn_readfirstlane(dix);
        if (dix >= 0 && 2 * (int64_t)dix < a.n_mm) j = 2 * (int64_t)dix;       // otherwise a.j: the launch stays inside the state
        if constexpr (kSharded) pv.Ij = j >> st.tm.shift;                       // (k_rowpanel<.., kDev> laid the panel out for this landmark)
        if (blockIdx.x == 0 && tid == kGatherCols + 128) store_partial(dl.rec, dll, dix, dl.seq_rec);   // BEARING lane 0: it has slack
    }

    // Fetch every kernel argument this kernel uses NOW, in one burst of scalar loads: left to itself the compiler fetches
    // them lazily, right before their first use, which put three dependent round trips to the argument block at the head of
    // the kernel's critical path.
    {
        const double *t_tiles = (const double *)st.tiles;
        asm volatile("" :: "s"(st.ldm), "s"(st.pair_stride), "s"(st.pcap), "s"(st.Gp), "s"(st.Kp), "s"(t_tiles), "s"(st.small),
                     "s"(st.tm.T), "s"(st.tm.shift), "s"(st.tm.world), "s"(st.tm.rank), "s"(a.j), "s"(a.n_mm), "s"(a.npend),
                     "s"(a.pstart));
    }
    const bool do_patch = !kSharded || !pv.patched;       // base values in hand: apply the pending pairs here
#ifdef EKF_GATHER_STAMPS
    long long stamp[12]; int nst = 0;
#define EKF_STAMP() do { stamp[nst++] = clock64(); } while (0)
    EKF_STAMP();
#else
#define EKF_STAMP() do { } while (0)
#endif

    if (role >= 4) {
        // =========================================== helper wavefronts ===========================================
        // (1h) the small operands, one per lane: the CHAIN wavefront takes what the previous kernel wrote a moment ago (robot block,
        //      strip columns j, j+1, pose, landmark: 20 doubles, cache-resident), the DIAG wavefront the landmark's own 2x2 block
        //      (the live F64 copy) -- two wavefronts, two load queues, so the chain's sincos starts when the
        //      POSE has arrived, not when the slowest of 24 loads has (vector-memory results return in order per wavefront).
        //      Unconditional selected addresses, see the column path.
        // Synchronisation: ONE early hardware barrier ("0", right after everyone has REQUESTED its loads, so that the two LDS flags
        // below are known to be reset) and one at the end ("B").  In between the wavefronts meet through LDS flags only, each waiting
        // for exactly what it needs: CHAIN, BEARING and DIAG for their own loads.
        double small_v = 0.0;
        if (role == 4) {
            const double *sp = prr_cur;                                  // idle lanes re-read Prr(1,1), unused
            if (lane < 9) sp = prr_cur + lane;
            else if (lane < 15) { const int t = (lane - 9) >> 1, b = (lane - 9) & 1; sp = strip + t * ldm + j + b; }
            else if (lane >= 19 && lane < 22) sp = x + (lane - 19);
            else if (lane >= 22 && lane < 24) sp = x + 3 + j + (lane - 22);
            small_v = *sp;
            if (lane == 24) { diag_ready = 0; staged_cnt = 0; }
        } else if (role == 5) {
            // lanes 0..3: canonical P(j+t, j+b) of the landmark's own 2x2 block -- from the LIVE F64 copy (DevState::diag): every
            // correction so far has applied its pair to it already, so there is no chain of pending pairs to re-run here (that chain,
            // serial in slot order, was what bounded this kernel from ~28 pending pairs on)
            const int t = (lane >> 1) & 1, b = lane & 1;
            small_v = st.diag[st.dcur][3 * (j >> 1) + (t > b ? t : b) + (t > b ? b : t)];
        } else {
            small_v = x[lane < 3 ? lane : 3 + j + ((lane - 3) & 1)];       // BEARING: lanes 0..2 the pose, 3..4 the landmark
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");                       // barrier 0: no waitcnt -- the loads stay in flight across it
        EKF_STAMP();                                                  // 1
        if (role == 4 && (lane < 15 || (lane >= 19 && lane < 24))) pss[lane] = small_v;      // (the column lanes read these after B)
        if (role == 5 && lane < 4) pss[15 + lane] = small_v;
        EKF_STAMP();                                                  // 2: own operands arrived
        if (role == 5) {
            // ---- DIAG: the block is in pss[15..18] (written above): tell the CHAIN wavefront
            wave_lds_sync();
            if (lane == 0) *(volatile int *)&diag_ready = 1;              // (one wavefront: LDS order = program order)
            EKF_STAMP();                                                  // (probe, DIAG view) flag set
        } else if (role == 6) {
            // ---- BEARING: nu = z - z_k, z_k = [sqrt(q); wrapTo360(atan2d(dy,dx) - heading)]  (EKF_SLAM.m:125-130,144), from the
            //      PREDICTED pose when predict is folded in -- same expressions as the chain wavefront's, so the same bits
            const double bx = lane_bcast(small_v, 0), by = lane_bcast(small_v, 1), bth = lane_bcast(small_v, 2),
                         blx = lane_bcast(small_v, 3), bly = lane_bcast(small_v, 4);
            if (lane == 0) {
                double pose[3] = { bx, by, bth };
                if (kPredict) {
                    const double2 sc2 = sincosd_ni(pose[2] + pa.u1);
                    const double sn2 = sc2.x, cs2 = sc2.y;
                    const double p0 = predict_pose_entry(pose, 0, pa.u0, pa.u1, sn2, cs2), p1 = predict_pose_entry(pose, 1, pa.u0, pa.u1, sn2, cs2),
                                 p2 = predict_pose_entry(pose, 2, pa.u0, pa.u1, sn2, cs2);
                    pose[0] = p0; pose[1] = p1; pose[2] = p2;
                }
                const double d0 = blx - pose[0], d1 = bly - pose[1];
                const double bearing = bearing_ni(d1, d0, pose[2]);
                const double sq = sqrt(d0 * d0 + d1 * d1);
                sol.nu[0] = a.z0 - sq;                                    // :144 (bearing NOT wrapped)
                sol.nu[1] = a.z1 - bearing;
            }
        } else {
            // ---- CHAIN: each matrix entry of the solve is formed on its own lane.  Every lane holds the 24 small operands in
            //      registers (static indices only: an array indexed by the lane would live in scratch memory) and SELECTS the ones
            //      its entry needs; what every lane needs identically (pose, H_s, inv(phi)) is computed redundantly; entries travel
            //      between lanes by v_readlane (G(:,S), phi) or, where each lane needs a different subset, through `pss` in LDS.
            // every lane gets the 20 operands this wavefront loaded (lane i holds operand i) by v_readlane: wave-uniform values, no
            // LDS round trip; 15..18 (the landmark's own 2x2 block) belong to DIAG and are read later, from LDS
            double p[24];
#pragma unroll
            for (int i = 0; i < 24; ++i) p[i] = (i >= 15 && i < 19) ? 0.0 : lane_bcast(small_v, i);
            double fa = 0.0, fb = 0.0;
            double pose[3] = { p[19], p[20], p[21] };
            if (kPredict) {
                // predict(u) folded into this correction: same per-entry arithmetic as k_predict (predict_*_entry)
                const double2 sc_l = sincosd_ni((lane & 1) ? pose[2] + pa.u1 : pose[2]);   // lane 0: pre-motion heading, lane 1: + u2
                EKF_STAMP();                                          // (probe) sincos
                const double sn = lane_bcast(sc_l.x, 0), cs = lane_bcast(sc_l.y, 0), sn2 = lane_bcast(sc_l.x, 1), cs2 = lane_bcast(sc_l.y, 1);
                double W[3];
                predict_common(pa.u0, pa.u1, sn, cs, fa, fb, W);
                // lane l < 9: Prr'(l/3, l%3) and Q; 9..14: strip'(t, j+b), t = (l-9)>>1, b = (l-9)&1.  The three operands an entry needs
                // (a column of Prr, or strip(0..2, j+b)) are GATHERED from the lanes that loaded them (small_v: lane i holds operand
                // i) with ds_bpermute -- no LDS memory, no select chains -- and every lane runs both (short) forms, keeping its own
                const int ri = lane >= 6 ? 2 : lane >= 3 ? 1 : 0, rj = lane - 3 * ri;      // the entry this lane holds ...
                const int ei = ri > rj ? ri : rj, ej = ri > rj ? rj : ri;                   // ... evaluated as its lower-triangle mirror (Prr stays exactly symmetric)
                const int st_t = (lane - 9) >> 1, st_b = (lane - 9) & 1;
                const bool is_prr = lane < 9;
                const int g0 = is_prr ? ej : 9 + st_b, g1 = is_prr ? 3 + ej : 11 + st_b, g2 = is_prr ? 6 + ej : 13 + st_b;
                const double v0 = lane_gather(small_v, g0 & 63), v1 = lane_gather(small_v, g1 & 63), v2 = lane_gather(small_v, g2 & 63);
                const double cj[3] = { v0, v1, v2 };
                const double c2[3] = { p[2], p[5], p[8] };
                const double wi = EKF_SEL(ei == 0) ? W[0] : (EKF_SEL(ei == 1) ? W[1] : W[2]);
                const double wj = EKF_SEL(ej == 0) ? W[0] : (EKF_SEL(ej == 1) ? W[1] : W[2]);
                double e_prr, e_q;
                predict_prr_entry(ei, ej, cj, c2, fa, fb, wi, wj, pa.C, e_prr, e_q);
                double s0 = v0, s1 = v1;
                predict_strip(s0, s1, v2, fa, fb);
                const double e_strip = EKF_SEL(st_t == 0) ? s0 : (EKF_SEL(st_t == 1) ? s1 : v2);
                const double p0 = predict_pose_entry(pose, 0, pa.u0, pa.u1, sn2, cs2), p1 = predict_pose_entry(pose, 1, pa.u0, pa.u1, sn2, cs2),
                             p2 = predict_pose_entry(pose, 2, pa.u0, pa.u1, sn2, cs2);      // every lane (3 operations)
                pose[0] = p0; pose[1] = p1; pose[2] = p2;
                EKF_STAMP();                                          // (probe) entries formed
                if (lane < 15) pss[lane] = EKF_SEL(is_prr) ? e_prr : e_strip;   // the column lanes and the G(:,S) lanes read Prr', strip' from here
                if (lane < 9) ps.Q[lane] = e_q;
                if (lane == 0) { ps.fa = fa; ps.fb = fb; }
            }
            if (lane < 3) pose_sh[lane] = lane == 0 ? pose[0] : lane == 1 ? pose[1] : pose[2];
            SmallSolve so;
            double sq;
            solve_hs(p[22] - pose[0], p[23] - pose[1], sq, so.Hs);    // EKF_SLAM.m:125-127,137-138 (every lane, redundantly)
            EKF_STAMP();                                              // 3: H_s
            while (*(volatile int *)&diag_ready == 0) { }             // the DIAG wavefront's (patched) 2x2 block is in pss[15..18]
            wave_lds_sync();                                          // pss: predicted entries (own writes) and that block
            const int ra = lane >= 5 ? 1 : 0;                         // row of this lane's G(:,S) entry
            double hsel[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) hsel[t] = ra ? so.Hs[1][t] : so.Hs[0][t];
            const double e_gs = solve_gs_entry(pss, hsel, lane < 10 ? lane - 5 * ra : 0);
            double GS[2][5];
#pragma unroll
            for (int i = 0; i < 10; ++i) GS[i / 5][i % 5] = lane_bcast(e_gs, i);
            double e_phi;
            {
                const int aa = (lane >> 1) & 1, bb = lane & 1;
                double gsel[5], hb[5];
#pragma unroll
                for (int t = 0; t < 5; ++t) { gsel[t] = aa ? GS[1][t] : GS[0][t]; hb[t] = bb ? so.Hs[1][t] : so.Hs[0][t]; }
                const double Rab = aa == 0 ? (bb == 0 ? a.R00 : a.R01) : (bb == 0 ? a.R10 : a.R11);
                e_phi = solve_phi_entry(gsel, hb, Rab);                                      // :141
            }
            const double phi[4] = { lane_bcast(e_phi, 0), lane_bcast(e_phi, 1), lane_bcast(e_phi, 2), lane_bcast(e_phi, 3) };
            ekfm::inv2(phi, so.Phi);                                  // :143 phi_k^-1 (every lane, redundantly)
            EKF_STAMP();                                              // 4: solve
            // publish: K_r and G_r are formed by every lane (18 operations, static indices, no divergent branches -- the per-lane
            // form with its select chains was 200 instructions), then lane 0 stores the whole struct; nu is BEARING's
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                so.Gr[0][b] = GS[0][b]; so.Gr[1][b] = GS[1][b];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) so.Kr[b][cc] = solve_kr_entry(GS[0][b], GS[1][b], so.Phi[cc], so.Phi[2 + cc]);
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 10; ++i) sol.Hs[i / 5][i % 5] = so.Hs[i / 5][i % 5];
#pragma unroll
                for (int i = 0; i < 4; ++i) sol.Phi[i] = so.Phi[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) { sol.Kr[i / 2][i % 2] = so.Kr[i / 2][i % 2]; sol.Gr[i / 3][i % 3] = so.Gr[i / 3][i % 3]; }
            }
        }
        __syncthreads();                                              // barrier B: sol, ps, pss complete
        if (blockIdx.x == 0) {
            // the replicated small outputs of workgroup 0, beside the column lanes' outputs, not in front of them, one KIND per
            // helper wavefront (run by one wavefront the five kinds are five divergent branches back to back, ~1 900 clocks at the
            // tail of the kernel): x_r (x(3) NOT re-wrapped) | Prr | G_r, K_r, Q for the host-side getters
            if (role == 4) {
                if (lane < 3) x_nxt[lane] = pose_sh[lane] + (sol.Kr[lane][0] * sol.nu[0] + sol.Kr[lane][1] * sol.nu[1]);
            } else if (role == 5) {
                if (lane < 9) {
                    const int r = lane / 3, b = lane - 3 * r;
                    // Prr' = Prr - K_r G_r, kept EXACTLY symmetric: entry (r,b) and its mirror both take the lower-triangle entry's value.
                    // Evaluated entry by entry, K_r(r,:) G_r(:,b) and K_r(b,:) G_r(:,r) differ in the last bit; with the strip stored once
                    // (symmetry enforced there) the antisymmetric part this leaves in the 3x3 block is not damped but AMPLIFIED by the
                    // corrections that follow -- measured: 2e-15 after 250 SLAM iterations, 1.3e-7 after 3 000, the heading drifting from
                    // the dense restatement with it (scripts/soak_config2.py), where the reference's dense P stays symmetric to 1e-16.
                    const int rr = r > b ? r : b, bb = r > b ? b : r;
                    prr_nxt[3 * r + b] = pss[3 * rr + bb] - (sol.Kr[rr][0] * sol.Gr[0][bb] + sol.Kr[rr][1] * sol.Gr[1][bb]);
                }
            } else {
                if (lane < 6) { const int r = lane / 3, b = lane - 3 * r; st.small[3 * r + b] = sol.Gr[r][b]; }
                else if (lane < 12) { const int b = (lane - 6) >> 1, r = (lane - 6) & 1; st.small[6 + 2 * b + r] = sol.Kr[b][r]; }
#if !defined(EKF_GATHER_STAMPS) || EKF_GATHER_STAMPS < 2                  // (those probe builds return a helper's stamps in the Q slots)
                else if (kPredict && lane < 21) st.small[12 + (lane - 12)] = ps.Q[lane - 12];
#endif
            }
        }
#ifdef EKF_GATHER_STAMPS
        EKF_STAMP();                                                  // barrier B passed
#if EKF_GATHER_STAMPS == 2                                                // the CHAIN wavefront's view (the Q slots hold one view per build)
        if (blockIdx.x == 0 && tid == kGatherCols) for (int i = 0; i < 9; ++i) st.small[12 + i] = (double)(stamp[i] - stamp[0]);
#endif
#if EKF_GATHER_STAMPS == 3                                                // the DIAG wavefront's view
        if (blockIdx.x == 0 && tid == kGatherCols + 64) for (int i = 0; i < 9; ++i) st.small[12 + i] = (double)(stamp[i] - stamp[0]);
#endif
#endif
        return;
    }

    // ================================================ column lanes ================================================
    const int64_t c = (int64_t)blockIdx.x * kGatherCols + tid;
    const bool live = c < a.n_mm;
    // (1) Loads, in the order their consumers need them.  Vector-memory results return in order, so the wave-uniform operands of
    //     the pending pairs (K_i / G_i at rows / columns j, j+1), which the DIAG wavefront and everyone's patches need, go FIRST.
    //     Every load below is unconditional with a selected / clamped address: a predicated load is merged by the compiler
    //     with the predicated LDS write that consumes it, which puts a full memory round trip in front of everything else.
    static_assert(kMaxPending * 4 == 2 * kGatherCols, "two uniform operands per column lane");
    auto load_up = [&](int e0) {
        const int e = (do_patch && e0 < 4 * npend) ? e0 : 0;     // clamped: slot pstart always exists
        const int i = e >> 2, which = e & 3;
        const double *base = (which < 2 ? st.Kp : st.Gp) + (int64_t)ring_slot(pstart, i, st.pcap) * st.pair_stride;
        return reinterpret_cast<const double2 *>(base)[j + (which & 1)];
    };
    const double2 up0 = load_up(tid), up1 = load_up(tid + kGatherCols);
    __builtin_amdgcn_sched_barrier(0);      // keep these loads AHEAD of the per-column ones below (in-order return)
    // barrier 0 (see the helper path): flags reset.  HERE, before the per-column loads: their address arithmetic takes ~1 200
    // clocks, and the CHAIN wavefront -- the critical path of the launch -- would stand at this barrier for all of them (it did:
    // the column lanes then waited ~2 000 clocks for the solve at barrier B).  No waitcnt: the loads stay in flight across it.
    asm volatile("s_barrier" ::: "memory");
    EKF_STAMP();                                                  // a: uniform operands requested, barrier 0 passed
    //     Then what this column needs: the two landmark rows at column c (canonical lower-triangle entries: row part left of
    //     j, column part right of j+1 -- one 16-byte load there, j is even; from the tiles or from the exchanged row-panel),
    //     the strip column, x(c) ...
    double m0 = 0.0, m1 = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0, xc = 0.0;
    const bool rowpart = c <= j, colpart = c >= j + 2;
    if (live) {
        if (kSharded) { const double2 m = pv.at(st.tm, c); m0 = m.x; m1 = m.y; }
        else if (rowpart) {                                     // P(j, c), P(j+1, c): one tile (j is even), rows T apart
            const int64_t m = st.tm.T - 1;
            const TS *__restrict__ tp = tiles + st.tm.tile_offset(j >> st.tm.shift, c >> st.tm.shift) + ((j & m) << st.tm.shift) + (c & m);
            m0 = (double)tp[0]; m1 = (double)tp[st.tm.T];
        }
        else if (colpart) pmm_low_pair<TS>(tiles, st.tm, c, j, m0, m1);
        else pmm_low_pair<TS>(tiles, st.tm, j + 1, j, m0, m1);  // c == j + 1: canonical (j+1, j), (j+1, j+1)
        s0 = strip[c]; s1 = strip[ldm + c]; s2 = strip[2 * ldm + c];
        xc = x[3 + c];
    }
    // the column's own diagonal-block entries, live F64 copies (DevState::diag): even columns hold (2k,2k), odd ones (2k+1,2k) and
    // (2k+1,2k+1).  Read for three purposes: rows j, j+1 at columns j, j+1 ARE these entries (below); this correction's pair is
    // applied to them at the end of the kernel; and the kDev epilogue's association starts from the result.
    double dgc = 0.0, dgl = 0.0;
    if (live) {
        const double *__restrict__ dg = st.diag[st.dcur] + 3 * (c >> 1);
        if (c & 1) { dgl = dg[1]; dgc = dg[2]; } else dgc = dg[0];
    }
    __builtin_amdgcn_sched_barrier(0);
    EKF_STAMP();                                                  // b: all loads requested
    // (2) stage the uniform operands (waits for the FIRST group of loads only); the four column wavefronts and DIAG meet on a
    //     counter in LDS -- the CHAIN and BEARING wavefronts do not take part
    upatch[tid] = up0; upatch[tid + kGatherCols] = up1;           // unconditional (entries past 4*npend are never read)
    wave_lds_sync();
    if ((tid & 63) == 0) atomicAdd(&staged_cnt, 1);
    while (*(volatile int *)&staged_cnt < 4) { }
    wave_lds_sync();
    EKF_STAMP();                                                  // 1: uniform operands staged

    // (2b) this column's operands of the first kPre pending pairs (G_i(:,c) left of j, K_i(c,:) right of it), all in flight
    //     together (fetched 8 at a time inside the patch loop they cost one L2 round trip per 8 pairs).  Requested AFTER barrier
    //     A: issuing these up to 32 loads takes the column wavefronts ~2 000 clocks, and before the barrier that was 2 000 clocks
    //     the helper wavefronts -- the critical path -- spent waiting for them; behind it the column lanes have ~3 000 clocks of
    //     slack until the solve is published (scripts/probe_gather_phases.py).
    // 32 pairs: a 64-pair variant (344 VGPRs, one workgroup per CU) was slower under an asynchronous flush (tuning log, sweep 12)
    constexpr int kPre = 32;
    const int npre = do_patch ? (npend < kPre ? npend : kPre) : 0;
    const int64_t pad_cols = st.tm.padded(a.n_mm);
    const int64_t ps2 = st.pair_stride / 2;
    double2 pre[kPre];
    bool next_assoc = false;
    if constexpr (kDev) next_assoc = dl.parts_out != nullptr;
    {
        // unconditional, clamped addresses (a predicated form lets the compiler sink the loads below the barrier, next to their
        // use); slots past npend repeat the last pending one (cache hits), c is clamped into the padded vector
        // One uniform base (Gp; Kp follows it in the same allocation, abi.hip) + a 32-bit per-lane element offset: the
        // compiler can then use the scalar-base addressing form and the 32 loads cost one scalar add each.
        const uint32_t cc = (uint32_t)(c < pad_cols ? c : pad_cols - 1);
        const uint32_t krel = (uint32_t)((st.Kp - st.Gp) >> 1);
        const uint32_t lane_off = cc + (rowpart ? 0u : krel);
        const char *__restrict__ ub = reinterpret_cast<const char *>(st.Gp);
        const uint32_t lane_bytes = lane_off * 16u;              // < 2^32: see below
        // slot offsets advance incrementally around the ring (scalar unit: one add, one wrap test per pair)
        // (32-bit: 2 * pcap * pair_stride / 2 <= 256 * 2 * capacity elements of 16 bytes stays far below 2^32)
        const uint32_t step = (uint32_t)ps2, wrap = (uint32_t)st.pcap * step;
        uint32_t off = (uint32_t)pstart * step;
        // Groups of 8 are skipped when no pending pair falls into them (immediate mode, the start of every batch).  The group
        // test uses an OPAQUE copy of npre: with the same condition as at the use sites the compiler would merge each group
        // of loads into the block that consumes it, below the barrier.
        int npre_ld = npre;
        asm volatile("" : "+s"(npre_ld));
#pragma unroll
        for (int g0 = 0; g0 < kPre; g0 += 8) {
            if (g0 < npre_ld) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    pre[g0 + t] = *reinterpret_cast<const double2 *>(ub + (uint64_t)off * 16u + lane_bytes);
                    if (g0 + t + 1 < npre) { off += step; if (off == wrap) off = 0; }
                }
            }
        }
    }


    // (3) every lane applies the pending pairs to its own two row entries while the helper wavefronts run the solve
    if (live && do_patch) {
        if (rowpart) {
#pragma unroll
            for (int g0 = 0; g0 < kPre; g0 += 8)
                if (g0 < npre) {                                  // uniform; inside a group no branches: select
                    double2 ua[8], ub[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) { ua[t] = upatch[4 * (g0 + t) + 0]; ub[t] = upatch[4 * (g0 + t) + 1]; }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const double v0 = rank2_apply(m0, ua[t], pre[g0 + t]), v1 = rank2_apply(m1, ub[t], pre[g0 + t]);
                        m0 = g0 + t < npre ? v0 : m0; m1 = g0 + t < npre ? v1 : m1;
                    }
                }
            // more than kPre pending pairs (async flush, batch > kPre): chunks of 8 independent loads, applied in order
            const double2 *__restrict__ gp = reinterpret_cast<const double2 *>(st.Gp) + c;
            int i = npre;
            for (; i + 8 <= npend; i += 8) {
                double2 g[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) g[q] = gp[(int64_t)ring_slot(pstart, i + q, st.pcap) * ps2];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    m0 = rank2_apply(m0, upatch[4 * (i + q) + 0], g[q]);
                    m1 = rank2_apply(m1, upatch[4 * (i + q) + 1], g[q]);
                }
            }
            for (; i < npend; ++i) {
                const double2 g = gp[(int64_t)ring_slot(pstart, i, st.pcap) * ps2];
                m0 = rank2_apply(m0, upatch[4 * i + 0], g);
                m1 = rank2_apply(m1, upatch[4 * i + 1], g);
            }
        } else if (colpart) {
#pragma unroll
            for (int g0 = 0; g0 < kPre; g0 += 8)
                if (g0 < npre) {
                    double2 ua[8], ub[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) { ua[t] = upatch[4 * (g0 + t) + 2]; ub[t] = upatch[4 * (g0 + t) + 3]; }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const double v0 = rank2_apply(m0, pre[g0 + t], ua[t]), v1 = rank2_apply(m1, pre[g0 + t], ub[t]);
                        m0 = g0 + t < npre ? v0 : m0; m1 = g0 + t < npre ? v1 : m1;
                    }
                }
            const double2 *__restrict__ kp = reinterpret_cast<const double2 *>(st.Kp) + c;
            int i = npre;
            for (; i + 8 <= npend; i += 8) {
                double2 k[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) k[q] = kp[(int64_t)ring_slot(pstart, i + q, st.pcap) * ps2];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    m0 = rank2_apply(m0, k[q], upatch[4 * (i + q) + 2]);
                    m1 = rank2_apply(m1, k[q], upatch[4 * (i + q) + 3]);
                }
            }
            for (; i < npend; ++i) {
                const double2 k = kp[(int64_t)ring_slot(pstart, i, st.pcap) * ps2];
                m0 = rank2_apply(m0, k, upatch[4 * i + 2]);
                m1 = rank2_apply(m1, k, upatch[4 * i + 3]);
            }
        } else {                                               // c == j + 1: canonical (j+1,j) and (j+1,j+1)
            for (int i = 0; i < npend; ++i) {
                m0 = rank2_apply(m0, upatch[4 * i + 1], upatch[4 * i + 2]);
                m1 = rank2_apply(m1, upatch[4 * i + 1], upatch[4 * i + 3]);
            }
        }
    }
    // rows j, j+1 at columns j and j+1 are the landmark's own diagonal block: the live F64 copies (no pending pair to apply; the same bits
    // as the patched tile entries with F64 tiles, the unrounded values with F32 tiles)
    {
        const double dgl_p = lane_xor1(dgl);                   // the odd partner's (2k+1, 2k)
        if (c == j) { m0 = dgc; m1 = dgl_p; }                      // P(j, j), P(j+1, j)
        else if (c == j + 1) { m0 = dgl; m1 = dgc; }               // P(j+1, j), P(j+1, j+1)
    }
    EKF_STAMP();                                                  // 2: patches done
    __syncthreads();                                              // barrier B: the helpers' results are in LDS
    EKF_STAMP();                                                  // 3: solve available

    // (4) the column's share of G, K, x and the strip
    const int64_t pad_end = st.tm.padded(a.n_mm);
    const int64_t out_off = (int64_t)ring_slot(pstart, npend, st.pcap) * st.pair_stride;   // this correction's own pair
    double2 *__restrict__ Gout = reinterpret_cast<double2 *>(st.Gp + out_off);
    double2 *__restrict__ Kout = reinterpret_cast<double2 *>(st.Kp + out_off);
    double g[2] = { 0.0, 0.0 }, k0 = 0.0, k1 = 0.0;
    double xn = 0.0, t0 = 0.0, t1 = 0.0, t2 = 0.0;               // x'(c), strip'(0..2, c): stored, and read again by the kDev epilogue
    if (live) {
        if (kPredict) predict_strip(s0, s1, s2, ps.fa, ps.fb);
        for (int r = 0; r < 2; ++r)
            g[r] = sol.Hs[r][0] * s0 + sol.Hs[r][1] * s1 + sol.Hs[r][2] * s2 + sol.Hs[r][3] * m0 + sol.Hs[r][4] * m1;
        k0 = g[0] * sol.Phi[0] + g[1] * sol.Phi[2];
        k1 = g[0] * sol.Phi[1] + g[1] * sol.Phi[3];
        if (!kFused) {
            Gout[c] = make_double2(g[0], g[1]);
            Kout[c] = make_double2(k0, k1);
            if (st.Gp32) {                                          // uniform: the F32-arithmetic pass reads float copies, planar, K negated
                st.Gp32[out_off + c] = (float)g[0]; st.Gp32[out_off + ldm + c] = (float)g[1];
                st.Kp32[out_off + c] = -(float)k0; st.Kp32[out_off + ldm + c] = -(float)k1;
            }
        }
        xn = xc + (k0 * sol.nu[0] + k1 * sol.nu[1]);
        t0 = s0 - (sol.Kr[0][0] * g[0] + sol.Kr[0][1] * g[1]);
        t1 = s1 - (sol.Kr[1][0] * g[0] + sol.Kr[1][1] * g[1]);
        t2 = s2 - (sol.Kr[2][0] * g[0] + sol.Kr[2][1] * g[1]);
        x_nxt[3 + c] = xn;
        double *__restrict__ sn = strip_nxt;
        sn[c] = t0;
        sn[ldm + c] = t1;
        sn[2 * ldm + c] = t2;
    } else if (c < pad_end && !kFused) {
        // zero the tail of the last tile so the downdate leaves the unused part of edge tiles untouched
        Gout[c] = make_double2(0.0, 0.0);
        Kout[c] = make_double2(0.0, 0.0);
        if (st.Gp32) {
            st.Gp32[out_off + c] = 0.0f; st.Gp32[out_off + ldm + c] = 0.0f;
            st.Kp32[out_off + c] = -0.0f; st.Kp32[out_off + ldm + c] = -0.0f;
        }
    }
    // (4b) this correction's pair on the diagonal blocks, at once: P(I - K H) restricted to each landmark's own 2x2 block, rank2_apply in
    //      slot order like every pass -- the live copies never carry a pending pair.  ndc = the column's (c,c), ndl = (2k+1, 2k) on odd columns.
    double ndc, ndl;
    {
        const double2 kn = make_double2(k0, k1), gn = make_double2(g[0], g[1]);
        const double2 gl = make_double2(lane_xor1(gn.x), lane_xor1(gn.y));       // the partner column's G (odd lanes: G(:, 2k))
        ndc = rank2_apply(dgc, kn, gn);
        ndl = rank2_apply(dgl, kn, gl);
        if (live) {
            double *__restrict__ dn = st.diag[st.dcur ^ 1] + 3 * (c >> 1);
            if (c & 1) { dn[1] = ndl; dn[2] = ndc; } else dn[0] = ndc;
        }
    }
    if (kFused) {
        // P = (I - K H) P on the landmark block, here: one workgroup holds every K(r,:) and G(:,c).  K goes through LDS (`upatch` is
        // free: the patches that read it are behind barrier B; the helper wavefronts have left, a barrier counts live wavefronts
        // only), G(:,c) is this lane's own.  Lane c walks down column c from the first row of its diagonal tile (diagonal tiles are
        // updated whole, like k_downdate does); rows / columns beyond n_mm hold K = G = 0 there and are left alone -- same bits.
        upatch[tid] = make_double2(k0, k1);
        upatch[kGatherCols + tid] = make_double2(g[0], g[1]);
        __syncthreads();
        {
            // all 256 lanes share the n x n elements (element e = tid + 256 q -> row e / n, column e % n); a lane requests all of
            // its elements before it touches any (a serial walk down one column paid a memory round trip per row: measured
            // SLOWER than two launches).  Stored elements: tile (I,J) with I >= J, diagonal tiles whole.
            TS *__restrict__ tw = (TS *)st.tiles;
            const int sh = st.tm.shift, msk = st.tm.T - 1;
            const unsigned n = (unsigned)a.n_mm, total = n * n;
            TS *ptr[kFuseElems];
            double val[kFuseElems];
            unsigned rr[kFuseElems], cq[kFuseElems];
#pragma unroll
            for (int q = 0; q < kFuseElems; ++q) {
                const unsigned e = (unsigned)tid + 256u * q;
                const unsigned r = e / n, cx = e - r * n;
                const bool stored = e < total && (r >> sh) >= (cx >> sh);
                rr[q] = stored ? r : 0u; cq[q] = stored ? cx : 0u;
                ptr[q] = stored ? tw + st.tm.tile_offset(r >> sh, cx >> sh) + ((r & msk) << sh) + (cx & msk) : nullptr;
                val[q] = stored ? (double)*ptr[q] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < kFuseElems; ++q)
                if (ptr[q]) *ptr[q] = (TS)rank2_apply(val[q], upatch[rr[q]], upatch[kGatherCols + cq[q]]);
        }
    }
    if constexpr (kDev) {
        if (next_assoc) {                                         // uniform
            // ---- the NEXT observation's association (Correspondence.m:49-87) on the state this correction leaves.  Landmark
            //      k = c / 2 is scored by its even column lane; everything it needs is in this lane pair's registers (x', strip',
            //      the landmark's own 2x2 block after this correction) or in the workgroup's LDS (Prr before the correction, K_r,
            //      G_r, nu).
            __shared__ double na_ll[kGatherCols / 64];
            __shared__ int na_ix[kGatherCols / 64];
            const bool odd = (c & 1) != 0;
            const double dcc = ndc, dlo = ndl;                    // the landmark's own block after this correction: computed above, live
            // odd lane -> even lane
            const double xn_o = lane_xor1(xn), t0_o = lane_xor1(t0), t1_o = lane_xor1(t1), t2_o = lane_xor1(t2),
                         d10 = lane_xor1(dlo), d11 = lane_xor1(dcc);
            double ll = INFINITY;
            int64_t ix = INT64_MAX;
            if (live && !odd) {
                const int64_t k = c >> 1;
                double q[24];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {                 // Prr' as the DIAG wavefront stores it (lower-triangle value, mirrored)
                        const int rr = r > b ? r : b, bb = r > b ? b : r;
                        q[3 * r + b] = pss[3 * rr + bb] - (sol.Kr[rr][0] * sol.Gr[0][bb] + sol.Kr[rr][1] * sol.Gr[1][bb]);
                    }
                q[9] = t0; q[10] = t0_o; q[11] = t1; q[12] = t1_o; q[13] = t2; q[14] = t2_o;
                q[15] = dcc; q[16] = d10; q[17] = d10; q[18] = d11;
#pragma unroll
                for (int l = 0; l < 3; ++l)                       // x_r' as the CHAIN wavefront stores it (x(3) NOT re-wrapped)
                    q[19 + l] = pose_sh[l] + (sol.Kr[l][0] * sol.nu[0] + sol.Kr[l][1] * sol.nu[1]);
                q[22] = xn; q[23] = xn_o;
                SmallSolve so2;
                solve_small(q, dl.z0, dl.z1, dl.R00, dl.R01, dl.R10, dl.R11, so2);
                const double n0 = so2.nu[0], n1 = so2.nu[1];
                const double pc = (n0 * so2.Phi[0] + n1 * so2.Phi[2]) * n0 + (n0 * so2.Phi[1] + n1 * so2.Phi[3]) * n1;     // :69
                const double d = dl.z2 - st.s[k];
                const double sc = d * (1.0 / dl.s_cost) * d;                                                            // :71
                const double like = (dl.w_pos != 0.0) ? (dl.w_pos * pc + sc) : sc;                                      // :74-75
                if (like <= dl.s_thresh) { ll = like; ix = k; }                                                         // :78
            }
            // workgroup arg-min: butterflies, the four column wavefronts' winners through LDS (the helper wavefronts have left:
            // a barrier counts live wavefronts only), one entry per workgroup for the next launch's reduce_partials_wave
            wave_argmin_sparse(ll, ix);
            if ((tid & 63) == 0) { na_ll[tid >> 6] = ll; na_ix[tid >> 6] = ix == INT64_MAX ? -1 : (int)ix; }
            __syncthreads();
            if (tid < 64) {
                ll = tid < kGatherCols / 64 ? na_ll[tid] : INFINITY;
                ix = (tid < kGatherCols / 64 && na_ix[tid] >= 0) ? (int64_t)na_ix[tid] : INT64_MAX;
                if (ix == INT64_MAX) ll = INFINITY;
                wave_argmin_sparse(ll, ix);
                if (tid == 0) store_partial(dl.parts_out + blockIdx.x, ll, ix == INT64_MAX ? -1 : (int)ix, dl.seq_out);
            }
        }
    }
#ifdef EKF_GATHER_STAMPS
    EKF_STAMP();                                                  // 4: outputs issued
    __syncthreads();
#if EKF_GATHER_STAMPS == 1                                                // column lane 0's view
    if (c == 0) for (int i = 0; i < 7; ++i) st.small[12 + i] = (double)(stamp[i] - stamp[0]);
#endif
#endif
#undef EKF_STAMP
}

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


// ---------------------------------------------------------------------------------------------------
// association (Correspondence.m:49-87): one thread per landmark, block arg-min, then a one-block finish
// ---------------------------------------------------------------------------------------------------
// kPredict: a recorded predict(u) (ekf_predict is lazy) is applied to what the lanes read -- pose, Prr and the strip columns,
// through the same per-entry functions as k_predict -- AND written to the other state buffer (a.cur ^ 1): this launch is
// k_predict and the association of the scan's first row in one (lane k owns landmark k's two strip columns either way), so
// the correction that follows neither waits for a k_predict launch nor folds the predict into its own latency chain.
template <typename TS, bool kPredict>
__global__ __launch_bounds__(kAssocBlock) void k_associate(DevState st, AssocArgs a, double *__restrict__ pos_cost,
                                                           double *__restrict__ sig_cost,
                                                           AssocDecision *partial, int *ticket, AssocDecision *__restrict__ decision,
                                                           AssocHostPartial *host_partials, int seq, double *__restrict__ cand,
                                                           PredictArgs pa) {
    __shared__ double sh_ll[kAssocBlock / 64];
    __shared__ int64_t sh_ix[kAssocBlock / 64];
    __shared__ PredictSmall aps;
    const int tid = threadIdx.x;
    const int cur = a.cur;
    const int64_t k = (int64_t)blockIdx.x * kAssocBlock + tid;
    double ll = INFINITY;
    int64_t ix = INT64_MAX;
    if (kPredict) {
        // one lane per workgroup runs the 3x3 part (two sincos + 9 entries); the others have nothing to do before it anyway
        if (tid < 64) {
            // wavefront 0: the two sincos (pre-motion heading on even lanes, heading + u2 on odd ones) in ONE call, then lane 0 forms
            // the entries -- predict_small's arithmetic with half of its sincos latency
            const double pose[3] = { st.x[cur][0], st.x[cur][1], st.x[cur][2] };
            const double2 sc_l = sincosd_ni((tid & 1) ? pose[2] + pa.u1 : pose[2]);
            const double sn = lane_bcast(sc_l.x, 0), cs = lane_bcast(sc_l.y, 0), sn2 = lane_bcast(sc_l.x, 1), cs2 = lane_bcast(sc_l.y, 1);
            if (tid == 0) {
                double prr[9];
                for (int i = 0; i < 9; ++i) prr[i] = st.prr[cur][i];
                predict_finish(pose, prr, pa.u0, pa.u1, pa.C, sn, cs, sn2, cs2, aps);
            }
        }
        __syncthreads();
    }
    if (k < a.N) {
        const double *__restrict__ x = st.x[cur];
        const double *__restrict__ strip = st.strip[cur];
        const int64_t j = 2 * k;
        double pss[24];
        for (int i = 0; i < 9; ++i) pss[i] = kPredict ? aps.prr[i] : st.prr[cur][i];
        for (int t = 0; t < 3; ++t) for (int b = 0; b < 2; ++b) pss[9 + 2 * t + b] = strip[t * st.ldm + j + b];
        if (kPredict) {
            double *__restrict__ sn = st.strip[cur ^ 1];
            double *__restrict__ xn = st.x[cur ^ 1];
            for (int b = 0; b < 2; ++b) {
                predict_strip(pss[9 + b], pss[11 + b], pss[13 + b], aps.fa, aps.fb);
                sn[j + b] = pss[9 + b]; sn[st.ldm + j + b] = pss[11 + b]; sn[2 * st.ldm + j + b] = pss[13 + b];
                xn[3 + j + b] = x[3 + j + b];
            }
            if (k == 0) {
                for (int i = 0; i < 9; ++i) { st.prr[cur ^ 1][i] = aps.prr[i]; st.small[12 + i] = aps.Q[i]; }
                for (int i = 0; i < 3; ++i) xn[i] = aps.pose[i];
            }
        }
        // the landmark's own 2x2 block: the live F64 copy (DevState::diag) -- every correction so far has applied its pair to it, on
        // every shard, so there is neither a chain of pending pairs to run here nor a tile another shard holds.  have_diag (does this
        // shard hold the landmark's diagonal TILE) only decides which shard nominates the landmark in a sharded association's exchange.
        const bool have_diag = st.tm.mine(j >> st.tm.shift, j >> st.tm.shift);
        {
            const double *__restrict__ dg = st.diag[st.dcur] + 3 * k;
            pss[15] = dg[0]; pss[16] = dg[1]; pss[17] = dg[1]; pss[18] = dg[2];
        }
        for (int i = 0; i < 3; ++i) pss[19 + i] = kPredict ? aps.pose[i] : x[i];
        pss[22] = x[3 + j]; pss[23] = x[3 + j + 1];
        SmallSolve sol;
        solve_small(pss, a.z0, a.z1, a.R00, a.R01, a.R10, a.R11, sol);
        const double n0 = sol.nu[0], n1 = sol.nu[1];
        const double pc = (n0 * sol.Phi[0] + n1 * sol.Phi[2]) * n0 + (n0 * sol.Phi[1] + n1 * sol.Phi[3]) * n1;  // :69
        const double d = a.z2 - st.s[k];
        const double sc = d * (1.0 / a.s_cost) * d;                                                          // :71
        if (pos_cost) pos_cost[k] = pc;
        if (sig_cost) sig_cost[k] = sc;
        const double like = (a.w_pos != 0.0) ? (a.w_pos * pc + sc) : sc;                                      // :74-75
        // a.own_only (sharded association with an exchange): a shard only nominates landmarks whose diagonal
        // block it holds; the candidates of all shards meet in k_assoc_merge
        if (like <= a.s_thresh && (have_diag || !a.own_only)) { ll = like; ix = k; }                         // :78
    }
    // Workgroup arg-min: per wavefront (ballot + readlane for the usual lone candidate, butterflies otherwise -- no barrier), the
    // wavefronts' winners through LDS, once more in wavefront 0.
    wave_argmin_sparse(ll, ix);
    if ((tid & 63) == 0) { sh_ll[tid >> 6] = ll; sh_ix[tid >> 6] = ix; }
    __syncthreads();
    if (tid < 64) {
        ll = tid < kAssocBlock / 64 ? sh_ll[tid] : INFINITY;
        ix = tid < kAssocBlock / 64 ? sh_ix[tid] : INT64_MAX;
        wave_argmin_sparse(ll, ix);
    }
    if (host_partials) {
        // The HOST takes the arg-min over the workgroups' winners: ONE 16-byte store per workgroup into mapped host memory, payload
        // and sequence number together -- no ticket, no fence, no second reduction on the device.
        if (tid == 0) store_partial(host_partials + blockIdx.x, ll, ix == INT64_MAX ? -1 : (int)ix, seq);
        return;
    }
    if (gridDim.x > 1) {
        // several workgroups and a consumer on the DEVICE (the sharded exchange's candidate): the LAST workgroup to get here
        // reduces the per-workgroup minima.
        // Hand-over of the partials, release / acquire at agent scope around the ticket: ONE lane releases (write-back of this
        // XCD's L2, ~1.7 us) and, in the last workgroup, ONE wavefront acquires (L1 invalidate) before it reads them.  The full
        // __threadfence() on both sides that stood here first -- write-back AND invalidate, the second one by all 256 threads --
        // was most of this kernel's 9 us (MI355X_MICROARCH.md: ~3.5 us per fence, 2-3.8x that with a whole workgroup fencing).
        __shared__ int last;
        if (tid == 0) {
            partial[blockIdx.x].min_ll = ll; partial[blockIdx.x].index = ix;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // the partial is visible before the ticket is drawn
            last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
        }
        __syncthreads();
        if (!last) return;
        if (tid < 64) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // this wavefront's loads below see every workgroup's partial
            ll = INFINITY; ix = INT64_MAX;
            for (int64_t i = tid; i < (int64_t)gridDim.x; i += 64) {
                const double pl = ((volatile AssocDecision *)partial)[i].min_ll;
                const int64_t pi = ((volatile AssocDecision *)partial)[i].index;
                if (assoc_better(pl, pi, ll, ix)) { ll = pl; ix = pi; }
            }
            wave_argmin(ll, ix);
        }
    }
    // a map that fits one workgroup needs no partials, no ticket, no second reduction
    if (tid == 0) {
        const bool found = ix != INT64_MAX;           // something passed the threshold (min_ll starts at Inf, :43)
        AssocDecision d;
        d.is_new = found ? 0 : 1;
        d.index = found ? ix : a.N;                   // default index = numOfLandmarks + 1 (:40), 0-based here
        d.min_ll = ll;
        d.seq = seq;
        *decision = d;
        if (gridDim.x > 1) *ticket = 0;               // ready for the next launch (stream order)
        if (cand) { cand[0] = ll; cand[1] = found ? (double)ix : -1.0; cand[2] = 0.0; cand[3] = 0.0; }
    }
}

// Sharded association, after the all-gather: contribution r of `recv` holds shard r's candidate {likelihood, 0-based index or
// -1, 0, 0} and, if costs travel too, its position costs (4 + k; NaN where shard r does not hold landmark k's diagonal block).
// Every shard takes the same strict arg-min over the candidates (Correspondence.m:78-85: lowest likelihood, lowest index on
// ties -- what the unsharded kernel's reduction does) and assembles pos_cost from each landmark's owner.
__global__ __launch_bounds__(kBlock) void k_assoc_merge(TileMap tm, const double *__restrict__ recv, int world, int64_t count,
                                                        int64_t N, int want_costs, double *__restrict__ pos_cost,
                                                        AssocDecision *__restrict__ decision, AssocDecision *host_decision,
                                                        int seq) {
    if (want_costs)
        for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < N; k += (int64_t)gridDim.x * kBlock) {
            const int64_t I = (2 * k) >> tm.shift;
            pos_cost[k] = recv[(int64_t)tm.owner(I, I) * count + 4 + k];
        }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double ll = INFINITY;
        int64_t ix = INT64_MAX;
        for (int r = 0; r < world; ++r) {
            const double cl = recv[(int64_t)r * count], ci = recv[(int64_t)r * count + 1];
            if (ci >= 0.0 && assoc_better(cl, (int64_t)ci, ll, ix)) { ll = cl; ix = (int64_t)ci; }
        }
        const bool found = ix != INT64_MAX;
        AssocDecision d;
        d.is_new = found ? 0 : 1;
        d.index = found ? ix : N;
        d.min_ll = ll;
        d.seq = seq;
        *decision = d;
        if (host_decision) {
            volatile AssocDecision *hd = host_decision;
            hd->index = d.index; hd->is_new = d.is_new; hd->min_ll = d.min_ll;
            __threadfence_system();
            hd->seq = seq;
        }
    }
}

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

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace

int gather_fuse_max_rows() { return kFuseMaxRows; }

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
template <typename TS, int T, int kChunk, int kCols = 128, int kWpe = (kChunk <= 4 ? 4 : 3)>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kWpe, kWpe)))
void k_flush_mfma(const TS *__restrict__ tiles, TS *__restrict__ dst, const int2 *__restrict__ work, int64_t nwork,
                  const double *__restrict__ Kp, const double *__restrict__ Gp, int64_t pair_stride, int pstart, int pcap,
                  int npairs, TileMap tm) {
    constexpr int kRows = 64, kKPad = kRows + 16;
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
                const int i = (e >> 6) < cn ? (e >> 6) : cn - 1;
                tk[q] = reinterpret_cast<const double2 *>(Kp + (int64_t)ring_slot(pstart, c0 + i, pcap) * pair_stride)[krow0 + row];
            }
        };
        fetch(0, npairs < kChunk ? npairs : kChunk);
        for (int c0 = 0; c0 < npairs; c0 += kChunk) {
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
                const int e = tid + q * kBlock, i = e >> 6, row = e & (kRows - 1);
                if (i < cn) { Ks[2 * i][row] = -tk[q].x; Ks[2 * i + 1][row] = -tk[q].y; }
                else if (i == cn) { Ks[2 * i][row] = -0.0; Ks[2 * i + 1][row] = -0.0; }
            }
            __syncthreads();
            if (c0 + kChunk < npairs) fetch(c0 + kChunk, npairs - c0 - kChunk < kChunk ? npairs - c0 - kChunk : kChunk);
            const int ksteps = (cn + 1) >> 1;                         // two pairs = four rank-1 terms per MFMA
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

// what was launched, for the measurement hooks (ekf_downdate_kernel_name): "k_xxx<double,128,4,false>"
static void name_kernel(char *out, const char *base, size_t elt, int T, int p3, int xcd) {
    if (!out) return;
    if (xcd < 0) snprintf(out, 64, "%s<%s,%d,%d>", base, elt == 8 ? "double" : "float", T, p3);
    else snprintf(out, 64, "%s<%s,%d,%d,%s>", base, elt == 8 ? "double" : "float", T, p3, xcd ? "true" : "false");
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
            if (use_strip && arith == 1 && npairs > 56 && npairs <= 64 && aux && aux->segs && aux->nsegs > 0) {
                // 57-64 pairs (eight stages of eight): the strip form -- one persistent workgroup per CU walks row strips with -K in its
                // wavefronts' registers and a whole item's G double-buffered in LDS (flush32_pipe.h): 7.3 ms against 8.1-8.3 at 40 000
                // landmarks and 64 pairs, same bits
                static const hipError_t attr = hipFuncSetAttribute((const void *)ekf_pipe32::k_flush_strip32<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                   ekf_pipe32::lds_bytes_strip<8>());
                if (attr == hipSuccess) {
                    hipLaunchKernelGGL((ekf_pipe32::k_flush_strip32<8>), dim3((unsigned)aux->grid), dim3(512), ekf_pipe32::lds_bytes_strip<8>(), s,
                                       (const float *)st.tiles, (float *)dstv, aux->segs, aux->nsegs, (const float *)st.Kp32, (const float *)st.Gp32, st.pair_stride,
                                       st.ldm, pstart, st.pcap, npairs, st.tm, aux->dump, (unsigned long long *)nullptr);
                    if (kname) snprintf(kname, 64, "k_flush_strip32<8>");
                    return true;
                }
            }
            if (arith == 1 && npairs > 2) {                           // cfg.pass_arith = EKF_ARITH_F32: the f32 matrix pipe (one or two pairs: the pass is
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
                hipLaunchKernelGGL((k_flush_mfma<TS, T, 4, 64, 5>), dim3((unsigned)g2), dim3(kBlock), 0, s, (const TS *)st.tiles, (TS *)dstv,
                                   work_xcd, xcd_len, st.Kp, st.Gp, st.pair_stride, pstart, st.pcap, npairs, st.tm);
                if (kname) snprintf(kname, 64, "k_flush_mfma<double,%d,4,64>", T);
                return true;
            }
        }
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

