// Fragment of kernels.hip (included there, inside its anonymous namespace, after kernels.h / device_math.h): predict (EKF_SLAM.m:40-51,56-65): the shared 3x3 part, k_predict, k_predict_mfma.
#pragma once

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
