// Fragment of kernels.hip (included there, inside its anonymous namespace, after kernels.h / device_math.h): k_append (EKF_SLAM.m:67-98).
#pragma once

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
