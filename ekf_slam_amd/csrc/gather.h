// Fragment of kernels.hip (included there, inside its anonymous namespace, after kernels.h / device_math.h): k_gather: G, phi, K, x, strip of one correction (EKF_SLAM.m:125-144).
#pragma once

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
