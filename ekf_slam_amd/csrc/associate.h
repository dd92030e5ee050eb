// Fragment of kernels.hip (included there, inside its anonymous namespace, after kernels.h / device_math.h): k_associate, k_assoc_merge (Correspondence.m:49-87).
#pragma once

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
