// Fragment of kernels.hip (included there, inside its anonymous namespace, after kernels.h / device_math.h): association order and the per-workgroup winner entries of the device-resident measure loop.
#pragma once

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
