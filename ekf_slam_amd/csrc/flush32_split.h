// The pass over float tiles with the rank-2m product on the BF16 matrix pipe in SPLIT arithmetic (cfg.pass_arith = EKF_ARITH_SPLIT3).
//
//   tile(I,J)[r][c] += sum_i ( -K_i(I T + r, x) G_i(x, J T + c) + -K_i(.., y) G_i(y, ..) )        (EKF_SLAM.m:145, m corrections at once)
//
// Every float operand is cut EXACTLY into three bfloat16 pieces (a = a1 + a2 + a3: 3 x 8 significant bits = the float's 24, each piece
// the round-to-nearest bfloat16 of what the previous ones left), and a product a b is evaluated as the six partial products
//     a3 b1 + a2 b2 + a1 b3 + a2 b1 + a1 b2 + a1 b1        (dropped: a2 b3 + a3 b2 + a3 b3: at most 2^-23 |a b|, 0.09 x 2^-24 |a b| in the root mean square)
// each of them EXACT in float (8 x 8 bits), summed in float by v_mfma_f32_16x16x32_bf16, smallest class first, from a ZERO accumulator;
// the float tile value is added once at the end, as in the F32-arithmetic pass (flush32_mfma.h).  The pieces themselves lose nothing; the three
// dropped partial products are bounded by |a2| <= 2^-8 |a|, |a3| <= 2^-16 |a| (tests/test_split3_arith_cpu.py restates the cut in NumPy: exact sums,
// exact partial products, the dropped part measured); the rest is the accumulation error of a float sum -- 24 accumulator roundings per entry
// where the fmaf chain has 128.  Measured entry by entry against an F64 sum (scripts/probes/flush32_bench.hip ACC=1) the split sum is CLOSER than
// the fmaf chain's.  The result is NOT bit-identical to the fmaf chain.
// Why: the bf16 instruction retires 16x the flops per cycle of v_mfma_f32_16x16x4_f32; at six partial products that is 2.7x the F32
// matrix pipe, which moves the 64-pair pass from the matrix pipe's roof (7.3 ms at 40 000 landmarks) to the HBM roof.
//
// Operands: k_split_pairs cuts the float copies of the pending pairs (DevState::Kp32 / Gp32, written by the gather; K negated there) into
// bf16 planes in LOGICAL pair order (the ring is resolved here, pairs beyond npairs are zeros: the pass knows neither pstart nor npairs),
// once per pass -- ~0.25 GB of traffic at 50 000 landmarks, 1 % of the pass.  k index of the matrix instruction: k-block kb = 16 pairs = 32
// k; inside it k = 2 t + xy (pair t, plane xy); lane (lr, lc) of either operand holds k = 8 lr .. 8 lr + 7 (pairs 4 lr .. 4 lr + 3).
//   Kb  "row-major":       bf16 [(p kKB + kb) ldm + row][32]                    a wavefront's A fragment = 16 rows x 64 B, contiguous
//   Gb  "fragment-major":  bf16 [((p kKB + kb) ncg + cg) 4 + e][lr][lc][8]      column 64 cg + 4 lc + e: 1 KiB = one B fragment of a wavefront,
//                                                                              so LDS staging is a plain copy and a fragment read is lane x 16 B
// The kernel: flush32_pipe.h's strip form (one persistent workgroup of eight wavefronts per CU walking row strips; -K -- here its three
// planes, 96 registers -- in registers for a whole segment; the tile as plain 16-byte loads / stores spread over the item's first half;
// ARRIVE / WAIT instead of a barrier) with G staged per CHUNK of two k-blocks: 2 x 3 planes x 128 columns x 64 B = 48 KiB, double-buffered.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flush32_pipe.h"

namespace ekf_pipe32 {

typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));

constexpr int kKB = 4;                              // k-blocks of 16 pairs the operand planes hold (64 pairs)
constexpr int kSplitChunk = 2 * 3 * 2 * 4096;       // G of one chunk in LDS: [kbl 2][plane 3][column group 2][e 4][1 KiB]
constexpr int lds_bytes_split(int waves = 8) { return 2 * kSplitChunk * waves / 8; }
inline size_t split_plane_elems(int64_t ldm) { return (size_t)3 * kKB * (size_t)ldm * 32; }      // bf16 elements of Kb (and of Gb)

// float -> three bfloat16 pieces, exact sum (finite inputs; round-to-nearest-even on the bits)
__device__ __forceinline__ uint32_t bf16_rn_bits(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3(float v, uint32_t &b1, uint32_t &b2, uint32_t &b3) {
    b1 = bf16_rn_bits(v);
    const float r1 = v - __uint_as_float(b1 << 16);
    b2 = bf16_rn_bits(r1);
    const float r2 = r1 - __uint_as_float(b2 << 16);
    b3 = bf16_rn_bits(r2);
}

// grid (ceil(cols / 256), kKB, 2): blockIdx.z = 0 cuts K (src = Kp32) into Kb, 1 cuts G (src = Gp32) into Gb; one thread per (element, k-block)
__global__ __launch_bounds__(256) void k_split_pairs(const float *__restrict__ Kn, const float *__restrict__ G, uint16_t *__restrict__ Kb,
                                                     uint16_t *__restrict__ Gb, int64_t pair_stride, int64_t ldm, int64_t cols, int pstart, int pcap,
                                                     int npairs) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= cols) return;
    const int kb = blockIdx.y;
    const bool isg = blockIdx.z != 0;
    const float *__restrict__ src = isg ? G : Kn;
    float v[32];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int p = 16 * kb + t;
        int sl = pstart + (p < npairs ? p : 0);
        sl -= sl >= pcap ? pcap : 0;
        const float *s = src + (int64_t)sl * pair_stride + e;
        const float x = s[0], y = s[ldm];
        v[2 * t] = p < npairs ? x : 0.0f;
        v[2 * t + 1] = p < npairs ? y : 0.0f;
    }
    uint32_t w[3][16];                              // plane, dword q: k = 2 q (low half), 2 q + 1 (high half)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        uint32_t a1, a2, a3, c1, c2, c3;
        split3(v[2 * q], a1, a2, a3);
        split3(v[2 * q + 1], c1, c2, c3);
        w[0][q] = a1 | (c1 << 16); w[1][q] = a2 | (c2 << 16); w[2][q] = a3 | (c3 << 16);
    }
    const int64_t ncg = ldm >> 6;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int64_t pk = (int64_t)p * kKB + kb;
        uint16_t *d = isg ? Gb + (((pk * ncg + (e >> 6)) * 4 + (e & 3)) * 512 + ((e & 63) >> 2) * 8) : Kb + (pk * ldm + e) * 32;
        const int lrs = isg ? 128 : 8;              // bf16 elements between a record's four 16-byte pieces
#pragma unroll
        for (int lr = 0; lr < 4; ++lr)
            *reinterpret_cast<u4_t *>(d + lr * lrs) = u4_t{ w[p][4 * lr], w[p][4 * lr + 1], w[p][4 * lr + 2], w[p][4 * lr + 3] };
    }
}

// One work-list entry = a 128 x 128 item (flush32_pipe.h: strip_entry, the segments of build_strip_segments); an item = kNCH chunks of two
// k-blocks.  kNCH = 2: up to 64 pairs; kNCH = 1: up to 32 (the planes of k-blocks 2, 3 are not touched).
// kAbl (probe builds only, scripts/probes/flush32_bench.hip): 1 no tile stores, 2 no tile loads, 4 no G loads; 8 / 16 plain instead of nontemporal
// tile stores / loads; 64 s_memtime stamps -- never the product kernel.  (The schedule variants that were measured and lost -- the tile pieces two per
// group, the G block elsewhere in the chunk or staggered between a SIMD's wavefronts, later stores, results stored in one burst, a contiguous-item
// tile layout -- are in profiles/round4_tuning.md 53-56, not here.)
// kW: wavefronts per workgroup.  8: one workgroup per CU, an item = a work-list entry (128 x 128).  4: TWO independent workgroups per CU (the
// same two wavefronts per SIMD), an item = a 64-column half of an entry (both halves by the same workgroup, one after the other: -K stays):
// the two workgroups of a CU share no synchronisation, so their tile-traffic and matrix phases drift apart and interleave on each SIMD.
// kLd0: the group of the item's first tile LOAD (the stores of the previous item's results start at group 0, one per group).  0: a piece's
// register gives up its result and takes the new tile value in the same group (k_flush_strip32's scheme): 5.24 ms at 40 000 landmarks; 2 / 3 /
// 4 / 5 / 8: 4.99 / 4.94 / 4.95 / 4.99 / 5.05 (round4_tuning.md 56) -- the loads a few groups BEHIND the stores, closer to the epilogue that
// needs them and to the store that follows.  (kNCH = 1: the loads cannot leave the item's only chunk.)
template <int kNCH = 2, int kAbl = 0, int kW = 8, int kLd0 = (kNCH == 2 ? 4 : 0)>
__global__ __launch_bounds__(64 * kW)
void k_flush_split3(const float *__restrict__ tiles, float *__restrict__ dst, const int4 *__restrict__ segs, int64_t nsegs,
                    const uint16_t *__restrict__ Kb, const uint16_t *__restrict__ Gb, int64_t ldm, TileMap tm, float *__restrict__ dump) {
    constexpr int T = 256, NG = 8 * kNCH, NKBU = 2 * kNCH;                 // groups (k-block, column block) and k-blocks per item
    static_assert(kNCH == 1 || kNCH == 2, "chunks per item");
    static_assert(kW == 8 || kW == 4, "wavefronts per workgroup");
    static_assert(kLd0 >= 0 && kLd0 + 8 <= 8 * kNCH, "the eight tile loads of an item sit in groups kLd0 .. kLd0 + 7");
    constexpr int kCG = kW / 4;                                            // 64-column groups per item
    constexpr uint32_t kChunk = 2 * 3 * kCG * 4096;                        // G of one chunk in LDS: [kbl 2][plane 3][column group kCG][e 4][1 KiB]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane >> 4, lc = lane & 15;

    // ---- the work list (as k_flush_strip32) ----
    int nbase = 0;
    int4 ent;
    const int64_t myseg = nsegs > (int64_t)blockIdx.x ? (nsegs - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int nent = (int)(myseg * kSeg);
    auto load_entries = [&](int nb) {
        const int n = nb + lane;
        int4 e = make_int4(0, 0, -1, -1);
        if (n < nent) {
            int64_t sg = (int64_t)blockIdx.x + (int64_t)(n / kSeg) * gridDim.x;
            if (tm.reverse) sg = ((nsegs >> 3) - 1 - (sg >> 3)) * 8 + (sg & 7);
            e = segs[sg * kSeg + (n % kSeg)];
        }
        return e;
    };
    ent = load_entries(0);
    int nnext = 0;
    StripItem half2; half2.toff = 0; half2.krow0 = -1; half2.gcol0 = 0;    // kW == 4: the second 64-column half of the entry handed out last
    auto next_item = [&]() {
        StripItem q; q.toff = 0; q.krow0 = -1; q.gcol0 = 0;
        if (kW == 4 && half2.krow0 >= 0) { q = half2; half2.krow0 = -1; return q; }
        while (nnext < nent) {
            const int n = nnext++;
            if (n - nbase >= 64) { nbase = n; ent = load_entries(nbase); }
            const int kr = __builtin_amdgcn_readlane(ent.z, n - nbase);
            if (kr < 0) continue;
            q.toff = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(ent.y, n - nbase) << 32) | (uint32_t)__builtin_amdgcn_readlane(ent.x, n - nbase));
            q.krow0 = kr;
            q.gcol0 = __builtin_amdgcn_readlane(ent.w, n - nbase);
            if (kW == 4) { half2 = q; half2.toff += 64; half2.gcol0 += 64; }
            break;
        }
        return q;
    };
    StripItem cur = next_item();
    if (cur.krow0 < 0) return;
    StripItem nxt = next_item();

    // wavefront w: rows 32 (w >> 1) .. + 31, columns 64 (w & 1) .. + 63 of the item: two row blocks x four column blocks of 16 x 16; lane
    // (lr, lc): A = -K(row 32 wi + 16 rb + lc, k 8 lr ..) in ka[rb][kb][p]; B = G(k 8 lr .., column 64 wj + 4 lc + e); accumulators acc[rb][e][i] =
    // entry (row 32 wi + 16 rb + 4 lr + i, column 64 wj + 4 lc + e): the lane's tile piece 4 rb + i is sixteen consecutive bytes of that row
    const int wi = kW == 8 ? wave >> 1 : wave, wj = kW == 8 ? wave & 1 : 0;
    const uint32_t t_lane = (uint32_t)((32 * wi + 4 * lr) * T + 64 * wj + 4 * lc) * 4;
    bf8_t ka[2][NKBU][3];
    auto load_k = [&](const StripItem &q) {
        uint32_t kl = (uint32_t)((32 * wi + lc) * 64 + lr * 16);
        asm volatile("" : "+v"(kl));
        const char *kq = reinterpret_cast<const char *>(Kb) + (int64_t)q.krow0 * 64;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int kb = 0; kb < NKBU; ++kb) {
                const char *pq = kq + ((int64_t)p * kKB + kb) * ldm * 64;
                ka[0][kb][p] = *reinterpret_cast<const bf8_t *>(pq + kl);
                ka[1][kb][p] = *reinterpret_cast<const bf8_t *>(pq + kl + 16 * 64);
            }
    };
    // G on its way: chunk pieces t = wave, wave + 8, ... + 40 of the 48 KiB (piece t = 1 KiB: block t >> 2 = (kbl 3 + p) 2 + cgl, fragment e = t & 3)
    u4_t gq[6] = {};
    const int64_t ncg = ldm >> 6;
    auto g_src = [&](const StripItem &q, int ch, int j) {
        const int t = wave + kW * j, blk = t >> 2, e = t & 3;
        const int cgl = kCG == 2 ? blk & 1 : 0, kp = kCG == 2 ? blk >> 1 : blk, kbl = kp / 3, p = kp - 3 * kbl;
        const int64_t cg = ((q.krow0 >= 0 ? q.gcol0 : 0) >> 6) + cgl;
        return reinterpret_cast<const char *>(Gb) + ((((int64_t)p * kKB + 2 * ch + kbl) * ncg + cg) * 4 + e) * 1024 + lane * 16;
    };
    auto load_g = [&](const StripItem &q, int ch, int j) { if (!(kAbl & 4)) gq[j] = *reinterpret_cast<const u4_t *>(g_src(q, ch, j)); };
    // Items of odd checkerboard parity compute the update with the OPPOSITE sign (G's sign bits flipped on the way into LDS: exact) and subtract
    // it.  v_mfma_f32_16x16x32_bf16 aligns its 32 products to the largest and cuts the smaller ones off below 2^-24 of it TOWARDS MINUS
    // INFINITY (scripts/probes/mfma_bf16_round.hip), then adds C with round-to-nearest: every entry's sum carries a small bias of one sign
    // (-0.014 float ulps of sum |k g| against the mean error's 0.2) -- invisible per entry, coherent over the 10^10 entries of P and over the
    // passes (the sum-of-entries digest drifted 6e-8 over configs[4]'s 156 passes).  With the sign alternating from item to item the bias
    // alternates too.  Per entry nothing changes: the same products, the same rounding, mirrored.
    auto parity = [&](const StripItem &q) { return (uint32_t)(((q.krow0 ^ q.gcol0) >> 7) & 1); };
    auto write_g = [&](int buf, int j, uint32_t flip) {
        const uint32_t fm = flip ? 0x80008000u : 0u;
        *reinterpret_cast<u4_t *>(smem + (uint32_t)buf * kChunk + (uint32_t)(wave + kW * j) * 1024 + (uint32_t)lane * 16) = gq[j] ^ u4_t{ fm, fm, fm, fm };
    };
    // Tile traffic: ONE register buffer of eight 16-byte pieces carries two items at a time -- piece p's register gives up the PREVIOUS item's
    // finished entries in group p (a store) and takes the CURRENT item's tile value in group kLd0 + p (a load); at the item's end the
    // accumulators are added into it.  (Requesting the NEXT item's pieces at the epilogue instead -- eight stores and eight loads in one burst
    // per wavefront, all eight wavefronts at once -- measured slower: 5.8 against 5.2 ms, round4_tuning.md 53.)
    f4_t acc[2][4], tl[8];
    auto zero_acc = [&]() {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[rb][e] = f4_t{ 0.0f, 0.0f, 0.0f, 0.0f };
    };
    auto piece_off = [&](int p) { return (size_t)(((p >> 2) * 16 + (p & 3)) * T) * 4 + t_lane; };
    const char *out_base = reinterpret_cast<const char *>(dump + (size_t)blockIdx.x * (kItem * T)), *in_base = nullptr;     // dump: 128 KiB per workgroup
#pragma unroll
    for (int p = 0; p < 8; ++p) tl[p] = f4_t{ 0.0f, 0.0f, 0.0f, 0.0f };
    bf8_t fb[2][3];                                                        // fragment sets: group gi in set gi & 1
    auto read_frags = [&](int buf, int gl, bf8_t (&b)[3]) {                // gl: group inside the chunk (kbl = gl >> 2, e = gl & 3)
        const uint32_t o = (uint32_t)buf * kChunk + (uint32_t)((gl >> 2) * 3 * kCG + wj) * 4096 + (uint32_t)(gl & 3) * 1024 + (uint32_t)lane * 16;
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf8_t *>(smem + o + (uint32_t)p * (kCG * 4096));
    };
    auto mfma_group = [&](int kb, int e, const bf8_t (&b)[3]) {
        // the six partial products, smallest class first; the two row blocks alternate (two independent chains)
#define EKF_SP(PA, PB) do { acc[0][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[0][kb][PA], b[PB], acc[0][e], 0, 0, 0); \
                            acc[1][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[1][kb][PA], b[PB], acc[1][e], 0, 0, 0); } while (0)
        EKF_SP(2, 0); EKF_SP(1, 1); EKF_SP(0, 2); EKF_SP(1, 0); EKF_SP(0, 1); EKF_SP(0, 0);
#undef EKF_SP
    };

    // prologue: the first chunk's G straight into buffer 0; the second chunk's (this item's second half, or the next item's only one) into the registers
#pragma unroll
    for (int j = 0; j < 6; ++j) load_g(cur, 0, j);
    load_k(cur);
#pragma unroll
    for (int j = 0; j < 6; ++j) { write_g(0, j, parity(cur)); if (kNCH == 2) load_g(cur, 1, j); else load_g(nxt, 0, j); }
    zero_acc();
    __shared__ int arrived;
    if (tid == 0) __hip_atomic_store(&arrived, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int buf = 0, target = 0;
    StripItem nn = next_item();
    read_frags(buf, 0, fb[0]);
    // kAbl & 64: s_memtime stamps summed per wavefront -- first chunk | WAIT between the chunks | second chunk | epilogue (the wait for the tile
    // pieces, the adds) | WAIT at the item's end -- written over the workgroup's dump area at the end ([wave][8] 64-bit counts; [5] = items)
    unsigned long long seg[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tprev = 0;          // [6], [7]: inside the chunks -- the G block, the tile pieces
    auto sub_mark = [&](int which, unsigned long long t0) { if constexpr ((kAbl & 64) != 0) seg[which] += stamp_now() - t0; };
    auto sub_start = [&]() { unsigned long long t = 0; if constexpr ((kAbl & 64) != 0) t = stamp_now(); return t; };
    auto mark = [&](int which) { if constexpr ((kAbl & 64) != 0) { const unsigned long long t = stamp_now(); seg[which] += t - tprev; tprev = t; } };
    if constexpr ((kAbl & 64) != 0) tprev = stamp_now();
    for (;;) {
        const bool newk = nxt.krow0 >= 0 && nxt.krow0 != cur.krow0;
        in_base = reinterpret_cast<const char *>(tiles + cur.toff);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int ch = gi >> 3, gl = gi & 7;
            if (gl < 7) read_frags(buf, gl + 1, fb[(gi + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (gl == 7) {
                // ARRIVE: this wavefront has read the last of this chunk's G (its fragments are in registers) and written its share of the next chunk's
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(&arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (gl == 0) {
                // The next chunk's G into the buffer the last WAIT freed, the chunk after it on its way: all six pieces here, in front of the chunk's
                // tile traffic.  (Vector-memory operations retire in issue order, so a wait for a G piece also waits for every older load and store;
                // here the G pieces waited for are older than every tile piece in flight.  Spread over the chunk between the tile pieces --
                // k_flush_strip32's order -- the pass measured the same, 5.18 ms at 40 000 landmarks: round4_tuning.md 53.)
                const unsigned long long tg0 = sub_start();
                const uint32_t flip_next = (kNCH == 2 && ch == 0) ? parity(cur) : parity(nxt);      // whose G these pieces are
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    write_g(buf ^ 1, j, flip_next);
                    __builtin_amdgcn_sched_barrier(0);                     // (hipcc otherwise hoists the six loads over the writes, into six more registers each: spills)
                    if (kNCH == 2) { if (ch == 0) load_g(nxt, 0, j); else load_g(nxt, 1, j); }
                    else load_g(nn, 0, j);
                    __builtin_amdgcn_sched_barrier(0);
                }
                sub_mark(6, tg0);
            }
            mfma_group(2 * ch + (gl >> 2), gl & 3, fb[gi & 1]);
            if (gi < 8) {
                const unsigned long long tt0 = sub_start();
                // the tile pieces: the register first gives up the PREVIOUS item's finished entries, then takes this item's tile value
                {
                    const int p = gi;
                    f4_t *po = reinterpret_cast<f4_t *>(const_cast<char *>(out_base) + piece_off(p));
                    const f4_t *pi = reinterpret_cast<const f4_t *>(in_base + piece_off(p));
                    if (!(kAbl & 1)) { if (kAbl & 8) *po = tl[p]; else __builtin_nontemporal_store(tl[p], po); }
                    if (!(kAbl & 2) && kLd0 == 0) { if (kAbl & 16) tl[p] = *pi; else tl[p] = __builtin_nontemporal_load(pi); }
                }
                sub_mark(7, tt0);
            }
            if (kLd0 > 0 && gi >= kLd0 && gi < kLd0 + 8 && !(kAbl & 2))     // the tile loads, kLd0 groups behind the stores
                tl[gi - kLd0] = __builtin_nontemporal_load(reinterpret_cast<const f4_t *>(in_base + piece_off(gi - kLd0)));
            __builtin_amdgcn_sched_barrier(0);
            if (gl == 7 && gi + 1 < NG) {
                // WAIT between the chunks of an item
                mark(0);
                buf ^= 1;
                target += kW;
                while (__hip_atomic_load(&arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
                read_frags(buf, 0, fb[(gi + 1) & 1]);
                mark(1);
            }
        }
        mark(2);
        if (newk) load_k(nxt);
        const float sgn = parity(cur) ? -1.0f : 1.0f;
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) tl[p][e] = fmaf(acc[p >> 2][e][p & 3], sgn, tl[p][e]);     // (exact product: one rounding, as tl + acc)
        zero_acc();
        out_base = reinterpret_cast<const char *>(dst + cur.toff);
        mark(3);
        seg[5] += 1;
        if (nxt.krow0 < 0) break;
        cur = nxt;
        nxt = nn;
        nn = next_item();
        buf ^= 1;
        target += kW;
        while (__hip_atomic_load(&arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
        read_frags(buf, 0, fb[0]);
        mark(4);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p)                                            // the last item's result
        __builtin_nontemporal_store(tl[p], reinterpret_cast<f4_t *>(const_cast<char *>(out_base) + piece_off(p)));
    if constexpr ((kAbl & 64) != 0) {
        unsigned long long *st = reinterpret_cast<unsigned long long *>(dump + (size_t)blockIdx.x * (kItem * T)) + wave * 8;
        if (lane == 0)
            for (int q = 0; q < 8; ++q) st[q] = seg[q];
    }
}

}  // namespace ekf_pipe32