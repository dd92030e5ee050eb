// The pass over float tiles in F32 arithmetic at 57-64 pending pairs (cfg.pass_arith = EKF_ARITH_F32, cfg.batch = 64: configs[4]).
//
//   tile(I,J)[r][c] += sum_i ( -K_i(I T + r, x) G_i(x, J T + c) + -K_i(.., y) G_i(y, ..) )        (EKF_SLAM.m:145, m corrections at once)
//
// summed in float FROM ZERO in ring order on v_mfma_f32_16x16x4_f32 (a k-ordered chain of fmaf), the float tile value added once at the
// end: the arithmetic -- and therefore every bit of the result -- of k_flush_mfma32 (flush32_mfma.h; scripts/probes/flush32_bench.hip
// checks both against a scalar fmaf chain, every entry, at 40 000 landmarks).  What differs is how the bytes move: k_flush_mfma32 runs one
// work item per workgroup as  fetch operands -> matrix loop -> load tile -> add -> store  with three workgroups per CU hiding each
// other's latencies, and at 64 pairs its time is the SUM of its tile traffic and its matrix work (profiles/round3_tuning.md 39).
// Here one persistent workgroup per CU keeps the matrix pipe fed (k_flush_strip32 below; how it got there: profiles/round4_tuning.md).
//
// Operand layout (DevState::Kp32 / Gp32 as the gather writes them, "planar"): slot s, plane xy, element e at
// [s * pair_stride + xy * ldm + e]; the K copies are stored NEGATED (-(float)K: exact).  Pairs beyond npairs in the last stage of eight
// enter as -0.0f (-K) and +0.0f (G): fmaf(-0, +0, acc) == acc for every acc.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace ekf_pipe32 {

typedef float f4_t __attribute__((ext_vector_type(4)));

constexpr int kItem = 128;                          // a work item: 128 rows x 128 columns of a 256 x 256 float tile
constexpr int kPiece = 1024;                        // G of one pair for an item's 128 columns: 2 planes x 128 floats
constexpr int kDumpFloats = kItem * 256;            // per workgroup: where its first item's meaningless first stores go

// kStamp (diagnostic builds only, scripts/probes/flush32_bench.hip): s_memtime stamps around the segments of a stage, summed per
// wavefront in scalar registers and written to `stamps` ([workgroup][wave][8] cycles) at the end; never the product kernel.
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Row strips: eight wavefronts per CU, -K in their REGISTERS, G of a whole item double-buffered in LDS, ONE barrier per item.
// How it got here (profiles/round4_tuning.md; scripts/probes/glds_issue_rate.hip, glds_coexec.hip, mfma_issue_rate.hip):
//   * LDS-DMA is out: an LDS-DMA instruction issued from inside an MFMA stream stalls its wavefront for hundreds of cycles, and a CU keeps
//     only ~8 of them in flight (60-78 GB/s at L2 latency, ~10 GB/s beyond it).  The TILE (half of all bytes, always an HBM miss) travels as
//     plain loads into the registers of the wavefront that will add and store it, with half an item's matrix work to hide behind; G --
//     asked for by every CU of an XCD at the same moment, so every request waits out the same miss -- as plain loads TWO ITEMS ahead;
//   * a barrier per 8-pair stage (-K in LDS, a ring of G stages) left the matrix pipe idle for 590 of a stage's 2 670 cycles.  As the A
//     operand of v_mfma_f32_16x16x4_f32, the -K of 32 rows for all 64 pairs is 64 registers, constant along a row strip: with -K out of LDS
//     a whole item's G (64 KiB at 64 pairs) fits twice, and the wavefronts run an item's 256 MFMAs each with no synchronisation at all;
//   * every LDS read costs its wavefront ~28 cycles of MFMA issue, prefetched or not (8 MFMAs + two 16-byte reads: 39 cycles per MFMA
//     instead of 32), and the SIMD's arbiter serves the older of its two wavefronts first, strictly: the younger only fills gaps.  So a
//     wavefront owns 32 rows x 64 columns -- ONE 16-byte read per eight MFMAs;
//   * that shape needs ~175 registers, more than the 168 a third wavefront per SIMD would leave: there are no loader wavefronts.  Each
//     wavefront carries an eighth of the NEXT-BUT-ONE item's G in registers (eight 16-byte pieces, loaded during this item, written to the
//     free LDS buffer during the next one's first half): a whole item (~8 us) for the round trip;
//   * operand bytes: a workgroup walks ALONG A ROW STRIP -- up to kSeg consecutive 128-column items of one 128-row slab; -K is fetched once
//     per strip segment (64 dword loads per wavefront), per item only G.
// LDS: G of the item in work | G of the next item (kS x 8 KiB each).
constexpr int kSeg = 16;                            // items per strip segment (the work list is cut into segments of kSeg entries)
template <int kS> constexpr int lds_bytes_strip() { return 2 * kS * 8 * kPiece; }

struct StripItem { int64_t toff; int krow0, gcol0; };      // krow0 < 0: none  (no padding bytes: a padded struct is copied through scratch)

// one work-list entry for the item (tile (I, J), row half `slab`, column half `cpart`); (0, 0, -1, -1) pads a short segment
inline int4 strip_entry(const TileMap &tm, int I, int J, int slab, int cpart) {
    const int64_t toff = tm.tile_offset(I, J) + (int64_t)(slab * kItem) * 256 + cpart * kItem;
    return make_int4((int)(uint32_t)(toff & 0xffffffffll), (int)(uint32_t)((uint64_t)toff >> 32), I * 256 + slab * kItem, J * 256 + cpart * kItem);
}

// kS: stages of eight pairs per item = ceil(npairs / 8) (a template parameter: the k-step loop is unrolled, see the tile traffic below)
// kLd: the k-step of the item's first tile LOAD (piece p's load at k-step kLd + 2 p; its store -- the previous item's result -- stays at k-step 2 p)
template <int kS = 8, bool kStamp = false, int kLd = 0>
__global__ __launch_bounds__(512)
void k_flush_strip32(const float *__restrict__ tiles, float *__restrict__ dst, const int4 *__restrict__ segs, int64_t nsegs,
                     const float *__restrict__ Kn, const float *__restrict__ G, int64_t pair_stride, int64_t ldm, int pstart, int pcap,
                     int npairs, TileMap tm, float *__restrict__ dump, unsigned long long *__restrict__ stamps = nullptr) {
    constexpr int T = 256, NP = 8 * kS, NK = 4 * kS;                       // pair pieces and k-steps (of v_mfma_f32_16x16x4_f32) per item
    constexpr uint32_t kBuf = (uint32_t)NP * kPiece;                       // one item's G: pair piece p at p KiB, plane xy at + 512 xy
    static_assert(kS >= 5 && kS <= 8, "stages per item (the G pieces and the tile pieces move in the first 16 k-steps)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane >> 4, lc = lane & 15;
    unsigned long long seg[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tprev = 0;
    auto mark = [&](int which) { if constexpr (kStamp) { const unsigned long long t = stamp_now(); seg[which] += t - tprev; tprev = t; } };

    // ---- the work list: segment sg = blockIdx.x + k gridDim.x, entries segs[sg * kSeg ..), each (tile-store offset of the item's first
    // entry: low, high word; first landmark-block row; first landmark-block column), precomputed by the host (strip_entry above): the index
    // arithmetic from (I, J, sub-block) costs ~1 200 scalar cycles per item and wavefront, with the matrix pipe idle ----
    int nbase = 0;                                                        // this workgroup's entry numbers nbase .. nbase + 63 sit in `ent`
    int4 ent;
    const int64_t myseg = nsegs > (int64_t)blockIdx.x ? (nsegs - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int nent = (int)(myseg * kSeg);
    auto load_entries = [&](int nb) {
        const int n = nb + lane;
        int4 e = make_int4(0, 0, -1, -1);
        if (n < nent) {
            int64_t sg = (int64_t)blockIdx.x + (int64_t)(n / kSeg) * gridDim.x;          // nsegs is a multiple of 8: position sg >> 3 of stream sg & 7
            if (tm.reverse) sg = ((nsegs >> 3) - 1 - (sg >> 3)) * 8 + (sg & 7);           // alternate passes walk every stream backwards
            e = segs[sg * kSeg + (n % kSeg)];
        }
        return e;
    };
    ent = load_entries(0);
    int nnext = 0;
    auto next_item = [&]() {
        StripItem q; q.toff = 0; q.krow0 = -1; q.gcol0 = 0;
        while (nnext < nent) {
            const int n = nnext++;
            if (n - nbase >= 64) { nbase = n; ent = load_entries(nbase); }
            const int kr = __builtin_amdgcn_readlane(ent.z, n - nbase);
            if (kr < 0) continue;                                          // padding of a short segment
            q.toff = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(ent.y, n - nbase) << 32) | (uint32_t)__builtin_amdgcn_readlane(ent.x, n - nbase));
            q.krow0 = kr;
            q.gcol0 = __builtin_amdgcn_readlane(ent.w, n - nbase);
            break;
        }
        return q;
    };
    StripItem cur = next_item();
    if (cur.krow0 < 0) return;
    StripItem nxt = next_item(), nn = next_item();                         // the items after this one
    auto slot_of = [&](int p) { int sl = pstart + p; if (sl >= pcap) sl -= pcap; return sl; };

    // wavefront w: rows 32 (w >> 1) .. + 31 of the item, columns 64 (w & 1) .. + 63: two row blocks x four column blocks of 16 x 16.  Lane
    // (lr, lc) holds, of k-step g (k = 4 g + lr: pair 2 g + (lr >> 1), plane lr & 1): A = -K(row 32 wi + 16 rb + lc, k) in ka[rb][g];
    // B = G(k, columns 64 wj + 4 lc + e) in fb[.][e] -- ONE 16-byte LDS read per k-step feeds eight MFMAs; accumulators acc[rb][e][i] = entry
    // (row 32 wi + 16 rb + 4 lr + i, column 64 wj + 4 lc + e): the lane's tile piece p = 4 rb + i is sixteen consecutive bytes of that row.
    const int wi = wave >> 1, wj = wave & 1;
    const uint32_t b_off = (uint32_t)(lr >> 1) * kPiece + (uint32_t)(lr & 1) * 512 + (uint32_t)wj * 256 + (uint32_t)lc * 16;       // + buf kBuf + g 2 KiB
    const uint32_t t_lane = (uint32_t)((32 * wi + 4 * lr) * T + 64 * wj + 4 * lc) * 4;
    float ka[2][NK];
    const uint32_t k_lane = (uint32_t)(lr & 1) * (uint32_t)ldm + (uint32_t)(32 * wi + lc);
    auto load_k = [&](const StripItem &q) {
        // (32-bit element offsets from a scalar base, recomputed here each time: hoisted out of the item loop, the addresses spill)
        uint32_t kl = k_lane;
        int ps = pstart, h = lr >> 1;
        asm volatile("" : "+v"(kl), "+v"(h), "+s"(ps));
        const float *kq = Kn + q.krow0;
#pragma unroll
        for (int g = 0; g < NK; ++g) {                                     // all 8 kS loads in flight at once (pairs beyond npairs: pair 0's, masked below)
            const int p = 2 * g + h;
            int sl = ps + (p < npairs ? p : 0);
            sl -= sl >= pcap ? pcap : 0;
            const uint32_t o = (uint32_t)sl * (uint32_t)pair_stride + kl;
            ka[0][g] = kq[o];
            ka[1][g] = kq[o + 16];
        }
    };
    auto mask_k = [&]() {
#pragma unroll
        for (int g = 0; g < NK; ++g) {
            const bool in = 2 * g + (lr >> 1) < npairs;
            ka[0][g] = in ? ka[0][g] : -0.0f;
            ka[1][g] = in ? ka[1][g] : -0.0f;
        }
    };
    // G on its way: this wavefront's pair pieces w, w + 8, ... (lane l: plane l >> 5, columns 4 (l & 31) ..) of the item after next, in
    // registers from one item's first half to the next one's, where they are written to the LDS buffer the barrier in between has freed
    constexpr int GQ = NP / 8;
    f4_t gq[GQ];
    const uint32_t g_lane = ((uint32_t)(lane >> 5) * (uint32_t)ldm + 4u * (uint32_t)(lane & 31)) * 4u;
    const f4_t gzero = { 0.0f, 0.0f, 0.0f, 0.0f };
    auto load_g = [&](const StripItem &q, int j) {
        const int p = wave + 8 * j;
        gq[j] = *reinterpret_cast<const f4_t *>(reinterpret_cast<const char *>(G + (int64_t)slot_of(p < npairs ? p : 0) * pair_stride + (q.krow0 >= 0 ? q.gcol0 : 0)) + g_lane);
    };
    auto write_g = [&](int buf, int j) {
        const int p = wave + 8 * j;
        // (p < NP = 8 GQ; only the last stage has pairs beyond npairs -- the launcher guarantees kS == ceil(npairs / 8))
        *reinterpret_cast<f4_t *>(smem + (uint32_t)buf * kBuf + (uint32_t)p * kPiece + (uint32_t)lane * 16) = (j < GQ - 1 || p < npairs) ? gq[j] : gzero;
    };
    // Tile traffic, spread over the item instead of bursting at its ends.  ONE register buffer of eight 16-byte pieces carries two items at
    // a time: in the item's first half each piece's register first gives up the PREVIOUS item's finished entries (a store), then takes the
    // CURRENT item's tile value (a load, which has at least half an item to arrive); at the item's end the accumulators are added into it.
    // For hipcc to get the waits right the K-STEP LOOP IS UNROLLED (kS a template parameter): with a run-time index it cannot tell which
    // piece registers have a load pending and puts s_waitcnt vmcnt(0) in front of every store -- which then waits out the load issued two
    // instructions earlier, a full HBM round trip per piece (450 cycles each, measured).  Unrolled, it sees that no store names a register
    // with a pending load and the only wait is the one in front of the final adds.  (Loads hidden in asm are not an option: the register
    // allocator moves the destination registers of asm outputs around before the data has landed.)  The G pieces move the same way.
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
    // fragment sets: the LDS read of k-step g + 2 is issued in front of the eight MFMAs of k-step g; the sched_barriers keep hipcc from
    // sinking the reads to their uses
    f4_t fb[4];
    auto read_frag = [&](int buf, int g, f4_t &b) {
        b = *reinterpret_cast<const f4_t *>(smem + b_off + (uint32_t)buf * kBuf + (uint32_t)g * (2 * kPiece));
    };
    auto mfma_step = [&](int g, const f4_t &b) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[rb][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[rb][g], b[e], acc[rb][e], 0, 0, 0);
    };
    // where a wavefront ARRIVES at the item's barrier: behind the MFMAs of the last k-step but two (its last fragment read has been issued)
    constexpr int kBar = NK - 3;
    constexpr int kG0 = kBar - 2 * (GQ - 1);                                         // the G pieces move at k-steps kG0, kG0 + 2, ... <= kBar: the tile pieces own the first half
    static_assert(kG0 >= 0 && kG0 + 2 * (GQ - 1) <= kBar, "the G pieces are in LDS before the item's barrier");
    static_assert(NK % 4 == 0, "k-steps per item");
    static_assert(kLd >= 0 && kLd % 2 == 0 && kLd + 16 <= NK, "the tile loads end before the item does");

    // prologue: the first item's G straight into buffer 0, the second's into the registers
#pragma unroll
    for (int j = 0; j < GQ; ++j) load_g(cur, j);
    load_k(cur);
#pragma unroll
    for (int j = 0; j < GQ; ++j) { write_g(0, j); load_g(nxt, j); }
    mask_k();
    zero_acc();
    // The item's barrier, in two halves.  ARRIVE (k-step kBar: this wavefront has read the last of this item's G and written its share of
    // the next item's): one LDS add.  WAIT (after the item's epilogue, in front of the next item's first fragment read): spin until all
    // eight have arrived.  Between the two a wavefront runs its last k-steps and its epilogue (wait for the tile, 32 adds) -- with one
    // s_barrier in place of the pair all eight did that at the same moment, the matrix pipe idle for ~1 400 cycles per item (STAMP=2).
    __shared__ int arrived;
    if (tid == 0) __hip_atomic_store(&arrived, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // (P)
    int buf = 0, target = 0;                                               // G buffer of the item in work; arrivals that open the next one
    read_frag(buf, 0, fb[0]);
    read_frag(buf, 1, fb[1]);
    if constexpr (kStamp) tprev = stamp_now();
    for (;;) {
        const bool newk = nxt.krow0 >= 0 && nxt.krow0 != cur.krow0;
        in_base = reinterpret_cast<const char *>(tiles + cur.toff);
        StripItem n3 = nn;
#pragma unroll
        for (int g = 0; g < NK; ++g) {
            // on entry: the fragments of k-steps g and g + 1 are in (or on their way to) sets g & 3 and (g + 1) & 3
            if (g + 2 < NK) read_frag(buf, g + 2, fb[(g + 2) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(g, fb[g & 3]);
            if (g < 16 && g % 2 == 0) {
                // (unconditional on purpose -- the workgroup's FIRST item stores its still meaningless registers to a dump area: with a
                // conditional store hipcc loads into a temporary, waits for it at once and copies)
                const int p = g / 2;
                __builtin_nontemporal_store(tl[p], reinterpret_cast<f4_t *>(const_cast<char *>(out_base) + piece_off(p)));
                if (kLd == 0) tl[p] = __builtin_nontemporal_load(reinterpret_cast<const f4_t *>(in_base + piece_off(p)));
            }
            if (kLd > 0 && g >= kLd && g < kLd + 16 && (g - kLd) % 2 == 0)
                tl[(g - kLd) / 2] = __builtin_nontemporal_load(reinterpret_cast<const f4_t *>(in_base + piece_off((g - kLd) / 2)));
            if (g >= kG0 && (g - kG0) % 2 == 0 && (g - kG0) / 2 < GQ) {     // the next item's G into the buffer the last barrier freed; the one after it on its way
                write_g(buf ^ 1, (g - kG0) / 2);
                load_g(nn, (g - kG0) / 2);
            }
            if (g == 7) mark(6);
            if (g == 15) mark(7);
            if (g == kBar) mark(0);
            if (g == 17) n3 = next_item();                                 // (scalar work, between the MFMAs)
            if (g == kBar) {                                               // ARRIVE: this item's last read (k-step NK - 1) has retired, the next item's G pieces are written
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(&arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        mark(2);
        // a new row slab: its -K replaces this one's as soon as the last MFMA has been issued
        if (newk) load_k(nxt);
        // the item's result takes its tile's place in the registers; it leaves during the next item's first half
        mark(5);                                                           // (hipcc counts the wait for the tile pieces itself: vmcnt(8 + ...), the G pieces stay in flight)
        if (newk) mask_k();
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) tl[p][e] += acc[p >> 2][e][p & 3];
        zero_acc();
        out_base = reinterpret_cast<const char *>(dst + cur.toff);
        mark(3);
        if (nxt.krow0 < 0) break;
        buf ^= 1;
        cur = nxt;
        nxt = nn;
        nn = n3;
        mark(4);
        target += 8;                                                       // WAIT
        while (__hip_atomic_load(&arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
        mark(1);
        read_frag(buf, 0, fb[0]);
        read_frag(buf, 1, fb[1]);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p)                                            // the last item's result
        __builtin_nontemporal_store(tl[p], reinterpret_cast<f4_t *>(const_cast<char *>(out_base) + piece_off(p)));
    if constexpr (kStamp) {
        if (lane == 0)
            for (int q = 0; q < 8; ++q) stamps[((int64_t)blockIdx.x * 8 + wave) * 8 + q] = seg[q];
    }
}

}  // namespace ekf_pipe32
