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
// [s * pair_stride + xy * ldm + e]; the K copies are stored NEGATED (-(float)K: exact), so the pieces are plain copies.  Pairs beyond
// npairs in the last stage come from a page of zeros (-0.0f for -K, +0.0f for G: fmaf(-0, +0, acc) == acc for every acc).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace ekf_pipe32 {

typedef float f4_t __attribute__((ext_vector_type(4)));

constexpr int kItem = 128;                          // a work item: 128 rows x 128 columns of a 256 x 256 float tile
constexpr int kPiece = 1024;                        // one LDS-DMA wave instruction: 64 lanes x 16 bytes
constexpr int kKBytes = 64 * kPiece;                // -K of one 128-row slab for 64 pairs: one piece per pair
constexpr int kZeroFloats = 512;                    // the page of zeros: [0, 256) -0.0f, [256, 512) +0.0f
constexpr int kDumpFloats = kItem * 256;            // per workgroup: where its first item's meaningless first stores go

// one 1 KiB piece: lane l's 16 bytes at base + voff -> LDS [lds_dst + 16 l]  (M0 is compiler-reserved: saved and restored)
__device__ __forceinline__ void glds16(const void *base, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform n known only at run time (the immediate must be a literal)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define EKF_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        EKF_VM(0) EKF_VM(1) EKF_VM(2) EKF_VM(3) EKF_VM(4) EKF_VM(5) EKF_VM(6) EKF_VM(7) EKF_VM(8) EKF_VM(9) EKF_VM(10) EKF_VM(11) EKF_VM(12) EKF_VM(13) EKF_VM(14) EKF_VM(15) EKF_VM(16) EKF_VM(17) EKF_VM(18) EKF_VM(19) EKF_VM(20) EKF_VM(21) EKF_VM(22) EKF_VM(23) EKF_VM(24) EKF_VM(25) EKF_VM(26) EKF_VM(27) EKF_VM(28) EKF_VM(29) EKF_VM(30) EKF_VM(31) EKF_VM(32) EKF_VM(33) EKF_VM(34) EKF_VM(35) EKF_VM(36) EKF_VM(37) EKF_VM(38) EKF_VM(39) EKF_VM(40) EKF_VM(41) EKF_VM(42) EKF_VM(43) EKF_VM(44) EKF_VM(45) EKF_VM(46) EKF_VM(47) EKF_VM(48)
#undef EKF_VM
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// every LDS read of this wavefront retired, then the workgroup's barrier (raw: __syncthreads() would drain the LDS-DMA queue)
__device__ __forceinline__ void lds_done_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

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
// Row strips with -K RESIDENT in LDS, dedicated LOADER wavefronts, eight consumer wavefronts.
// What its two predecessors taught (a 512-thread workgroup per CU with every wavefront issuing its own pieces; 256-thread workgroups,
// two or three per CU; profiles/round4_tuning.md, scripts/probes/glds_issue_rate.hip, glds_coexec.hip):
//   * an LDS-DMA instruction issued from inside an MFMA stream stalls its wavefront for hundreds of cycles, and both wavefronts of a SIMD
//     do it at the same moment (lockstep behind the stage's barrier): no consumer may issue operand loads;
//   * a CU keeps only ~24 LDS-DMA pieces in flight: at L2 latency that is 60-80 GB/s, at HBM latency under this kernel's load ~12 GB/s --
//     so the TILE (half of all bytes, always an HBM miss) must not travel by LDS-DMA.  Each consumer loads the sixteen-byte pieces its own
//     lanes will add and store straight into registers at the item's start (plain loads, eight per wavefront; their latency has the whole
//     item's matrix work to hide behind), and LDS-DMA carries only the L2-resident operands;
//   * operand bytes: a workgroup walks ALONG A ROW STRIP -- up to kSeg consecutive 128-column items of one 128-row slab -- and keeps that
//     slab's -K for all pending pairs in LDS (64 KiB at 64 pairs), loaded once per strip segment: per item only G (64 KiB at 64 pairs).
// LDS: -K 2 x 64 KiB (the strip segment in work, and the next one's, requested during this one's last item) | G ring (D + 1) x 8 KiB.
constexpr int kSeg = 16;                            // items per strip segment (the work list is cut into segments of kSeg entries)
template <int D> constexpr int lds_bytes_strip() { return 2 * kKBytes + (D + 1) * 8 * kPiece; }

struct StripItem { int64_t toff, krow0, gcol0; bool ok; };

// one work-list entry for the item (tile (I, J), row half `slab`, column half `cpart`); (0, 0, -1, -1) pads a short segment
inline int4 strip_entry(const TileMap &tm, int I, int J, int slab, int cpart) {
    const int64_t toff = tm.tile_offset(I, J) + (int64_t)(slab * kItem) * 256 + cpart * kItem;
    return make_int4((int)(uint32_t)(toff & 0xffffffffll), (int)(uint32_t)((uint64_t)toff >> 32), I * 256 + slab * kItem, J * 256 + cpart * kItem);
}

// kS: stages per item = ceil(npairs / 8) (a template parameter: see the tile traffic below); NL loader wavefronts (1, 2, 4 or 8): piece x
// of a stage / of -K is loader x % NL's
template <int D, int NL = 2, int kS = 8, bool kStamp = false>
__global__ __launch_bounds__(512 + 64 * NL)
void k_flush_strip32(const float *__restrict__ tiles, float *__restrict__ dst, const int4 *__restrict__ segs, int64_t nsegs,
                     const float *__restrict__ Kn, const float *__restrict__ G, int64_t pair_stride, int64_t ldm, int pstart, int pcap,
                     int npairs, TileMap tm, const float *__restrict__ zeros, float *__restrict__ dump, unsigned long long *__restrict__ stamps = nullptr) {
    constexpr int R = D + 1, T = 256, P = 8;      // a stage: eight pairs = four k-steps of v_mfma_f32_16x16x4_f32
    constexpr uint32_t kKres = 0, kRing = 2 * kKBytes, kSlot = 8 * kPiece;
    static_assert(NL == 1 || NL == 2 || NL == 4 || NL == 8, "loader wavefronts");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 8;
    const int lid = wave - 8;                                             // which loader
    const int lr = lane >> 4, lc = lane & 15;
    constexpr int S = kS;                                                 // stages per item; the launcher guarantees kS == ceil(npairs / 8) > D
    static_assert(kS > D + 1 && kS <= 8, "stages per item");
    unsigned long long seg[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tprev = 0;
    auto mark = [&](int which) { if constexpr (kStamp) { const unsigned long long t = stamp_now(); seg[which] += t - tprev; tprev = t; } };

    // ---- the work list: segment sg = blockIdx.x + k gridDim.x, entries segs[sg * kSeg ..), each (tile-store offset of the item's first
    // entry: low, high word; first landmark-block row; first landmark-block column), precomputed by the host (strip_entry below): the index
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
        StripItem q; q.ok = false; q.toff = 0; q.krow0 = 0; q.gcol0 = 0;
        while (nnext < nent) {
            const int n = nnext++;
            if (n - nbase >= 64) { nbase = n; ent = load_entries(nbase); }
            const int kr = __builtin_amdgcn_readlane(ent.z, n - nbase);
            if (kr < 0) continue;                                          // padding of a short segment
            q.ok = true;
            q.toff = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(ent.y, n - nbase) << 32) | (uint32_t)__builtin_amdgcn_readlane(ent.x, n - nbase));
            q.krow0 = kr;
            q.gcol0 = __builtin_amdgcn_readlane(ent.w, n - nbase);
            break;
        }
        return q;
    };
    StripItem cur = next_item();
    if (!cur.ok) return;
    StripItem nxt = next_item(), nn = nxt;                                 // nn: the item after next, looked up in the middle of each item

    if (loader) {
        // =========================== the loader wavefronts ===========================
#if !defined(EKF_LOADER_PRIO0)
        __builtin_amdgcn_s_setprio(3);     // the youngest wavefront of its SIMD would lose every issue arbitration to the two MFMA streams beside it
#endif
        const uint32_t g_lane = ((uint32_t)(lane >> 5) * (uint32_t)ldm + 4u * (uint32_t)(lane & 31)) * 4u;
        const uint32_t k_lane = ((uint32_t)(lane & 1) * (uint32_t)ldm + 4u * (uint32_t)(lane >> 1)) * 4u;
        const uint32_t z_lane = (uint32_t)lane * 16;
        auto slot_of = [&](int p) { int sl = pstart + p; if (sl >= pcap) sl -= pcap; return sl; };
        auto issue_k = [&](const StripItem &q, int kb) {                   // all pairs' -K for q's row slab into buffer kb: one piece per pair
            for (int p = lid; p < S * P; p += NL) {
                const bool real = p < npairs;
                glds16(real ? Kn + (int64_t)slot_of(real ? p : 0) * pair_stride + q.krow0 : zeros, real ? k_lane : z_lane,
                       lds0 + kKres + (uint32_t)kb * kKBytes + (uint32_t)p * kPiece);
            }
        };
        constexpr int kKPieces = (S * P + NL - 1) / NL;                    // (an upper bound on this loader's share; waits only get stricter)
        auto issue_g = [&](const StripItem &q, int s, int slot) {          // this loader's share of the eight pairs of stage s
#pragma unroll
            for (int ii = 0; ii < P / NL; ++ii) {
                const int i = lid + NL * ii, p = P * s + i;
                const bool real = q.ok && p < npairs;
                glds16(real ? G + (int64_t)slot_of(real ? p : 0) * pair_stride + q.gcol0 : zeros + 256, real ? g_lane : z_lane,
                       lds0 + kRing + (uint32_t)slot * kSlot + (uint32_t)i * kPiece);
            }
        };
        int kb = 0;                                                        // -K buffer of the item in work
        issue_k(cur, kb);
#pragma unroll
        for (int s = 0; s <= D; ++s) issue_g(cur, s, s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");                            // (P) the first item's -K and first stages are in LDS
        int slot = 0;
        if constexpr (kStamp) tprev = stamp_now();
        for (;;) {
            const bool newk = nxt.ok && nxt.krow0 != cur.krow0;
            for (int s = 0; s < S; ++s) {
                // stage s + 1 (of this item, or the next one's first) was requested D iterations ago, at the end of iteration s - D; younger
                // than its pieces: the (D - 1) P / NL pieces of the iterations since, and -- in iterations 1 .. D of an item whose successor
                // starts a new row slab -- this loader's share of that slab's -K, requested at the end of iteration 0.  (From iteration
                // D + 1 on those pieces are OLDER than what is waited for: in-order return has them in LDS before barrier D + 1, seven
                // stages before the first consumer reads them.)
                mark(5);
                wait_vmcnt((D - 1) * (P / NL) + ((newk && s >= 1 && s <= D) ? kKPieces : 0) > 48 ? 48 : (D - 1) * (P / NL) + ((newk && s >= 1 && s <= D) ? kKPieces : 0));
                mark(6);
                asm volatile("s_barrier" ::: "memory");                    // (S) publishes stage s + 1; every consumer has read stage s
                mark(7);
                {
                    const int sn = s + 1 + D;
                    if (sn < S) issue_g(cur, sn, slot); else issue_g(nxt, sn - S, slot);
                }
                if (newk && s == 0) issue_k(nxt, kb ^ 1);                  // the other buffer's last reader finished an item ago at least
                if (s == 0) nn = next_item();
                slot = slot + 1 == R ? 0 : slot + 1;
            }
            if (!nxt.ok) break;
            if (newk) kb ^= 1;
            cur = nxt;
            nxt = nn;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (kStamp) {
            if (lane == 0)
                for (int q = 0; q < 8; ++q) stamps[((int64_t)blockIdx.x * (8 + NL) + wave) * 8 + q] = seg[q];
        }
        return;
    }

    // =========================== the eight consumer wavefronts ===========================
    const uint32_t b_off = kRing + (uint32_t)(lr >> 1) * kPiece + (uint32_t)(lr & 1) * 512 + (uint32_t)lc * 16;       // + slot kSlot + ks 2 KiB + bp 256
    const int arow = 16 * wave + lc;
    const uint32_t a_off = kKres + (uint32_t)(lr >> 1) * kPiece + (uint32_t)(((arow >> 2) * 8) + (lr & 1) * 4 + (arow & 3)) * 4;   // + (8 s + 2 ks) KiB
    const uint32_t t_lane = (uint32_t)((16 * wave + 4 * lr) * T + 4 * lc) * 4;
    // Tile traffic, spread over the item instead of bursting at its ends (8 loads + 8 stores per wavefront, both wavefronts of a SIMD at
    // once, cost 5 500 of an item's 27 000 cycles: STAMP=2).  ONE register buffer of eight 16-byte pieces carries two items at a time: in
    // the item's first half each piece's register first gives up the PREVIOUS item's finished entries (a store), then takes the CURRENT
    // item's tile value (a load, which has at least half an item to arrive); at the item's end the accumulators are added into it.
    // For hipcc to get the waits right the STAGE LOOP IS UNROLLED (kS stages, a template parameter): with a run-time stage index it cannot
    // tell which piece registers have a load pending and puts s_waitcnt vmcnt(0) in front of every store -- which then waits out the load
    // issued two instructions earlier, a full HBM round trip per piece (450 cycles each, measured).  Unrolled, it sees that no store names
    // a register with a pending load and the only wait is the one in front of the final adds.  (Loads hidden in asm are not an option: the
    // register allocator moves the destination registers of asm outputs around before the data has landed.)
    // (Moving the partner wavefronts' pieces to another k-step of the stage -- two positions in one instruction stream -- makes hipcc
    // assume the first position's loads pending at the second and wait; two copies of the unrolled stage loop spill.  Not done.)
    f4_t acc[2][4], tl[8];
    auto zero_acc = [&]() {
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[bp][e] = f4_t{ 0.0f, 0.0f, 0.0f, 0.0f };
    };
    auto piece_off = [&](int p) { return (size_t)((p & 3) * T + 64 * (p >> 2)) * 4 + t_lane; };
    constexpr int half = kS / 2 > 0 ? kS / 2 : 1;
    constexpr int pps = (8 + half - 1) / half;                             // pieces per stage: 2 at 57-64 pairs (stages 0-3 of 8)
    const char *out_base = reinterpret_cast<const char *>(dump + (size_t)blockIdx.x * (kItem * T)), *in_base = nullptr;     // dump: 128 KiB per workgroup
#pragma unroll
    for (int p = 0; p < 8; ++p) tl[p] = f4_t{ 0.0f, 0.0f, 0.0f, 0.0f };
    auto move_pieces = [&](int s) {                      // (s is a compile-time constant after unrolling)
#pragma unroll
        for (int p = 0; p < 8; ++p)
            if (p >= s * pps && p < (s + 1) * pps) {
                // (unconditional on purpose -- the workgroup's FIRST item stores its still meaningless registers to a dump area: with a
                // conditional store hipcc loads into a temporary, waits for it at once and copies)
#if defined(EKF_TILE_PLAIN)
                *reinterpret_cast<f4_t *>(const_cast<char *>(out_base) + piece_off(p)) = tl[p];
                tl[p] = *reinterpret_cast<const f4_t *>(in_base + piece_off(p));
#else
                __builtin_nontemporal_store(tl[p], reinterpret_cast<f4_t *>(const_cast<char *>(out_base) + piece_off(p)));
                tl[p] = __builtin_nontemporal_load(reinterpret_cast<const f4_t *>(in_base + piece_off(p)));
#endif
            }
    };
    // four fragment sets, one per k-step of a stage: every LDS read is issued two k-steps (sixteen MFMAs) ahead of its use, so that the
    // lgkmcnt(0) in front of the stage's barrier finds its reads long retired (read one k-step ahead, the last read of a stage sat
    // directly in front of the barrier and its latency was paid there by every wavefront, every stage)
    float fa[4];
    f4_t fb[4][2];
    auto read_frag = [&](int slot, int kbuf, int s, int ks, float &a, f4_t (&b)[2]) {
        a = *reinterpret_cast<const float *>(smem + a_off + (uint32_t)kbuf * kKBytes + (uint32_t)(P * s + 2 * ks) * kPiece);
        const uint32_t so = (uint32_t)slot * kSlot + (uint32_t)ks * (2 * kPiece);
        b[0] = *reinterpret_cast<const f4_t *>(smem + b_off + so);
        b[1] = *reinterpret_cast<const f4_t *>(smem + b_off + so + 256);
    };
    auto mfma_step = [&](float a, const f4_t (&b)[2]) {
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[bp][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[bp][e], acc[bp][e], 0, 0, 0);
    };
    zero_acc();
    asm volatile("s_barrier" ::: "memory");                                // (P)
    int slot = 0, kb = 0;                                                  // G ring slot, -K buffer of the item in work
    read_frag(slot, kb, 0, 0, fa[0], fb[0]);
    read_frag(slot, kb, 0, 1, fa[1], fb[1]);
    if constexpr (kStamp) tprev = stamp_now();
    for (;;) {
        const bool newk = nxt.ok && nxt.krow0 != cur.krow0;
        const int kbn = newk ? kb ^ 1 : kb;                                // the next item's (a new row slab's -K arrived during this item)
        in_base = reinterpret_cast<const char *>(tiles + cur.toff);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            // on entry: fragments of k-steps 0 and 1 are in (or on their way to) sets 0 and 1
            read_frag(slot, kb, s, 2, fa[2], fb[2]);
            mfma_step(fa[0], fb[0]);                                       // k-step 0
            move_pieces(s);
            if (s == S / 2) nn = next_item();                              // (scalar work, between the MFMAs)
            read_frag(slot, kb, s, 3, fa[3], fb[3]);                       // the stage's last LDS read: after the barrier its slot is refilled
            mfma_step(fa[1], fb[1]);                                       // k-step 1
            mark(0);
            lds_done_barrier();                                            // (S) stage s + 1 is visible
            mark(1);
            const int nslot = slot + 1 == R ? 0 : slot + 1;
            read_frag(nslot, s + 1 < S ? kb : kbn, s + 1 < S ? s + 1 : 0, 0, fa[0], fb[0]);
            mfma_step(fa[2], fb[2]);                                       // k-step 2
            read_frag(nslot, s + 1 < S ? kb : kbn, s + 1 < S ? s + 1 : 0, 1, fa[1], fb[1]);
            mfma_step(fa[3], fb[3]);                                       // k-step 3
            slot = nslot;
            mark(2);
        }
        // the item's result takes its tile's place in the registers; it leaves during the next item's first stages
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mark(5);
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) tl[p][e] += acc[p >> 2][e][p & 3];
        zero_acc();
        out_base = reinterpret_cast<const char *>(dst + cur.toff);
        mark(3);
        if (!nxt.ok) break;
        kb = kbn;
        cur = nxt;
        nxt = nn;
        mark(4);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p)                                            // the last item's result
        __builtin_nontemporal_store(tl[p], reinterpret_cast<f4_t *>(const_cast<char *>(out_base) + piece_off(p)));
    if constexpr (kStamp) {
        if (lane == 0)
            for (int q = 0; q < 8; ++q) stamps[((int64_t)blockIdx.x * (8 + NL) + wave) * 8 + q] = seg[q];
    }
}

}  // namespace ekf_pipe32
