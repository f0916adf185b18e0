// astar.hip -- batched grid A*, one wavefront per query (gfx950).
//
// Takes over planning_space::fast_marching_trees (sea_current.hpp:1339-1407): (start, goal) ->
// optional waypoint list.  Semantics (costs 10/14, octile heuristic, no corner cutting, canonical
// g field and parent rule) are written in oracle/sc_oracle.h; results are bit-exact against it.
//
// Open list = bucket queue.  With integer costs and a consistent heuristic every open node has
// f in [fmin, fmin + 28] (a successor's f exceeds its parent's by at most 2*14), so the open list is
// 32 circular buckets indexed by f & 31 -- no heap, no comparisons.  Every node in the fmin bucket
// already has its optimal g, so the whole bucket is expanded in parallel, 64 nodes per step.
// Successors are relaxed with a returning atomicMin on the query's private g array (exactly one lane
// sees old > new, so each (node, g) is queued exactly once) and appended to their bucket by
// wavefront-ballot compaction: lanes that improved a node with the same f' take consecutive slots
// (popcount of the ballot below the lane).  No inter-wave communication exists: a query's g array,
// buckets and counters are private to its wave.
//
// Latency is what bounds a query (the kernel ends with its slowest query), so the critical path of
// a step is kept to ONE memory round trip:
//   - successors with f' == f (the long equal-f chains along a corridor) go to a ring in LDS, not to
//     HBM.  Entries of that ring can never be stale (any later improvement would have f < fmin), and
//     their g is simply f - h: no load is needed to validate them;
//   - the move mask of every successor is fetched together with the atomics and stored with the
//     entry, so the next step starts straight at its atomics.
// Entries that come back from the HBM buckets (f' > f at insertion time) are validated against g.
//
// g values carry a descending epoch tag in their top bits: a slot left over from an earlier launch
// compares greater than anything written now, so atomicMin treats it as "unset" and the per-batch
// memset of the g arrays (4 MiB per query at 1024^2) disappears.
//
// The search runs until the f = C* bucket is exhausted (not merely until the goal is popped), which
// makes the final g of every expanded node -- and therefore the parent chain extracted from it --
// independent of the expansion order.
//
// A full chip is bound by the rate of L2-missing atomics (profiles/r01_astar_saturation_pmc.json), so
// the remaining design choices are about issuing fewer of them: g arrays are laid out in 4 x 4-cell
// tiles (gix), successors that cannot receive their optimal g through this node are pruned when the
// node is queued (prune_always / entry_prune: the neighbour pruning of jump point search), and the
// relaxations of a step go out back to back under explicit EXEC masks (masked_atomic_min*).
#include "sc_internal.h"
#include <stdlib.h>
#include <type_traits>

#define NBUCKET 32
#define Q_OVERFLOW 100  // internal: bucket ring overflow, retried by the host with a larger ring
#define CQ 2048         // LDS ring entries for the current-f queue (power of two)

struct astar_args {
    const uint8_t* moves;
    const int32_t* d2;
    int W, H;
    int32_t rmin;
    const int32_t* start;
    const int32_t* goal;
    int q0;    // first query of this launch
    int nq;    // queries in this launch (== slots)
    int Lmax;
    int32_t* path;
    int32_t* len;
    int32_t* cost;
    int32_t* status;
    uint32_t* g;        // [slots][cells], epoch-tagged
    uint32_t* buckets;  // [slots][NBUCKET][cap]
    int cap;            // power of two
    int32_t* expanded;  // [Q]
    int32_t* dbg;       // [Q][2] sub-iterations, kilo-cycles (may be null)
    const int32_t* redo;  // optional: only run queries whose status == Q_OVERFLOW
    int tw;               // g layout: 4 x 4-cell tiles (one 64-byte sector each), tw tiles per tile row
    uint32_t epoch_tag;   // epoch << shift
    uint32_t gmask;       // (1 << shift) - 1, or 0xFFFFFFFF when epochs are off
};

__device__ __forceinline__ int octile(int x, int y, int gx, int gy) {
    int dx = abs(x - gx), dy = abs(y - gy);
    return 10 * max(dx, dy) + 4 * min(dx, dy);
}

// Index of cell (x, y) in a query's g array.  The array is laid out in tiles of 4 x 4 cells = one 64-byte sector, so
// the 3 x 3 neighbourhood of a node touches 2.25 sectors on average instead of 3.4 with rows of W cells; the rate of
// L2-missing atomics is what bounds a full chip (tools/microbench/relax_mb.hip: +36 % steps/s with this layout).
// The two halves are additive: gix = g_xpart(x) + g_ypart(y).
__device__ __forceinline__ uint32_t g_xpart(int x) { return ((uint32_t)(x >> 2) << 4) | (uint32_t)(x & 3); }
__device__ __forceinline__ uint32_t g_ypart(int y, int tw) { return (((uint32_t)(y >> 2) * (uint32_t)tw) << 4) | ((uint32_t)(y & 3) << 2); }
__device__ __forceinline__ uint32_t gix(int x, int y, int tw) { return g_xpart(x) + g_ypart(y, tw); }

__device__ __forceinline__ uint32_t g_load(const uint32_t* p) {
    // agent-scope relaxed load: served by L2, where this wave's atomicMin results live
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// N returning atomic-min operations issued back to back, each only on the lanes whose operand is not
// 0xFFFFFFFF (a no-op for min).  hipcc drains vmcnt in front of every divergent `if (...) atomic`,
// which turns N relaxations into N dependent round trips; issuing them unconditionally instead
// doubles the scarce scattered-atomic request rate.  One asm statement with explicit EXEC masks
// gives both: one round trip, no wasted requests.  The statement waits for its own results
// (s_waitcnt vmcnt(0)), as required for asm-issued memory operations.
#define ASTAR_MASKED_ATOMIC(k)                                              \
    "s_mov_b64 exec, %[sv]\n\t"                                             \
    "v_cmp_ne_u32_e32 vcc, -1, %[d" #k "]\n\t"                              \
    "s_and_b64 exec, %[sv], vcc\n\t"                                        \
    "global_atomic_umin %[o" #k "], %[a" #k "], %[d" #k "], off sc0\n\t"
__device__ __forceinline__ void masked_atomic_min8(uint32_t* const (&ad)[8], const uint32_t (&dv)[8], uint32_t (&o)[8]) {
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 ASTAR_MASKED_ATOMIC(0) ASTAR_MASKED_ATOMIC(1) ASTAR_MASKED_ATOMIC(2) ASTAR_MASKED_ATOMIC(3)
                 ASTAR_MASKED_ATOMIC(4) ASTAR_MASKED_ATOMIC(5) ASTAR_MASKED_ATOMIC(6) ASTAR_MASKED_ATOMIC(7)
                 "s_mov_b64 exec, %[sv]\n\t"
                 "s_waitcnt vmcnt(0)"
                 : [sv] "=&s"(sv), [o0] "+v"(o[0]), [o1] "+v"(o[1]), [o2] "+v"(o[2]), [o3] "+v"(o[3]), [o4] "+v"(o[4]),
                   [o5] "+v"(o[5]), [o6] "+v"(o[6]), [o7] "+v"(o[7])
                 : [a0] "v"(ad[0]), [a1] "v"(ad[1]), [a2] "v"(ad[2]), [a3] "v"(ad[3]), [a4] "v"(ad[4]), [a5] "v"(ad[5]),
                   [a6] "v"(ad[6]), [a7] "v"(ad[7]), [d0] "v"(dv[0]), [d1] "v"(dv[1]), [d2] "v"(dv[2]), [d3] "v"(dv[3]),
                   [d4] "v"(dv[4]), [d5] "v"(dv[5]), [d6] "v"(dv[6]), [d7] "v"(dv[7])
                 : "vcc", "memory");
}
__device__ __forceinline__ void masked_atomic_min2(uint32_t* const (&ad)[2], const uint32_t (&dv)[2], uint32_t (&o)[2]) {
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 ASTAR_MASKED_ATOMIC(0) ASTAR_MASKED_ATOMIC(1)
                 "s_mov_b64 exec, %[sv]\n\t"
                 "s_waitcnt vmcnt(0)"
                 : [sv] "=&s"(sv), [o0] "+v"(o[0]), [o1] "+v"(o[1])
                 : [a0] "v"(ad[0]), [a1] "v"(ad[1]), [d0] "v"(dv[0]), [d1] "v"(dv[1])
                 : "vcc", "memory");
}

// Relaxations that can never give a node its optimal g are pruned at insertion time (the neighbour pruning of jump
// point search, without the jumps).  c was reached from its parent p by move d; the relaxation c -> n is dropped when
// a route p -> m -> n exists that is legal whenever c -> n is, costs no more, and -- on a tie -- starts with the
// diagonal move (so the justifications cannot be circular):
//   ALWAYS[d]  n == p, the two cells that touch p (m == n: one move from p), and for diagonal d the two cells two
//              straight steps from p (p -> m -> n costs 20 < 14 + 14; the diagonal being legal means m is free);
//   straight d, per side: if p's diagonal move e on that side is legal, the cell beside c (one move e from p, 14 < 20)
//              and the cell diagonally ahead of c (p -e-> m -d-> n, 14 + 10 on both routes).
// Every node of E still receives g*: the cheaper-or-equal route runs through nodes whose f does not exceed f(n).  The
// values left in cells OUTSIDE E (never expanded) are upper bounds that depend on which relaxations were dropped.
__device__ __forceinline__ uint32_t prune_always(int d) {
    // d: 0 E, 1 W, 2 S(+y), 3 N, 4 SE, 5 SW, 6 NE, 7 NW
    const unsigned long long T = 0x75B6D9EA34C851A2ull;  // bytes: A2 51 C8 34 EA D9 B6 75
    return (uint32_t)(T >> (8 * d)) & 0xFFu;
}

// HBM bucket entry: y << 19 | side flags << 16 | arrival move << 13 | x  (x, y < 8192).  The side flags keep the
// parent's two diagonal-move bits that the conditional pruning of a straight arrival needs (see above).
__device__ __forceinline__ uint32_t entry_pack(int x, int y, int d, uint32_t pmv) {
    const int e0 = (int)((0x6476u >> (4 * (d & 3))) & 7u), e1 = (int)((0x7554u >> (4 * (d & 3))) & 7u);
    const uint32_t side = d < 4 ? (((pmv >> e0) & 1u) | (((pmv >> e1) & 1u) << 1)) : 0u;
    return (uint32_t)y << 19 | side << 16 | (uint32_t)d << 13 | (uint32_t)x;
}
// moves of a popped entry that are pruned: ALWAYS[d] plus, for straight d, the two cells per side flag
__device__ __forceinline__ uint32_t entry_prune(uint32_t e) {
    const int d = (int)((e >> 13) & 7u);
    uint32_t prune = prune_always(d);
    const int k0 = d < 2 ? 3 : 0, k1 = d < 2 ? 2 : 1;
    const int e0 = (int)((0x6476u >> (4 * (d & 3))) & 7u), e1 = (int)((0x7554u >> (4 * (d & 3))) & 7u);
    if ((e >> 16) & 1u) prune |= (1u << k0) | (1u << e0);
    if ((e >> 17) & 1u) prune |= (1u << k1) | (1u << e1);
    return prune;
}

__global__ void __launch_bounds__(64) astar_kernel(astar_args a) {
    // LDS ring of the current f level (indexed directly so that the accesses stay ds_* instructions)
    __shared__ uint32_t qxy[CQ];
    __shared__ uint8_t qmv[CQ];
    // head/tail of the 32 HBM bucket rings: lanes that insert take their slot with one LDS atomic add
    // on the tail (no ballot loop per f class); the pop side reads them with plain ds_reads
    __shared__ int s_head[NBUCKET];
    __shared__ int s_tail[NBUCKET];
#define HEAD(bb) __hip_atomic_load(&s_head[(bb)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)
#define TAIL(bb) __hip_atomic_load(&s_tail[(bb)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)
    const int lane = threadIdx.x;
    const int slot = blockIdx.x;
    const int q = a.q0 + slot;
    if (a.redo && a.redo[q] != Q_OVERFLOW) return;
    const int W = a.W, H = a.H;
    const size_t cells = (size_t)W * H;
    const int s = a.start[q], t = a.goal[q];
    int32_t* path = a.path + (size_t)q * a.Lmax;
    const uint32_t etag = a.epoch_tag, gmask = a.gmask;

    int dbg_iter = 0;
#ifdef ASTAR_COUNT_POPS
    int dbg_pops = 0;
#endif
    unsigned long long dbg_t0 = 0;
    auto finish = [&](int st, int ln, int cs, int ex) {
        if (lane == 0) {
            a.status[q] = st; a.len[q] = ln; a.cost[q] = cs; a.expanded[q] = ex;
#ifdef ASTAR_COUNT_POPS
            if (a.dbg) { a.dbg[2 * q] = dbg_iter; a.dbg[2 * q + 1] = dbg_pops; }
#else
            if (a.dbg) { a.dbg[2 * q] = dbg_iter; a.dbg[2 * q + 1] = (int)((__builtin_amdgcn_s_memtime() - dbg_t0) >> 10); }
#endif
        }
    };
    if (s < 0 || t < 0 || (size_t)s >= cells || (size_t)t >= cells || a.d2[s] < a.rmin || a.d2[t] < a.rmin) {
        finish(SC_Q_BAD_ENDPOINT, 0, -1, 0);
        return;
    }
    if (s == t) {
        if (lane == 0) path[0] = s;
        finish(SC_Q_OK, 1, 0, 0);
        return;
    }
    const int tw = a.tw;
    uint32_t* g = a.g + (size_t)slot * ((size_t)tw * ((H + 3) >> 2) * 16);
    uint32_t* bk = a.buckets + (size_t)slot * NBUCKET * a.cap;
    const int cap = a.cap, capm = a.cap - 1;
    const int sx = s % W, sy = s / W, gx = t % W, gy = t / W;
    const int ddx[8] = {1, -1, 0, 0, 1, -1, 1, -1};
    const int ddy[8] = {0, 0, 1, -1, 1, 1, -1, -1};

    // Lane layout: 8 lanes per node, one per move.  `d` is this lane's move for the whole kernel.
    const int d = lane & 7, sub = lane >> 3;
    const int mdx = (int)((0x2252u >> (2 * d)) & 3u) - 1;   // {1,-1,0,0,1,-1,1,-1}
    const int mdy = (int)((0x0A25u >> (2 * d)) & 3u) - 1;   // {0,0,1,-1,1,1,-1,-1}
    const uint32_t mw = d < 4 ? 10u : 14u;
    const int moff = mdy * W + mdx;
    // pruning of the successor this lane creates (it arrives by move d)
    const uint32_t p_always = prune_always(d);
    // straight d: neighbour k of the successor is also reachable from the parent by diagonal e
    const int ck0 = d < 2 ? 3 : 0, ce0 = (int)((0x6476u >> (4 * (d & 3))) & 7u);   // E:(N,NE) W:(N,NW) S:(E,SE) N:(E,NE)
    const int ck1 = d < 2 ? 2 : 1, ce1 = (int)((0x7554u >> (4 * (d & 3))) & 7u);   // E:(S,SE) W:(S,SW) S:(W,SW) N:(W,NW)
    const bool p_cond = d < 4;

    if (lane < NBUCKET) { s_head[lane] = 0; s_tail[lane] = 0; }
    int fcur = octile(sx, sy, gx, gy);
    if (lane == 0) {
        __hip_atomic_store(&g[gix(sx, sy, tw)], etag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // g(start) = 0
        qxy[0] = (uint32_t)(sy << 16 | sx);
        qmv[0] = a.moves[s];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int lh = 0, lt = 1;  // LDS ring of the current f (wave-uniform)
    // Register copy of the 32 HBM ring heads / tails, bucket i in lane i: the drain loop and the search for the next
    // non-empty level read them with v_readlane / one ballot instead of a chain of dependent LDS reads per level.
    // Heads only move in wave-uniform code; tails move by LDS atomics during a step and are re-read once after it.
    int rh = 0, rt = 0;

    bool found = false, overflow = false, ovf = false;
    int nexp = 0, niter = 0;
#ifdef ASTAR_STAMPS
    unsigned long long st_pop = 0, st_mem = 0, st_rest = 0, st_wide = 0;
    int n_wide = 0, n_hbm = 0;
#define STAMP() __builtin_amdgcn_s_memtime()
#endif
    dbg_t0 = __builtin_amdgcn_s_memtime();

    for (;;) {
        const int b = fcur & 31;
        uint32_t* bq = bk + (size_t)b * cap;
        // drain everything with f == fcur: the LDS ring first, then what earlier levels left in HBM
        for (;;) {
            int n;
            bool from_lds;
            int hd = 0;
            if (lt != lh) { n = min(64, lt - lh); from_lds = true; }
            else {
                hd = __builtin_amdgcn_readlane(rh, b);
                const int tl = __builtin_amdgcn_readlane(rt, b);
                if (hd == tl) break;
                n = min(64, tl - hd);
                from_lds = false;
            }
            const int K = (n + 7) >> 3;  // groups of 8 nodes (x 8 moves = 64 lanes)
            ++niter;
#ifdef ASTAR_COUNT_POPS
            dbg_pops += n;
#endif
            // The step is instantiated for 1, 2, 4 or 8 groups: narrow frontiers (the common case on
            // dense maps) then run ~1/8 of the instructions of a full 64-node step.
            auto step = [&](auto km_tag) {
                constexpr int KM = decltype(km_tag)::value;
#ifdef ASTAR_STAMPS
                const unsigned long long ts0 = STAMP();
#endif
                int cx[KM], cy[KM];
                uint32_t gc[KM], mv[KM];
                bool valid[KM];
                if (from_lds) {
#pragma unroll
                    for (int k = 0; k < KM; ++k) {
                        valid[k] = false; cx[k] = cy[k] = 0; gc[k] = 0; mv[k] = 0;
                        if (8 * k + sub < n) {
                            const uint32_t xy = qxy[(lh + 8 * k + sub) & (CQ - 1)];
                            mv[k] = qmv[(lh + 8 * k + sub) & (CQ - 1)];
                            cx[k] = xy & 0xFFFF; cy[k] = xy >> 16;
                            gc[k] = (uint32_t)(fcur - octile(cx[k], cy[k], gx, gy));  // never stale: g = f - h
                            valid[k] = true;
                        }
                    }
                    lh += n;
                } else {
                    uint32_t gv[KM], mm[KM], de[KM];
#pragma unroll
                    for (int k = 0; k < KM; ++k) {
                        valid[k] = false; cx[k] = cy[k] = 0; gv[k] = 0; mm[k] = 0; de[k] = 0;
                        if (8 * k + sub < n) {
                            const uint32_t e = bq[(hd + 8 * k + sub) & capm];
                            cx[k] = e & 0x1FFF; de[k] = e; cy[k] = e >> 19;
                            const int c = cy[k] * W + cx[k];
                            gv[k] = g_load(&g[gix(cx[k], cy[k], tw)]);
                            mm[k] = a.moves[c];
                            valid[k] = true;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KM; ++k) {
                        gc[k] = gv[k] & gmask;
                        // stale unless it still carries this launch's tag and the g that put it in this bucket
                        valid[k] = valid[k] && (gv[k] & ~gmask) == etag && (int)(gc[k] + octile(cx[k], cy[k], gx, gy)) == fcur;
                        mv[k] = mm[k] & ~entry_prune(de[k]);
                    }
                    if (lane == 0) __hip_atomic_store(&s_head[b], hd + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    if (lane == b) rh = hd + n;
                }
#ifdef ASTAR_STAMPS
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const unsigned long long ts1 = STAMP();
#endif
                // ---- relax: every lane handles move d of its node; all groups' memory operations first ----
                uint32_t old[KM], nmv[KM];
                bool legal[KM];
                if constexpr (KM == 1) {
                    // a single conditional block costs nothing extra
                    old[0] = 0; nmv[0] = 0;
                    legal[0] = valid[0] && ((mv[0] >> d) & 1);
                    if (legal[0]) {
                        const int nidx = cy[0] * W + cx[0] + moff;
                        old[0] = __hip_atomic_fetch_min(&g[gix(cx[0] + mdx, cy[0] + mdy, tw)], etag | (gc[0] + mw), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
                        nmv[0] = a.moves[nidx];
                    }
                } else {
                    uint32_t* ad[KM];
                    uint32_t dv[KM];
#pragma unroll
                    for (int k = 0; k < KM; ++k) {
                        legal[k] = valid[k] && ((mv[k] >> d) & 1);
                        const int nidx = legal[k] ? cy[k] * W + cx[k] + moff : 0;
                        ad[k] = &g[legal[k] ? gix(cx[k] + mdx, cy[k] + mdy, tw) : 0u];
                        dv[k] = legal[k] ? (etag | (gc[k] + mw)) : 0xFFFFFFFFu;
                        old[k] = 0;
                        nmv[k] = a.moves[nidx];   // plain loads: unconditional is harmless
                    }
                    masked_atomic_min2(ad, dv, old);
                }
#ifdef ASTAR_STAMPS
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const unsigned long long ts2 = STAMP();
#endif
#pragma unroll
                for (int k = 0; k < KM; ++k) {
                    if (KM > 1 && k >= K) break;
                    nexp += __popcll(__ballot(valid[k] && d == 0));
                    if (__ballot(valid[k] && cy[k] * W + cx[k] == t)) found = true;
                    const uint32_t ng = gc[k] + mw;
                    const bool imp = legal[k] && old[k] > (etag | ng);
                    const int nx = cx[k] + mdx, ny = cy[k] + mdy;
                    const int df = (int)ng + octile(nx, ny, gx, gy) - fcur;  // in {0,6,8,14,20,28}
                    // moves worth trying from the successor
                    uint32_t prune = p_always;
                    if (p_cond) prune |= (((mv[k] >> ce0) & 1u) << ck0) | (((mv[k] >> ce1) & 1u) << ck1) | (mv[k] & ((1u << ce0) | (1u << ce1)));
                    const uint32_t smv = nmv[k] & ~prune;
                    // same-f successors: consecutive slots of the LDS ring by ballot rank
                    const bool same = imp && df == 0;
                    const unsigned long long m0 = __ballot(same);
                    bool spill = false;
                    if (m0) {
                        const int cnt = __popcll(m0);
                        if (lt - lh + cnt <= CQ) {
                            if (same) {
                                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
                                qxy[(lt + rank) & (CQ - 1)] = (uint32_t)(ny << 16 | nx);
                                qmv[(lt + rank) & (CQ - 1)] = (uint8_t)smv;
                            }
                            lt += cnt;
                        } else spill = true;   // ring full: park them in the HBM bucket of this level
                    }
                    // other levels: one LDS atomic per inserting lane hands out its slot in the HBM ring
                    if (imp && (df != 0 || spill)) {
                        const int bb = (fcur + df) & 31;
                        const int pos = __hip_atomic_fetch_add(&s_tail[bb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        if (pos - HEAD(bb) >= cap) ovf = true;
                        else bk[(size_t)bb * cap + (pos & capm)] = entry_pack(nx, ny, d, mv[k]);
                    }
                }
#ifdef ASTAR_STAMPS
                const unsigned long long ts3 = STAMP();
                st_pop += ts1 - ts0; st_mem += ts2 - ts1; st_rest += ts3 - ts2;
#endif
            };
            // Wide frontiers (more than 16 nodes): one lane per node, the 8 moves unrolled.  The cost of
            // this form does not depend on the node count, so it wins as soon as 3+ groups are needed.
            auto step_wide = [&]() {
                int cx = 0, cy = 0;
                uint32_t gc = 0, mv = 0;
                bool valid = false;
                if (from_lds) {
                    if (lane < n) {
                        const uint32_t xy = qxy[(lh + lane) & (CQ - 1)];
                        mv = qmv[(lh + lane) & (CQ - 1)];
                        cx = xy & 0xFFFF; cy = xy >> 16;
                        gc = (uint32_t)(fcur - octile(cx, cy, gx, gy));
                        valid = true;
                    }
                    lh += n;
                } else {
                    uint32_t gv = 0, mm = 0, de = 0;
                    if (lane < n) {
                        const uint32_t e = bq[(hd + lane) & capm];
                        cx = e & 0x1FFF; de = e; cy = e >> 19;
                        gv = g_load(&g[gix(cx, cy, tw)]);
                        mm = a.moves[cy * W + cx];
                        valid = true;
                    }
                    gc = gv & gmask;
                    valid = valid && (gv & ~gmask) == etag && (int)(gc + octile(cx, cy, gx, gy)) == fcur;
                    mv = mm & ~entry_prune(de);
                    if (lane == 0) __hip_atomic_store(&s_head[b], hd + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    if (lane == b) rh = hd + n;
                }
                if (!valid) mv = 0;
                const int c = cy * W + cx;
                nexp += __popcll(__ballot(valid));
                if (__ballot(valid && c == t)) found = true;
                const int off8[8] = {1, -1, W, -W, W + 1, W - 1, -W + 1, -W - 1};
                uint32_t old[8], nmv[8], dv[8];
                uint32_t* ad[8];
                const uint32_t xp[3] = {g_xpart(cx - 1), g_xpart(cx), g_xpart(cx + 1)};
                const uint32_t yp[3] = {g_ypart(cy - 1, tw), g_ypart(cy, tw), g_ypart(cy + 1, tw)};
#pragma unroll
                for (int dd = 0; dd < 8; ++dd) {
                    const bool lg = (mv >> dd) & 1;
                    const int nidx = lg ? c + off8[dd] : 0;
                    ad[dd] = &g[lg ? xp[ddx[dd] + 1] + yp[ddy[dd] + 1] : 0u];
                    dv[dd] = lg ? (etag | (gc + (dd < 4 ? 10u : 14u))) : 0xFFFFFFFFu;
                    old[dd] = 0;
                    nmv[dd] = a.moves[nidx];   // plain loads: unconditional is harmless
                }
                masked_atomic_min8(ad, dv, old);
#pragma unroll
                for (int dd = 0; dd < 8; ++dd) {
                    const uint32_t ng = gc + (dd < 4 ? 10u : 14u);
                    const bool imp = ((mv >> dd) & 1) && old[dd] > (etag | ng);
                    const int nx = cx + ddx[dd], ny = cy + ddy[dd];
                    const int df = (int)ng + octile(nx, ny, gx, gy) - fcur;
                    uint32_t prune = prune_always(dd);
                    if (dd < 4) {
                        const int k0 = dd < 2 ? 3 : 0, e0 = (0x6476 >> (4 * dd)) & 7, k1 = dd < 2 ? 2 : 1, e1 = (0x7554 >> (4 * dd)) & 7;
                        prune |= (((mv >> e0) & 1u) << k0) | (((mv >> e1) & 1u) << k1) | (mv & ((1u << e0) | (1u << e1)));
                    }
                    const uint32_t smv = nmv[dd] & ~prune;
                    // same-f successors: consecutive slots of the LDS ring by ballot rank
                    const bool same = imp && df == 0;
                    const unsigned long long m0 = __ballot(same);
                    bool spill = false;
                    if (m0) {
                        const int cnt = __popcll(m0);
                        if (lt - lh + cnt <= CQ) {
                            if (same) {
                                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
                                qxy[(lt + rank) & (CQ - 1)] = (uint32_t)(ny << 16 | nx);
                                qmv[(lt + rank) & (CQ - 1)] = (uint8_t)smv;
                            }
                            lt += cnt;
                        } else spill = true;   // ring full: park them in the HBM bucket of this level
                    }
                    // other levels: one LDS atomic per inserting lane hands out its slot in the HBM ring
                    if (imp && (df != 0 || spill)) {
                        const int bb = (fcur + df) & 31;
                        const int pos = __hip_atomic_fetch_add(&s_tail[bb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        if (pos - HEAD(bb) >= cap) ovf = true;
                        else bk[(size_t)bb * cap + (pos & capm)] = entry_pack(nx, ny, dd, mv);
                    }
                }
            };
#ifdef ASTAR_STAMPS
            if (!from_lds) ++n_hbm;
            const unsigned long long tw0 = STAMP();
#endif
            if (K == 1) step(std::integral_constant<int, 1>{});
            else if (K == 2) step(std::integral_constant<int, 2>{});
            else step_wide();
#ifdef ASTAR_STAMPS
            if (K > 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st_wide += STAMP() - tw0; ++n_wide; }
#endif
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            rt = TAIL(lane & 31);
            if (__ballot(ovf)) { overflow = true; break; }
        }
        if (overflow || found) break;
        // level fcur is exhausted: advance to the next non-empty bucket
        const uint32_t nonempty = (uint32_t)__ballot(lane < NBUCKET && rh != rt);   // bit i = bucket i
        if (nonempty == 0) break;  // open list empty: no path
        const int r0 = (fcur + 1) & 31;
        const uint32_t rot = r0 ? (nonempty >> r0) | (nonempty << (32 - r0)) : nonempty;
        fcur += 1 + (__ffs((int)rot) - 1);
    }
    dbg_iter = niter;
#ifdef ASTAR_STAMPS
    if (lane == 0 && a.dbg) {
        a.dbg[2 * q] = (int)(st_pop >> 10); a.dbg[2 * q + 1] = (int)(st_mem >> 10); a.expanded[q] = (int)(st_rest >> 10); a.status[q] = niter;
        a.cost[q] = (int)((__builtin_amdgcn_s_memtime() - dbg_t0) >> 10);   // whole search, same clock
        a.len[q] = n_wide; path[0] = (int)(st_wide >> 10); path[1] = n_hbm;
    }
    return;
#endif

    if (overflow) { finish(Q_OVERFLOW, 0, -1, nexp); return; }
    if (!found) { finish(SC_Q_NO_PATH, 0, -1, nexp); return; }

    // ---- canonical parent chain, goal -> start, written right-aligned then shifted left ----
    const int Lmax = a.Lmax;
    int cx = gx, cy = gy, L = 1;
    uint32_t gc = (uint32_t)fcur;  // g(goal) = C*
    if (lane == 0) path[Lmax - 1] = t;
    bool broken = false;
    while (cx != sx || cy != sy) {
        bool ok = false;
        if (lane < 8) {
            const int d = lane;
            const int nx = cx - ddx[d], ny = cy - ddy[d];
            if (nx >= 0 && ny >= 0 && nx < W && ny < H) {
                const int n = ny * W + nx;
                if ((a.moves[n] >> d) & 1) {
                    const uint32_t gn = g_load(&g[gix(nx, ny, tw)]);
                    ok = gn == (etag | (gc - (d < 4 ? 10u : 14u)));  // tagged and g[n] + w == g[c]
                }
            }
        }
        const unsigned long long m = __ballot(ok);
        if (!m) { broken = true; break; }
        const int d = __ffsll((long long)m) - 1;
        cx -= ddx[d]; cy -= ddy[d];
        gc -= (d < 4 ? 10u : 14u);
        ++L;
        if (lane == 0 && L <= Lmax) path[Lmax - L] = cy * W + cx;
    }
    if (broken) { finish(SC_Q_NO_PATH, 0, -1, nexp); return; }
    if (L > Lmax) { finish(SC_Q_TRUNCATED, L, fcur, nexp); return; }
    const int shift = Lmax - L;
    if (shift > 0) {
        // lane 0's stores must have reached L2 before every lane reads them back (L1-bypassing loads)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        for (int i = 0; i < L; i += 64) {
            int v = 0;
            if (i + lane < L) v = __hip_atomic_load(&path[shift + i + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i + lane < L) path[i + lane] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // chunk j written before chunk j+1 is read
        }
    }
    finish(SC_Q_OK, L, fcur, nexp);
}

// strip the epoch tags of one slot: canonical g field (0xFFFFFFFF = unreached)
__global__ void __launch_bounds__(256) gfield_untag_kernel(const uint32_t* g, int W, int H, int tw, uint32_t etag, uint32_t gmask, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)W * H) {
        const uint32_t v = g[gix((int)(i % W), (int)(i / W), tw)];
        out[i] = (v & ~gmask) == etag && (gmask != 0xFFFFFFFFu || v != 0xFFFFFFFFu) ? (v & gmask) : 0xFFFFFFFFu;
    }
}

static int astar_run(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2, const int32_t* start,
                     const int32_t* goal, int Q, int Lmax, int32_t* path, int32_t* len, int32_t* cost,
                     int32_t* status) {
    const size_t cells = (size_t)W * H;
    const int tw = (W + 3) >> 2;
    const size_t gcells = (size_t)tw * ((H + 3) >> 2) * 16;   // g array: whole 4 x 4 tiles
    const int32_t rmin = r2 > 1 ? r2 : 1;
    int r = sc_scratch_reserve(ctx, &ctx->moves, cells);
    if (r != SC_OK) return r;
    r = sc_launch_moves(ctx, d2, W, H, r2, (uint8_t*)ctx->moves.p);
    if (r != SC_OK) return r;
    r = sc_scratch_reserve(ctx, &ctx->qstats, (size_t)Q * 3 * sizeof(int32_t));
    if (r != SC_OK) return r;
    ctx->last_Q = Q;
    // epoch layout: the g field needs log2(14 * cells) bits
    int shift = cells * 14 < (1u << 24) ? 24 : cells * 14 < (1u << 28) ? 28 : 32;

    // Ring entries per f level: a level holds at most one frontier "ring" of nodes.  Measured worst cases (open and 5 %
    // maps): 16 k at 1024^2, 164 k at 4096^2; an overflow is detected and the batch rerun with 4x the rings, so this only
    // sets the starting point.
    int cap = ctx->astar_cap;
    if (cells > ((size_t)1 << 21)) {
        int want = 1 << 15;
        while (want < 64 * (W > H ? W : H)) want <<= 1;
        if (cap < want) cap = want;
    }
    const int32_t* redo = nullptr;
    for (int attempt = 0; attempt < 6; ++attempt) {
        const size_t per_slot = gcells * 4 + (size_t)NBUCKET * cap * 4;
        size_t slots = ctx->astar_slot_budget / per_slot;
        if (slots < 1) slots = 1;
        if (slots > (size_t)Q) slots = Q;
        const size_t g_bytes = slots * gcells * 4;
        if (g_bytes > ctx->gslots.bytes || shift != ctx->astar_shift) ctx->astar_epoch = 0;  // fresh or re-laid-out memory
        r = sc_scratch_reserve(ctx, &ctx->gslots, g_bytes);
        if (r != SC_OK) return r;
        r = sc_scratch_reserve(ctx, &ctx->buckets, slots * NBUCKET * (size_t)cap * 4);
        if (r != SC_OK) return r;
        ctx->astar_shift = shift;
        for (int q0 = 0; q0 < Q; q0 += (int)slots) {
            const int nq = (int)((size_t)(Q - q0) < slots ? (size_t)(Q - q0) : slots);
            uint32_t etag, gmask;
            if (shift == 32) {
                SC_HIP(ctx, hipMemsetAsync(ctx->gslots.p, 0xFF, ctx->gslots.bytes, ctx->stream));
                etag = 0; gmask = 0xFFFFFFFFu;
            } else {
                // epochs count down from (all ones) - 1; 0 is never used so that tag | g != 0xFFFFFFFF is not needed
                if (ctx->astar_epoch <= 1) {
                    SC_HIP(ctx, hipMemsetAsync(ctx->gslots.p, 0xFF, ctx->gslots.bytes, ctx->stream));
                    ctx->astar_epoch = (1u << (32 - shift)) - 1;
                }
                ctx->astar_epoch -= 1;
                etag = ctx->astar_epoch << shift;
                gmask = (1u << shift) - 1;
            }
            ctx->astar_last_tag = etag; ctx->astar_last_mask = gmask;
            astar_args a{(const uint8_t*)ctx->moves.p, d2, W, H, rmin, start, goal, q0, nq, Lmax, path, len, cost,
                         status, (uint32_t*)ctx->gslots.p, (uint32_t*)ctx->buckets.p, cap, (int32_t*)ctx->qstats.p,
                         getenv("SC_ASTAR_DEBUG") ? (int32_t*)ctx->qstats.p + Q : nullptr, redo,
                         tw, etag, gmask};
            int tk = sc_time_begin(ctx, SC_K_ASTAR);
            hipLaunchKernelGGL(astar_kernel, dim3(nq), dim3(64), 0, ctx->stream, a);
            sc_time_end(ctx, tk);
            SC_HIP(ctx, hipGetLastError());
        }
        // Bucket-ring overflow is rare (cap is generous) but must be seen on the host to retry.
        std::vector<int32_t> st(Q);
        SC_HIP(ctx, hipMemcpyAsync(st.data(), status, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream));
        r = sc_stream_wait(ctx);   // milliseconds: sleep, do not spin
        if (r != SC_OK) return r;
        bool any = false;
        for (int q = 0; q < Q; ++q) any |= st[q] == Q_OVERFLOW;
        if (!any) return SC_OK;
        cap *= 4;
        ctx->astar_cap = cap;
        redo = status;
    }
    snprintf(ctx->err, sizeof(ctx->err), "A* bucket ring overflow persists at cap=%d", cap);
    return SC_ERR_NOMEM;
}

extern "C" int sc_astar_batch(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2_clear,
                              const int32_t* start, const int32_t* goal, int Q, int Lmax,
                              int32_t* path, int32_t* len, int32_t* cost, int32_t* status) {
    if (!ctx || !d2 || !start || !goal || !path || !len || !cost || !status || W <= 0 || H <= 0 || Q < 0 ||
        Lmax <= 0 || W > SC_MAX_DIM || H > SC_MAX_DIM)
        return SC_ERR_INVALID;
    if (Q == 0) return SC_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    return astar_run(ctx, d2, W, H, r2_clear, start, goal, Q, Lmax, path, len, cost, status);
}

extern "C" int sc_astar_last_expansions(sc_ctx* ctx, int64_t* expansions) {
    if (!ctx || !expansions) return SC_ERR_INVALID;
    *expansions = 0;
    if (ctx->last_Q <= 0) return SC_OK;
    std::vector<int32_t> ex(ctx->last_Q);
    SC_HIP(ctx, hipMemcpyAsync(ex.data(), ctx->qstats.p, (size_t)ctx->last_Q * 4, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int64_t tot = 0;
    for (int v : ex) tot += v;
    *expansions = tot;
    return SC_OK;
}

extern "C" int sc_astar_gfield(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2_clear,
                               int32_t start, int32_t goal, uint32_t* gfield, int32_t* cost, int32_t* status) {
    if (!ctx || !d2 || !gfield || !cost || !status || W <= 0 || H <= 0 || W > SC_MAX_DIM || H > SC_MAX_DIM)
        return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)W * H;
    // scratch: start, goal, len, path(1)
    int r = sc_scratch_reserve(ctx, &ctx->staging[7], 64);
    if (r != SC_OK) return r;
    int32_t* sg = (int32_t*)ctx->staging[7].p;
    int32_t h[2] = {start, goal};
    SC_HIP(ctx, hipMemcpyAsync(sg, h, 8, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // Lmax = 1: the path is not wanted; a found path reports SC_Q_TRUNCATED
    r = astar_run(ctx, d2, W, H, r2_clear, sg, sg + 1, 1, 1, sg + 4, sg + 2, cost, status);
    if (r != SC_OK) return r;
    hipLaunchKernelGGL(gfield_untag_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)ctx->gslots.p, W, H, (W + 3) >> 2, ctx->astar_last_tag, ctx->astar_last_mask, gfield);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// debug: per-query {expansions, sub-iterations, kilo-cycles} of the last batch (needs SC_ASTAR_DEBUG=1)
extern "C" int sc_astar_debug_stats(sc_ctx* ctx, int32_t* out3, int Q) {
    if (!ctx || !out3 || Q != ctx->last_Q) return SC_ERR_INVALID;
    SC_HIP(ctx, hipMemcpyAsync(out3, ctx->qstats.p, (size_t)Q * 3 * 4, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SC_OK;
}
