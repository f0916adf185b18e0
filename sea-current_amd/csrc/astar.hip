// astar.hip -- batched grid A*, persistent wavefronts pulling queries from a device-side queue (gfx950).
//
// Takes over planning_space::fast_marching_trees (sea_current.hpp:1339-1407): (start, goal) ->
// optional waypoint list.  Semantics (costs 10/14, octile heuristic, no corner cutting, canonical
// g field and parent rule) are written in oracle/sc_oracle.h; results are bit-exact against it.
//
// Open list = bucket queue.  With integer costs and a consistent heuristic every open node has
// f in [fmin, fmin + 28] (a successor's f exceeds its parent's by at most 2*14), so the open list is
// 32 circular buckets in HBM indexed by f & 31 -- no heap, no comparisons.  Levels are processed in
// strictly increasing f, so the FIRST time a node is popped its g = f - h is optimal.
//
// Closed set = one BIT per cell (tiles of 32 x 16 cells = one 64-byte line), set with a returning
// atomic OR when a node is popped: the lane that sees the bit clear expands the node, every later
// entry of the same node (a duplicate pushed by another parent, or a dearer route in a later
// bucket) is dropped.  Nothing is compared or updated when a successor is PUSHED, so a step of the
// search is one memory round trip (the atomic OR of the popped nodes and the load of their legal-move
// bytes go out together), and that round trip is served by L2: the lines of a query's bitmap around
// its frontier are a few KiB, where a 4-byte g per cell (the previous design: atomic min on
// successors) made 56 % of the atomics miss L2 on a full chip.  The g values are still written
// (plain stores, nobody waits for them) because the canonical parent chain is read from them at the
// end; a g value counts only if the cell's closed bit is set, so the g slots are never cleared.
//
// A step pops up to 64 entries of the current level from a ring in LDS.  Same-f successors (the long
// equal-f chains towards the goal) are appended to that ring by wavefront-ballot compaction
// (consecutive slots by mbcnt rank of the ballot); successors of other levels take their slot in
// the HBM ring of their level with one LDS atomic add.  When the LDS ring runs empty it is refilled
// from the level's HBM ring with coalesced loads, so every pop is an LDS read.
// Lane layout: frontiers of <= 8 nodes use 8 lanes per node (one per move); wider ones one lane per
// node for the pop, then the legal successors of all nodes are compacted into a list and pushed 64 at
// a time (1-3 successors per node survive the pruning below, so that is 2-3 passes instead of 8).
//
// Successors that cannot receive their optimal g through this node are pruned when the node is
// expanded (prune_always / entry_prune: the neighbour pruning of jump point search without the jumps).
//
// The search runs until the f = C* bucket is exhausted (not merely until the goal is popped), which
// makes the set of expanded nodes E = {g* + h <= C*}, their g, and therefore the parent chain
// extracted from it, independent of the expansion order.
//
// Scheduling: the launch is a fixed set of wavefronts (one per slot of scratch: g array, closed
// bitmap, bucket rings); each pulls the next query index from a device counter until the batch is
// exhausted, longest predicted searches first (queries are counting-sorted by octile(start, goal) when
// there are more queries than slots).  A launch therefore has one tail, not one per sub-batch, and
// sc_astar_batch never synchronises the host: a bucket-ring overflow (rare; rings are generous) is
// recorded on the device and the overflowed queries are rerun by a second launch that is always
// enqueued behind the first, with 16x the ring space per query (a handful of wavefronts; it returns
// at once when the list is empty).
//
// Two kernels share astar_query: astar_kernel_dual (two wavefronts per query, see astar_query) runs every batch -- it is
// faster for one call (the longest search is shorter) and at saturation (fewer instructions per expansion);
// astar_kernel (one wavefront per query, the design described above) runs sc_astar_gfield, the overflow retry pass and
// SC_ASTAR_DUAL=0.
#include "sc_internal.h"
#include <stdlib.h>

#define NBUCKET 32
// The LDS ring of the current f level holds CQ entries (a power of two) and is refilled from the level's HBM ring CQ / 2
// entries at a time, RU x 64 loads under way at once.  Two builds of the two-wavefront kernel (astar_run picks):
//   throughput  CQ 1024, RU 4: 13.1 KiB of LDS and <= 80 VGPRs, 12 searches per CU -- batches larger than the chip holds;
//   latency     CQ 2048, RU 8: 17.2 KiB and 81 VGPRs, 9 per CU -- every search of the batch resident at once, where
//               the call lasts as long as its longest search and a refill of a wide level is half as many round trips.
#ifndef ASTAR_CQ_THROUGHPUT
#define ASTAR_CQ_THROUGHPUT 1024
#define ASTAR_RU_THROUGHPUT 4
#endif
#ifndef ASTAR_CQ_LATENCY
#define ASTAR_CQ_LATENCY 2048
#define ASTAR_RU_LATENCY 8
#endif
#define E_RUN (1u << 18)       // entry flag: the node's continuation in its arrival direction is already queued
#define E_START (4u << 13 | 3u << 16)   // the start node: no parent (a diagonal arrival never has side flags)
#define RUNK 8                 // cells of a same-f straight or diagonal run queued at once
#ifndef WIDE_RUNS
#define WIDE_RUNS 0            // runs in wide steps too (measured: fewer steps, but 1.6x the entries popped; slower on all maps but blocks)
#endif

struct astar_args {
    const uint8_t* moves;
    const int32_t* d2;
    int W, H;
    int32_t rmin;
    const int32_t* start;
    const int32_t* goal;
    const int32_t* qgrid;    // optional: grid of every query (several grids in one launch); NULL = all on grid 0
    const int32_t* order;    // optional: query index of the i-th queue position
    const int32_t* nq_dev;   // optional: number of queue positions, read from the device (retry pass)
    int nq;                  // number of queue positions otherwise
    int Lmax;
    int32_t* path;
    int32_t* len;
    int32_t* cost;
    int32_t* status;
    void* g;             // [slots][gcells] of GT (astar_kernel<GT>): valid where the closed bit is set
    uint32_t* closed;    // [slots][bwords], 32 x 16-cell tiles
    uint32_t* buckets;   // [slots][NBUCKET][cap]
    int cap;             // power of two
    int32_t* expanded;   // [Q] nodes expanded; [Q .. 2Q) entries popped; [2Q .. 3Q) kilo-cycles of the search; [3Q .. 4Q) steps
    int32_t* counter;    // queue position handed out next
    int32_t* ovf_list;   // queries whose rings overflowed ...
    int32_t* ovf_count;  // ... and how many
    int32_t* ovf_sticky; // set on any overflow, cleared by the host when it next synchronises
    int nstat;           // Q of the batch (stride of the statistics arrays)
    int lazy;            // two-wavefront kernel: more queries than slots -- wavefront 1 waits for fuller batches
    int tw, bw;          // tiles per row of the g array / of the closed bitmap
    size_t gcells, bwords;
};

__device__ __forceinline__ int octile(int x, int y, int gx, int gy) {
    int dx = abs(x - gx), dy = abs(y - gy);
    return 10 * max(dx, dy) + 4 * min(dx, dy);
}

// Index of cell (x, y) in a query's g array: tiles of 4 x 4 cells = one 64-byte sector, so that the stores of
// a frontier and the 8 loads of a parent-chain step touch few lines.
__device__ __forceinline__ uint32_t gix(int x, int y, int tw) {
    return ((((uint32_t)(y >> 2) * (uint32_t)tw + (uint32_t)(x >> 2)) << 4) | ((uint32_t)(y & 3) << 2)) | (uint32_t)(x & 3);
}
// The batch kernel keeps g modulo 256, one byte per cell in tiles of 8 x 8 cells (one 64-byte line): the only reader is
// the parent-chain extraction, which asks whether g[n] + w == g[c] for a CLOSED neighbour n of c, and closed neighbours
// joined by a legal move differ by at most 14 in g (the move is legal both ways), so equality modulo 256 is equality.
// A quarter of the scratch and of the L2 footprint of full words; sc_astar_gfield runs the kernel with full words.
__device__ __forceinline__ uint32_t gix8(int x, int y, int tw8) {
    return ((((uint32_t)(y >> 3) * (uint32_t)tw8 + (uint32_t)(x >> 3)) << 6) | ((uint32_t)(y & 7) << 3)) | (uint32_t)(x & 7);
}
template <typename GT> __device__ __forceinline__ uint32_t gidx(int x, int y, int tw);
template <> __device__ __forceinline__ uint32_t gidx<uint32_t>(int x, int y, int tw) { return gix(x, y, tw); }
template <> __device__ __forceinline__ uint32_t gidx<uint8_t>(int x, int y, int tw) { return gix8(x, y, tw); }
// word of cell (x, y) in the closed bitmap (bit x & 31): tiles of 32 x 16 cells = one 64-byte line
__device__ __forceinline__ uint32_t cix(int x, int y, int bw) {
    return (((uint32_t)(y >> 4) * (uint32_t)bw + (uint32_t)(x >> 5)) << 4) | (uint32_t)(y & 15);
}

template <typename T>
__device__ __forceinline__ T g_load(const T* p) {
    // agent-scope relaxed load: served by L2, where this wave's stores have landed
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Relaxations that can never give a node its optimal g are pruned when the node is expanded (the neighbour pruning
// of jump point search, without the jumps).  c was reached from its parent p by move d; the relaxation c -> n is
// dropped when a route p -> m -> n exists that is legal whenever c -> n is, costs no more, and -- on a tie -- starts
// with the diagonal move (so the justifications cannot be circular):
//   ALWAYS[d]  n == p, the two cells that touch p (m == n: one move from p), and for diagonal d the two cells two
//              straight steps from p (p -> m -> n costs 20 < 14 + 14; the diagonal being legal means m is free);
//   straight d, per side: if p's diagonal move e on that side is legal, the cell beside c (one move e from p, 14 < 20)
//              and the cell diagonally ahead of c (p -e-> m -d-> n, 14 + 10 on both routes).
// Every node of E still receives g*: the cheaper-or-equal route runs through nodes whose f does not exceed f(n).
// Any parent through which c has its optimal g justifies the pruning, so it does not matter which of c's duplicate
// entries is popped first.
__device__ __forceinline__ uint32_t prune_always(int d) {
    // d: 0 E, 1 W, 2 S(+y), 3 N, 4 SE, 5 SW, 6 NE, 7 NW
    const unsigned long long T = 0x75B6D9EA34C851A2ull;  // bytes: A2 51 C8 34 EA D9 B6 75
    return (uint32_t)(T >> (8 * d)) & 0xFFu;
}

// Queue entry: y << 19 | run flag << 18 | side flags << 16 | arrival move << 13 | x  (x, y < 8192).  The side
// flags keep the parent's two diagonal-move bits that the conditional pruning of a straight arrival needs.
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
    if (e & E_RUN) prune |= 1u << d;                   // its continuation was queued together with it
    return (e & 0x3E000u) == E_START ? 0u : prune;
}

// Returning atomic OR on the lanes whose operand is not 0 only, without a branch: hipcc drains vmcnt in front of a
// divergent `if (...) atomic`, which would put the load issued beside it on a round trip of its own.  The caller
// waits with atomic_wait() (asm-issued memory operations are invisible to the compiler's counters).
__device__ __forceinline__ void masked_atomic_or_issue(uint32_t* addr, uint32_t bits, uint32_t& old) {
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmp_ne_u32_e32 vcc, 0, %[b]\n\t"
                 "s_and_b64 exec, %[sv], vcc\n\t"
                 "global_atomic_or %[o], %[a], %[b], off sc0\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(sv), [o] "+v"(old)
                 : [a] "v"(addr), [b] "v"(bits)
                 : "vcc", "scc", "memory");
}
__device__ __forceinline__ void atomic_wait(uint32_t& old) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(old) : : "memory"); }
// The pair of a step: the legal-move byte of every popped node (a plain load, back after an L2 hit) and the masked
// returning atomic OR on its closed bit (about twice as long under way).  Both are issued here, in that order, so that
// the load can be waited for alone (vmcnt counts in order): everything that does not depend on who won the node --
// pruning, successor coordinates, f levels, queue entries -- is computed while the atomic is still in flight.
__device__ __forceinline__ void pop_pair_issue(const uint8_t* mv_addr, uint32_t* bit_addr, uint32_t bits, uint32_t& pmv, uint32_t& old) {
    unsigned long long sv;
    asm volatile("global_load_ubyte %[m], %[ma], off\n\t"
                 "s_mov_b64 %[sv], exec\n\t"
                 "v_cmp_ne_u32_e32 vcc, 0, %[b]\n\t"
                 "s_and_b64 exec, %[sv], vcc\n\t"
                 "global_atomic_or %[o], %[a], %[b], off sc0\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(sv), [o] "+v"(old), [m] "=&v"(pmv)
                 : [a] "v"(bit_addr), [b] "v"(bits), [ma] "v"(mv_addr)
                 : "vcc", "scc", "memory");
}
// The same with the legal-move bytes of the next RUNK - 1 cells along this lane's move in front of the pair (see "runs"
// in astar_query): nine memory operations, one round trip.
__device__ __forceinline__ void pop_run_issue(const uint8_t* const (&ra)[RUNK - 1], const uint8_t* mv_addr, uint32_t* bit_addr, uint32_t bits,
                                              uint32_t (&rm)[RUNK - 1], uint32_t& pmv, uint32_t& old) {
    static_assert(RUNK == 8, "seven run loads are spelled out");
    unsigned long long sv;
    asm volatile("global_load_ubyte %[r0], %[p0], off\n\t"
                 "global_load_ubyte %[r1], %[p1], off\n\t"
                 "global_load_ubyte %[r2], %[p2], off\n\t"
                 "global_load_ubyte %[r3], %[p3], off\n\t"
                 "global_load_ubyte %[r4], %[p4], off\n\t"
                 "global_load_ubyte %[r5], %[p5], off\n\t"
                 "global_load_ubyte %[r6], %[p6], off\n\t"
                 "global_load_ubyte %[m], %[ma], off\n\t"
                 "s_mov_b64 %[sv], exec\n\t"
                 "v_cmp_ne_u32_e32 vcc, 0, %[b]\n\t"
                 "s_and_b64 exec, %[sv], vcc\n\t"
                 "global_atomic_or %[o], %[a], %[b], off sc0\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(sv), [o] "+v"(old), [m] "=&v"(pmv), [r0] "=&v"(rm[0]), [r1] "=&v"(rm[1]), [r2] "=&v"(rm[2]), [r3] "=&v"(rm[3]),
                   [r4] "=&v"(rm[4]), [r5] "=&v"(rm[5]), [r6] "=&v"(rm[6])
                 : [a] "v"(bit_addr), [b] "v"(bits), [ma] "v"(mv_addr), [p0] "v"(ra[0]), [p1] "v"(ra[1]), [p2] "v"(ra[2]), [p3] "v"(ra[3]),
                   [p4] "v"(ra[4]), [p5] "v"(ra[5]), [p6] "v"(ra[6])
                 : "vcc", "scc", "memory");
}
__device__ __forceinline__ void pop_run_wait_loads(uint32_t (&rm)[RUNK - 1], uint32_t& pmv, uint32_t& old) {
    asm volatile("s_waitcnt vmcnt(1)" : "+v"(pmv), "+v"(old), "+v"(rm[0]), "+v"(rm[1]), "+v"(rm[2]), "+v"(rm[3]), "+v"(rm[4]), "+v"(rm[5]), "+v"(rm[6]) : : "memory");
}
// The legal-move bytes of the next RUNK - 1 cells of a wide step's successor lanes (issued behind the step's pair).
__device__ __forceinline__ void run_loads_issue(const uint8_t* const (&ra)[RUNK - 1], uint32_t (&rm)[RUNK - 1]) {
    asm volatile("global_load_ubyte %[r0], %[p0], off\n\t"
                 "global_load_ubyte %[r1], %[p1], off\n\t"
                 "global_load_ubyte %[r2], %[p2], off\n\t"
                 "global_load_ubyte %[r3], %[p3], off\n\t"
                 "global_load_ubyte %[r4], %[p4], off\n\t"
                 "global_load_ubyte %[r5], %[p5], off\n\t"
                 "global_load_ubyte %[r6], %[p6], off"
                 : [r0] "=&v"(rm[0]), [r1] "=&v"(rm[1]), [r2] "=&v"(rm[2]), [r3] "=&v"(rm[3]), [r4] "=&v"(rm[4]), [r5] "=&v"(rm[5]), [r6] "=&v"(rm[6])
                 : [p0] "v"(ra[0]), [p1] "v"(ra[1]), [p2] "v"(ra[2]), [p3] "v"(ra[3]), [p4] "v"(ra[4]), [p5] "v"(ra[5]), [p6] "v"(ra[6])
                 : "memory");
}
__device__ __forceinline__ void run_loads_wait(uint32_t (&rm)[RUNK - 1]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rm[0]), "+v"(rm[1]), "+v"(rm[2]), "+v"(rm[3]), "+v"(rm[4]), "+v"(rm[5]), "+v"(rm[6]) : : "memory");
}
// wait for the atomic issued in front of RUNK - 1 run loads (memory operations return in order)
__device__ __forceinline__ void atomic_wait_but(uint32_t& old, uint32_t (&rm)[RUNK - 1]) {
    asm volatile("s_waitcnt vmcnt(7)" : "+v"(old), "+v"(rm[0]), "+v"(rm[1]), "+v"(rm[2]), "+v"(rm[3]), "+v"(rm[4]), "+v"(rm[5]), "+v"(rm[6]) : : "memory");
}
__device__ __forceinline__ void pop_pair_wait_load(uint32_t& pmv, uint32_t& old) { asm volatile("s_waitcnt vmcnt(1)" : "+v"(pmv), "+v"(old) : : "memory"); }

// Two-wavefront kernel: a query's closed bitmap has ONE writer (its wavefront 0), so the test-and-set of a popped node need
// not be a returning atomic (about twice a load's latency): the word is LOADED (agent scope: from L2) beside the legal-move
// byte, duplicates inside the step are settled in LDS while the loads are under way (dup_settle), and the bits of the
// nodes that were won are set by an atomic OR nobody waits for: the OR of one step and the load of a later one travel the
// same path to the same L2 channel in issue order, so the load sees the bit without waiting for the OR's acknowledgement
// (waiting for it put the atomic's latency back on the path; a stale word would show as a node expanded twice, which
// the expansion counts in tests/test_gpu_astar.py, equal to the oracle's query by query, would catch).
__device__ __forceinline__ void pop_loads_issue(const uint8_t* mv_addr, const uint32_t* w_addr, uint32_t& byte, uint32_t& word) {
    asm volatile("global_load_ubyte %0, %2, off\n\t"
                 "global_load_dword %1, %3, off sc1"
                 : "=&v"(byte), "=&v"(word) : "v"(mv_addr), "v"(w_addr) : "memory");
}
__device__ __forceinline__ void pop_loads_wait(uint32_t& byte, uint32_t& word) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(byte), "+v"(word) : : "memory"); }
__device__ __forceinline__ void masked_or_noret(uint32_t* addr, uint32_t bits) {
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmp_ne_u32_e32 vcc, 0, %[b]\n\t"
                 "s_and_b64 exec, %[sv], vcc\n\t"
                 "global_atomic_or %[a], %[b], off\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(sv) : [a] "v"(addr), [b] "v"(bits) : "vcc", "scc", "memory");
}

// value of lane 8k (the only non-zero one of its group) in all 8 lanes of the group
__device__ __forceinline__ uint32_t bcast_group8(uint32_t v) {
    const uint32_t q = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00 /*quad_perm:[0,0,0,0]*/, 0xF, 0xF, false);
    const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q, 0x114 /*row_shr:4*/, 0xF, 0xF, false);
    return q | r;
}

#ifdef ASTAR_MARKERS   // bring-up aid: progress words the host can read while a launch is still running
#define MARK(i, v) do { if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(&a.counter[8 + (i)], (int)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
#else
#define MARK(i, v) do { } while (0)
#endif
#ifdef ASTAR_STAMPS    // measurement aid: cycles per phase of the wide steps, left in the first words of the query's path
#define STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp[i] += (long long)(t_ - t_last); t_last = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// One query in scratch slot `slot`: by one wavefront (DUAL = false), or by two (DUAL = true, astar_kernel_dual).
// A search is bound by what ONE wavefront can issue per
// frontier step, so the second wavefront takes the part of a wide step that the next step does not wait for:
//   wavefront 0 pops, closes (the atomic OR), and queues only the SAME-f successors -- at most two moves per node are
//     same-f (the diagonal towards the goal and the straight move along the major axis), known from the geometry without
//     any list -- then hands every node it won to wavefront 1 through a ring in LDS and goes on to the next step;
//   wavefront 1 stores g and pushes the successors of later levels (successor list, HBM rings), one batch of up to 64
//     handed-over nodes at a time, whenever there are any.
// They meet only when a level is exhausted: wavefront 0 waits until wavefront 1 has nothing left (everything for the
// later levels is then in their rings) before it picks the next level.  No barrier per step -- the two earlier
// multi-wavefront forms (four wavefronts per query, a master with workers; round 2, in the history) lost exactly there.
#ifndef HQ
#define HQ 256   // records of the hand-over ring
#endif
#define ASTAR_SPIN_LIMIT (1 << 20)   // polls (about 100 cycles each) wavefront 0 waits for wavefront 1 before it gives the query up
template <typename GT, bool DUAL, int CQ, int REFILL_UNROLL>
__device__ __forceinline__ void astar_query(const astar_args& a, const int q, const int slot) {
    constexpr int REFILL = CQ / 2;         // entries moved from the level's HBM ring into the LDS ring at a time
    __shared__ uint32_t qe[CQ + 64];       // LDS ring of the current f level (+ one spare word per lane: lanes with nothing to append write there)
    __shared__ int s_head[NBUCKET];        // head / tail of the 32 HBM rings: lanes that insert take their slot with one
    __shared__ int s_tail[NBUCKET];        // LDS atomic add on the tail; heads only move in wave-uniform code
    __shared__ uint32_t nd_xy[64];         // wide steps: the popped nodes (y << 16 | x) and their legal-move bytes
    __shared__ uint32_t nd_mv[64];
    __shared__ uint16_t succ[512];         // wide steps: compacted successor list (node << 3 | move)
    __shared__ uint8_t prune_tbl[64];      // entry_prune by the entry's bits 13..18 (arrival move, side flags, run flag)
    __shared__ uint32_t dup_tbl[DUAL ? 1024 + 64 : 1];  // DUAL: the node last popped under hash (x + 32 y) mod 1024 (y << 16 | x, lane in the spare bits); + a spare word per lane
    __shared__ uint2 hq_rec[DUAL ? HQ + 64 : 1];   // hand-over ring: (node y << 16 | x, moves | legal moves << 8); + a spare record per lane
    __shared__ int hq_tail, hq_head, hq_clean, hq_stop, hq_ovf, hq_fcur, hq_found, hq_flush;   // hq_fcur: the level of everything in the ring (it is empty whenever the level changes)
    constexpr int SCOPE = DUAL ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_WAVEFRONT;
    const int lane = threadIdx.x & 63;
    const int wv = DUAL ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int W = a.W, H = a.H;
    const size_t cells = (size_t)W * H;
    const int s = a.start[q], t = a.goal[q];
    int32_t* path = a.path + (size_t)q * a.Lmax;

    // single exit: the outcome is collected here and written once at the end (early returns inside the query loop
    // of the kernel gave the compiler an irreducible region to structurise)
    int out_st = SC_Q_OK, out_len = 0, out_cost = -1, nexp = 0, npop = 0, kcyc = 0, nstep = 0;
    const size_t grid_off = a.qgrid ? (size_t)a.qgrid[q] * cells : 0;   // this query's grid among the launch's
    const uint8_t* const mvs = a.moves + grid_off;
    const bool bad = s < 0 || t < 0 || (size_t)s >= cells || (size_t)t >= cells || a.d2[grid_off + s] < a.rmin || a.d2[grid_off + t] < a.rmin;
    if (bad) out_st = SC_Q_BAD_ENDPOINT;
    else if (s == t) {
        if (lane == 0 && wv == 0) path[0] = s;
        out_len = 1; out_cost = 0; nexp = 1;   // the oracle counts the start node of a trivial query as expanded
    } else {
    MARK(0, 1);
    const int tw = a.tw, bw = a.bw;
    GT* g = static_cast<GT*>(a.g) + (size_t)slot * a.gcells;
    uint32_t* cl = a.closed + (size_t)slot * a.bwords;
    uint32_t* bk = a.buckets + (size_t)slot * NBUCKET * a.cap;
    const int cap = a.cap, capm = a.cap - 1;
    const int sx = s % W, sy = s / W, gx = t % W, gy = t / W;
    const int ddx[8] = {1, -1, 0, 0, 1, -1, 1, -1};
    const int ddy[8] = {0, 0, 1, -1, 1, 1, -1, -1};

    // closed bitmap of this slot: all clear (bwords is a multiple of 16)
    {
        uint4* p = reinterpret_cast<uint4*>(cl);
        const size_t n4 = a.bwords >> 2;
        for (size_t i = DUAL ? threadIdx.x : lane; i < n4; i += DUAL ? 128 : 64) p[i] = make_uint4(0, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();   // the previous query's last LDS reads are done before the rings are reset
    if (wv == 0 && lane < NBUCKET) { s_head[lane] = 0; s_tail[lane] = 0; }
    if (wv == 0) prune_tbl[lane] = (uint8_t)entry_prune((uint32_t)lane << 13);
    if (DUAL)
        for (int i = threadIdx.x; i < 1024 + 64; i += 128) dup_tbl[i] = 0xFFFFFFFFu;   // no node popped yet in this search
    int fcur = octile(sx, sy, gx, gy);
    if (wv == 0 && lane == 0) {
        qe[0] = (uint32_t)sy << 19 | E_START | (uint32_t)sx;
        hq_tail = 0; hq_head = 0; hq_clean = 0; hq_stop = 0; hq_ovf = 0; hq_fcur = fcur; hq_found = 0; hq_flush = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bitmap is clear before the first atomic OR
    wave_lds_sync();
    if (DUAL) __syncthreads();
    int lh = 0, lt = 1;  // LDS ring of the current f (wave-uniform)
    MARK(0, 2);

    bool found = false, ovf = false;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#ifdef ASTAR_STAMPS
    long long stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = t_begin;
    int nwide = 0, nrounds = 0, n_qfull = 0, n_lvl = 0, b_batches = 0, b_records = 0, b_idle = 0, n_refill = 0;
#endif
    // Every step pops at least one entry and a search pushes at most 8 entries per cell: a bound that a correct search
    // cannot reach.  It bounds wavefront 0's pop loop; its two waits on wavefront 1 (a full hand-over ring, the end of a
    // level) give up after ASTAR_SPIN_LIMIT polls -- wavefront 1 needs microseconds for a full ring -- and end the query as
    // SC_Q_RING_OVERFLOW, which the retry pass takes over; wavefront 1's own poll loop ends when wavefront 0 sets hq_stop,
    // which it does on every way out.  So no wavefront can spin forever whatever the state of its scratch memory.
    int steps_left = (int)(8 * cells + 1024 < 0x7FFFFFFF ? 8 * cells + 1024 : 0x7FFFFFFF);

    // this lane's move in the 8-lanes-per-node form
    const int mdx = (int)((0x2252u >> (2 * (lane & 7))) & 3u) - 1;   // {1,-1,0,0,1,-1,1,-1}
    const int mdy = (int)((0x0A25u >> (2 * (lane & 7))) & 3u) - 1;   // {0,0,1,-1,1,1,-1,-1}
    const uint32_t mw = (lane & 7) < 4 ? 10u : 14u;
    const int moff = mdy * W + mdx;

    // Queue entry `ne` of a successor whose f exceeds the current level's by df -- `act` lanes only.
    auto push_entry = [&](const bool act, const uint32_t ne, const int df) {
        // same-f successors: consecutive slots of the LDS ring by ballot rank
        const bool same = act && df == 0;
        const unsigned long long m0 = __ballot(same);
        bool spill = false;
        if (m0) {
            const int cnt = __popcll(m0);
            if (__builtin_expect(lt - lh + cnt <= CQ, 1)) {
                if (same) {
                    const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
                    qe[(lt + rank) & (CQ - 1)] = ne;
                }
                lt += cnt;
            } else spill = true;   // ring full: park them in the HBM ring of this level
        }
        // other levels: one LDS atomic per inserting lane hands out its slot in the HBM ring
        if (act && (df != 0 || spill)) {
            const int bb = (fcur + df) & 31;
            const int pos = __hip_atomic_fetch_add(&s_tail[bb], 1, __ATOMIC_RELAXED, SCOPE);
            if (__builtin_expect(pos - __hip_atomic_load(&s_head[bb], __ATOMIC_RELAXED, SCOPE) >= cap, 0)) ovf = true;
            else bk[(size_t)bb * cap + (pos & capm)] = ne;
        }
        lt = __builtin_amdgcn_readfirstlane(lt);   // wave-uniform by construction; says so to the compiler
    };
    int hq_tl = 0, hq_hd = 0;   // DUAL, wavefront 0: records handed over so far; wavefront 1's progress as last seen
    // DUAL, wavefront 0: hand the nodes of the `rec` lanes (node (hx, hy), moves left to push hm, legal moves hp) to wavefront 1
    auto hand_over = [&](const bool rec, const int hx, const int hy, const uint32_t hm, const uint32_t hp) {
        const unsigned long long wm = __ballot(rec);
        if (wm) {
            const int cnt = __popcll(wm);
            int waited = 0;
            while (__builtin_expect(hq_tl + cnt - hq_hd > HQ, 0)) {     // wavefront 1 is a whole ring behind (as far as we know): look again
                hq_hd = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&hq_head, __ATOMIC_RELAXED, SCOPE));
                if (hq_tl + cnt - hq_hd > HQ) {
                    __builtin_amdgcn_s_sleep(1);
#ifdef ASTAR_STAMPS
                    ++n_qfull;
#endif
                    if (++waited > ASTAR_SPIN_LIMIT) {   // a lost hand-over must not hang the queue: the query ends as a ring overflow
                        ovf = true; steps_left = -1;
                        return wm;
                    }
                }
            }
            {
                const int r = (hq_tl + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(wm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wm, 0u))) & (HQ - 1);
                hq_rec[rec ? r : HQ + lane] = make_uint2((uint32_t)hy << 16 | (uint32_t)hx, hm | hp << 8);   // lanes without a record: their spare one
            }
            hq_tl += cnt;
            wave_lds_sync();
            __hip_atomic_store(&hq_tail, hq_tl, __ATOMIC_RELAXED, SCOPE);   // every lane the same word and value: no EXEC juggling for one lane
        }
        return wm;
    };
    // DUAL, wavefront 0: among the `is` lanes (each pops node (hx, hy), entry he) find one representative per node.
    // 0 = representative, 1 = dropped: the same node is popped by another lane of this step, or it is the node that was
    // popped under this hash most recently (closed then, whatever the bitmap word read in this step says), 2 = lost its
    // table slot to a DIFFERENT node: its entry goes back into the ring for the next step.
    // The table keeps the exact node per slot across steps.  That closes the one window in which the closed test would
    // rest on how the memory system orders a wave's own operations: the OR that sets a node's bit in step t is not waited
    // for, and a duplicate of that node popped in step t + 1 loads the word in the same step-t+1 burst -- it is caught
    // here instead (its slot still holds the node: a step admits one node per slot).  From step t + 2 on the load is issued
    // behind step t + 1's s_waitcnt vmcnt(0), which the OR of step t has completed by then (vmcnt counts atomics).
    constexpr uint32_t DUP_KEY = 0x1FFF1FFFu, DUP_EMPTY = 0xFFFFFFFFu;
    auto dup_settle = [&](const bool is, const int hx, const int hy, const uint32_t he) {
        // Lanes without a node work on a spare word of their own: the three LDS operations then run under the full EXEC mask,
        // back to back (LDS executes a wave's operations in order: the first read sees what earlier steps left, the second
        // what this step's writes left), with ONE wait -- no s_and_saveexec / s_or pairs and no second round trip.
        const int h = is ? ((hx + 32 * hy) & 1023) : 1024 + lane;
        const uint32_t key = (uint32_t)hy << 16 | (uint32_t)hx;
        const uint32_t mine = key | ((uint32_t)(lane & 7) << 13) | ((uint32_t)(lane >> 3) << 29);
        const uint32_t before = dup_tbl[h];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    // compiler only: keep the order read, write, read
        dup_tbl[h] = mine;
        wave_lds_sync();
        const uint32_t now = dup_tbl[h];
        const bool closed_before = is && before != DUP_EMPTY && ((before ^ key) & DUP_KEY) == 0u;
        int res = closed_before ? 1 : 0;
        if (__ballot(is && now != mine)) {
            if (is && !closed_before && now != mine) res = ((now ^ key) & DUP_KEY) == 0u ? 1 : 2;
            const bool back = res == 2;
            const unsigned long long mb = __ballot(back);
            if (mb) {
                const int c = __popcll(mb);
                if (__builtin_expect(lt - lh + c <= CQ, 1)) {
                    qe[back ? ((lt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u))) & (CQ - 1)) : CQ + lane] = he;
                    lt += c;
                } else push_entry(back, he, 0);
                npop -= c;   // they are popped again
            }
        }
        wave_lds_sync();
        return res;
    };
    if (DUAL && wv == 1) {
        // ---- wavefront 1: g and the later-level successors of the nodes wavefront 0 has won ----
        int hd = 0;
        bool dirty = false;
        for (;;) {
            const int tl = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&hq_tail, __ATOMIC_RELAXED, SCOPE));
            const int stop = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&hq_stop, __ATOMIC_RELAXED, SCOPE));
            wave_lds_sync();
            if (tl == hd) {
                if (dirty) {
                    // nothing left: once the stores have landed, say so (wavefront 0 waits for this at the end of a level)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&hq_clean, hd, __ATOMIC_RELAXED, SCOPE);
                    dirty = false;
                }
                if (stop) break;
                __builtin_amdgcn_s_sleep(4);
#ifdef ASTAR_STAMPS
                ++b_idle;
#endif
                continue;
            }
            if (a.lazy && tl - hd < 48 && !stop &&
                __builtin_amdgcn_readfirstlane(__hip_atomic_load(&hq_flush, __ATOMIC_RELAXED, SCOPE)) != tl) {
                // a busy chip: wait for a fuller batch (fuller rounds of 64 successors) unless wavefront 0 is waiting for us
                __builtin_amdgcn_s_sleep(4);
                continue;
            }
#ifdef ASTAR_STAMPS
            ++b_batches; b_records += min(tl - hd, 64);
#endif
            const int n = min(tl - hd, 64);
            const bool valid = lane < n;
            const int r = (hd + lane) & (HQ - 1);
            const uint2 rec = valid ? hq_rec[r] : make_uint2(0u, 0u);
            const uint32_t xy = rec.x, m = rec.y;
            const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane(__hip_atomic_load(&hq_fcur, __ATOMIC_RELAXED, SCOPE));
            hd += n;
            wave_lds_sync();
            if (lane == 0) __hip_atomic_store(&hq_head, hd, __ATOMIC_RELAXED, SCOPE);   // the records are read: their places are free
            dirty = true;
            const int x = xy & 0xFFFF, y = xy >> 16;
            const int hc = octile(x, y, gx, gy);
            if (valid) g[gidx<GT>(x, y, tw)] = (GT)(f - (uint32_t)hc);
            if (__ballot(valid && x == gx && y == gy) && lane == 0) __hip_atomic_store(&hq_found, 1, __ATOMIC_RELAXED, SCOPE);   // the goal has been expanded
            uint32_t mv = m & 0xFFu;
            nd_xy[lane] = xy;
            nd_mv[lane] = ((m >> 8) & 0xFFu) | ((f - (uint32_t)hc) << 8);     // legal moves | g << 8
            const uint32_t cnt = (uint32_t)__popc(mv);
            int base = 0, total = 0;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const unsigned long long mm = __ballot((cnt >> bb) & 1u);
                base += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u)) << bb;
                total += __popcll(mm) << bb;
            }
            while (mv) {
                const int dd = __ffs((int)mv) - 1;
                mv &= mv - 1;
                succ[base++] = (uint16_t)(lane << 3 | dd);
            }
            wave_lds_sync();
            for (int j0 = 0; j0 < total; j0 += 64) {
                const bool act = j0 + lane < total;
                const uint32_t sd = act ? succ[j0 + lane] : 0u;
                const int par = (int)(sd >> 3), d = (int)(sd & 7u);
                const uint32_t pxy = nd_xy[par], pm = nd_mv[par];
                const int px = pxy & 0xFFFF, py = pxy >> 16;
                const int nx = px + (int)((0x2252u >> (2 * d)) & 3u) - 1, ny = py + (int)((0x0A25u >> (2 * d)) & 3u) - 1;
                const uint32_t fs = (pm >> 8) + (d < 4 ? 10u : 14u) + (uint32_t)octile(nx, ny, gx, gy);
                if (act) {
                    const int bb = (int)(fs & 31u);
                    const int pos = __hip_atomic_fetch_add(&s_tail[bb], 1, __ATOMIC_RELAXED, SCOPE);
                    if (pos - __hip_atomic_load(&s_head[bb], __ATOMIC_RELAXED, SCOPE) >= cap) __hip_atomic_store(&hq_ovf, 1, __ATOMIC_RELAXED, SCOPE);
                    else bk[(size_t)bb * cap + (pos & capm)] = entry_pack(nx, ny, d, pm & 0xFFu);
                }
            }
            wave_lds_sync();
        }
#ifdef ASTAR_STAMPS
        if (lane == 0 && a.Lmax >= 32) { path[16] = b_batches; path[17] = b_records; path[18] = b_idle; }
#endif
    } else {
    for (;;) {
        const int b = fcur & 31;
        const uint32_t* bq = bk + (size_t)b * cap;
        // drain everything with f == fcur: the LDS ring, refilled from what earlier levels left in HBM
        for (;;) {
            if (__builtin_expect(lt == lh, 0)) {
                STAMP(0);   // loop overhead
                wave_lds_sync();
                const int hd = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_head[b], __ATOMIC_RELAXED, SCOPE));
                const int tl = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_tail[b], __ATOMIC_RELAXED, SCOPE));
                if (hd == tl) break;
                const int n = min(tl - hd, REFILL);
                // REFILL_UNROLL loads under way before the first is waited for (one after the other, a chunk of 512 entries
                // was eight dependent round trips)
                for (int base = lane; base < n; base += 64 * REFILL_UNROLL) {
                    uint32_t chunk[REFILL_UNROLL];
#pragma unroll
                    for (int k = 0; k < REFILL_UNROLL; ++k)
                        if (base + 64 * k < n) chunk[k] = bq[(hd + base + 64 * k) & capm];
#pragma unroll
                    for (int k = 0; k < REFILL_UNROLL; ++k)
                        if (base + 64 * k < n) qe[(lt + base + 64 * k) & (CQ - 1)] = chunk[k];
                }
                lt += n;
                if (lane == 0) __hip_atomic_store(&s_head[b], hd + n, __ATOMIC_RELAXED, SCOPE);
                wave_lds_sync();
                STAMP(10);  // refills
#ifdef ASTAR_STAMPS
                ++n_refill;
#endif
            }
            const int n = min(64, lt - lh);
            npop += n; ++nstep;
            STAMP(0);   // loop overhead, refills
            if (DUAL && n <= 16) {
                // ---- narrow step, two wavefronts: lane (node, direction, k) for the two same-f directions of a node and
                // k = 0 .. KL-1 cells along them (KL = 8 for up to 4 nodes, 4 / 2 / 1 for up to 8 / 16 / 32): ONE load per lane (the
                // legal-move byte of cell k; cell 0 is the node), the run of a direction is the number of leading lanes of its
                // group whose cell continues legally, and every queued cell of every run goes into the ring in one append.
                // The moves that leave the level go to wavefront 1 with the node, as in the wide step.
                const int kl = n <= 4 ? 3 : n <= 8 ? 2 : n <= 16 ? 1 : 0, KL = 1 << kl, GL = 2 * KL;   // log2 lanes per direction; lanes per direction, per node
                const int sub = lane >> (kl + 1), j = (lane >> kl) & 1, k = lane & (KL - 1);
                const bool valid = sub < n;
                const uint32_t e_raw = qe[(lh + sub) & (CQ - 1)];   // read under the full EXEC mask (the index is always inside the ring)
                const uint32_t e = valid ? e_raw : 0u;
                lh += n;
                const int x = e & 0x1FFF, y = e >> 19;
                const uint32_t bit = 1u << (x & 31);
                const int dxg = gx - x, dyg = gy - y, adx = abs(dxg), ady = abs(dyg);
                const int dD = 4 + ((dxg < 0 ? 1 : 0) | (dyg < 0 ? 2 : 0));
                const int dS = adx > ady ? (dxg < 0 ? 1 : 0) : (dyg < 0 ? 3 : 2);
                const uint32_t bD = adx >= 1 && ady >= 1 ? 1u << dD : 0u, bS = adx != ady ? 1u << dS : 0u;
                const int d = j ? dS : dD;
                const bool has = valid && (j ? bS : bD) != 0u;
                const int lgeom = j ? abs(adx - ady) : min(adx, ady);            // cells that keep f along d (>= 1 when `has`)
                const int ddx_ = (int)((0x2252u >> (2 * d)) & 3u) - 1, ddy_ = (int)((0x0A25u >> (2 * d)) & 3u) - 1;
                const uint8_t* const addr = mvs + (y * W + x) + (has ? min(k, lgeom - 1) : 0) * (ddy_ * W + ddx_);   // invalid lanes read cell 0: harmless
                const bool head = valid && j == 0 && k == 0;                      // the node's own lane: atomic, hand-over
                uint32_t old, byte;
                uint32_t* const wa = &cl[cix(x, y, bw)];
                pop_loads_issue(addr, wa, byte, old);
                const uint32_t prune = prune_tbl[(e >> 13) & 63u];
                const int dup = dup_settle(head, x, y, e);
                pop_loads_wait(byte, old);
                // the link cell k -> cell k + 1 along d (from the node itself: unless pruned)
                const bool link = has && k < lgeom && ((byte >> d) & 1u) && (k > 0 || !((prune >> d) & 1u));
                const unsigned long long lm = __ballot(link);
                const uint32_t grp = (uint32_t)(lm >> (lane & ~(KL - 1))) & ((1u << KL) - 1u);
                const int run = __ffs((int)~grp) - 1;                             // leading lanes of the group that link
                const uint32_t ne = entry_pack(x + (k + 1) * ddx_, y + (k + 1) * ddy_, d, byte) | (k + 1 < run ? E_RUN : 0u);
                const bool won1 = head && dup == 0 && !(old & bit);
                masked_or_noret(wa, won1 ? bit : 0u);
                const unsigned long long wm1 = __ballot(won1);
                const bool act = ((wm1 >> (lane & ~(GL - 1))) & 1ull) && k < run;
                {
                    const unsigned long long ma = __ballot(act);
                    if (ma) {
                        const int c = __popcll(ma);
                        if (__builtin_expect(lt - lh + c <= CQ, 1)) {
                            qe[act ? ((lt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0u))) & (CQ - 1)) : CQ + lane] = ne;
                            lt += c;
                        } else push_entry(act, ne, 0);                            // ring full: the general path parks them in the level's HBM ring
                    }
                }
                hand_over(won1, x, y, (byte & ~prune) & ~(bD | bS), byte);
                nexp += __popcll(wm1);
                STAMP(8);   // narrow steps
            } else if (!DUAL && n <= 8) {
                // ---- narrow step: 8 lanes per node, lane (sub, d) handles move d of node sub ----
                const int sub = lane >> 3, d = lane & 7;
                const bool valid = sub < n;
                const uint32_t e = valid ? qe[(lh + sub) & (CQ - 1)] : 0u;
                lh += n;
                const int x = e & 0x1FFF, y = e >> 19;
                const uint32_t bit = 1u << (x & 31);
                uint32_t old = 0, pmv;
                const int hc = octile(x, y, gx, gy);
                const int nx = x + mdx, ny = y + mdy;
                const int df = (int)mw + octile(nx, ny, gx, gy) - hc;   // in {0,6,8,14,20,28}
                // Runs.  A same-f move keeps its f for as long as it keeps lowering h by its cost: a straight move for
                // |major| - |minor| cells, a diagonal one for min(|dx|, |dy|) cells (goal offsets), every cell of the run
                // reached with g = f - h, which is optimal at the current level whoever expands its predecessor.  So the
                // lane of a same-f move looks up to RUNK cells ahead (their legal-move bytes travel with this step's
                // pair) and queues the whole legal stretch at once; each queued cell but the last carries E_RUN and
                // leaves its continuation out when it is expanded.  A chain of single nodes -- one round trip per cell
                // otherwise -- then advances RUNK cells per step.
                const int adx = abs(gx - x), ady = abs(gy - y);
                const int lgeom = d >= 4 ? min(adx, ady) : abs(adx - ady);
                const int lmax = valid && df == 0 ? min(RUNK, lgeom) : 1;     // df == 0 implies lgeom >= 1
                const uint8_t* const cell = mvs + (y * W + x);             // invalid lanes read cell 0: harmless
                const uint8_t* ra[RUNK - 1];
                uint32_t rm[RUNK - 1];
#pragma unroll
                for (int k = 1; k < RUNK; ++k) ra[k - 1] = cell + min(k, lmax - 1) * moff;
                pop_run_issue(ra, cell, &cl[cix(x, y, bw)], valid && d == 0 ? bit : 0u, rm, pmv, old);
                const uint32_t prune = prune_tbl[(e >> 13) & 63u];
                pop_run_wait_loads(rm, pmv, old);
                // while the atomic is under way: this lane's successor(s), should its node be won
                const bool cand = valid && (((pmv & ~prune) >> d) & 1u);
                int run = cand ? 1 : 0;                                        // cells to queue along this move
#pragma unroll
                for (int k = 1; k < RUNK; ++k)
                    if (run == k && k < lmax && ((rm[k - 1] >> d) & 1u)) run = k + 1;
                const uint32_t ne = entry_pack(nx, ny, d, pmv) | (run > 1 ? E_RUN : 0u);
                atomic_wait(old);
                old = bcast_group8(old);
                const bool won = valid && !(old & bit);
                push_entry(won && cand, ne, df);
#pragma unroll
                for (int k = 2; k <= RUNK; ++k) {
                    if (__ballot(won && run >= k) == 0) break;
                    push_entry(won && run >= k, entry_pack(x + k * mdx, y + k * mdy, d, rm[k - 2]) | (run > k ? E_RUN : 0u), 0);
                }
                nexp += __popcll(__ballot(won && d == 0));
                if (__ballot(won && x == gx && y == gy)) found = true;
                if (won && d == 0) g[gidx<GT>(x, y, tw)] = (GT)(fcur - hc);
                STAMP(8);   // narrow steps
            } else if (DUAL) {
                // ---- wide step, two wavefronts: close the nodes, queue their same-f successors, hand the rest over ----
#ifdef ASTAR_STAMPS
                ++nwide;
#endif
                const bool valid = lane < n;
                const uint32_t e_raw = qe[(lh + lane) & (CQ - 1)];   // read under the full EXEC mask
                const uint32_t e = valid ? e_raw : 0u;
                lh += n;
                const int x = e & 0x1FFF, y = e >> 19;
                const uint32_t bit = 1u << (x & 31);
                uint32_t old, pmv;
                uint32_t* const wa = &cl[cix(x, y, bw)];
                pop_loads_issue(mvs + (y * W + x), wa, pmv, old);   // invalid lanes read cell 0: harmless
                const int dup = dup_settle(valid, x, y, e);
                const uint32_t prune = prune_tbl[(e >> 13) & 63u];
                // The moves that keep f: the diagonal towards the goal while both offsets are non-zero (h falls by 14), and
                // the straight move along the larger offset while the offsets differ (h falls by 10).  Every other move
                // raises f by 6, 8, 14, 20 or 28.
                const int dxg = gx - x, dyg = gy - y, adx = abs(dxg), ady = abs(dyg);
                const int dD = 4 + ((dxg < 0 ? 1 : 0) | (dyg < 0 ? 2 : 0));
                const int dS = adx > ady ? (dxg < 0 ? 1 : 0) : (dyg < 0 ? 3 : 2);
                const uint32_t bD = adx >= 1 && ady >= 1 ? 1u << dD : 0u, bS = adx != ady ? 1u << dS : 0u;
                const int nxD = x + (dxg < 0 ? -1 : 1), nyD = y + (dyg < 0 ? -1 : 1);
                const int nxS = x + (adx > ady ? (dxg < 0 ? -1 : 1) : 0), nyS = y + (adx > ady ? 0 : (dyg < 0 ? -1 : 1));
                pop_loads_wait(pmv, old);
                const uint32_t cand = valid ? (pmv & ~prune) : 0u;
                const uint32_t neD = (uint32_t)nyD << 19 | (uint32_t)dD << 13 | (uint32_t)nxD;   // a diagonal arrival has no side flags
                const uint32_t neS = entry_pack(nxS, nyS, dS, pmv);
                const bool won = valid && dup == 0 && !(old & bit);
                masked_or_noret(wa, won ? bit : 0u);
                {
                    // both same-f entries of a node in one go: the diagonal ones first, then the straight ones
                    const bool pD = won && (cand & bD), pS = won && (cand & bS);
                    const unsigned long long mD = __ballot(pD), mS = __ballot(pS);
                    const int cD = __popcll(mD), c2 = cD + __popcll(mS);
                    if (c2) {
                        if (__builtin_expect(lt - lh + c2 <= CQ, 1)) {
                            qe[pD ? ((lt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mD >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mD, 0u))) & (CQ - 1)) : CQ + lane] = neD;
                            qe[pS ? ((lt + cD + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mS >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mS, 0u))) & (CQ - 1)) : CQ + lane] = neS;
                            lt += c2;
                        } else {
                            push_entry(pD, neD, 0);      // ring full: the general path parks them in the level's HBM ring
                            push_entry(pS, neS, 0);
                        }
                    }
                }
                nexp += __popcll(hand_over(won, x, y, cand & ~(bD | bS), pmv));
                STAMP(6);
            } else {
                // ---- wide step: one lane per node, then the legal successors of all nodes 64 at a time ----
                // The successor list is built while the atomic is still under way, as if every node were won (most are);
                // the successors of a node that turns out to be somebody else's are dropped when they would be pushed.
                // Same-f successors look RUNK cells ahead exactly as in the narrow step (their loads go out before the
                // atomic is waited for), so the equal-f chains of a wide frontier advance RUNK cells per step too.
                STAMP(0);   // everything outside the wide steps
                const bool valid = lane < n;
                const uint32_t e = valid ? qe[(lh + lane) & (CQ - 1)] : 0u;
                lh += n;
                const int x = e & 0x1FFF, y = e >> 19;
                const uint32_t bit = 1u << (x & 31);
                uint32_t old = 0, pmv;
                pop_pair_issue(mvs + (y * W + x), &cl[cix(x, y, bw)], valid ? bit : 0u, pmv, old);   // invalid lanes read cell 0: harmless
                const uint32_t prune = prune_tbl[(e >> 13) & 63u];
                const int hc = octile(x, y, gx, gy);
                STAMP(1);   // pop and issue
                pop_pair_wait_load(pmv, old);
                STAMP(2);   // the load
                uint32_t mv = valid ? (pmv & ~prune) : 0u;
                nd_xy[lane] = (uint32_t)y << 16 | (uint32_t)x;
                nd_mv[lane] = pmv;
                // exclusive prefix of the successor counts (<= 8 each) from four ballots
                const uint32_t cnt = (uint32_t)__popc(mv);
                int base = 0, total = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const unsigned long long m = __ballot((cnt >> bb) & 1u);
                    base += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << bb;
                    total += __popcll(m) << bb;
                }
                while (mv) {
                    const int dd = __ffs((int)mv) - 1;
                    mv &= mv - 1;
                    succ[base++] = (uint16_t)(lane << 3 | dd);
                }
                wave_lds_sync();
                STAMP(3);   // successor list
                bool won = false;
#ifdef ASTAR_STAMPS
                ++nwide;
#endif
                for (int j0 = 0; j0 < total; j0 += 64) {
#ifdef ASTAR_STAMPS
                    ++nrounds;
#endif
                    const bool act = j0 + lane < total;
                    const uint32_t sd = act ? succ[j0 + lane] : 0u;
                    const int par = (int)(sd >> 3), d = (int)(sd & 7u);
                    const uint32_t xy = nd_xy[par], pmvp = nd_mv[par];
                    const int px = xy & 0xFFFF, py = xy >> 16;
                    const int ddx_ = (int)((0x2252u >> (2 * d)) & 3u) - 1, ddy_ = (int)((0x0A25u >> (2 * d)) & 3u) - 1;
                    const int nx = px + ddx_, ny = py + ddy_;
                    const int df = (d < 4 ? 10 : 14) + octile(nx, ny, gx, gy) - octile(px, py, gx, gy);
                    const int adx = abs(gx - px), ady = abs(gy - py);
                    const int lgeom = d >= 4 ? min(adx, ady) : abs(adx - ady);
                    const int lmax = WIDE_RUNS && act && df == 0 ? min(RUNK, lgeom) : 1;
                    const int so = ddy_ * W + ddx_;
                    const uint8_t* const cell = mvs + (py * W + px);
                    const uint8_t* ra[RUNK - 1];
                    uint32_t rm[RUNK - 1];
#pragma unroll
                    for (int k = 1; k < RUNK; ++k) ra[k - 1] = cell + min(k, lmax - 1) * so;
                    if (WIDE_RUNS) run_loads_issue(ra, rm);
                    if (j0 == 0) {
                        // the pop's atomic is older than the run loads: in-order return, RUNK - 1 operations may remain
                        STAMP(4);   // first round's preparation
                        if (WIDE_RUNS) atomic_wait_but(old, rm); else atomic_wait(old);
                        STAMP(5);   // the atomic
                        won = valid && !(old & bit);   // duplicates inside one pop: the atomics serialise, one lane wins
                        nexp += __popcll(__ballot(won));
                        if (__ballot(won && x == gx && y == gy)) found = true;
                        nd_mv[lane] = pmv | (won ? 0x100u : 0u);
                        wave_lds_sync();
                    }
                    const bool go = act && ((nd_mv[par] >> 8) & 1u);
                    int run = 1;
                    if (WIDE_RUNS) {
                        run_loads_wait(rm);
#pragma unroll
                        for (int k = 1; k < RUNK; ++k)
                            if (run == k && k < lmax && ((rm[k - 1] >> d) & 1u)) run = k + 1;
                    }
                    push_entry(go, entry_pack(nx, ny, d, pmvp) | (run > 1 ? E_RUN : 0u), df);
                    if (WIDE_RUNS) {
#pragma unroll
                        for (int k = 2; k <= RUNK; ++k) {
                            if (__ballot(go && run >= k) == 0) break;
                            push_entry(go && run >= k, entry_pack(px + k * ddx_, py + k * ddy_, d, rm[k - 2]) | (run > k ? E_RUN : 0u), 0);
                        }
                    }
                }
                if (total == 0) {
                    atomic_wait(old);
                    won = valid && !(old & bit);
                    nexp += __popcll(__ballot(won));
                    if (__ballot(won && x == gx && y == gy)) found = true;
                }
                if (won) g[gidx<GT>(x, y, tw)] = (GT)(fcur - hc);
                STAMP(6);   // pushes
            }
            // wave-uniform by construction (ballots, popcounts); the joins above hide that from the compiler, which would
            // otherwise run these loops under EXEC masks with the counters in VGPRs
            lt = __builtin_amdgcn_readfirstlane(lt);
            lh = __builtin_amdgcn_readfirstlane(lh);
            nexp = __builtin_amdgcn_readfirstlane(nexp);
            found = __builtin_amdgcn_readfirstlane((int)found) != 0;
            wave_lds_sync();
            MARK(1, steps_left); MARK(2, fcur); MARK(3, nexp); MARK(4, lt - lh);
            if (--steps_left < 0) ovf = true;
            if (__builtin_expect(DUAL ? steps_left < 0 : __ballot(ovf) != 0, 0)) break;
        }
        if (DUAL) {
            // every node of this level has been handed over: wait until wavefront 1 has pushed what follows from them
            STAMP(0);
            if (a.lazy && lane == 0) __hip_atomic_store(&hq_flush, hq_tl, __ATOMIC_RELAXED, SCOPE);
            for (int waited = 0; __builtin_amdgcn_readfirstlane(__hip_atomic_load(&hq_clean, __ATOMIC_RELAXED, SCOPE)) != hq_tl;) {
                __builtin_amdgcn_s_sleep(1);
#ifdef ASTAR_STAMPS
                ++n_lvl;
#endif
                if (++waited > ASTAR_SPIN_LIMIT) { ovf = true; break; }   // as above: never expected, never a hang
            }
            wave_lds_sync();
            if (__hip_atomic_load(&hq_ovf, __ATOMIC_RELAXED, SCOPE)) ovf = true;
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&hq_found, __ATOMIC_RELAXED, SCOPE))) found = true;
            STAMP(7);
        }
        if (__ballot(ovf) || found) break;
        // level fcur is exhausted: advance to the next non-empty bucket
        const int rh = __hip_atomic_load(&s_head[lane & 31], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const int rt = __hip_atomic_load(&s_tail[lane & 31], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const uint32_t nonempty = (uint32_t)__ballot(lane < NBUCKET && rh != rt);   // bit i = bucket i
        if (nonempty == 0) break;  // open list empty: no path
        const int r0 = (fcur + 1) & 31;
        const uint32_t rot = r0 ? (nonempty >> r0) | (nonempty << (32 - r0)) : nonempty;
        fcur += 1 + (__ffs((int)rot) - 1);
        if (DUAL && lane == 0) __hip_atomic_store(&hq_fcur, fcur, __ATOMIC_RELAXED, SCOPE);   // the ring is empty: wavefront 1 reads it with the next records
        STAMP(9);   // next level
    }
    if (DUAL && lane == 0) __hip_atomic_store(&hq_stop, 1, __ATOMIC_RELAXED, SCOPE);   // wavefront 1 has nothing left (level-end wait): it leaves
    }   // wavefront 0
    if (DUAL) __syncthreads();   // wavefront 1's g stores have landed (it waits for them before it reports an empty ring)

    const bool overflow = __ballot(ovf) != 0;
    if (overflow) {
        if (lane == 0 && wv == 0) {
            const int i = atomicAdd(a.ovf_count, 1);
            a.ovf_list[i] = q;
            *a.ovf_sticky = 1;
        }
        out_st = SC_Q_RING_OVERFLOW;
    } else if (!found) {
        out_st = SC_Q_NO_PATH;
    } else {
        // ---- canonical parent chain, goal -> start, written right-aligned then shifted left ----
        STAMP(9);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every g store of this wave has reached L2
        const int Lmax = a.Lmax;
        int cx = gx, cy = gy, L = 1;
        uint32_t gc = (uint32_t)fcur;  // g(goal) = C*
        if (lane == 0) path[Lmax - 1] = t;
        bool broken = false;
        // One round trip fetches the legal moves, the closed bit and g of the 7 x 7 cells around the current cell (lane
        // l < 49: cell (l % 7 - 3, l / 7 - 3)); the next three parents are then found from registers (the candidates of
        // move d sit in lane d, which pulls its cell's data with ds_bpermute) -- not two dependent round trips per cell.
        const int wox = lane % 7 - 3, woy = lane / 7 - 3;
        const int md = lane & 7;
        const int mdx_ = (int)((0x2252u >> (2 * md)) & 3u) - 1, mdy_ = (int)((0x0A25u >> (2 * md)) & 3u) - 1;
        const uint32_t mw_ = md < 4 ? 10u : 14u;
        while (!broken && (cx != sx || cy != sy)) {
            uint32_t info = 0u, gw = 0u;   // legal moves | closed << 8 (0 outside the grid), g
            {
                const int nx = cx + wox, ny = cy + woy;
                if (lane < 49 && nx >= 0 && ny >= 0 && nx < W && ny < H) {
                    const uint32_t mvb = mvs[ny * W + nx];
                    const uint32_t cw = g_load(&cl[cix(nx, ny, bw)]);
                    gw = (uint32_t)g_load(&g[gidx<GT>(nx, ny, tw)]);
                    info = mvb | ((cw >> (nx & 31)) & 1u) << 8;
                }
            }
            int ox = 0, oy = 0;     // the current cell inside the window
            for (int k = 0; k < 3 && !broken && (cx != sx || cy != sy); ++k) {
                const int src = (oy - mdy_ + 3) * 7 + (ox - mdx_ + 3);           // the cell move md arrives from
                const uint32_t ni = (uint32_t)__shfl((int)info, src, 64), ng = (uint32_t)__shfl((int)gw, src, 64);
                // closed in this search, the move legal from there, and g[n] + w == g[c]
                const bool ok = lane < 8 && ((ni >> md) & 1u) && ((ni >> 8) & 1u) && (GT)ng == (GT)(gc - mw_);
                const unsigned long long m = __ballot(ok);
                if (!m) broken = true;
                else {
                    const int d = __ffsll((long long)m) - 1;
                    cx -= ddx[d]; cy -= ddy[d];
                    ox -= ddx[d]; oy -= ddy[d];
                    gc -= (d < 4 ? 10u : 14u);
                    ++L;
                    if (lane == 0 && L <= Lmax) path[Lmax - L] = cy * W + cx;
                }
            }
        }
        if (broken) out_st = SC_Q_NO_PATH;
        else if (L > Lmax) { out_st = SC_Q_TRUNCATED; out_len = L; out_cost = fcur; }
        else {
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
            out_len = L; out_cost = fcur;
        }
        STAMP(11);  // path extraction
    }
    kcyc = (int)((__builtin_amdgcn_s_memtime() - t_begin) >> 10);
#ifdef ASTAR_STAMPS
    if (lane == 0 && wv == 0 && a.Lmax >= 16) {
        for (int i = 0; i < 12; ++i) path[i] = (int)(stamp[i] >> 10);
        path[12] = nwide; path[13] = nrounds; path[14] = n_qfull; path[15] = n_lvl; path[19] = n_refill;
    }
#endif
    }   // search
    if (lane == 0 && wv == 0) {
        a.status[q] = out_st; a.len[q] = out_len; a.cost[q] = out_cost;
        a.expanded[q] = nexp; a.expanded[a.nstat + q] = npop; a.expanded[2 * a.nstat + q] = kcyc; a.expanded[3 * a.nstat + q] = nstep;
    }
}

template <typename GT>
__global__ void __launch_bounds__(64) astar_kernel(astar_args a) {
    const int nq = a.nq_dev ? *a.nq_dev : a.nq;
    for (;;) {
        // every lane adds 1: the compiler folds that into ONE atomic add of 64 by one lane (no divergent branch in the
        // queue loop); the queue position is the counter in units of a wavefront
        const int qi = __builtin_amdgcn_readfirstlane(atomicAdd(a.counter, 1)) >> 6;
        if (qi >= nq) break;
        astar_query<GT, false, ASTAR_CQ_THROUGHPUT, ASTAR_RU_THROUGHPUT>(a, a.order ? a.order[qi] : qi, (int)blockIdx.x);
    }
}

// Two wavefronts per query (see astar_query): for batches that leave the chip room for them.
template <typename GT, int CQ, int RU>
__global__ void __launch_bounds__(128) astar_kernel_dual(astar_args a) {
    __shared__ int s_qi;
    const int nq = a.nq_dev ? *a.nq_dev : a.nq;
    for (;;) {
        if (threadIdx.x == 0) s_qi = atomicAdd(a.counter, 64) >> 6;   // the queue position is the counter in units of 64, as above
        __syncthreads();
        const int qi = __builtin_amdgcn_readfirstlane(s_qi);
        __syncthreads();
        if (qi >= nq) break;
        astar_query<GT, true, CQ, RU>(a, a.order ? a.order[qi] : qi, (int)blockIdx.x);
        __syncthreads();
    }
}

// Queue order and counters of a launch: queries by descending octile(start, goal) (a proxy for the size of the
// search: the longest ones start first, so the launch does not end on a long query that started late) by a counting
// sort on 1024 bins; also resets the launch's counters (ctr[0] queue position of the main pass, ctr[1] its overflow
// count, ctr[2] / ctr[3] the same for the retry pass).  One workgroup.  order == NULL: counters only.
__global__ void __launch_bounds__(1024) astar_prep_kernel(const int32_t* __restrict__ start, const int32_t* __restrict__ goal, int Q, int W,
                                                          int H, int32_t* __restrict__ order, int32_t* __restrict__ ctr) {
    __shared__ int hist[1024];
    __shared__ int scan[1024];
    const int tid = threadIdx.x;
    if (tid < 4) ctr[tid] = 0;
    if (!order) return;
    hist[tid] = 0;
    __syncthreads();
    const int hmax = 14 * max(W, H) + 1;
    auto key = [&](int q) {
        const int s = start[q], t = goal[q];
        if (s < 0 || t < 0 || s >= W * H || t >= W * H) return 1023;
        const int h = octile(s % W, s / W, t % W, t / W);
        return 1023 - min(1023, (int)(((long long)h * 1024) / hmax));
    };
    for (int q = tid; q < Q; q += 1024) atomicAdd(&hist[key(q)], 1);
    __syncthreads();
    // inclusive scan (Hillis-Steele) -> exclusive offsets
    scan[tid] = hist[tid];
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    hist[tid] = scan[tid] - hist[tid];
    __syncthreads();
    for (int q = tid; q < Q; q += 1024) order[atomicAdd(&hist[key(q)], 1)] = q;
}

// g field of one slot: g where the cell was closed, 0xFFFFFFFF elsewhere
__global__ void __launch_bounds__(256) gfield_kernel(const uint32_t* g, const uint32_t* cl, int W, int H, int tw, int bw, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)W * H) {
        const int x = (int)(i % W), y = (int)(i / W);
        out[i] = (cl[cix(x, y, bw)] >> (x & 31)) & 1u ? g[gix(x, y, tw)] : 0xFFFFFFFFu;
    }
}

// Wavefronts a launch keeps resident: what fits on the chip (LDS-limited), unless SC_ASTAR_WAVES overrides it.
static int astar_resident_waves(sc_ctx* ctx) {
    if (ctx->astar_waves > 0) return ctx->astar_waves;
    int per_cu = 0, dev = ctx->device;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, astar_kernel<uint8_t>, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 8;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) prop.multiProcessorCount = 256;
    int w = per_cu * prop.multiProcessorCount;
    if (const char* e = getenv("SC_ASTAR_WAVES")) { const int v = atoi(e); if (v > 0) w = v; }
    ctx->astar_waves = w;
    return w;
}
// Queries a build of the two-wavefront kernel keeps resident (blocks of 128 threads); 0 = switched off (SC_ASTAR_DUAL=0)
static int astar_resident_dual(sc_ctx* ctx, bool latency) {
    int& cached = latency ? ctx->astar_dual_lat : ctx->astar_dual;
    if (cached >= 0) return cached;
    int per_cu = 0, n = 0;
    hipDeviceProp_t prop;
    const hipError_t e_ = latency ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, astar_kernel_dual<uint8_t, ASTAR_CQ_LATENCY, ASTAR_RU_LATENCY>, 128, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, astar_kernel_dual<uint8_t, ASTAR_CQ_THROUGHPUT, ASTAR_RU_THROUGHPUT>, 128, 0);
    if (e_ == hipSuccess && per_cu > 0 && hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0)
        n = per_cu * prop.multiProcessorCount;
    if (const char* e = getenv("SC_ASTAR_DUAL")) { const int v = atoi(e); if (v == 0) n = 0; else if (v > 1 && v < n) n = v; }   // 0 = off, N > 1 = at most N resident
    if (latency) { if (const char* e = getenv("SC_ASTAR_LATENCY")) { if (atoi(e) == 0) n = 0; } }   // SC_ASTAR_LATENCY=0: the throughput build for every batch
    cached = n;
    return n;
}

static int astar_run(sc_ctx* ctx, const int32_t* d2, int G, const int32_t* qgrid, int W, int H, int32_t r2, const int32_t* start,
                     const int32_t* goal, int Q, int Lmax, int32_t* path, int32_t* len, int32_t* cost,
                     int32_t* status, bool full_g = false) {
    const size_t cells = (size_t)W * H;
    const int tw = full_g ? (W + 3) >> 2 : (W + 7) >> 3, bw = (W + 31) >> 5;
    // g array: whole 4 x 4 tiles of words (sc_astar_gfield), whole 8 x 8 tiles of bytes otherwise
    const size_t gcells = full_g ? (size_t)tw * ((H + 3) >> 2) * 16 : (size_t)tw * ((H + 7) >> 3) * 64;
    const size_t gsz = full_g ? 4 : 1;
    const size_t bwords = (size_t)bw * ((H + 15) >> 4) * 16;  // closed bitmap: whole 32 x 16 tiles
    const int32_t rmin = r2 > 1 ? r2 : 1;
    int r = sc_scratch_reserve(ctx, &ctx->moves, cells * (size_t)G);
    if (r != SC_OK) return r;
    r = sc_launch_moves(ctx, d2, W, H * G, H, r2, (uint8_t*)ctx->moves.p);   // G grids stacked: one launch
    if (r != SC_OK) return r;
    // qstats: expanded[Q] | popped[Q] | kilo-cycles[Q] | steps[Q] | order[Q] | ovf_list[Q]
    r = sc_scratch_reserve(ctx, &ctx->qstats, (size_t)Q * 6 * sizeof(int32_t));
    if (r != SC_OK) return r;
    int32_t* expanded = (int32_t*)ctx->qstats.p;
    int32_t* order = expanded + 4 * (size_t)Q;
    int32_t* ovf_list = order + Q;
    ctx->last_Q = Q;
    // Ring entries per f level.  Measured worst cases (open and 5 % maps): 16 k at 1024^2, 164 k at 4096^2 with one
    // entry per node; duplicates add a few tens of per cent.  Overflowed queries are rerun with 16x the space.
    int cap = ctx->astar_cap;
    if (const char* e = getenv("SC_ASTAR_CAP")) { const int v = atoi(e); if (v >= 1024 && (v & (v - 1)) == 0) cap = v; }
    if (cells > ((size_t)1 << 21)) {
        int want = 1 << 15;
        while (want < 64 * (W > H ? W : H)) want <<= 1;
        if (cap < want) cap = want;
    }
    const size_t per_slot = gcells * gsz + bwords * 4 + (size_t)NBUCKET * cap * 4;
    size_t slots = ctx->astar_slot_budget / per_slot;
    // Two wavefronts per query (astar_kernel_dual) for every batch: it wins on one call's time (its longest search is
    // shorter) and on the saturated rate (fewer instructions per expansion); SC_ASTAR_DUAL=0 selects the one-wavefront kernel.
    // Batches that fit the chip whole run the latency build (every search resident at once: the call lasts as long as
    // its longest search), larger ones the throughput build (more searches per CU).
    const size_t rlat = (size_t)astar_resident_dual(ctx, true);
    const bool latency = !full_g && rlat > 0 && (size_t)Q <= rlat && (size_t)Q <= slots;
    const size_t rdual = latency ? rlat : (size_t)astar_resident_dual(ctx, false);
    const bool dual = !full_g && rdual > 0;
    const size_t resident = dual ? rdual : (size_t)astar_resident_waves(ctx);
    if (slots > resident) slots = resident;
    if (slots > (size_t)Q) slots = Q;
    if (slots < 1) slots = 1;
    // The budget above is per context; the memory is the device's.  When the slot scratch has to grow, what the device
    // has free right now (plus what this context's own slot scratch gives back when it is reallocated) bounds it too --
    // two contexts planning on 4096^2 grids would otherwise ask for 2 x 96 of the 288 GB -- and an allocation that
    // fails all the same halves the slots and tries again: fewer resident searches, never an error while one fits.
    auto ring_slots = [](size_t n) { return n > 16 ? n : (size_t)16; };   // >= 16 slots' worth of rings: the retry pass's 16x
    if (slots * gcells * gsz > ctx->gslots.bytes || slots * bwords * 4 > ctx->closed.bytes || ring_slots(slots) * NBUCKET * (size_t)cap * 4 > ctx->buckets.bytes) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t avail = (size_t)(0.9 * (double)(free_b + ctx->gslots.bytes + ctx->closed.bytes + ctx->buckets.bytes));
            while (slots > 1 && slots * (gcells * gsz + bwords * 4) + ring_slots(slots) * NBUCKET * (size_t)cap * 4 > avail) slots = (slots + 1) / 2;
        }
    }
    for (;;) {
        r = sc_scratch_reserve(ctx, &ctx->gslots, slots * gcells * gsz);
        if (r == SC_OK) r = sc_scratch_reserve(ctx, &ctx->closed, slots * bwords * 4);
        if (r == SC_OK) r = sc_scratch_reserve(ctx, &ctx->buckets, ring_slots(slots) * NBUCKET * (size_t)cap * 4);
        if (r == SC_OK) break;
        if (r != SC_ERR_NOMEM || slots <= 1) return r;
        (void)hipGetLastError();   // the failed allocation is handled here
        slots = (slots + 1) / 2;
    }
    if (!ctx->actr.p) {
        r = sc_scratch_reserve(ctx, &ctx->actr, 16 * sizeof(int32_t));
        if (r != SC_OK) return r;
        SC_HIP(ctx, hipMemsetAsync(ctx->actr.p, 0, 16 * sizeof(int32_t), ctx->stream));
    }
    int32_t* ctr = (int32_t*)ctx->actr.p;   // [0] main queue, [1] main overflow count, [2] retry queue, [3] retry overflow count, [4] sticky

    const bool sorted = (size_t)Q > slots;   // every query starts at once otherwise
    int tk = sc_time_begin(ctx, SC_K_ASTAR);
    hipLaunchKernelGGL(astar_prep_kernel, dim3(1), dim3(1024), 0, ctx->stream, start, goal, Q, W, H, sorted ? order : (int32_t*)nullptr, ctr);
    astar_args a{(const uint8_t*)ctx->moves.p, d2, W, H, rmin, start, goal, qgrid, sorted ? order : nullptr, nullptr, Q, Lmax, path, len,
                 cost, status, ctx->gslots.p, (uint32_t*)ctx->closed.p, (uint32_t*)ctx->buckets.p, cap, expanded, ctr,
                 ovf_list, ctr + 1, ctr + 4, Q, (dual && (size_t)Q > slots) ? 1 : 0, tw, bw, gcells, bwords};
    if (full_g) hipLaunchKernelGGL(astar_kernel<uint32_t>, dim3((unsigned)slots), dim3(64), 0, ctx->stream, a);
    else if (dual && latency && (size_t)Q <= slots)
        hipLaunchKernelGGL((astar_kernel_dual<uint8_t, ASTAR_CQ_LATENCY, ASTAR_RU_LATENCY>), dim3((unsigned)slots), dim3(128), 0, ctx->stream, a);
    else if (dual)
        hipLaunchKernelGGL((astar_kernel_dual<uint8_t, ASTAR_CQ_THROUGHPUT, ASTAR_RU_THROUGHPUT>), dim3((unsigned)slots), dim3(128), 0, ctx->stream, a);
    else hipLaunchKernelGGL(astar_kernel<uint8_t>, dim3((unsigned)slots), dim3(64), 0, ctx->stream, a);
    // retry pass over the overflow list (normally empty: the wavefronts read the count and leave)
    {
        const size_t rslots = slots >= 16 ? slots / 16 : 1;
        const int rcap = cap * 16;
        astar_args b = a;
        b.order = ovf_list; b.nq_dev = ctr + 1; b.nq = 0; b.cap = rcap; b.counter = ctr + 2;
        b.ovf_list = order; b.ovf_count = ctr + 3;   // a second overflow stays in status (SC_Q_RING_OVERFLOW); the list is scratch
        if (full_g) hipLaunchKernelGGL(astar_kernel<uint32_t>, dim3((unsigned)rslots), dim3(64), 0, ctx->stream, b);
        else hipLaunchKernelGGL(astar_kernel<uint8_t>, dim3((unsigned)rslots), dim3(64), 0, ctx->stream, b);
    }
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

extern "C" int sc_astar_batch(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2_clear,
                              const int32_t* start, const int32_t* goal, int Q, int Lmax,
                              int32_t* path, int32_t* len, int32_t* cost, int32_t* status) {
    if (!ctx || !d2 || !start || !goal || !path || !len || !cost || !status || W <= 0 || H <= 0 || Q < 0 ||
        Lmax <= 0 || W > SC_MAX_DIM || H > SC_MAX_DIM)
        return SC_ERR_INVALID;
    if (Q == 0) return SC_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    return astar_run(ctx, d2, 1, nullptr, W, H, r2_clear, start, goal, Q, Lmax, path, len, cost, status);
}

extern "C" int sc_astar_batch_multi(sc_ctx* ctx, const int32_t* d2, int G, const int32_t* qgrid, int W, int H, int32_t r2_clear,
                                    const int32_t* start, const int32_t* goal, int Q, int Lmax,
                                    int32_t* path, int32_t* len, int32_t* cost, int32_t* status) {
    if (!ctx || !d2 || !qgrid || !start || !goal || !path || !len || !cost || !status || G <= 0 || W <= 0 || H <= 0 || Q < 0 ||
        Lmax <= 0 || W > SC_MAX_DIM || H > SC_MAX_DIM || (long long)H * G > 0x7FFFFFFF / (W > 0 ? W : 1))
        return SC_ERR_INVALID;
    if (Q == 0) return SC_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    return astar_run(ctx, d2, G, qgrid, W, H, r2_clear, start, goal, Q, Lmax, path, len, cost, status);
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
    r = astar_run(ctx, d2, 1, nullptr, W, H, r2_clear, sg, sg + 1, 1, 1, sg + 4, sg + 2, cost, status, /*full_g=*/true);
    if (r != SC_OK) return r;
    // one query: it ran in slot 0 of the main pass, or of the retry pass (same slot 0)
    hipLaunchKernelGGL(gfield_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)ctx->gslots.p, (const uint32_t*)ctx->closed.p, W, H, (W + 3) >> 2, (W + 31) >> 5, gfield);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// debug: the 16 counter / marker words of the A* launches, read WITHOUT waiting for the context's stream
extern "C" int sc_astar_debug_peek(sc_ctx* ctx, int32_t* out16) {
    if (!ctx || !out16 || !ctx->actr.p) return SC_ERR_INVALID;
    hipStream_t s;
    SC_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    SC_HIP(ctx, hipMemcpyAsync(out16, ctx->actr.p, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SC_HIP(ctx, hipStreamSynchronize(s));
    SC_HIP(ctx, hipStreamDestroy(s));
    return SC_OK;
}

// debug: per-query {expansions, popped entries, kilo-cycles, steps} of the last batch, int32 [4][Q]
extern "C" int sc_astar_debug_stats(sc_ctx* ctx, int32_t* out, int Q) {
    if (!ctx || !out || Q != ctx->last_Q) return SC_ERR_INVALID;
    SC_HIP(ctx, hipMemcpyAsync(out, ctx->qstats.p, (size_t)Q * 16, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SC_OK;
}
