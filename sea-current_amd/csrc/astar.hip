// astar.hip -- batched grid A*, one wavefront per query (gfx950).
//
// Takes over planning_space::fast_marching_trees (sea_current.hpp:1339-1407): (start, goal) ->
// optional waypoint list.  Semantics (costs 10/14, octile heuristic, no corner cutting, canonical
// g field and parent rule) are written in oracle/sc_oracle.h; results are bit-exact against it.
//
// Why a bucket queue: with integer costs and a consistent heuristic every open node has
// f in [fmin, fmin + 28] (a successor's f exceeds its parent's by at most 2*14), so the open list
// is 32 circular buckets indexed by f & 31 -- no heap, no comparisons.  Every node in the fmin
// bucket already has its optimal g, so the whole bucket is expanded in parallel, 64 nodes per
// step.  Successors are relaxed with atomicMin on the query's private g array (exactly one lane
// sees old > new, so each (node, g) is queued once) and appended to their buckets by wavefront
// ballot compaction: lanes that improved a node with the same f' take consecutive slots
// (popcount of the ballot below the lane); the ring tails live in LDS.  No inter-wave
// communication exists: a query's g array, buckets and counters are private to its wave.
//
// The search runs until the f = C* bucket is exhausted (not merely until the goal is popped), which
// makes the final g field -- and therefore the parent chain extracted from it -- independent of the
// expansion order.
#include "sc_internal.h"

#define NBUCKET 32
#define G_UNSET 0xFFFFFFFFu
#define Q_OVERFLOW 100  // internal: bucket ring overflow, retried by the host with a larger ring

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
    uint32_t* g;        // [slots][cells]
    uint32_t* buckets;  // [slots][NBUCKET][cap]
    int cap;            // power of two
    int32_t* expanded;  // [Q]
    const int32_t* redo;  // optional: only run queries whose status == Q_OVERFLOW
};

__device__ __forceinline__ int octile(int x, int y, int gx, int gy) {
    int dx = abs(x - gx), dy = abs(y - gy);
    return 10 * max(dx, dy) + 4 * min(dx, dy);
}

__device__ __forceinline__ uint32_t g_load(const uint32_t* p) {
    // agent-scope relaxed load: served by L2, where this wave's atomicMin results live
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(64) astar_kernel(astar_args a) {
    __shared__ int s_head[NBUCKET];
    __shared__ int s_tail[NBUCKET];
    volatile int* head = s_head;
    volatile int* tail = s_tail;
    const int lane = threadIdx.x;
    const int slot = blockIdx.x;
    const int q = a.q0 + slot;
    if (a.redo && a.redo[q] != Q_OVERFLOW) return;
    const int W = a.W, H = a.H;
    const size_t cells = (size_t)W * H;
    const int s = a.start[q], t = a.goal[q];
    int32_t* path = a.path + (size_t)q * a.Lmax;

    auto finish = [&](int st, int ln, int cs, int ex) {
        if (lane == 0) { a.status[q] = st; a.len[q] = ln; a.cost[q] = cs; a.expanded[q] = ex; }
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
    uint32_t* g = a.g + (size_t)slot * cells;
    uint32_t* bk = a.buckets + (size_t)slot * NBUCKET * a.cap;
    const int cap = a.cap, capm = a.cap - 1;
    const int sx = s % W, sy = s / W, gx = t % W, gy = t / W;
    const int off[8] = {1, -1, W, -W, W + 1, W - 1, -W + 1, -W - 1};
    const int ddx[8] = {1, -1, 0, 0, 1, -1, 1, -1};
    const int ddy[8] = {0, 0, 1, -1, 1, 1, -1, -1};

    if (lane < NBUCKET) { head[lane] = 0; tail[lane] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int fcur = octile(sx, sy, gx, gy);
    if (lane == 0) {
        __hip_atomic_store(&g[s], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bk[(size_t)(fcur & 31) * cap] = (uint32_t)(sy << 16 | sx);
        tail[fcur & 31] = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    bool found = false, overflow = false;
    int nexp = 0;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    for (;;) {
        const int b = fcur & 31;
        uint32_t* bq = bk + (size_t)b * cap;
        int hd = head[b];
        // drain the current bucket; same-f successors are appended while we go
        for (;;) {
            const int tl = tail[b];
            if (hd >= tl) break;
            for (int base0 = hd; base0 < tl && !overflow; base0 += 64) {
                const int idx = base0 + lane;
                const bool act = idx < tl;
                uint32_t xy = act ? bq[idx & capm] : 0u;
                const int cx = xy & 0xFFFF, cy = xy >> 16;
                const int c = cy * W + cx;
                uint32_t gc = act ? g_load(&g[c]) : 0u;
                uint32_t mv = act ? a.moves[c] : 0u;
                const bool valid = act && (int)(gc + octile(cx, cy, gx, gy)) == fcur;  // else stale entry
                if (!valid) mv = 0;
                nexp += __popcll(__ballot(valid));
                if (__ballot(valid && c == t)) found = true;
                uint32_t old[8];
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    old[d] = 0;
                    if ((mv >> d) & 1)
                        old[d] = __hip_atomic_fetch_min(&g[c + off[d]], gc + (d < 4 ? 10u : 14u), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const uint32_t ng = gc + (d < 4 ? 10u : 14u);
                    const bool imp = ((mv >> d) & 1) && old[d] > ng;
                    const int nx = cx + ddx[d], ny = cy + ddy[d];
                    const int df = (int)ng + octile(nx, ny, gx, gy) - fcur;  // in {0,6,8,14,20,28}
                    unsigned long long rem = __ballot(imp);
                    while (rem) {
                        const int leader = __ffsll((long long)rem) - 1;
                        const int v = __builtin_amdgcn_readlane(df, leader);
                        const unsigned long long m = __ballot(imp && df == v);
                        const int bb = (fcur + v) & 31;
                        const int base = tail[bb];
                        const int cnt = __popcll(m);
                        if (base + cnt - head[bb] > cap) overflow = true;
                        else {
                            if (imp && df == v)
                                bk[(size_t)bb * cap + ((base + __popcll(m & lt_mask)) & capm)] = (uint32_t)(ny << 16 | nx);
                            if (lane == 0) tail[bb] = base + cnt;
                        }
                        rem &= ~m;
                    }
                }
            }
            if (overflow) break;
            hd = tl;
            if (lane == 0) head[b] = hd;
        }
        if (overflow || found) break;
        // bucket fcur is empty: advance to the next non-empty one
        int step = 1;
        for (; step < NBUCKET; ++step)
            if (head[(fcur + step) & 31] != tail[(fcur + step) & 31]) break;
        if (step == NBUCKET) break;  // open list empty: no path
        fcur += step;
    }

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
                    const uint32_t gn = g_load(&g[n]);
                    ok = gn != G_UNSET && gn + (d < 4 ? 10u : 14u) == gc;
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

static int astar_run(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2, const int32_t* start,
                     const int32_t* goal, int Q, int Lmax, int32_t* path, int32_t* len, int32_t* cost,
                     int32_t* status) {
    const size_t cells = (size_t)W * H;
    const int32_t rmin = r2 > 1 ? r2 : 1;
    int r = sc_scratch_reserve(ctx, &ctx->moves, cells);
    if (r != SC_OK) return r;
    r = sc_launch_moves(ctx, d2, W, H, r2, (uint8_t*)ctx->moves.p);
    if (r != SC_OK) return r;
    r = sc_scratch_reserve(ctx, &ctx->qstats, (size_t)Q * sizeof(int32_t));
    if (r != SC_OK) return r;
    ctx->last_Q = Q;

    int cap = ctx->astar_cap;
    const int32_t* redo = nullptr;
    for (int attempt = 0; attempt < 6; ++attempt) {
        const size_t per_slot = cells * 4 + (size_t)NBUCKET * cap * 4;
        size_t slots = ctx->astar_slot_budget / per_slot;
        if (slots < 1) slots = 1;
        if (slots > (size_t)Q) slots = Q;
        r = sc_scratch_reserve(ctx, &ctx->gslots, slots * cells * 4);
        if (r != SC_OK) return r;
        r = sc_scratch_reserve(ctx, &ctx->buckets, slots * NBUCKET * (size_t)cap * 4);
        if (r != SC_OK) return r;
        for (int q0 = 0; q0 < Q; q0 += (int)slots) {
            const int nq = (int)((size_t)(Q - q0) < slots ? (size_t)(Q - q0) : slots);
            SC_HIP(ctx, hipMemsetAsync(ctx->gslots.p, 0xFF, (size_t)nq * cells * 4, ctx->stream));
            astar_args a{(const uint8_t*)ctx->moves.p, d2, W, H, rmin, start, goal, q0, nq, Lmax, path, len, cost,
                         status, (uint32_t*)ctx->gslots.p, (uint32_t*)ctx->buckets.p, cap, (int32_t*)ctx->qstats.p, redo};
            int tk = sc_time_begin(ctx, SC_K_ASTAR);
            hipLaunchKernelGGL(astar_kernel, dim3(nq), dim3(64), 0, ctx->stream, a);
            sc_time_end(ctx, tk);
            SC_HIP(ctx, hipGetLastError());
        }
        // Ring overflow is rare (cap is generous); detecting it needs the statuses on the host.
        // Only pay the synchronisation when a previous call on this context ever overflowed or
        // on the first call with this grid size.
        std::vector<int32_t> st(Q);
        SC_HIP(ctx, hipMemcpyAsync(st.data(), status, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream));
        SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
    // Lmax = 1: the path is not wanted; a found path reports SC_Q_TRUNCATED, mapped back to OK below
    r = astar_run(ctx, d2, W, H, r2_clear, sg, sg + 1, 1, 1, sg + 4, sg + 2, cost, status);
    if (r != SC_OK) return r;
    SC_HIP(ctx, hipMemcpyAsync(gfield, ctx->gslots.p, cells * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return SC_OK;
}
