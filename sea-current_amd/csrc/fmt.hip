// fmt.hip -- the reference's own planner on the GPU, batched over queries.
//
// planning_space::fast_marching_trees (sea_current.hpp:1339-1407): FMT* over a shared set of free samples, with
// near (:1328-1337: distance compared with the SQUARE of the radius), cost (:1315-1326) = segment length unless the
// segment crosses an obstacle edge (intersects :142-178, float arithmetic, colinear = no hit) and pt_dist (:86-88).
// Restated in oracle/fmt_oracle.c; where the reference's result depends on unordered_set iteration order (equal
// costs) the lowest node index wins.  Built with -ffp-contract=off: every float operation is the reference's.
//
// One wavefront per query, the whole node table in LDS (coordinates, cost-to-come, parent, state).  Per iteration:
// the unvisited neighbours of z are compacted by ballot rank; for each of them the open nodes in range are compacted
// too, their segments are tested against the obstacle edges with a lane per (node, edge) pair, and a wave reduction
// picks the cheapest parent (cost-to-come + length); the next z is a wave arg-min over the open set.  Nodes: 0..n-1 samples, n goal, n+1 start.
#include "sc_internal.h"

#include <cfloat>

struct fmt_args {
    const float* samples; int n;
    const float* starts; const float* goals; int Q;
    float rn;
    const float* lines; int E;
    int Lmax;
    float* path; int32_t* len; float* cost; int32_t* status;
    const uint16_t* nbr; const int32_t* nbr_cnt; const int32_t* nbr_ovf;   // per sample: the samples in range, ascending (fmt_neighbors_kernel)
};
#define FMT_NBR_CAP 256

__device__ __forceinline__ float fmt_dist(float ax, float ay, float bx, float by) {
    const float dx = bx - ax, dy = by - ay;
    return (float)sqrt((double)dx * dx + (double)dy * dy);
}
__device__ __forceinline__ float fmt_cross(float ux, float uy, float vx, float vy) { return ux * vy - uy * vx; }
__device__ __forceinline__ bool fmt_hit(float l0x, float l0y, float l1x, float l1y, const float4 k) {
    const float a = fmt_cross(k.x - l0x, k.y - l0y, l1x - l0x, l1y - l0y);
    const float b = fmt_cross(l1x - l0x, l1y - l0y, k.z - k.x, k.w - k.y);
    if (b == 0) return false;
    const float u = a / b;
    const float c = fmt_cross(k.x - l0x, k.y - l0y, k.z - k.x, k.w - k.y);
    const float t = c / b;
    return 0 <= u && u <= 1 && 0 <= t && t <= 1;
}
__device__ __forceinline__ float fmt_edge_cost(float ax, float ay, float bx, float by, const float4* lines, int E) {
    for (int e = 0; e < E; ++e)
        if (fmt_hit(ax, ay, bx, by, lines[e])) return FLT_MAX;
    return fmt_dist(ax, ay, bx, by);
}

// wave arg-min of (v, i): smallest v, then smallest i; i < 0 means "no candidate"
__device__ __forceinline__ void fmt_argmin(float& v, int& i) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(i, o);
        if (oi >= 0 && (i < 0 || ov < v || (ov == v && oi < i))) { v = ov; i = oi; }
    }
}

// The samples are shared by the queries of a batch and `near` depends on positions only: the samples in range of every
// sample, once per batch (one wavefront per sample, ascending by ballot rank, the predicate of the scans below).  A search
// then reads a node's list instead of scanning all n samples for every z and every x (6 scans of 16 steps per iteration
// at n = 1000).  A sample with more than FMT_NBR_CAP in range sets *ovf and the batch falls back to the scans.
__global__ void __launch_bounds__(64) fmt_neighbors_kernel(const float* __restrict__ samples, int n, float rn, uint16_t* __restrict__ nbr,
                                                            int32_t* __restrict__ cnt_out, int32_t* __restrict__ ovf) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const float zx = samples[2 * i], zy = samples[2 * i + 1];
    const double r2 = (double)rn * (double)rn;
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        bool ok = false;
        if (j < n) {
            const float jx = samples[2 * j], jy = samples[2 * j + 1];
            ok = (double)fmt_dist(jx, jy, zx, zy) <= r2 && !(jx == zx && jy == zy);
        }
        const unsigned long long m = __ballot(ok);
        const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (ok && pos < FMT_NBR_CAP) nbr[(size_t)i * FMT_NBR_CAP + pos] = (uint16_t)j;
        cnt += __popcll(m);
    }
    if (lane == 0) {
        cnt_out[i] = cnt;
        if (cnt > FMT_NBR_CAP) *ovf = 1;
    }
}

__global__ void __launch_bounds__(64) fmt_kernel(fmt_args a) {
    extern __shared__ float4 smem4[];
    const int lane = threadIdx.x, q = blockIdx.x;
    const int n = a.n, N = n + 2, GOAL = n, INIT = n + 1, E = a.E;
    float4* lines = smem4;                                   // [E]
    float* px = reinterpret_cast<float*>(lines + E);         // [N]
    float* py = px + N;
    float* cst = py + N;
    uint16_t* par = reinterpret_cast<uint16_t*>(cst + N);    // [N]
    uint16_t* xl = par + N;                                  // [N] compacted neighbours of z
    uint16_t* yl = xl + N;                                   // [N] compacted open nodes near the current x
    uint8_t* st = reinterpret_cast<uint8_t*>(yl + N);        // [N] 0 unvisited, 1 open, 2 closed, 5 opened this iteration
    uint8_t* hit = st + N;                                   // [N] per entry of yl: its segment to x crosses an obstacle edge
    const float sx = a.starts[2 * q], sy = a.starts[2 * q + 1], gx = a.goals[2 * q], gy = a.goals[2 * q + 1];
    for (int e = lane; e < E; e += 64) lines[e] = reinterpret_cast<const float4*>(a.lines)[e];
    for (int i = lane; i < N; i += 64) {
        px[i] = i < n ? a.samples[2 * i] : i == GOAL ? gx : sx;
        py[i] = i < n ? a.samples[2 * i + 1] : i == GOAL ? gy : sy;
        cst[i] = i == INIT ? 0.f : FLT_MAX;
        par[i] = (uint16_t)i;
        st[i] = i == INIT ? 1 : 0;
    }
    __syncthreads();
    const double r2 = (double)a.rn * (double)a.rn;
    const bool lists = a.nbr != nullptr && *a.nbr_ovf == 0;
    // the nodes in range of node c with state `want`, compacted in index order into `dst`: from c's list (samples) plus the
    // two nodes of the query itself, or by a scan over all nodes (c is the start or the goal; no lists)
    auto in_range = [&](const int c, const float cx, const float cy, const uint8_t want, uint16_t* dst) {
        int cnt = 0;
        auto take = [&](const bool ok, const int i) {
            const unsigned long long m = __ballot(ok);
            if (ok) dst[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)i;
            cnt += __popcll(m);
        };
        if (lists && c < n) {
            const int nc = a.nbr_cnt[c];
            const uint16_t* l = a.nbr + (size_t)c * FMT_NBR_CAP;
            for (int base = 0; base < nc; base += 64) {
                const int k = base + lane;
                const int i = k < nc ? (int)l[k] : 0;
                take(k < nc && st[i] == want, i);
            }
            const int i = n + lane;                          // goal, start
            bool ok = false;
            if (lane < 2 && st[i] == want) ok = (double)fmt_dist(px[i], py[i], cx, cy) <= r2 && !(px[i] == cx && py[i] == cy);
            take(ok, i);
        } else {
            for (int base = 0; base < N; base += 64) {
                const int i = base + lane;
                bool ok = false;
                if (i < N && st[i] == want) ok = (double)fmt_dist(px[i], py[i], cx, cy) <= r2 && !(px[i] == cx && py[i] == cy);
                take(ok, i);
            }
        }
        return cnt;
    };
    int z = INIT;
    bool found = true;
    while (!(px[z] == gx && py[z] == gy)) {
        const float zx = px[z], zy = py[z];
        const int cnt = in_range(z, zx, zy, 0, xl);
        __syncthreads();
        for (int k = 0; k < cnt; ++k) {
            const int x = xl[k];
            const float xx = px[x], xy = py[x];
            // the open nodes near x, compacted in index order ...
            const int m = in_range(x, xx, xy, 1, yl);
            for (int j = lane; j < m; j += 64) hit[j] = 0;
            wave_lds_sync();
            // ... their segments to x against every obstacle edge, a lane per (node, edge) pair (the same float operations per
            // pair as one lane walking a node's edges; that way the walk ran for a whole wavefront whenever one lane had a
            // node in range: 62 edges x 16 scan steps per x, mostly idle) ...
            for (int p = lane; p < m * E; p += 64) {
                const int j = p / E, e = p - j * E;
                const int y = yl[j];
                if (fmt_hit(xx, xy, px[y], py[y], lines[e])) hit[j] = 1;
            }
            wave_lds_sync();
            // ... and the cheapest parent: cost-to-come + length, FLT_MAX through an obstacle; equal costs: the lowest index
            float best = 0.f, bec = 0.f;
            int bj = -1;
            for (int j = lane; j < m; j += 64) {
                const int y = yl[j];
                const float ec = hit[j] ? FLT_MAX : fmt_dist(xx, xy, px[y], py[y]);
                const float cy = cst[y] + ec;
                if (bj < 0 || cy < best) { best = cy; bj = j; bec = ec; }
            }
            {
                int wj = bj;
                fmt_argmin(best, wj);
                // the winner's edge cost: from the lane that held it
                const unsigned long long own = __ballot(bj == wj && wj >= 0);
                if (wj >= 0) {
                    const float ec = __shfl(bec, __ffsll((long long)own) - 1);
                    const int bi = yl[wj];
                    if (ec != FLT_MAX && lane == 0) { par[x] = (uint16_t)bi; cst[x] = cst[bi] + ec; st[x] = 5; }
                }
            }
            wave_lds_sync();
        }
        __syncthreads();
        float zb = 0.f;
        int zi = -1;
        for (int i = lane; i < N; i += 64) {
            uint8_t s = st[i];
            if (i == z) s = 2;
            else if (s == 5) s = 1;
            st[i] = s;
            if (s == 1 && (zi < 0 || cst[i] < zb)) { zb = cst[i]; zi = i; }
        }
        fmt_argmin(zb, zi);
        __syncthreads();
        if (zi < 0) { found = false; break; }
        z = zi;
    }
    if (lane == 0) {
        if (!found) { a.status[q] = SC_Q_NO_PATH; a.len[q] = 0; a.cost[q] = -1.f; }
        else {
            int L = 1;
            for (int p = z; p != INIT; p = par[p]) ++L;
            a.len[q] = L;
            a.cost[q] = cst[z];
            if (L > a.Lmax) a.status[q] = SC_Q_TRUNCATED;
            else {
                float* out = a.path + (size_t)q * a.Lmax * 2;
                int k = L - 1;
                for (int p = z;; p = par[p]) { out[2 * k] = px[p]; out[2 * k + 1] = py[p]; if (p == INIT) break; --k; }
                a.status[q] = SC_Q_OK;
            }
        }
    }
}

extern "C" int sc_fmt_star_batch(sc_ctx* ctx, const float* samples, int n, const float* starts, const float* goals, int Q, float rn,
                                 const float* lines, int E, int Lmax, float* path, int32_t* len, float* cost, int32_t* status) {
    if (!ctx || !samples || !starts || !goals || !path || !len || !cost || !status || n < 0 || n > SC_FMT_MAX_SAMPLES || Q < 0 || E < 0 ||
        E > SC_FMT_MAX_EDGES || (E > 0 && !lines) || Lmax <= 0 || !(rn > 0))
        return SC_ERR_INVALID;
    if (Q == 0) return SC_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const int N = n + 2;
    const size_t lds = (size_t)E * 16 + (size_t)N * (3 * 4 + 3 * 2 + 2) + 16;
    // neighbour lists of the samples: fmt_nbr = uint16 [n][FMT_NBR_CAP] | int32 count [n] | int32 overflow
    const size_t nb_bytes = (size_t)n * FMT_NBR_CAP * sizeof(uint16_t);
    int r = sc_scratch_reserve(ctx, &ctx->fmt_nbr, nb_bytes + ((size_t)n + 1) * sizeof(int32_t) + 16);
    if (r != SC_OK) return r;
    uint16_t* nbr = (uint16_t*)ctx->fmt_nbr.p;
    int32_t* ncnt = (int32_t*)((char*)ctx->fmt_nbr.p + ((nb_bytes + 15) & ~(size_t)15));
    int32_t* novf = ncnt + n;
    fmt_args a{samples, n, starts, goals, Q, rn, lines, E, Lmax, path, len, cost, status, n > 0 ? nbr : nullptr, ncnt, novf};
    int tk = sc_time_begin(ctx, SC_K_FMT);
    if (n > 0) {
        SC_HIP(ctx, hipMemsetAsync(novf, 0, sizeof(int32_t), ctx->stream));
        hipLaunchKernelGGL(fmt_neighbors_kernel, dim3(n), dim3(64), 0, ctx->stream, samples, n, rn, nbr, ncnt, novf);
    }
    hipLaunchKernelGGL(fmt_kernel, dim3(Q), dim3(64), lds, ctx->stream, a);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}
