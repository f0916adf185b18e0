// fmt.hip -- the reference's own planner on the GPU, batched over queries.
//
// planning_space::fast_marching_trees (sea_current.hpp:1339-1407): FMT* over a shared set of free samples, with
// near (:1328-1337: distance compared with the SQUARE of the radius), cost (:1315-1326) = segment length unless the
// segment crosses an obstacle edge (intersects :142-178, float arithmetic, colinear = no hit) and pt_dist (:86-88).
// Restated in oracle/fmt_oracle.c; where the reference's result depends on unordered_set iteration order (equal
// costs) the lowest node index wins.  Built with -ffp-contract=off: every float operation is the reference's.
//
// One wavefront per query, the whole node table in LDS (coordinates, cost-to-come, parent, state).  Per iteration:
// the unvisited neighbours of z are compacted by ballot rank; for each of them the lanes scan the open set in
// parallel (radius test, collision test against all obstacle edges, cost-to-come + length) and a wave reduction picks
// the cheapest parent; the next z is a wave arg-min over the open set.  Nodes: 0..n-1 samples, n goal, n+1 start.
#include "sc_internal.h"

#include <cfloat>

struct fmt_args {
    const float* samples; int n;
    const float* starts; const float* goals; int Q;
    float rn;
    const float* lines; int E;
    int Lmax;
    float* path; int32_t* len; float* cost; int32_t* status;
};

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
    uint8_t* st = reinterpret_cast<uint8_t*>(xl + N);        // [N] 0 unvisited, 1 open, 2 closed, 5 opened this iteration
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
    int z = INIT;
    bool found = true;
    while (!(px[z] == gx && py[z] == gy)) {
        const float zx = px[z], zy = py[z];
        int cnt = 0;
        for (int base = 0; base < N; base += 64) {
            const int i = base + lane;
            bool ok = false;
            if (i < N && st[i] == 0) ok = (double)fmt_dist(px[i], py[i], zx, zy) <= r2 && !(px[i] == zx && py[i] == zy);
            const unsigned long long m = __ballot(ok);
            if (ok) xl[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)i;
            cnt += __popcll(m);
        }
        __syncthreads();
        for (int k = 0; k < cnt; ++k) {
            const int x = xl[k];
            const float xx = px[x], xy = py[x];
            float best = 0.f;
            int bi = -1;
            for (int y = lane; y < N; y += 64) {
                if (st[y] != 1) continue;
                const float yx = px[y], yy = py[y];
                if (!((double)fmt_dist(yx, yy, xx, xy) <= r2) || (yx == xx && yy == xy)) continue;
                const float cy = cst[y] + fmt_edge_cost(xx, xy, yx, yy, lines, E);
                if (bi < 0 || cy < best) { best = cy; bi = y; }
            }
            fmt_argmin(best, bi);
            if (bi >= 0) {
                const float ec = fmt_edge_cost(xx, xy, px[bi], py[bi], lines, E);
                if (ec != FLT_MAX && lane == 0) { par[x] = (uint16_t)bi; cst[x] = cst[bi] + ec; st[x] = 5; }
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
    const size_t lds = (size_t)E * 16 + (size_t)N * (3 * 4 + 2 * 2 + 1) + 16;
    fmt_args a{samples, n, starts, goals, Q, rn, lines, E, Lmax, path, len, cost, status};
    int tk = sc_time_begin(ctx, SC_K_FMT);
    hipLaunchKernelGGL(fmt_kernel, dim3(Q), dim3(64), lds, ctx->stream, a);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}
