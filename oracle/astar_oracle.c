/*
 * astar_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Optimal 8-connected integer-cost grid A* with the canonical, order-independent
 * g field and parent rule stated in sc_oracle.h.  The reference has no A*
 * (SURVEY.md section 0): the slot is planning_space::fast_marching_trees
 * (sea_current.hpp:1339-1407), whose result type this mirrors
 * (start->goal waypoint list, nullopt == SCO_NO_PATH, :1383-1385, :1397-1406).
 * Definitional oracle -- "parity unpinned" versus the reference.
 */
#include "sc_oracle.h"
#include <stdlib.h>
#include <string.h>

static const int DX[8] = {1, -1, 0, 0, 1, -1, 1, -1};
static const int DY[8] = {0, 0, 1, -1, 1, 1, -1, -1};
static const uint32_t WCOST[8] = {10, 10, 10, 10, 14, 14, 14, 14};

static inline int trav(const int32_t* d2, int32_t c, int32_t rmin) { return d2[c] >= rmin; }

void sco_moves(const int32_t* d2, int W, int H, int32_t r2, uint8_t* moves) {
    int32_t rmin = r2 > 1 ? r2 : 1;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int32_t c = y * W + x;
            uint8_t m = 0;
            if (trav(d2, c, rmin)) {
                for (int d = 0; d < 8; ++d) {
                    int nx = x + DX[d], ny = y + DY[d];
                    if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
                    if (!trav(d2, ny * W + nx, rmin)) continue;
                    if (d >= 4 && (!trav(d2, y * W + nx, rmin) || !trav(d2, ny * W + x, rmin))) continue;
                    m |= (uint8_t)(1u << d);
                }
            }
            moves[c] = m;
        }
}

static inline uint32_t octile(int x, int y, int gx, int gy) {
    int dx = abs(x - gx), dy = abs(y - gy);
    int mx = dx > dy ? dx : dy, mn = dx > dy ? dy : dx;
    return (uint32_t)(10 * mx + 4 * mn);
}

typedef struct { uint32_t f, g; int32_t c; } hent;
typedef struct { hent* a; size_t n, cap; } heap;

static void hpush(heap* h, hent e) {
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 1024; h->a = (hent*)realloc(h->a, h->cap * sizeof(hent)); }
    size_t i = h->n++;
    while (i) {
        size_t p = (i - 1) >> 1;
        if (h->a[p].f <= e.f) break;
        h->a[i] = h->a[p]; i = p;
    }
    h->a[i] = e;
}
static hent hpop(heap* h) {
    hent top = h->a[0], e = h->a[--h->n];
    size_t i = 0;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m;
        if (l >= h->n) break;
        m = (r < h->n && h->a[r].f < h->a[l].f) ? r : l;
        if (h->a[m].f >= e.f) break;
        h->a[i] = h->a[m]; i = m;
    }
    h->a[i] = e;
    return top;
}

/* move d from n to c legal?  (n, c both traversable assumed checked by caller for c) */
static int move_legal(const int32_t* d2, int W, int H, int32_t rmin, int nx, int ny, int d) {
    int cx = nx + DX[d], cy = ny + DY[d];
    if (cx < 0 || cy < 0 || cx >= W || cy >= H) return 0;
    if (!trav(d2, ny * W + nx, rmin) || !trav(d2, cy * W + cx, rmin)) return 0;
    if (d >= 4 && (!trav(d2, ny * W + cx, rmin) || !trav(d2, cy * W + nx, rmin))) return 0;
    return 1;
}

int sco_astar(const int32_t* d2, int W, int H, int32_t r2, int32_t start,
              int32_t goal, int Lmax, int32_t* path, int32_t* len,
              int32_t* cost, uint32_t* gfield, int64_t* expanded) {
    size_t n = (size_t)W * H;
    int32_t rmin = r2 > 1 ? r2 : 1;
    if (len) *len = 0;
    if (cost) *cost = -1;
    if (expanded) *expanded = 0;
    int own = gfield == NULL;
    uint32_t* g = own ? (uint32_t*)malloc(n * sizeof(uint32_t)) : gfield;
    memset(g, 0xFF, n * sizeof(uint32_t));
    if (start < 0 || goal < 0 || (size_t)start >= n || (size_t)goal >= n ||
        !trav(d2, start, rmin) || !trav(d2, goal, rmin)) {
        if (own) free(g);
        return SCO_BAD_ENDPOINT;
    }
    int gx = goal % W, gy = goal / W;
    heap hp = {0, 0, 0};
    g[start] = 0;
    hent e0 = {octile(start % W, start / W, gx, gy), 0, start};
    hpush(&hp, e0);
    uint32_t cstar = SCO_G_INF;
    int64_t nexp = 0;
    while (hp.n) {
        if (hp.a[0].f > cstar) break; /* plateau f == C* exhausted */
        hent e = hpop(&hp);
        if (e.g != g[e.c]) continue; /* stale */
        if (e.c == goal && cstar == SCO_G_INF) cstar = e.g;
        ++nexp;
        int x = e.c % W, y = e.c / W;
        for (int d = 0; d < 8; ++d) {
            if (!move_legal(d2, W, H, rmin, x, y, d)) continue;
            int32_t c2 = (y + DY[d]) * W + (x + DX[d]);
            uint32_t ng = e.g + WCOST[d];
            if (ng < g[c2]) {
                g[c2] = ng;
                hent e2 = {ng + octile(x + DX[d], y + DY[d], gx, gy), ng, c2};
                hpush(&hp, e2);
            }
        }
    }
    free(hp.a);
    if (expanded) *expanded = nexp;
    int status = SCO_OK;
    if (cstar == SCO_G_INF) status = SCO_NO_PATH;
    else {
        if (cost) *cost = (int32_t)cstar;
        /* count length by following canonical parents from goal */
        int32_t L = 1, c = goal;
        while (c != start) {
            int cx = c % W, cy = c / W, found = 0;
            for (int d = 0; d < 8 && !found; ++d) {
                int nx = cx - DX[d], ny = cy - DY[d];
                if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
                if (!move_legal(d2, W, H, rmin, nx, ny, d)) continue;
                int32_t nn = ny * W + nx;
                if (g[nn] != SCO_G_INF && g[nn] + WCOST[d] == g[c]) { c = nn; found = 1; }
            }
            if (!found) { status = SCO_NO_PATH; break; } /* cannot happen */
            ++L;
        }
        if (len) *len = L;
        if (L > Lmax) status = SCO_PATH_TRUNCATED;
        else if (path) {
            int32_t i = L - 1;
            c = goal;
            path[i] = c;
            while (c != start) {
                int cx = c % W, cy = c / W;
                for (int d = 0; d < 8; ++d) {
                    int nx = cx - DX[d], ny = cy - DY[d];
                    if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
                    if (!move_legal(d2, W, H, rmin, nx, ny, d)) continue;
                    int32_t nn = ny * W + nx;
                    if (g[nn] != SCO_G_INF && g[nn] + WCOST[d] == g[c]) { c = nn; break; }
                }
                path[--i] = c;
            }
        }
    }
    if (own) free(g);
    return status;
}

void sco_astar_batch(const int32_t* d2, int W, int H, int32_t r2,
                     const int32_t* start, const int32_t* goal, int Q, int Lmax,
                     int32_t* path, int32_t* len, int32_t* cost,
                     int32_t* status, int64_t* expanded, int nthreads) {
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int q = 0; q < Q; ++q) {
        int32_t l = 0, c = -1;
        int64_t ex = 0;
        int st = sco_astar(d2, W, H, r2, start[q], goal[q], Lmax,
                           path ? path + (size_t)q * Lmax : NULL, &l, &c, NULL, &ex);
        if (len) len[q] = l;
        if (cost) cost[q] = c;
        if (status) status[q] = st;
        if (expanded) expanded[q] = ex;
    }
}
