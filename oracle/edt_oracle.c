/*
 * edt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Exact squared Euclidean distance transform of an occupancy grid.
 * The reference has no grid EDT: the slot is occupied by
 * obstacle::contains (sea_current.hpp:201-251) and planning_space::is_obstacle
 * (:1274-1280).  This oracle is definitional ("parity unpinned" vs reference):
 * sco_edt_brute IS the definition; sco_edt_exact is checked against it.
 */
#include "sc_oracle.h"
#include <stdlib.h>
#include <string.h>

void sco_edt_brute(const uint8_t* occ, int W, int H, int32_t* d2) {
    size_t n = (size_t)W * H;
    int32_t* ox = (int32_t*)malloc(n * sizeof(int32_t));
    int32_t* oy = (int32_t*)malloc(n * sizeof(int32_t));
    size_t m = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            if (occ[(size_t)y * W + x]) { ox[m] = x; oy[m] = y; ++m; }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int64_t best = INT64_MAX;
            for (size_t k = 0; k < m; ++k) {
                int64_t dx = x - ox[k], dy = y - oy[k];
                int64_t d = dx * dx + dy * dy;
                if (d < best) best = d;
            }
            d2[(size_t)y * W + x] = m ? (int32_t)best : SCO_EDT_INF;
        }
    free(ox); free(oy);
}

/* floor(a / b) for b > 0 */
static int64_t floor_div(int64_t a, int64_t b) {
    int64_t q = a / b, r = a % b;
    return (r != 0 && r < 0) ? q - 1 : q;
}

#define BIGG ((int64_t)1 << 20) /* "no obstacle in this column"; BIGG^2 > any real d2 */

static inline int64_t fval(const int64_t* g, int64_t x, int64_t i) {
    return (x - i) * (x - i) + g[i] * g[i];
}
static inline int64_t sep(const int64_t* g, int64_t i, int64_t u) {
    return floor_div(u * u - i * i + g[u] * g[u] - g[i] * g[i], 2 * (u - i));
}

/* Meijster, Roerdink & Hesselink (2000), all-integer two-phase algorithm. */
void sco_edt_exact(const uint8_t* occ, int W, int H, int32_t* d2) {
    size_t n = (size_t)W * H;
    int64_t* G = (int64_t*)malloc(n * sizeof(int64_t));
    /* phase 1: vertical distance to nearest occupied cell in the same column */
    for (int x = 0; x < W; ++x) {
        int64_t last = -1;
        for (int y = 0; y < H; ++y) {
            if (occ[(size_t)y * W + x]) last = y;
            G[(size_t)y * W + x] = last < 0 ? BIGG : y - last;
        }
        last = -1;
        for (int y = H - 1; y >= 0; --y) {
            if (occ[(size_t)y * W + x]) last = y;
            if (last >= 0 && last - y < G[(size_t)y * W + x]) G[(size_t)y * W + x] = last - y;
        }
    }
    /* phase 2: lower envelope of parabolas along each row */
    int64_t* s = (int64_t*)malloc((size_t)W * sizeof(int64_t));
    int64_t* t = (int64_t*)malloc((size_t)W * sizeof(int64_t));
    for (int y = 0; y < H; ++y) {
        const int64_t* g = G + (size_t)y * W;
        int64_t q = 0;
        s[0] = 0; t[0] = 0;
        for (int64_t u = 1; u < W; ++u) {
            while (q >= 0 && fval(g, t[q], s[q]) > fval(g, t[q], u)) --q;
            if (q < 0) { q = 0; s[0] = u; }
            else {
                int64_t w = 1 + sep(g, s[q], u);
                if (w < W) { ++q; s[q] = u; t[q] = w; }
            }
        }
        for (int64_t u = W - 1; u >= 0; --u) {
            int64_t v = fval(g, u, s[q]);
            d2[(size_t)y * W + u] = v >= BIGG * BIGG ? SCO_EDT_INF : (int32_t)v;
            if (u == t[q]) --q;
        }
    }
    free(s); free(t); free(G);
}

/* nearest[c] = linear index of the occupied cell closest to c; among equidistant ones the smallest index; -1 when the
 * grid is empty.  Definition by exhaustive search (small grids). */
void sco_edt_nearest_brute(const uint8_t* occ, int W, int H, int32_t* nearest) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int64_t best = INT64_MAX;
            int32_t arg = -1;
            for (int yy = 0; yy < H; ++yy)
                for (int xx = 0; xx < W; ++xx)
                    if (occ[(size_t)yy * W + xx]) {
                        const int64_t dx = x - xx, dy = y - yy, d = dx * dx + dy * dy;
                        if (d < best) { best = d; arg = yy * W + xx; }   /* scan order = index order: first hit is the smallest */
                    }
            nearest[(size_t)y * W + x] = arg;
        }
}

/* Same result from the exact d2: the nearest cell lies on the circle dx^2 + dy^2 = d2 around c, so only the integer
 * points of that circle are tested (O(sqrt d2) per cell; any grid size). */
void sco_edt_nearest(const uint8_t* occ, const int32_t* d2, int W, int H, int32_t* nearest) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int32_t D = d2[(size_t)y * W + x];
            int32_t arg = -1;
            if (D != SCO_EDT_INF) {
                int64_t best = INT64_MAX;
                for (int64_t dx = 0; dx * dx <= D; ++dx) {
                    const int64_t rem = D - dx * dx;
                    int64_t dy = 0;
                    while ((dy + 1) * (dy + 1) <= rem) ++dy;
                    if (dy * dy != rem) continue;
                    for (int sx = -1; sx <= 1; sx += 2)
                        for (int sy = -1; sy <= 1; sy += 2) {
                            const int64_t xx = x + sx * dx, yy = y + sy * dy;
                            if (xx < 0 || yy < 0 || xx >= W || yy >= H || !occ[(size_t)yy * W + xx]) continue;
                            if (yy * W + xx < best) best = yy * W + xx;
                        }
                }
                arg = (int32_t)best;
            }
            nearest[(size_t)y * W + x] = arg;
        }
}
