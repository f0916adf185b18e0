/*
 * bezier_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restates the path-smoothing step that follows the planner on every example path of the
 * reference (examples/zmq_test.cpp:66-68): bezier_spline::from_path (sea_current.hpp:599-683) with
 * the Lau09 tangent heuristics calc_start_tangent / calc_tangent / calc_end_tangent (:343-377),
 * shrink_tangent against obstacle edges (:575-596), the cubic Bezier and its hodograph (:1041-1053)
 * and bezier_spline::arclength (:767-896: 32-point Gauss-Legendre on 1/precision sub-intervals per
 * segment, cumulative table per segment).  Curve evaluation uses the Bernstein form; the reference
 * evaluates the same polynomial through its Bernstein-Fourier form (:700-763).
 * PINNED by examples/output.json: total arclength and both 101-entry cumulative tables
 * (tests/test_oracle_bezier.py).
 */
#include "sc_oracle.h"
#include <math.h>
#include <stdlib.h>

static double dist2(double ax, double ay, double bx, double by) { return hypot(bx - ax, by - ay); }

/* proper segment intersection (colinear counts as no hit, sea_current.hpp:142-178) */
static int seg_hit(double p1x, double p1y, double p2x, double p2y, const float* l, double* ix, double* iy) {
    double rx = p2x - p1x, ry = p2y - p1y, sx = l[2] - l[0], sy = l[3] - l[1];
    double den = rx * sy - ry * sx;
    if (den == 0.0) return 0;
    double qx = l[0] - p1x, qy = l[1] - p1y;
    double t = (qx * sy - qy * sx) / den, u = (qx * ry - qy * rx) / den;
    if (t < 0 || t > 1 || u < 0 || u > 1) return 0;
    *ix = p1x + t * rx; *iy = p1y + t * ry;
    return 1;
}

/* shrink_tangent (:575-596): cut the tangent where W +- T crosses an obstacle edge */
static void shrink(double* tx, double* ty, double wx, double wy, const float* lines, int nlines) {
    for (int i = 0; i < nlines; ++i) {
        double ix, iy;
        if (seg_hit(wx + *tx, wy + *ty, wx, wy, lines + 4 * i, &ix, &iy)) { *tx = ix - wx; *ty = iy - wy; }
        if (seg_hit(wx - *tx, wy - *ty, wx, wy, lines + 4 * i, &ix, &iy)) { *tx = wx - ix; *ty = wy - iy; }
    }
}

/* path: n waypoints (x,y); lines: nlines obstacle edges (x0,y0,x1,y1); start_angle NaN = along the first
 * leg (:605-609).  ctrl: [n-1][4][2] control points of the cubic per leg. */
void sco_bezier_from_path(const float* path, int n, float start_angle, const float* lines, int nlines, float* ctrl) {
    double* T = (double*)malloc(sizeof(double) * 2 * n);
#define PX(i) ((double)path[2 * (i)])
#define PY(i) ((double)path[2 * (i) + 1])
    double th = start_angle;
    if (isnan(start_angle)) th = atan2(PY(1) - PY(0), PX(1) - PX(0));
    /* start: magnitude 0.5*min(|W0W1|,|W1W0|) along theta (:371-373) */
    double m0 = 0.5 * dist2(PX(0), PY(0), PX(1), PY(1));
    T[0] = m0 * cos(th); T[1] = m0 * sin(th);
    shrink(&T[0], &T[1], PX(0), PY(0), lines, nlines);
    /* end: along the last leg (:375-377) */
    double ex = PX(n - 1) - PX(n - 2), ey = PY(n - 1) - PY(n - 2), el = hypot(ex, ey);
    double me = 0.5 * el;
    T[2 * (n - 1)] = me * ex / el; T[2 * (n - 1) + 1] = me * ey / el;
    shrink(&T[2 * (n - 1)], &T[2 * (n - 1) + 1], PX(n - 1), PY(n - 1), lines, nlines);
    /* interior: perpendicular to the angular bisector, pointing towards the next waypoint (:347-369) */
    for (int i = 1; i < n - 1; ++i) {
        double ux = PX(i - 1) - PX(i), uy = PY(i - 1) - PY(i), vx = PX(i + 1) - PX(i), vy = PY(i + 1) - PY(i);
        double theta = acos((ux * vx + uy * vy) / (hypot(ux, uy) * hypot(vx, vy))) / 2;
        double off = atan2(uy, ux), toff = atan2(vy, vx);
        int mult = (toff - off < 0) ? -1 : 1;
        double lx = sin(off + mult * theta), ly = -cos(off + mult * theta), ll = hypot(lx, ly);
        lx /= ll; ly /= ll;
        int mult2 = dist2(PX(i) + lx, PY(i) + ly, PX(i + 1), PY(i + 1)) < dist2(PX(i) - lx, PY(i) - ly, PX(i + 1), PY(i + 1)) ? 1 : -1;
        double mag = 0.5 * fmin(hypot(ux, uy), hypot(vx, vy));
        T[2 * i] = mag * mult2 * lx; T[2 * i + 1] = mag * mult2 * ly;
        shrink(&T[2 * i], &T[2 * i + 1], PX(i), PY(i), lines, nlines);
    }
    for (int i = 0; i < n - 1; ++i) {
        float* c = ctrl + 8 * i;
        c[0] = (float)PX(i); c[1] = (float)PY(i);
        c[2] = (float)(PX(i) + T[2 * i]); c[3] = (float)(PY(i) + T[2 * i + 1]);
        c[4] = (float)(PX(i + 1) - T[2 * (i + 1)]); c[5] = (float)(PY(i + 1) - T[2 * (i + 1) + 1]);
        c[6] = (float)PX(i + 1); c[7] = (float)PY(i + 1);
    }
    free(T);
}

/* point (order 0), hodograph (1) or second derivative (2) of cubic segments at parameters t */
void sco_bezier_eval(const float* ctrl, const int* seg, const double* t, int m, int order, double* out) {
    for (int i = 0; i < m; ++i) {
        const float* c = ctrl + 8 * seg[i];
        double s = t[i], r = 1 - s;
        for (int a = 0; a < 2; ++a) {
            double p0 = c[a], p1 = c[2 + a], p2 = c[4 + a], p3 = c[6 + a], v;
            if (order == 0) v = r * r * r * p0 + 3 * r * r * s * p1 + 3 * r * s * s * p2 + s * s * s * p3;
            else if (order == 1) v = 3 * (r * r * (p1 - p0) + 2 * r * s * (p2 - p1) + s * s * (p3 - p2));
            else v = 6 * (r * (p2 - 2 * p1 + p0) + s * (p3 - 2 * p2 + p1));
            out[2 * i + a] = v;
        }
    }
}

/* 32-point Gauss-Legendre nodes/weights on [-1,1] by Newton iteration on P_32 */
static void gl32(double* x, double* w) {
    const int N = 32;
    for (int i = 0; i < N; ++i) {
        double z = cos(acos(-1.0) * (i + 0.75) / (N + 0.5)), pp = 0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1, p2 = 0;
            for (int j = 1; j <= N; ++j) { double p3 = p2; p2 = p1; p1 = ((2.0 * j - 1) * z * p2 - (j - 1.0) * p3) / j; }
            pp = N * (z * p1 - p2) / (z * z - 1);
            double z1 = z;
            z = z1 - p1 / pp;
            if (fabs(z - z1) < 1e-16) break;
        }
        x[i] = z; w[i] = 2 / ((1 - z * z) * pp * pp);
    }
}

/* cum: [nseg][nsub+1] cumulative arclength of each segment at t = k/nsub (cum[s][0] = 0); returns total */
double sco_bezier_arclength(const float* ctrl, int nseg, int nsub, double* cum) {
    double x[32], w[32], total = 0;
    gl32(x, w);
    for (int s = 0; s < nseg; ++s) {
        double acc = 0;
        cum[(size_t)s * (nsub + 1)] = 0;
        for (int k = 0; k < nsub; ++k) {
            double a = (double)k / nsub, b = (double)(k + 1) / nsub, sum = 0;
            for (int i = 0; i < 32; ++i) {
                double t = 0.5 * (b - a) * x[i] + 0.5 * (b + a), d[2];
                int sg = s;
                sco_bezier_eval(ctrl, &sg, &t, 1, 1, d);
                sum += w[i] * hypot(d[0], d[1]);
            }
            acc += sum * 0.5 * (b - a);
            cum[(size_t)s * (nsub + 1) + k + 1] = acc;
        }
        total += acc;
    }
    return total;
}
