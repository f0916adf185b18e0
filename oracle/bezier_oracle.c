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

/* shrink_tangent on its own (:575-596): out[i] = k * T[i] cut against the edges at waypoint Wp[i] */
void sco_bezier_shrink_tangent(const float* T, const float* Wp, int M, float k, const float* lines, int nlines, float* out) {
    for (int i = 0; i < M; ++i) {
        double tx = (double)k * (double)T[2 * i], ty = (double)k * (double)T[2 * i + 1];
        shrink(&tx, &ty, (double)Wp[2 * i], (double)Wp[2 * i + 1], lines, nlines);
        out[2 * i] = (float)tx; out[2 * i + 1] = (float)ty;
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

/* ---- resample (:898-1005) with chebfit / chebeval (:1109-1170) and curvature (:1017-1039) ----------------
 * PINNED by examples/output.json: pos_x / pos_y (1.1e-5 absolute on coordinates up to 11) and ang_vel = vel * curvature
 * (9e-8) of the recorded run (tests/test_oracle_bezier.py).  The reference fits in float32 (Eigen HouseholderQR);
 * this restatement runs the same Householder least-squares fit in fp64 on the float32 tables. */

/* least-squares Chebyshev fit of y(x), columns T_0..T_{deg-1} of the normalised abscissa; coef [deg] */
static void chebfit(const float* x, const double* y, int m, int deg, double* coef, double* xmin_o, double* xmax_o) {
    double xmin = x[0], xmax = x[0];
    for (int r = 1; r < m; ++r) { if (x[r] < xmin) xmin = x[r]; if (x[r] > xmax) xmax = x[r]; }
    const int nc = deg + 1;                       /* [T | y] */
    double* A = (double*)malloc(sizeof(double) * m * nc);
    for (int r = 0; r < m; ++r) {
        const double xn = (2 * (double)x[r] - (xmax + xmin)) / (xmax - xmin);
        double* a = A + (size_t)r * nc;
        a[0] = 1;
        if (deg > 1) a[1] = xn;
        for (int j = 2; j < deg; ++j) a[j] = 2 * xn * a[j - 1] - a[j - 2];
        a[deg] = y[r];
    }
    for (int k = 0; k < deg && k < m; ++k) {      /* Householder reflections, column by column */
        double nrm = 0;
        for (int r = k; r < m; ++r) nrm += A[(size_t)r * nc + k] * A[(size_t)r * nc + k];
        nrm = sqrt(nrm);
        if (nrm == 0) continue;
        const double akk = A[(size_t)k * nc + k], alpha = akk > 0 ? -nrm : nrm;
        const double vk = akk - alpha, vtv = nrm * nrm - akk * akk + vk * vk;    /* v = a_k - alpha e_k */
        for (int j = k + 1; j < nc; ++j) {
            double dot = vk * A[(size_t)k * nc + j];
            for (int r = k + 1; r < m; ++r) dot += A[(size_t)r * nc + k] * A[(size_t)r * nc + j];
            const double f = 2 * dot / vtv;
            A[(size_t)k * nc + j] -= f * vk;
            for (int r = k + 1; r < m; ++r) A[(size_t)r * nc + j] -= f * A[(size_t)r * nc + k];
        }
        A[(size_t)k * nc + k] = alpha;
    }
    for (int k = deg - 1; k >= 0; --k) {          /* R c = (Q^T y)[0:deg] */
        double s = A[(size_t)k * nc + deg];
        for (int j = k + 1; j < deg; ++j) s -= A[(size_t)k * nc + j] * coef[j];
        coef[k] = s / A[(size_t)k * nc + k];
    }
    free(A);
    *xmin_o = xmin; *xmax_o = xmax;
}

static double chebeval(double x, const double* coef, int deg, double xmin, double xmax) {
    const double xn = (2 * x - (xmax + xmin)) / (xmax - xmin);
    double t0 = 1, t1 = xn, y = coef[0];
    if (deg > 1) y += coef[1] * t1;
    for (int j = 2; j < deg; ++j) { const double t2 = 2 * xn * t1 - t0; y += coef[j] * t2; t0 = t1; t1 = t2; }
    return y;
}

/* One spline of nseg cubic segments with arclength tables cum [nseg][nsub+1] (float, as arclength_data holds them) and
 * total `arclength`; profile_pos [n] are arclength positions over time (modified in place when nudge != 0, as the
 * reference's by-reference argument is).  Outputs (each may be NULL): pts [n][2], tpar [n] the curve parameter,
 * seg [n] the segment, curv [n] the signed curvature.  Returns 0, or 1 when some segment received no sample (the
 * reference indexes out of range there); outputs are then unspecified. */
int sco_bezier_resample(const float* ctrl, int nseg, int nsub, const float* cum, float arclength, float* pp, int n, int nudge,
                        float* pts, float* tpar, int32_t* seg, float* curv) {
    if (n <= 0 || nseg <= 0) return 1;
    int status = 0;
    if (nudge) {                                  /* :902-913 */
        pp[0] = 0; pp[n - 1] = arclength;
        for (int i = 1; i < n - 1; ++i) {
            if (pp[i] < pp[i - 1] || pp[i] > pp[i + 1]) pp[i] = (pp[i - 1] + pp[i + 1]) / 2;
            if (pp[i] < 0) pp[i] = 0;
            if (pp[i] > arclength) pp[i] = arclength;
        }
    }
    const int m = nsub + 1, deg = m < 10 ? m : 10;    /* :951 */
    double* y = (double*)malloc(sizeof(double) * m);
    const float prec = 1.0f / (float)nsub;
    for (int k = 0; k < m; ++k) { const float v = (float)k * prec; y[k] = v < 1.0f ? v : 1.0f; }   /* arclength_data::positions */
    int j = 0;
    float offset = 0;
    for (int i = 0; i < nseg; ++i) {
        const float* tab = cum + (size_t)i * m;
        const float last = tab[m - 1];
        const int start = j;
        for (; j < n && (pp[j] - offset <= last); ++j) {}
        if (i + 1 == nseg && i == 0) j = n;
        else if (i + 1 == nseg) j = n - 1;
        j -= 1;
        if (j < start) { status = 1; j = start; }
        offset = pp[j];
        double coef[10], xmin, xmax;
        chebfit(tab, y, m, deg, coef, &xmin, &xmax);
        for (int k = start; k <= j; ++k) {
            const int o = k + i;                   /* blocks overlap by one sample; the surplus is cut off the end (:983-993) */
            if (o >= n) break;
            const float xb = pp[k] - pp[start];
            float t = (float)chebeval((double)xb, coef, deg, xmin, xmax);
            if (t < 0) t = 0;
            if (t > 1) t = 1;
            double p[2], d1[2], d2[2];
            const double td = t;
            sco_bezier_eval(ctrl, &i, &td, 1, 0, p);
            if (pts) { pts[2 * o] = (float)p[0]; pts[2 * o + 1] = (float)p[1]; }
            if (tpar) tpar[o] = t;
            if (seg) seg[o] = i;
            if (curv) {
                sco_bezier_eval(ctrl, &i, &td, 1, 1, d1);
                sco_bezier_eval(ctrl, &i, &td, 1, 2, d2);
                curv[o] = (float)((d1[0] * d2[1] - d1[1] * d2[0]) / pow(d1[0] * d1[0] + d1[1] * d1[1], 1.5));
            }
        }
    }
    free(y);
    return status;
}

/* ---- public forms of the pieces above, for the free functions of the reference's header ---------------------- */
/* general degree curve (bezier_curve :700-763): de Casteljau in fp64; ctrl [nseg][deg+1][2] */
void sco_bezier_curve(const float* ctrl, int deg, const int* seg, const double* t, int m, double* out) {
    for (int i = 0; i < m; ++i) {
        double bx[64], by[64];
        const float* c = ctrl + (size_t)seg[i] * (deg + 1) * 2;
        for (int j = 0; j <= deg; ++j) { bx[j] = c[2 * j]; by[j] = c[2 * j + 1]; }
        const double s = t[i], r = 1.0 - s;
        for (int lvl = 1; lvl <= deg; ++lvl)
            for (int j = 0; j + lvl <= deg; ++j) { bx[j] = r * bx[j] + s * bx[j + 1]; by[j] = r * by[j] + s * by[j + 1]; }
        out[2 * i] = bx[0]; out[2 * i + 1] = by[0];
    }
}
/* chebfit / chebeval (:1109-1170) on float abscissae and ordinates; coef [degree] */
void sco_chebfit(const float* x, const float* y, int m, int degree, double* coef, double* xmin, double* xmax) {
    double* yd = (double*)malloc(sizeof(double) * (m > 0 ? m : 1));
    for (int r = 0; r < m; ++r) yd[r] = y[r];
    for (int k = 0; k < degree; ++k) coef[k] = 0;
    chebfit(x, yd, m, degree, coef, xmin, xmax);
    free(yd);
}
void sco_chebeval(const float* x, int m, int degree, const double* coef, double xmin, double xmax, double* y) {
    for (int r = 0; r < m; ++r) y[r] = chebeval((double)x[r], coef, degree, xmin, xmax);
}
