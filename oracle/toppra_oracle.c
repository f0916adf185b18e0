/*
 * toppra_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * fp64 restatement of the TOPP-RA computation that the reference performs in
 * gen_vel_prof<N> (sea_current.hpp:1191-1265) through the third-party library
 * hungpham2511/toppra (cpp/ subtree; un-vendored, version unpinned:
 * .gitmodules:4-6, CMakeLists.txt:36).  Its source is NOT in /root/reference,
 * so this follows the published algorithm (Pham & Pham, "A New Approach to
 * Time-Optimal Path Parameterization based on Reachability Analysis", T-RO
 * 2018) and the reference's call site:
 *   - path: PiecewisePolyPath::CubicHermiteSpline, knots {0,1}     (:1213-1220)
 *   - constraints: LinearJointVelocity (limits re-evaluated per gridpoint by
 *     LinearJointVelocityVarying::computeVelocityLimits, :1177-1188) and
 *     LinearJointAcceleration with DiscretizationType::Interpolation (:1209-1210)
 *   - computePathParametrization(0, 0)                              (:1225)
 *   - parametrizer::Spline + uniform sampling at dt                 (:1233-1243)
 * PINNED for dof = 1 by examples/output.json (tests/test_oracle_toppra.py);
 * dof > 1: parity unpinned.
 *
 * Stage LP (variables u = sddot, x = sdot^2), rows  alpha*u + beta*x <= gamma:
 *   accel, collocation at i      :  +-(a_i u + b_i x) <= +-alim, a_i = q'(s_i), b_i = q''(s_i)
 *   accel, interpolation (i<N)   :  +-((a_{i+1} + 2 D_i b_{i+1}) u + b_{i+1} x) <= +-alim
 *                       (i == N) :  the collocation rows repeated
 *   next controllable set        :  K_lo <= x + 2 D_i u <= K_hi
 *   velocity                     :  xlo_i <= x <= xhi_i,  xhi = min(1e8, min_k vlim_k/q'_k)^2
 * The two-variable LPs are solved exactly by eliminating u (Fourier-Motzkin).
 */
#include "sc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXSD 1e8
#define NEARLY_ZERO 1e-8
#define LP_TOL 1e-9

typedef struct { double al, be, ga; } row_t;

static void hermite_coef(int dof, const double* p0, const double* p1, const double* v0,
                         const double* v1, double* c2, double* c3) {
    for (int k = 0; k < dof; ++k) {
        double d = p1[k] - p0[k];
        c2[k] = 3.0 * d - 2.0 * v0[k] - v1[k];
        c3[k] = -2.0 * d + v0[k] + v1[k];
    }
}
static inline double hq(double p0, double v0, double c2, double c3, double s) { return p0 + s * (v0 + s * (c2 + s * c3)); }
static inline double hqs(double v0, double c2, double c3, double s) { return v0 + s * (2.0 * c2 + s * 3.0 * c3); }
static inline double hqss(double c2, double c3, double s) { return 2.0 * c2 + 6.0 * c3 * s; }

/* x-interval for which some u satisfies all rows; returns 0 if empty */
static int x_interval(const row_t* R, int nr, double* lo, double* hi) {
    double l = *lo, h = *hi;
    for (int i = 0; i < nr; ++i) {
        if (R[i].al == 0.0) {
            if (R[i].be > 0) { double v = R[i].ga / R[i].be; if (v < h) h = v; }
            else if (R[i].be < 0) { double v = R[i].ga / R[i].be; if (v > l) l = v; }
            else if (R[i].ga < -LP_TOL) return 0;
            continue;
        }
        if (R[i].al < 0) continue;
        for (int j = 0; j < nr; ++j) {
            if (!(R[j].al < 0)) continue;
            /* upper row i (al>0): u <= (ga_i - be_i x)/al_i ; lower row j (al<0): u >= (ga_j - be_j x)/al_j */
            double cf = R[j].al * R[i].be - R[i].al * R[j].be;
            double rhs = R[j].al * R[i].ga - R[i].al * R[j].ga;
            if (cf > 0) { double v = rhs / cf; if (v > l) l = v; }
            else if (cf < 0) { double v = rhs / cf; if (v < h) h = v; }
            else if (rhs > LP_TOL) return 0;
        }
    }
    *lo = l; *hi = h;
    return 1;
}

int sco_toppra(int dof, int N, const double* p0, const double* p1,
               const double* v0, const double* v1, const double* vlim_lo,
               const double* vlim_hi, const double* alim_lo,
               const double* alim_hi, double sd_start, double sd_end,
               double* K, double* x, double* u, double* t) {
    double* c2 = (double*)malloc(sizeof(double) * dof * 2);
    double* c3 = c2 + dof;
    hermite_coef(dof, p0, p1, v0, v1, c2, c3);
    double* xb = (double*)malloc(sizeof(double) * 2 * (N + 1));
    row_t* R = (row_t*)malloc(sizeof(row_t) * (4 * dof + 2));
    int status = 0;
    /* velocity constraint -> xbound (computeParams of LinearJointVelocity) */
    for (int i = 0; i <= N; ++i) {
        double s = (double)i / N, sdmin = -MAXSD, sdmax = MAXSD;
        for (int k = 0; k < dof; ++k) {
            double v = hqs(v0[k], c2[k], c3[k], s);
            double lo = vlim_lo[i * dof + k], hi = vlim_hi[i * dof + k];
            if (v > 0) { sdmax = fmin(hi / v, sdmax); sdmin = fmax(lo / v, sdmin); }
            else if (v < 0) { sdmax = fmin(lo / v, sdmax); sdmin = fmax(hi / v, sdmin); }
        }
        xb[2 * i] = sdmin > 0 ? sdmin * sdmin : 0.0;
        xb[2 * i + 1] = sdmax * sdmax;
    }
#define BUILD_ROWS(i)                                                                     \
    do {                                                                                  \
        double s_ = (double)(i) / N, s1_ = (double)((i) + 1) / N, D_ = s1_ - s_;          \
        nr = 0;                                                                           \
        for (int k = 0; k < dof; ++k) {                                                   \
            double a = hqs(v0[k], c2[k], c3[k], s_), b = hqss(c2[k], c3[k], s_);          \
            double a2 = a, b2 = b;                                                        \
            if ((i) < N) {                                                                \
                double an = hqs(v0[k], c2[k], c3[k], s1_), bn = hqss(c2[k], c3[k], s1_);  \
                a2 = an + 2.0 * D_ * bn; b2 = bn;                                         \
            }                                                                             \
            R[nr++] = (row_t){a, b, alim_hi[k]};  R[nr++] = (row_t){-a, -b, -alim_lo[k]}; \
            R[nr++] = (row_t){a2, b2, alim_hi[k]}; R[nr++] = (row_t){-a2, -b2, -alim_lo[k]}; \
        }                                                                                 \
    } while (0)
    /* backward pass: controllable sets */
    int nr;
    K[2 * N] = K[2 * N + 1] = sd_end * sd_end;
    for (int i = N - 1; i >= 0 && !status; --i) {
        double D = (double)(i + 1) / N - (double)i / N;
        BUILD_ROWS(i);
        R[nr++] = (row_t){2.0 * D, 1.0, K[2 * (i + 1) + 1]};
        R[nr++] = (row_t){-2.0 * D, -1.0, -K[2 * (i + 1)]};
        double lo = xb[2 * i], hi = xb[2 * i + 1];
        if (!x_interval(R, nr, &lo, &hi) || lo > hi + LP_TOL) { status = 1; break; }
        if (lo > hi) lo = hi;
        K[2 * i] = lo > 0 ? lo : 0.0;
        K[2 * i + 1] = hi;
    }
    /* forward pass: greedy maximal u */
    if (!status) {
        x[0] = sd_start * sd_start;
        if (x[0] < K[0] - LP_TOL || x[0] > K[1] + LP_TOL) status = 2;
    }
    for (int i = 0; i < N && !status; ++i) {
        double D = (double)(i + 1) / N - (double)i / N;
        BUILD_ROWS(i);
        R[nr++] = (row_t){2.0 * D, 1.0, K[2 * (i + 1) + 1]};
        R[nr++] = (row_t){-2.0 * D, -1.0, -K[2 * (i + 1)]};
        double umax = INFINITY, umin = -INFINITY;
        for (int r = 0; r < nr; ++r) {
            double num = R[r].ga - R[r].be * x[i];
            if (R[r].al > 0) umax = fmin(umax, num / R[r].al);
            else if (R[r].al < 0) umin = fmax(umin, num / R[r].al);
        }
        if (!(umax >= umin - 1e-6) || !isfinite(umax)) { status = 2; break; }
        u[i] = umax;
        double xn = x[i] + 2.0 * D * umax;
        if (xn > K[2 * (i + 1) + 1]) xn = K[2 * (i + 1) + 1];
        if (xn < K[2 * (i + 1)]) xn = K[2 * (i + 1)];
        x[i + 1] = xn;
    }
    /* knot times of parametrizer::Spline: dt_i = ds / mean(sd_i, sd_{i+1}) */
    if (!status && t) {
        t[0] = 0.0;
        for (int i = 1; i <= N; ++i) {
            double D = (double)i / N - (double)(i - 1) / N;
            double sda = 0.5 * (sqrt(fmax(x[i - 1], 0.0)) + sqrt(fmax(x[i], 0.0)));
            t[i] = t[i - 1] + (sda > NEARLY_ZERO ? D / sda : 5.0);
        }
    }
    free(R); free(xb); free(c2);
    return status;
}

int sco_toppra_sample(int dof, int N, const double* p0, const double* p1,
                      const double* v0, const double* v1, const double* x,
                      const double* t, double dt, int max_len, float* pos,
                      float* vel, float* acc, double* times) {
    double* c2 = (double*)malloc(sizeof(double) * dof * 2);
    double* c3 = c2 + dof;
    hermite_coef(dof, p0, p1, v0, v1, c2, c3);
    /* drop knots whose time increment is ~0 (parametrizer::Spline) */
    int* idx = (int*)malloc(sizeof(int) * (N + 1));
    int n = 0;
    for (int i = 0; i <= N; ++i)
        if (i == 0 || t[i] - t[i - 1] >= NEARLY_ZERO) idx[n++] = i;
    double* tk = (double*)malloc(sizeof(double) * n * 5);
    double *yk = tk + n, *M = yk + n, *cp = M + n, *dp = cp + n;
    for (int j = 0; j < n; ++j) tk[j] = t[idx[j]];
    double T = tk[n - 1];
    int length = (int)ceil(T / dt);
    int wl = length < max_len ? length : max_len;
    for (int j = 0; j < wl; ++j)
        times[j] = length > 1 ? (j == length - 1 ? T : (T * j) / (length - 1)) : 0.0;
    for (int k = 0; k < dof; ++k) {
        for (int j = 0; j < n; ++j) yk[j] = hq(p0[k], v0[k], c2[k], c3[k], (double)idx[j] / N);
        double d0 = hqs(v0[k], c2[k], c3[k], 0.0) * sqrt(fmax(x[0], 0.0));
        double d1 = hqs(v0[k], c2[k], c3[k], 1.0) * sqrt(fmax(x[N], 0.0));
        /* clamped cubic spline: tridiagonal system for second derivatives M */
        if (n == 1) M[0] = 0;
        else {
            /* row j: lo_j M_{j-1} + di_j M_j + up_j M_{j+1} = rhs_j ; Thomas */
            for (int j = 0; j < n; ++j) {
                double lo, di, up, rhs;
                if (j == 0) {
                    double h = tk[1] - tk[0];
                    lo = 0; di = 2 * h; up = h; rhs = 6 * ((yk[1] - yk[0]) / h - d0);
                } else if (j == n - 1) {
                    double h = tk[j] - tk[j - 1];
                    lo = h; di = 2 * h; up = 0; rhs = 6 * (d1 - (yk[j] - yk[j - 1]) / h);
                } else {
                    double h0 = tk[j] - tk[j - 1], h1 = tk[j + 1] - tk[j];
                    lo = h0; di = 2 * (h0 + h1); up = h1;
                    rhs = 6 * ((yk[j + 1] - yk[j]) / h1 - (yk[j] - yk[j - 1]) / h0);
                }
                if (j == 0) { cp[0] = up / di; dp[0] = rhs / di; }
                else {
                    double m = di - lo * cp[j - 1];
                    cp[j] = up / m; dp[j] = (rhs - lo * dp[j - 1]) / m;
                }
            }
            M[n - 1] = dp[n - 1];
            for (int j = n - 2; j >= 0; --j) M[j] = dp[j] - cp[j] * M[j + 1];
        }
        int seg = 0;
        for (int j = 0; j < wl; ++j) {
            double tt = times[j];
            while (seg < n - 2 && tt > tk[seg + 1]) ++seg;
            double P, V, A;
            if (n == 1) { P = yk[0]; V = 0; A = 0; }
            else {
                double h = tk[seg + 1] - tk[seg], a = tk[seg + 1] - tt, b = tt - tk[seg];
                double ca = yk[seg] / h - M[seg] * h / 6, cb = yk[seg + 1] / h - M[seg + 1] * h / 6;
                P = M[seg] * a * a * a / (6 * h) + M[seg + 1] * b * b * b / (6 * h) + ca * a + cb * b;
                V = -M[seg] * a * a / (2 * h) + M[seg + 1] * b * b / (2 * h) - ca + cb;
                A = M[seg] * a / h + M[seg + 1] * b / h;
            }
            pos[(size_t)k * max_len + j] = (float)P;
            vel[(size_t)k * max_len + j] = (float)V;
            acc[(size_t)k * max_len + j] = (float)A;
        }
    }
    free(tk); free(idx); free(c2);
    return length;
}
