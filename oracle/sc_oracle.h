/*
 * sc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the semantics the HIP kernels in sea-current_amd/csrc
 * are graded against.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path never does.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - EDT, A*: the reference (/root/reference/sea_current.hpp) has NO grid EDT
 *     and NO A* (its planner is FMT* over polygon obstacles, sea_current.hpp:
 *     1339-1407; collision by ray casting, :201-251).  These two oracles are
 *     DEFINITIONAL: exact squared Euclidean distance (unique), and optimal
 *     integer-cost A* with the canonical g-field / parent rule written below.
 *     "parity unpinned" versus the reference -- there is nothing to pin to.
 *   - TOPP-RA: restates hungpham2511/toppra (cpp/, un-vendored and unpinned in
 *     the reference: .gitmodules:4-6) as called by gen_vel_prof<N>
 *     (sea_current.hpp:1191-1265).  PINNED for dof=1 by the reference's only
 *     recorded output, examples/output.json (tests/golden/toppra_1dof_*.npz):
 *     duration, vel, acc reproduce to float32 round-off.  dof>1 / varying
 *     limits: parity unpinned.
 */
#ifndef SC_ORACLE_H
#define SC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SCO_EDT_INF INT32_MAX /* d2 of every cell when the grid has no obstacle */

/* ---- EDT -------------------------------------------------------------- */
/* occ[y*W+x] != 0 means occupied.  d2[y*W+x] = min over occupied (x',y') of
 * (x-x')^2 + (y-y')^2, exact, int32; SCO_EDT_INF if no cell is occupied. */
void sco_edt_brute(const uint8_t* occ, int W, int H, int32_t* d2);
/* Same result by the exact separable algorithm (column scan + lower envelope
 * of parabolas, all-integer comparisons).  O(W*H). */
void sco_edt_exact(const uint8_t* occ, int W, int H, int32_t* d2);

/* nearest[y*W+x] = linear index of the nearest occupied cell (smallest index among equidistant ones), -1 if the grid is
 * empty.  _brute is the definition; sco_edt_nearest walks the integer points of the circle of radius^2 d2. */
void sco_edt_nearest_brute(const uint8_t* occ, int W, int H, int32_t* nearest);
void sco_edt_nearest(const uint8_t* occ, const int32_t* d2, int W, int H, int32_t* nearest);

/* ---- A* --------------------------------------------------------------- */
/* Move d in 0..7: dx = {1,-1,0,0,1,-1,1,-1}, dy = {0,0,1,-1,1,1,-1,-1};
 * cost 10 for d<4, 14 for d>=4.  Cell c traversable iff d2[c] >= max(r2,1).
 * A diagonal move needs the target and BOTH orthogonal side cells traversable
 * (no corner cutting).  h(c) = 10*max(|dx|,|dy|) + 4*min(|dx|,|dy|) to goal.
 *
 * Canonical result (order independent):
 *   C* = optimal cost.  E = { n : g*(n) + h(n) <= C* }  (every node A* may
 *   expand, i.e. the search runs until the f = C* plateau is exhausted).
 *   g[n] = g*(n) for n in E; for n not in E: min over p in E, move p->n legal,
 *   of g*(p) + w; SCO_G_INF if none.  No path: E = the whole component.
 *   parent(c) = smallest d such that n = c - (dx_d,dy_d) is traversable, the
 *   move n->c is legal and g[n] + w_d == g[c].
 *   path = start..goal obtained by following parent() from goal.
 * Parity contract of the HIP kernel: status, cost, len, path, and g on E are
 * bit-exact; outside E the kernel's successor pruning leaves values >= the g
 * defined above (tests/test_gpu_astar.py::test_astar_gfield_bit_exact).
 */
#define SCO_G_INF 0xFFFFFFFFu
enum { SCO_OK = 0, SCO_NO_PATH = 1, SCO_BAD_ENDPOINT = 2, SCO_PATH_TRUNCATED = 3 };

/* moves[c]: bit d set iff move d out of c is legal (0 for blocked cells). */
void sco_moves(const int32_t* d2, int W, int H, int32_t r2, uint8_t* moves);

/* One query.  gfield (W*H uint32, may be NULL) receives the canonical g field.
 * path (Lmax int32, may be NULL if Lmax==0) receives start..goal cell indices;
 * *len = number of cells on the path even when it exceeds Lmax (status 3).
 * *expanded (may be NULL) = |E| (number of expansions).  Returns status. */
int sco_astar(const int32_t* d2, int W, int H, int32_t r2, int32_t start,
              int32_t goal, int Lmax, int32_t* path, int32_t* len,
              int32_t* cost, uint32_t* gfield, int64_t* expanded);

/* Batch: loops sco_astar; path is [Q][Lmax].  nthreads<=1: serial. */
void sco_astar_batch(const int32_t* d2, int W, int H, int32_t r2,
                     const int32_t* start, const int32_t* goal, int Q, int Lmax,
                     int32_t* path, int32_t* len, int32_t* cost,
                     int32_t* status, int64_t* expanded, int nthreads);

/* ---- TOPP-RA ---------------------------------------------------------- */
/* Path: 2-knot cubic Hermite on s in [0,1] (sea_current.hpp:1213-1220):
 *   q(s) = p0 + v0 s + (3(p1-p0) - 2 v0 - v1) s^2 + (-2(p1-p0) + v0 + v1) s^3
 * Grid: N uniform intervals (toppra default N = 100, confirmed by the fixture).
 * vlim_lo/hi: [N+1][dof] velocity limits per gridpoint (the reference evaluates
 * its vel_lim_func at the gridpoint value, sea_current.hpp:1185-1186);
 * alim_lo/hi: [dof].  sd_start = sd_end = 0 in the reference (:1225).
 * Outputs: K [N+1][2] controllable sets, x [N+1] = sdot^2, u [N] = sddot,
 * t [N+1] knot times of the spline parametrizer.  Returns 0 ok, 1 controllable
 * set empty (backward pass failed), 2 forward pass failed. */
int sco_toppra(int dof, int N, const double* p0, const double* p1,
               const double* v0, const double* v1, const double* vlim_lo,
               const double* vlim_hi, const double* alim_lo,
               const double* alim_hi, double sd_start, double sd_end,
               double* K, double* x, double* u, double* t);

/* Spline parametrizer + uniform sampling (sea_current.hpp:1233-1262):
 * clamped cubic spline through (t_i, q(s_i)) with end slopes q'(s)*sdot,
 * length = ceil(T/dt) samples at linspace(0,T,length).  pos/vel/acc are
 * [dof][max_len] float (the reference casts to VectorXf), times [max_len]
 * double.  Returns length (may exceed max_len; only max_len are written). */
int sco_toppra_sample(int dof, int N, const double* p0, const double* p1,
                      const double* v0, const double* v1, const double* x,
                      const double* t, double dt, int max_len, float* pos,
                      float* vel, float* acc, double* times);

/* ---- Bezier smoothing + arclength ------------------------ */
/* from_path (sea_current.hpp:599-683): n waypoints (x,y) -> ctrl [n-1][4][2]; lines [nlines][4] obstacle edges
 * for shrink_tangent (:575-596); start_angle NaN = along the first leg. */
void sco_bezier_from_path(const float* path, int n, float start_angle, const float* lines, int nlines, float* ctrl);
/* bezier_spline::shrink_tangent (:575-596) on its own: out[i] = k * T[i] cut where Wp[i] +- that tangent crosses an edge
 * (edges in order, the shortened tangent carried along).  No recorded output: checked against closed forms. */
void sco_bezier_shrink_tangent(const float* T, const float* Wp, int M, float k, const float* lines, int nlines, float* out);
/* order 0: point, 1: hodograph (:1041-1053), 2: second derivative; out [m][2] */
void sco_bezier_eval(const float* ctrl, const int* seg, const double* t, int m, int order, double* out);
/* arclength (:767-896): GL-32 on nsub sub-intervals per segment; cum [nseg][nsub+1]; returns the total */
double sco_bezier_arclength(const float* ctrl, int nseg, int nsub, double* cum);
/* resample (:898-1005; chebfit/chebeval :1109-1170; curvature :1017-1039): map arclength positions over time
 * (profile_pos [n], nudged in place when nudge != 0) to curve points.  cum: float tables [nseg][nsub+1].
 * pts [n][2], tpar [n], seg [n], curv [n] (each may be NULL).  Returns 0, or 1 if a segment got no sample. */
int sco_bezier_resample(const float* ctrl, int nseg, int nsub, const float* cum, float arclength, float* profile_pos, int n,
                        int nudge, float* pts, float* tpar, int32_t* seg, float* curv);

/* general-degree curve (bezier_curve :700-763; de Casteljau, ctrl [nseg][deg+1][2]) and the free chebfit / chebeval
 * (:1109-1170: `degree` Chebyshev columns of x normalised by its own range, Householder least squares in fp64) */
void sco_bezier_curve(const float* ctrl, int deg, const int* seg, const double* t, int m, double* out);
void sco_chebfit(const float* x, const float* y, int m, int degree, double* coef, double* xmin, double* xmax);
void sco_chebeval(const float* x, int m, int degree, const double* coef, double xmin, double xmax, double* y);

/* ---- the reference's own planner: FMT* over Halton samples --------------------------------
 * Restates sea_current.hpp:100-132 (halton), :1294-1313 (sample_free), :1328-1337 (near), :1315-1326 + :142-178 (cost),
 * :1339-1407 (fast_marching_trees); see fmt_oracle.c for the tie-break and free-space conventions.  PARITY UNPINNED.
 * lines [E][4] obstacle edges, obs_off [nobs+1] edge ranges per obstacle, rect = (x_min, x_max, y_min, y_max),
 * hstate = (f2, i2, f3, i3) Halton states of bases 2 and 3 (advanced). */
void sco_halton(int b, int n, int* f_state, int* i_state, float* out);
int sco_point_in_obstacles(float px, float py, const float* lines, const int* obs_off, int nobs);
int sco_sample_free(int n, const float* rect, const float* lines, const int* obs_off, int nobs, int* hstate, float* pts);
int sco_fmt_star(const float* samples, int n, float sx, float sy, float gx, float gy, float rn, const float* lines, int E, int Lmax,
                 float* path, int32_t* len, float* cost_out);

#ifdef __cplusplus
}
#endif
#endif
