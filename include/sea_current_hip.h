/*
 * sea_current_hip.h -- C ABI of libsea_current_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for sea-current's planning hot path.  The
 * reference (turtle-robotics/sea-current @ 2024-10-16) is a header-only C++20
 * library with NO FFI of its own; the boundary it exposes is the `turtle::sc`
 * API of sea_current.hpp.  Each entry point below names the reference
 * interface whose work it takes over (file:line relative to the reference
 * root).  The C++ successor header sea-current_amd/sea_current.hpp keeps the
 * reference signatures and calls these functions; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types.
 *   - functions without a suffix take DEVICE pointers and enqueue work on the
 *     context's stream without synchronising (inputs already resident in HBM);
 *     `_host` variants take host pointers, copy, run, copy back and synchronise.
 *   - every function returns an sc_status (0 = ok) and never exits the process
 *     (the reference's SC_ASSERT calls std::exit(1), sea_current.hpp:34-52).
 *   - a context is bound to one GPU and one stream; calls on one context are
 *     serialised on its stream, contexts are independent (one per host
 *     thread / rank).  No allocation happens in steady state: scratch is owned
 *     by the context and only grows.
 */
#ifndef SEA_CURRENT_HIP_H
#define SEA_CURRENT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SC_ABI_VERSION 1

typedef struct sc_ctx sc_ctx;

typedef enum {
    SC_OK = 0,
    SC_ERR_INVALID = 1,  /* bad argument (null pointer, non-positive size, W or H too large) */
    SC_ERR_HIP = 2,      /* a HIP runtime call failed; see sc_last_error */
    SC_ERR_NOMEM = 3,    /* device allocation failed */
    SC_ERR_NO_DEVICE = 4 /* no usable gfx950 device */
} sc_status;

/* per-query status written by sc_astar_batch (mirrors the reference planner's
 * std::optional result: nullopt == SC_Q_NO_PATH, sea_current.hpp:1383-1385) */
typedef enum {
    SC_Q_OK = 0,
    SC_Q_NO_PATH = 1,
    SC_Q_BAD_ENDPOINT = 2, /* start/goal out of range or not traversable */
    SC_Q_TRUNCATED = 3,    /* path longer than Lmax: len holds the needed length, path is unspecified */
    SC_Q_RING_OVERFLOW = 4 /* the search's frontier outgrew the device queues twice (16x the usual space on the second
                            * attempt); no result for this query.  sc_ctx_synchronize enlarges the queues of later calls */
} sc_query_status;

/* kernels timed by sc_ctx_set_timing (indices for sc_ctx_get_timing) */
typedef enum {
    SC_K_EDT_COLBITS = 0, /* occupancy bytes -> transposed per-band column bit words */
    SC_K_EDT_BAND = 1,    /* per 32-row band: vertical distances + exact row envelope -> d2 */
    SC_K_MOVES = 2,       /* d2 + clearance -> legal-move byte per cell */
    SC_K_ASTAR = 3,       /* batched A*: two wavefronts per query (prep + search + retry launches) */
    SC_K_TOPPRA = 4,      /* batched TOPP-RA: computeParams + backward + forward sweep */
    SC_K_TOPPRA_SAMPLE = 5,
    SC_K_BEZIER = 6,      /* tangents + control points, curve evaluation */
    SC_K_ARCLENGTH = 7,   /* GL-32 arclength tables */
    SC_K_RESAMPLE = 8,    /* resample: nudge + split, Chebyshev fit + evaluation */
    SC_K_OCC = 9,         /* occupancy grid from a rectangle list (dynamic-obstacle frames) */
    SC_K_NEAREST = 10,    /* nearest obstacle cell from d2 */
    SC_K_FMT = 11,        /* FMT* over Halton samples (the reference's own planner), one wavefront per query */
    SC_K_GATHER = 12,     /* gather of result paths: pack, ncclAllGather, unpack */
    SC_K_COUNT = 13
} sc_kernel_id;

#define SC_EDT_INF INT32_MAX /* d2 of every cell of a grid without obstacles */
#define SC_MAX_DIM 8192      /* W, H <= SC_MAX_DIM (LDS row buffers) */

int sc_abi_version(void);
const char* sc_status_string(int status);
/* message of the last failing HIP call on this context ("" if none) */
const char* sc_last_error(const sc_ctx* ctx);

/* ---- context ----------------------------------------------------------- */
int sc_ctx_create(int device, sc_ctx** out);
int sc_ctx_destroy(sc_ctx* ctx);
/* Use the caller's hipStream_t (e.g. torch's current stream) instead of the
 * context's own; NULL means HIP's null (legacy default) stream. */
int sc_ctx_set_stream(sc_ctx* ctx, void* hip_stream);
/* Go back to the context's own non-blocking stream (the default after create). */
int sc_ctx_use_own_stream(sc_ctx* ctx);
/* Wait for everything enqueued on the context's stream.  Also where a context adapts to what its launches met: A* rings
 * that overflowed (later calls start with larger ones), maps of open space at widths of 513 .. 1024 (later EDTs run the
 * band kernel's build with the site search, until a launch meets none), and where a wide-row EDT whose bounded waits ran
 * out is reported (SC_ERR_HIP).  Results never depend on the adapted state. */
int sc_ctx_synchronize(sc_ctx* ctx);
/* Kernel timing: when enabled every kernel launch is bracketed by HIP events on
 * the context's stream; sc_ctx_get_timing synchronises and returns the summed
 * device time and the number of launches since the last reset. */
int sc_ctx_set_timing(sc_ctx* ctx, int enable);
int sc_ctx_reset_timing(sc_ctx* ctx);
int sc_ctx_get_timing(sc_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);
/* Bytes of device scratch currently owned by the context. */
int sc_ctx_scratch_bytes(sc_ctx* ctx, int64_t* bytes);

/* ---- EDT ---------------------------------------------------------------
 * Exact squared Euclidean distance transform of `batch` occupancy grids.
 *   occ : uint8 [batch][H][W] row-major, != 0 means occupied
 *   d2  : int32 [batch][H][W]; SC_EDT_INF everywhere if a grid has no obstacle
 * Takes over the collision/clearance queries the reference answers by ray
 * casting and segment tests: obstacle::contains (sea_current.hpp:201-251),
 * planning_space::is_obstacle (:1274-1280), ::is_free (:1289-1292), ::cost
 * (:1315-1326).  The reference has no grid; the result is defined
 * mathematically (oracle/sc_oracle.h). */
int sc_edt_u8_i32(sc_ctx* ctx, const uint8_t* occ, int W, int H, int batch, int32_t* d2);
int sc_edt_u8_i32_host(sc_ctx* ctx, const uint8_t* occ, int W, int H, int batch, int32_t* d2);

/* Optional second output of the EDT: nearest[b][y][x] = linear index (y' * W + x') of the occupied cell
 * nearest to (x, y) in grid b, the smallest index among equidistant ones, -1 when the grid has no obstacle.  Needs
 * the d2 of sc_edt_u8_i32 for the same occ.  9 B/cell with d2 instead of 5.  Device pointers. */
int sc_edt_nearest_i32(sc_ctx* ctx, const uint8_t* occ, const int32_t* d2, int W, int H, int batch, int32_t* nearest);

/* Occupancy grid of a frame of the dynamic-obstacle replan loop (BASELINE.json configs[4]): occ = base (or all free
 * when base is NULL; base == occ paints in place) with R cell rectangles (x0, y0, x1, y1; x1/y1 exclusive; clipped)
 * painted as occupied; free_border != 0 keeps the outermost ring of cells free.  Stands where the
 * reference edits planning_space::obstacles (sea_current.hpp:314) between plans; the EDT is then recomputed in full
 * by sc_edt_u8_i32 (exact, ~10 us at 1024^2).  Device pointers; rects int32 [R][4]. */
int sc_occ_from_rects(sc_ctx* ctx, const uint8_t* base, const int32_t* rects, int R, int W, int H, int free_border, uint8_t* occ);

/* Legal-move mask per cell: bit d set iff move d (dx={1,-1,0,0,1,-1,1,-1},
 * dy={0,0,1,-1,1,1,-1,-1}) out of the cell is allowed: both cells have
 * d2 >= max(r2_clear,1) and, for diagonals, both side cells too. */
int sc_moves_i32_u8(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2_clear, uint8_t* moves);

/* ---- batched A* --------------------------------------------------------
 * Q independent start->goal queries on one grid; 8-connected, costs 10/14,
 * octile heuristic, no corner cutting, canonical parent rule (sc_oracle.h).
 *   d2          : int32 [H][W] from sc_edt_u8_i32
 *   start, goal : int32 [Q] linear cell indices (y*W + x)
 *   path        : int32 [Q][Lmax], cells start..goal in path[q][0..len[q])
 *   len, cost, status : int32 [Q]  (cost = -1, len = 0 when there is no path)
 * Takes over planning_space::fast_marching_trees (sea_current.hpp:1339-1407)
 * with its helpers near (:1328-1337) and sample_free (:1294-1313): same role
 * (start, goal) -> optional waypoint list, on a grid instead of Halton samples.
 * Scratch: every search in flight owns W*H bytes of g, W*H/8 of closed bits and its open-list rings (9 MiB at 1024^2,
 * 50 MiB at 4096^2); a context takes at most 96 GiB for them (environment SC_ASTAR_SLOT_GB when the context is created)
 * and serves longer query lists from the same slots.  Only enqueues: a full ring is handled on the device (a second,
 * normally empty, launch with 16x the ring space); what that cannot hold either reports SC_Q_RING_OVERFLOW. */
int sc_astar_batch(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2_clear,
                   const int32_t* start, const int32_t* goal, int Q, int Lmax,
                   int32_t* path, int32_t* len, int32_t* cost, int32_t* status);
int sc_astar_batch_host(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2_clear,
                        const int32_t* start, const int32_t* goal, int Q, int Lmax,
                        int32_t* path, int32_t* len, int32_t* cost, int32_t* status);
/* Several grids in one launch: d2 int32 [G][H][W] (e.g. from sc_edt_u8_i32 with batch = G), qgrid int32 [Q] the grid of
 * every query.  Independent problems (several maps, several robots' local maps, consecutive frames) then share ONE
 * launch and its single tail: the persistent wavefronts pull queries of all grids from one queue, longest first. */
int sc_astar_batch_multi(sc_ctx* ctx, const int32_t* d2, int G, const int32_t* qgrid, int W, int H, int32_t r2_clear,
                         const int32_t* start, const int32_t* goal, int Q, int Lmax,
                         int32_t* path, int32_t* len, int32_t* cost, int32_t* status);
/* Node expansions of the last sc_astar_batch call on this context (synchronises). */
int sc_astar_last_expansions(sc_ctx* ctx, int64_t* expansions);
/* Debug: per-query expansions, popped queue entries, kilo-cycles and steps (int32 [4][Q]) of the last sc_astar_batch
 * (synchronises).  sc_astar_debug_peek: the launch counters, read without waiting for the stream (bring-up aid). */
int sc_astar_debug_stats(sc_ctx* ctx, int32_t* stats4, int Q);
int sc_astar_debug_peek(sc_ctx* ctx, int32_t* out16);
/* Debug/parity: g field of ONE query (uint32 [H][W]): the optimal cost-to-come g* of every node the search
 * expanded -- E = {n : g*(n) + h(n) <= C*}, all that paths and parents are read from -- and 0xFFFFFFFF elsewhere. */
int sc_astar_gfield(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2_clear,
                    int32_t start, int32_t goal, uint32_t* gfield, int32_t* cost, int32_t* status);

/* ---- batched TOPP-RA ---------------------------------------------------
 * P independent plans; path of plan p is the 2-knot cubic Hermite spline the
 * reference builds in gen_vel_prof<N> (sea_current.hpp:1213-1220) from
 * p0,p1 (positions) and v0,v1 (path tangents), all [P][dof] fp64.
 *   vlim_lo/hi : [P][dof] if vlim_per_stage == 0, else [P][N+1][dof] (limits
 *                evaluated at each gridpoint, LinearJointVelocityVarying,
 *                sea_current.hpp:1177-1188)
 *   alim_lo/hi : [P][dof]
 *   K [P][N+1][2], x [P][N+1] (= sdot^2), u [P][N], t [P][N+1], status [P]
 *   status: 0 ok, 1 backward pass infeasible, 2 forward pass infeasible
 * One kernel fuses LinearConstraint::computeParams of both constraints with
 * the backward (controllable sets) and forward sweeps of
 * TOPPRA::computePathParametrization(0,0) (:1224-1225) and the knot times of
 * parametrizer::Spline (:1233).  dof <= 16. */
int sc_toppra_hermite_batch(sc_ctx* ctx, int P, int dof, int N,
                            const double* p0, const double* p1, const double* v0, const double* v1,
                            const double* vlim_lo, const double* vlim_hi, int vlim_per_stage,
                            const double* alim_lo, const double* alim_hi,
                            double sd_start, double sd_end,
                            double* K, double* x, double* u, double* t, int32_t* status);
int sc_toppra_hermite_batch_host(sc_ctx* ctx, int P, int dof, int N,
                                 const double* p0, const double* p1, const double* v0, const double* v1,
                                 const double* vlim_lo, const double* vlim_hi, int vlim_per_stage,
                                 const double* alim_lo, const double* alim_hi,
                                 double sd_start, double sd_end,
                                 double* K, double* x, double* u, double* t, int32_t* status);
/* Spline parametrizer + uniform sampling (sea_current.hpp:1233-1262):
 *   length[p] = ceil(T_p/dt); samples at linspace(0, T_p, length[p]);
 *   pos/vel/acc float32 [P][dof][max_len] (the reference casts to VectorXf),
 *   times fp64 [P][max_len].  Only min(length, max_len) samples are written. */
int sc_toppra_sample_batch(sc_ctx* ctx, int P, int dof, int N,
                           const double* p0, const double* p1, const double* v0, const double* v1,
                           const double* x, const double* t, double dt, int max_len,
                           float* pos, float* vel, float* acc, double* times, int32_t* length);
int sc_toppra_sample_batch_host(sc_ctx* ctx, int P, int dof, int N,
                                const double* p0, const double* p1, const double* v0, const double* v1,
                                const double* x, const double* t, double dt, int max_len,
                                float* pos, float* vel, float* acc, double* times, int32_t* length);

/* ---- path smoothing and arclength ----------------
 * P paths of up to n_max waypoints (float [P][n_max][2], npts [P] valid counts).
 * ctrl: float [P][n_max-1][4][2], the cubic Bezier of every leg (zeros past a
 * path's last leg).  lines: float [nlines][4] obstacle edges (x0,y0,x1,y1) used
 * to shrink tangents; start_angle NaN = along the first leg.  Takes over
 * bezier_spline::from_path (sea_current.hpp:599-683) with calc_start_tangent /
 * calc_tangent / calc_end_tangent (:343-377) and shrink_tangent (:575-596). */
int sc_bezier_from_path_batch(sc_ctx* ctx, const float* path, const int32_t* npts, int P, int n_max, float start_angle,
                              const float* lines, int nlines, float* ctrl);
int sc_bezier_from_path_batch_host(sc_ctx* ctx, const float* path, const int32_t* npts, int P, int n_max, float start_angle,
                                   const float* lines, int nlines, float* ctrl);
/* bezier_spline::shrink_tangent (sea_current.hpp:575-596) on its own: tangent i becomes k * T[i], cut where the stretch
 * Wp[i] +- k T[i] crosses an obstacle edge (edges in order, the shortened tangent carried from edge to edge as the
 * reference does).  T, Wp, out float [M][2]; lines float [nlines][4]. */
int sc_bezier_shrink_tangent_batch(sc_ctx* ctx, const float* T, const float* Wp, int M, float k, const float* lines, int nlines, float* out);
int sc_bezier_shrink_tangent_batch_host(sc_ctx* ctx, const float* T, const float* Wp, int M, float k, const float* lines, int nlines, float* out);
/* Point (order 0), hodograph (1) or second derivative (2) of segment seg[i] (index
 * into ctrl viewed as [S][4][2]) at parameter t[i]; out float [M][2].  Takes over
 * bezier_spline::bezier_curve (:700-763) and ::hodograph (:1041-1053). */
int sc_bezier_eval_batch(sc_ctx* ctx, const float* ctrl, const int32_t* seg, const float* t, int M, int order, float* out);
/* host-pointer form; S = number of segments in ctrl */
int sc_bezier_eval_batch_host(sc_ctx* ctx, const float* ctrl, int S, const int32_t* seg, const float* t, int M, int order, float* out);
/* General degree (1 .. SC_BEZIER_MAX_DEGREE): point of segment seg[i] of ctrl viewed as [S][degree+1][2] at parameter
 * t[i].  Takes over bezier_spline::bezier_curve for any control polygon (:700-763) and, fed with the derivative control
 * points degree * (P[j+1] - P[j]), ::hodograph (:1041-1053) and the hodograph of a hodograph (::curvature :1017-1039). */
#define SC_BEZIER_MAX_DEGREE 15
int sc_bezier_curve_batch(sc_ctx* ctx, const float* ctrl, int degree, const int32_t* seg, const float* t, int M, float* out);
int sc_bezier_curve_batch_host(sc_ctx* ctx, const float* ctrl, int S, int degree, const int32_t* seg, const float* t, int M, float* out);
/* The free functions chebfit / chebeval (:1109-1170): B independent least-squares fits y(x) over the Chebyshev columns
 * T_0 .. T_{degree-1} of x normalised to [-1, 1] by its own range (the reference builds `degree` columns).  Problem b owns
 * rows off[b] .. off[b+1]-1 of x / y (total = off[B] rows); coef float [B][degree], xrange float [B][2] = (xmin, xmax) --
 * the three members of the reference's chebpoly.  chebeval evaluates problem b's polynomial at its rows of x. */
#define SC_CHEB_MAX_DEGREE 32
int sc_chebfit_batch(sc_ctx* ctx, const float* x, const float* y, const int32_t* off, int B, int total, int degree, float* coef, float* xrange);
int sc_chebfit_batch_host(sc_ctx* ctx, const float* x, const float* y, const int32_t* off, int B, int degree, float* coef, float* xrange);
int sc_chebeval_batch(sc_ctx* ctx, const float* x, const int32_t* off, int B, int degree, const float* coef, const float* xrange, float* y);
int sc_chebeval_batch_host(sc_ctx* ctx, const float* x, const int32_t* off, int B, int degree, const float* coef, const float* xrange, float* y);
/* bezier_spline::arclength (:767-896): 32-point Gauss-Legendre on nsub (= 1/precision)
 * sub-intervals of every segment.  cum float [S][nsub+1] cumulative length at
 * t = k/nsub, seg_len float [S] (the reference's arclength_data: segments, and
 * arclength = sum of seg_len over a path). */
int sc_bezier_arclength_batch(sc_ctx* ctx, const float* ctrl, int S, int nsub, float* cum, float* seg_len);
int sc_bezier_arclength_batch_host(sc_ctx* ctx, const float* ctrl, int S, int nsub, float* cum, float* seg_len);
/* bezier_spline::resample (:898-1005) with chebfit / chebeval (:1109-1170) and ::curvature (:1017-1039): map the
 * arclength positions of a velocity profile back onto the curve.  B splines; spline b owns segments
 * seg_off[b] .. seg_off[b+1]-1 (S = seg_off[B] in total) of ctrl [S][4][2] and of the tables cum [S][nsub+1]
 * (sc_bezier_arclength_batch), has total length arclength[b], and samples prof_off[b] .. prof_off[b+1]-1 of
 * profile_pos (M = prof_off[B] in total).  nudge != 0 first repairs profile_pos IN PLACE the way the reference does
 * (:902-913: ends pinned to 0 / arclength, non-monotone samples averaged, clamped).  Per sample: pts float [M][2],
 * tpar float [M] (curve parameter), seg int32 [M] (segment within the spline), curvature float [M] -- each may be
 * NULL.  status int32 [B]: 0, or 1 when a segment received no sample (the reference indexes out of range there; the
 * spline's outputs are then unspecified).  nsub <= SC_RESAMPLE_MAX_NSUB. */
#define SC_RESAMPLE_MAX_NSUB 512
int sc_bezier_resample_batch(sc_ctx* ctx, const float* ctrl, const float* cum, const float* arclength, const int32_t* seg_off,
                             int B, int S, int nsub, float* profile_pos, const int32_t* prof_off, int nudge, float* pts,
                             float* tpar, int32_t* seg, float* curvature, int32_t* status);
int sc_bezier_resample_batch_host(sc_ctx* ctx, const float* ctrl, const float* cum, const float* arclength, const int32_t* seg_off,
                                  int B, int S, int nsub, float* profile_pos, const int32_t* prof_off, int nudge, float* pts,
                                  float* tpar, int32_t* seg, float* curvature, int32_t* status);

/* ---- the reference's own planner, batched -----------------------------------------------
 * planning_space::fast_marching_trees (sea_current.hpp:1339-1407) with near (:1328-1337), cost (:1315-1326) and
 * intersects (:142-178): FMT* from starts[q] to goals[q] (float [Q][2]) over n shared free samples (float [n][2], e.g.
 * from sample_free :1294-1313; the reference draws them inside the call) with connection radius rn (the reference
 * compares distances with rn squared; so does this) around obstacle edges lines (float [E][4]).  path float
 * [Q][Lmax][2] start..goal, len [Q], cost float [Q] (cost-to-come of the goal), status [Q] (SC_Q_OK / SC_Q_NO_PATH /
 * SC_Q_TRUNCATED).  Equal costs are resolved towards the lowest node index (samples in order, then goal, then start);
 * the reference resolves them by unordered_set iteration order. */
#define SC_FMT_MAX_SAMPLES 2046
#define SC_FMT_MAX_EDGES 512
int sc_fmt_star_batch(sc_ctx* ctx, const float* samples, int n, const float* starts, const float* goals, int Q, float rn,
                      const float* lines, int E, int Lmax, float* path, int32_t* len, float* cost, int32_t* status);
int sc_fmt_star_batch_host(sc_ctx* ctx, const float* samples, int n, const float* starts, const float* goals, int Q, float rn,
                           const float* lines, int E, int Lmax, float* path, int32_t* len, float* cost, int32_t* status);

/* ---- multi-GPU: query sharding and the gather of result paths --------------------------------
 * One process (context) per GPU.  Queries shard in contiguous blocks: rank r of `world` owns [q0, q1) as sc_rank_range
 * says, plans them with sc_astar_batch on its own replica of the grid (every rank recomputes the EDT: cheaper than
 * moving 4 B/cell), and sc_allgather_paths leaves EVERY rank with every query's result, in query order.  The reference
 * has no collective of any kind (its only transport is the ZMQ REP loop, examples/zmq_test.cpp:18-22); this is the
 * exchange BASELINE.json names ("RCCL all-gather of result paths over xGMI").
 *
 * Communicator: sc_comm_unique_id on one rank, the 128 bytes handed to all ranks by whatever launched them, then
 * sc_comm_init on each (ncclCommInitRank); or sc_comm_adopt of an ncclComm_t the caller already has.  RCCL is loaded
 * at run time (dlopen), so single-GPU users need no librccl.
 *
 * sc_allgather_paths (device pointers, enqueued on the context's stream, no host synchronisation):
 *   in   path [Q_local][Lmax], len / cost / status [Q_local]   this rank's sc_astar_batch results
 *   out  len_all / cost_all / status_all [Q_total], offsets_all int64 [Q_total + 1] (cells in front of each query;
 *        only paths with status SC_Q_OK count), cells_all [cells_capacity] the paths back to back (may be NULL),
 *        path_all [Q_total][Lmax] the fixed-stride parity layout (may be NULL), *truncated != 0 if some rank's paths
 *        exceeded cap_cells (bit 0: cells beyond it read -1; gather again with a larger cap_cells) or cells_all is too
 *        small (bit 1).
 *   cap_cells: cells a rank's message can carry (the same on all ranks).  One ncclAllGather of
 *        sc_gather_msg_words(Q_total, world, cap_cells) = 2 + 3 ceil(Q_total / world) + cap_cells (rounded up to even)
 *        int32 per rank. */
void sc_rank_range(int Q, int world, int rank, int* q0, int* q1);
int sc_comm_unique_id(void* id128);
int sc_comm_init(sc_ctx* ctx, const void* id128, int nranks, int rank);
int sc_comm_adopt(sc_ctx* ctx, void* nccl_comm, int nranks, int rank);
int sc_comm_destroy(sc_ctx* ctx);
int sc_allgather_paths(sc_ctx* ctx, const int32_t* path, const int32_t* len, const int32_t* cost, const int32_t* status,
                       int Q_local, int Q_total, int Lmax, int cap_cells, int32_t* len_all, int32_t* cost_all, int32_t* status_all,
                       int64_t* offsets_all, int32_t* cells_all, int64_t cells_capacity, int32_t* path_all, int32_t* truncated);
/* bytes every rank received in the last sc_allgather_paths on this context */
int sc_allgather_last_bytes(sc_ctx* ctx, int64_t* bytes);
/* The two halves of sc_allgather_paths on their own, for a caller whose transport is not RCCL (the reference's is ZMQ,
 * examples/zmq_test.cpp:18-22) and for tests that play several ranks on one GPU.  Device pointers, enqueued, no host
 * synchronisation, no communicator needed.
 *   sc_gather_msg_words : int32 words of one rank's message for (Q_total, world, cap_cells); 0 for bad arguments.
 *   sc_gather_pack      : rank `rank` of `world` packs its Q_local results into msg [sc_gather_msg_words] (Q_local must be
 *                         what sc_rank_range gives that rank; 0 is allowed and the input pointers may then be NULL).
 *   sc_gather_unpack    : msgs = the `world` messages back to back in rank order -> the outputs of sc_allgather_paths.
 *   *truncated: bit 0 = some rank's paths exceeded cap_cells (cells beyond it read -1), bit 1 = cells_all is smaller
 *   than offsets_all[Q_total] (cells beyond cells_capacity are not written).  sc_allgather_paths sets the same bits. */
int64_t sc_gather_msg_words(int Q_total, int world, int cap_cells);
int sc_gather_pack(sc_ctx* ctx, const int32_t* path, const int32_t* len, const int32_t* cost, const int32_t* status, int Q_local,
                   int Q_total, int world, int rank, int Lmax, int cap_cells, int32_t* msg);
int sc_gather_unpack(sc_ctx* ctx, const int32_t* msgs, int world, int Q_total, int Lmax, int cap_cells, int32_t* len_all,
                     int32_t* cost_all, int32_t* status_all, int64_t* offsets_all, int32_t* cells_all, int64_t cells_capacity,
                     int32_t* path_all, int32_t* truncated);

#ifdef __cplusplus
}
#endif
#endif
