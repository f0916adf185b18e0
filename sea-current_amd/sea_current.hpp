// sea_current.hpp -- MI355X-native successor of turtle-robotics/sea-current's single header.
//
// Keeps the `turtle::sc` names, argument order and result types of the reference's planning path
// and velocity-profile path (citations: file:line in the reference's sea_current.hpp), and routes
// the work to libsea_current_hip.so through the C ABI in include/sea_current_hip.h:
//
//   reference                                             here
//   ---------------------------------------------------   ------------------------------------------
//   bounding_rect                      :61-83             same members / constructor order
//   obstacle (polygon, ray casting)    :193-284           same members; rasterised into occupancy_grid
//   planning_space::is_obstacle        :1274-1280         same signature (polygon test, host)
//   planning_space::is_free            :1289-1292         same signature
//   planning_space::cost               :1315-1326         same signature and FLT_MAX convention
//   planning_space::fast_marching_trees:1339-1407         same signature; grid EDT + batched A* on the GPU
//   (new) planning_space::fast_marching_trees_sampled      the reference's FMT* itself (Halton samples, radius), batched on the GPU
//   bezier_spline::from_path / arclength :599-683,:767-896  same signatures, GPU tangents + GL-32 tables
//   bezier_spline::resample            :898-1005          same signature; nudge, split, Chebyshev fit and evaluation on the GPU
//   bezier_spline::curvature / angular_velocity :1017-1067  same signatures
//   velocity_profile                   :379-386           same members
//   vel_lim_func                       :1175              same shape
//   gen_vel_prof<N>                    :1191-1265         same argument order (END before START), GPU TOPP-RA
//   (new) occupancy_grid, planning_space::plan_batch, gen_vel_prof_batch: the batched entry points
//
// Differences that are deliberate: obstacle::closed is initialised (the reference leaves it
// uninitialised, :197); library code never prints or calls std::exit (SC_ASSERT throws in DEBUG);
// all functions are `inline` (the reference defines non-inline functions in a header).
// Also mirrored from the "next" rows: bezier_spline::from_path, ::arclength, ::resample,
// ::curvature, ::angular_velocity -- with these the reference's whole example pipeline (examples/zmq_test.cpp:66-93)
// runs through this header.  bezier_spline::pts is an n x 2 matrix (Eigen's when present) as in the reference (:390).
// The planner's sampling helpers keep their signatures as host functions (halton, sample_free, near, point_set,
// x_state / y_state); fast_marching_trees itself plans on the grid.  Q_cache (the reference's cached Fourier coefficients) is
// kept and filled on the host; not mirrored: the ZMQ transport.  The service's request text and JSON reply are parse_path_request / serialize_path_to_json.
//
// Eigen and toppra are NOT required: if <Eigen/Dense> is on the include path it is used for
// Vector2f / VectorXf, otherwise small stand-ins with the same accessors are provided.
// There is no CPU fallback: without a gfx950 GPU every planning call throws std::runtime_error.
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <unordered_set>
#include <vector>

#include "../include/sea_current_hip.h"

#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define SC_HAVE_EIGEN 1
#endif

#ifdef DEBUG
#define SC_ASSERT(cnd, msg)                                                                       \
    do {                                                                                          \
        if (!bool(cnd)) throw std::logic_error(std::string("SC_ASSERT failed: ") + #cnd + ": " + (msg)); \
    } while (0)
#else
#define SC_ASSERT(cnd, msg)
#endif

namespace turtle::sc {

#ifdef SC_HAVE_EIGEN
using Eigen::Vector2f;
using Eigen::VectorXf;
using VectorXd = Eigen::VectorXd;
template <int N> using VectorNd = Eigen::Matrix<double, N, 1>;
using points_matrix = Eigen::Matrix<float, Eigen::Dynamic, 2>;   // bezier_spline::pts (sea_current.hpp:390)
using fourier_matrix = Eigen::Matrix<std::complex<float>, Eigen::Dynamic, 2>;   // an entry of bezier_spline::Q_cache (:393)
#else
// minimal stand-ins with the accessors the API and the reference's call sites use
struct Vector2f {
    float v[2] = {0, 0};
    Vector2f() = default;
    Vector2f(float x, float y) : v{x, y} {}
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float& x() { return v[0]; }
    float& y() { return v[1]; }
    float operator()(int i) const { return v[i]; }
    float& operator()(int i) { return v[i]; }
    Vector2f operator+(const Vector2f& o) const { return {v[0] + o.v[0], v[1] + o.v[1]}; }
    Vector2f operator-(const Vector2f& o) const { return {v[0] - o.v[0], v[1] - o.v[1]}; }
    Vector2f operator-() const { return {-v[0], -v[1]}; }
    Vector2f operator*(float s) const { return {v[0] * s, v[1] * s}; }
    Vector2f operator/(float s) const { return {v[0] / s, v[1] / s}; }
    bool operator==(const Vector2f& o) const { return v[0] == o.v[0] && v[1] == o.v[1]; }
    float dot(const Vector2f& o) const { return v[0] * o.v[0] + v[1] * o.v[1]; }
    float norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1]); }
    Vector2f normalized() const { const float n = norm(); return n > 0 ? Vector2f(v[0] / n, v[1] / n) : *this; }
};
template <class S, class = std::enable_if_t<std::is_arithmetic_v<S>>>
inline Vector2f operator*(S s, const Vector2f& a) { return a * (float)s; }
// dynamic column vector: `Vec v{n}` / `Vec v(n)` give n elements (as Eigen's size constructor does), v(i), v(i, 0), v[i]
template <class T> struct VectorX_ {
    std::vector<T> d;
    VectorX_() = default;
    explicit VectorX_(size_t n) : d(n) {}
    static VectorX_ Zero(size_t n) { return VectorX_(n); }
    static VectorX_ Ones(size_t n) { VectorX_ v(n); std::fill(v.d.begin(), v.d.end(), T(1)); return v; }
    size_t rows() const { return d.size(); }
    size_t cols() const { return 1; }
    size_t size() const { return d.size(); }
    T& operator()(size_t i) { return d[i]; }
    const T& operator()(size_t i) const { return d[i]; }
    T& operator()(size_t i, size_t) { return d[i]; }
    const T& operator()(size_t i, size_t) const { return d[i]; }
    T& operator[](size_t i) { return d[i]; }
    const T& operator[](size_t i) const { return d[i]; }
    T* data() { return d.data(); }
    const T* data() const { return d.data(); }
    auto begin() { return d.begin(); }
    auto end() { return d.end(); }
    auto begin() const { return d.begin(); }
    auto end() const { return d.end(); }
    T minCoeff() const { return *std::min_element(d.begin(), d.end()); }
    T maxCoeff() const { return *std::max_element(d.begin(), d.end()); }
};
using VectorXf = VectorX_<float>;
using VectorXd = VectorX_<double>;
// fixed-size column vector: `Vec<1> v{x}` lists the coefficients
template <class T, int N> struct VectorN_ {
    T d[N] = {};
    VectorN_() = default;
    VectorN_(std::initializer_list<T> l) { int i = 0; for (T x : l) if (i < N) d[i++] = x; }
    int rows() const { return N; }
    T& operator()(int i) { return d[i]; }
    const T& operator()(int i) const { return d[i]; }
    T& operator()(int i, int) { return d[i]; }
    const T& operator()(int i, int) const { return d[i]; }
    T& operator[](int i) { return d[i]; }
    const T& operator[](int i) const { return d[i]; }
};
template <int N> using VectorNd = VectorN_<double, N>;
// n x 2 matrix of points, row-major: m(i, c), m.rows(), m.col(c) (a copy), m.row(i)
struct points_matrix {
    std::vector<float> d;
    points_matrix() = default;
    explicit points_matrix(size_t n, size_t = 2) : d(2 * n) {}
    static points_matrix Zero(size_t n, size_t = 2) { return points_matrix(n); }
    void resize(size_t n, size_t = 2) { d.resize(2 * n); }
    size_t rows() const { return d.size() / 2; }
    size_t cols() const { return 2; }
    float& operator()(size_t i, size_t c) { return d[2 * i + c]; }
    const float& operator()(size_t i, size_t c) const { return d[2 * i + c]; }
    Vector2f row(size_t i) const { return Vector2f(d[2 * i], d[2 * i + 1]); }
    VectorXf col(size_t c) const { VectorXf v(rows()); for (size_t i = 0; i < rows(); ++i) v(i) = d[2 * i + c]; return v; }
    const float* data() const { return d.data(); }
    float* data() { return d.data(); }
};
// (degree + 1) x 2 complex coefficients: an entry of bezier_spline::Q_cache (sea_current.hpp:393)
struct fourier_matrix {
    std::vector<std::complex<float>> d;
    fourier_matrix() = default;
    explicit fourier_matrix(size_t n, size_t = 2) : d(2 * n) {}
    static fourier_matrix Zero(size_t n, size_t = 2) { return fourier_matrix(n); }
    size_t rows() const { return d.size() / 2; }
    size_t cols() const { return 2; }
    std::complex<float>& operator()(size_t i, size_t c) { return d[2 * i + c]; }
    const std::complex<float>& operator()(size_t i, size_t c) const { return d[2 * i + c]; }
};
#endif

}  // namespace turtle::sc

// The names the reference's callers spell (examples/zmq_test.cpp:69-86, examples/test.cpp:184-204, examples/json_test.cpp:
// 29-47): toppra::Vector / toppra::value_type, and Eigen::Vector<value_type, N> for the arguments of gen_vel_prof.  With
// toppra / Eigen on the include path they are the real types; otherwise these aliases of the stand-ins above keep
// those call sites compiling unchanged (define SC_NO_NAMESPACE_STANDINS to keep namespaces `toppra` / `Eigen` untouched).
#if __has_include(<toppra/toppra.hpp>)
#include <toppra/toppra.hpp>
#elif !defined(SC_NO_NAMESPACE_STANDINS)
namespace toppra {
using value_type = double;
using Vector = turtle::sc::VectorXd;
}  // namespace toppra
#endif
#if !defined(SC_HAVE_EIGEN) && !defined(SC_NO_NAMESPACE_STANDINS)
namespace Eigen {
template <class T, int N> using Vector = turtle::sc::VectorN_<T, N>;
}  // namespace Eigen
#endif

namespace turtle::sc {

// the toppra types that appear in the reference's signatures (`using toppra::value_type`, sea_current.hpp:1172)
using value_type = double;
namespace toppra_compat { using Vector = VectorXd; }

// ---- GPU context -------------------------------------------------------------------------------
class gpu_context {
public:
    explicit gpu_context(int device = 0) {
        int st = sc_ctx_create(device, &h_);
        if (st != SC_OK)
            throw std::runtime_error(std::string("sea_current: no usable MI355X context (") + sc_status_string(st) +
                                     "); there is no CPU fallback");
    }
    ~gpu_context() { if (h_) sc_ctx_destroy(h_); }
    gpu_context(const gpu_context&) = delete;
    gpu_context& operator=(const gpu_context&) = delete;
    sc_ctx* get() const { return h_; }
    void check(int st, const char* what) const {
        if (st != SC_OK) throw std::runtime_error(std::string(what) + ": " + sc_status_string(st) + ": " + sc_last_error(h_));
    }
private:
    sc_ctx* h_ = nullptr;
};
// one context per host thread (contexts are not shared between threads: sea_current_hip.h)
inline gpu_context& default_context() {
    thread_local gpu_context ctx(0);
    return ctx;
}

// ---- geometry (sea_current.hpp:61-83) -------------------------------------------------------------
struct bounding_rect {
    float x_max;
    float x_min;
    float y_max;
    float y_min;
    bounding_rect(const float x_max, const float x_min, const float y_max, const float y_min)
        : x_max(x_max), x_min(x_min), y_max(y_max), y_min(y_min) {}
    inline bool contains(const Vector2f& p) const { return p.x() <= x_max && p.x() >= x_min && p.y() <= y_max && p.y() >= y_min; }
    inline void enclose_point(const Vector2f& p) {
        x_max = std::max(x_max, p.x()); x_min = std::min(x_min, p.x());
        y_max = std::max(y_max, p.y()); y_min = std::min(y_min, p.y());
    }
};

// L2 distance, in the reference's arithmetic (:86-88): float differences, squared and rooted in double -- the GPU planner
// (csrc/fmt.hip) and this host function then agree to the bit on the radius tests of near()
inline float pt_dist(const Vector2f& u, const Vector2f& v = {0, 0}) {
    const double dx = v.x() - u.x(), dy = v.y() - u.y();
    return (float)std::sqrt(dx * dx + dy * dy);
}
// distance from point a to the (infinite) line through p1 and p2 (:181-190)
inline float dist_pt_line(const Vector2f& p1, const Vector2f& p2, const Vector2f a) {
    const float twice_area = std::abs((p2.x() - p1.x()) * (p1.y() - a.y()) - (p1.x() - a.x()) * (p2.y() - p1.y()));
    return twice_area / (p2 - p1).norm();
}
inline float cross2d(const Vector2f& a, const Vector2f& b) { return a.x() * b.y() - a.y() * b.x(); }

// proper intersection of segments p1p2 and q1q2 (colinear overlap counts as no hit, as in :142-178)
inline std::tuple<bool, Vector2f> intersects(const Vector2f& p1, const Vector2f& p2, const Vector2f& q1, const Vector2f& q2) {
    const Vector2f r = p2 - p1, s = q2 - q1, qp = q1 - p1;
    const float den = cross2d(r, s);
    if (den == 0.0f) return {false, Vector2f(0, 0)};
    const float t = cross2d(qp, s) / den, u = cross2d(qp, r) / den;
    if (t < 0 || t > 1 || u < 0 || u > 1) return {false, Vector2f(0, 0)};
    return {true, p1 + r * t};
}

// ---- obstacle (sea_current.hpp:193-284) -----------------------------------------------------------
struct obstacle {
    std::vector<std::tuple<Vector2f, Vector2f>> lines;
    std::vector<Vector2f> vertices;
    bounding_rect bound_rect = {0, 0, 0, 0};
    bool closed = true;  // the reference never initialises this member
    int64_t id = 0;

    obstacle() {}
    // closed polygon through `vertices` (the last edge returns to the first vertex)
    obstacle(std::vector<Vector2f> verts) : vertices(std::move(verts)) {
        if (vertices.empty()) return;
        bound_rect = {vertices[0].x(), vertices[0].x(), vertices[0].y(), vertices[0].y()};
        for (size_t i = 0; i < vertices.size(); ++i) {
            lines.push_back({vertices[i], vertices[(i + 1) % vertices.size()]});
            bound_rect.enclose_point(vertices[i]);
        }
    }
    // explicit edge list (open or closed shapes)
    obstacle(std::vector<Vector2f> verts, std::vector<std::tuple<int, int>> edges) : vertices(std::move(verts)) {
        if (vertices.empty()) return;
        bound_rect = {vertices[0].x(), vertices[0].x(), vertices[0].y(), vertices[0].y()};
        for (const auto& e : edges) {
            SC_ASSERT(std::get<0>(e) < (int)vertices.size() && std::get<1>(e) < (int)vertices.size(), "edge index out of range");
            lines.push_back({vertices[std::get<0>(e)], vertices[std::get<1>(e)]});
            bound_rect.enclose_point(vertices[std::get<0>(e)]);
            bound_rect.enclose_point(vertices[std::get<1>(e)]);
        }
    }
    // even-odd rule on the edge list (closed) / distance-to-edge test (open), after the AABB reject
    bool contains(const Vector2f& p) const {
        if (!bound_rect.contains(p)) return false;
        if (!closed) {
            for (const auto& [a, b] : lines) {
                const Vector2f ab = b - a, ap = p - a;
                const float len = ab.norm();
                if (len > 0 && std::fabs(cross2d(ab, ap)) / len < 0.01f) return true;
            }
            return false;
        }
        bool inside = false;
        for (const auto& [a, b] : lines) {
            if ((a.y() > p.y()) != (b.y() > p.y())) {
                const float xi = a.x() + (p.y() - a.y()) * (b.x() - a.x()) / (b.y() - a.y());
                if (p.x() < xi) inside = !inside;
            }
        }
        return inside;
    }
};

// ---- occupancy grid (new): the world model the GPU path works on ----------------------------------
// Row-major uint8 occupancy over a bounding_rect; cell (ix, iy) covers
// [x_min + ix*res, x_min + (ix+1)*res) x [y_min + iy*res, ...).  d2 is the exact squared distance (in
// cells) to the nearest occupied cell, computed on the GPU (sc_edt_u8_i32).
class occupancy_grid {
public:
    int W = 0, H = 0;
    float resolution = 1.0f;
    bounding_rect bound_rect = {0, 0, 0, 0};
    std::vector<uint8_t> occ;
    std::vector<int32_t> d2;  // filled by edt()

    occupancy_grid() = default;
    occupancy_grid(const bounding_rect& br, int cells_x, int cells_y)
        : W(cells_x), H(cells_y), resolution((br.x_max - br.x_min) / (float)cells_x), bound_rect(br),
          occ((size_t)cells_x * cells_y, 0) {}

    int cell_x(float x) const { return std::clamp((int)std::floor((x - bound_rect.x_min) / resolution), 0, W - 1); }
    int cell_y(float y) const { return std::clamp((int)std::floor((y - bound_rect.y_min) / ((bound_rect.y_max - bound_rect.y_min) / (float)H)), 0, H - 1); }
    int32_t cell_of(const Vector2f& p) const { return cell_y(p.y()) * W + cell_x(p.x()); }
    Vector2f centre_of(int32_t c) const {
        const float ry = (bound_rect.y_max - bound_rect.y_min) / (float)H;
        return Vector2f(bound_rect.x_min + ((c % W) + 0.5f) * resolution, bound_rect.y_min + ((c / W) + 0.5f) * ry);
    }
    // mark every cell whose centre lies inside a closed obstacle or that an obstacle edge passes through
    void rasterize(const std::vector<obstacle>& obstacles) {
        const float ry = (bound_rect.y_max - bound_rect.y_min) / (float)H;
        for (const auto& ob : obstacles) {
            if (ob.lines.empty()) continue;
            if (ob.closed) {
                const int x0 = cell_x(ob.bound_rect.x_min), x1 = cell_x(ob.bound_rect.x_max);
                const int y0 = cell_y(ob.bound_rect.y_min), y1 = cell_y(ob.bound_rect.y_max);
                for (int iy = y0; iy <= y1; ++iy)
                    for (int ix = x0; ix <= x1; ++ix)
                        if (ob.contains(centre_of(iy * W + ix))) occ[(size_t)iy * W + ix] = 1;
            }
            for (const auto& [a, b] : ob.lines) {  // edges: sample at sub-cell steps
                const float len = (b - a).norm();
                const int n = std::max(1, (int)std::ceil(len / (0.5f * std::min(resolution, ry))));
                for (int k = 0; k <= n; ++k) occ[(size_t)cell_of(a + (b - a) * ((float)k / n))] = 1;
            }
        }
    }
    // exact squared EDT on the GPU
    void edt(gpu_context& ctx = default_context()) {
        d2.resize(occ.size());
        ctx.check(sc_edt_u8_i32_host(ctx.get(), occ.data(), W, H, 1, d2.data()), "sc_edt_u8_i32_host");
    }
    struct batch_result {
        std::vector<int32_t> path, len, cost, status;  // path is [Q][Lmax]
        int Lmax = 0;
    };
    // Q independent start->goal queries (linear cell indices); r2_clear = squared clearance in cells
    batch_result astar_batch(const std::vector<int32_t>& start, const std::vector<int32_t>& goal, int32_t r2_clear = 0,
                             int Lmax = 0, gpu_context& ctx = default_context()) {
        if (d2.size() != occ.size()) edt(ctx);
        batch_result r;
        const int Q = (int)start.size();
        r.Lmax = Lmax > 0 ? Lmax : 4 * (W + H);
        r.path.assign((size_t)Q * r.Lmax, -1); r.len.assign(Q, 0); r.cost.assign(Q, -1); r.status.assign(Q, SC_Q_NO_PATH);
        if (Q == 0) return r;
        ctx.check(sc_astar_batch_host(ctx.get(), d2.data(), W, H, r2_clear, start.data(), goal.data(), Q, r.Lmax,
                                      r.path.data(), r.len.data(), r.cost.data(), r.status.data()), "sc_astar_batch_host");
        return r;
    }
};

// ---- planning_space (sea_current.hpp:298-319, 1272-1407) --------------------------------------------
// ---- sampling helpers of the reference's planner (sea_current.hpp:90-132, 287-296) ----------------
// State of the incremental Halton generator: the last fraction was f / i.
struct halton_state {
    int f = 0;
    int i = 0;
    halton_state(int f, int i) : f(f), i(i) {}
    halton_state() : f(0), i(0) {}
};

// next n numbers of the base-b Halton sequence (1/b, 2/b, ..., 1/b^2, ...), continuing from `state` (same signature and
// float results as :100-132: integer numerator / denominator, one float division per number)
inline std::vector<float> halton(const int b, const int n, halton_state& state) {
    std::vector<float> nums(n);
    int num = 0, den = 1;
    if (state.i != 0 && state.f != 0) { den = state.i; num = state.f; }
    for (int j = 0; j < n; ++j) {
        const int gap = den - num;
        if (gap == 1) {              // the current power of b is used up: start the next one
            num = 1;
            den *= b;
        } else {
            int y = den / b;
            while (gap <= y) y /= b;
            num = (b + 1) * y - gap;
        }
        nums[j] = static_cast<float>(num) / den;
    }
    state.f = num;
    state.i = den;
    return nums;
}

// hash of a point by the bit patterns of its coordinates (:287-295)
struct hash_vector2f {
    size_t operator()(const Vector2f v) const {
        const float fa = v.x(), fb = v.y();
        int32_t a, b;
        std::memcpy(&a, &fa, 4);
        std::memcpy(&b, &fb, 4);
        return std::hash<int32_t>()(a) ^ std::hash<int32_t>()(b);
    }
};
struct equal_vector2f {
    bool operator()(const Vector2f& a, const Vector2f& b) const { return a.x() == b.x() && a.y() == b.y(); }
};
using point_set = std::unordered_set<Vector2f, hash_vector2f, equal_vector2f>;

class planning_space {
public:
    std::vector<obstacle> obstacles;
    std::vector<std::function<bool(Vector2f)>> free_space_allocations;
    bounding_rect bound_rect;
    halton_state x_state;       // Halton state of sample_free (bases 2 and 3), advanced by every call (:317-318)
    halton_state y_state;
    int grid_cells = 256;       // cells along the longer side of bound_rect (new knob)
    float clearance = 0.0f;     // required obstacle clearance in world units (new knob)

    planning_space(const bounding_rect& br) : bound_rect(br) {}

    std::tuple<bool, obstacle> is_obstacle(const Vector2f& p) {
        for (auto& ob : obstacles)
            if (ob.contains(p)) return {true, ob};
        return {false, obstacle()};
    }
    bool is_free_space_allocated(const Vector2f& p) {
        for (auto& f : free_space_allocations)
            if (f(p)) return true;
        return false;
    }
    bool is_free(const Vector2f& p) { return !std::get<0>(is_obstacle(p)) && is_free_space_allocated(p); }
    // n free points of the bounding rectangle from the Halton sequence, (0,0) included (:1294-1313; no stdout print).
    // The reference never returns when nothing can be free (no free-space allocation); this throws instead.
    point_set sample_free(const int n) {
        if (free_space_allocations.empty() && n > 1)
            throw std::logic_error("planning_space::sample_free: no free-space allocation, no point can be free");
        point_set pts = {Vector2f(0, 0)};
        pts.reserve(n);
        size_t tested = 0;
        while ((size_t)n > pts.size()) {
            const int want = n - (int)pts.size();
            const std::vector<float> xs = halton(2, want, x_state), ys = halton(3, want, y_state);
            for (int i = 0; i < want; ++i) {
                const Vector2f test((bound_rect.x_max - bound_rect.x_min) * xs[i] + bound_rect.x_min,
                                    (bound_rect.y_max - bound_rect.y_min) * ys[i] + bound_rect.y_min);
                if (is_free(test)) pts.insert(test);
            }
            tested += (size_t)want;
            if (tested > 1000 * (size_t)n + 100000 && pts.size() <= 1)
                throw std::runtime_error("planning_space::sample_free: no free point found");
        }
        return pts;
    }
    // points of `nodes` within `dist` of b, b itself excluded.  The reference compares the (unsquared) distance with
    // dist squared (:1331); kept, so that radii mean the same thing in both libraries.
    point_set near(const Vector2f b, const point_set& nodes, const float dist) const {
        point_set out;
        out.reserve(nodes.size());
        for (const auto& a : nodes)
            if (pt_dist(a, b) <= std::pow(dist, 2) && !(a.x() == b.x() && a.y() == b.y())) out.insert(a);
        return out;
    }
    // both points within epsilon of (the supporting lines of) one obstacle's edges (:444-463; the reference tests `a`
    // twice, so does this)
    bool is_same_obstacle_fuzzy(const Vector2f& a, const Vector2f& b, const float epsilon) {
        (void)b;
        for (const auto& ob : obstacles) {
            bool tag = false;
            for (const auto& [p1, p2] : ob.lines) {
                const Vector2f e = p2 - p1;
                const float len = e.norm();
                if (len > 0 && std::fabs((p2.x() - p1.x()) * (p1.y() - a.y()) - (p1.x() - a.x()) * (p2.y() - p1.y())) / len < epsilon) tag = true;
            }
            if (tag) return true;
        }
        return false;
    }
    // length of segment ab, FLT_MAX if it crosses any obstacle edge (:1315-1326)
    float cost(const Vector2f a, const Vector2f b) const {
        for (const auto& ob : obstacles)
            for (const auto& [p, q] : ob.lines)
                if (std::get<0>(intersects(a, b, p, q))) return FLT_MAX;
        return pt_dist(a, b);
    }
    occupancy_grid make_grid() const {
        const float wx = bound_rect.x_max - bound_rect.x_min, wy = bound_rect.y_max - bound_rect.y_min;
        const float res = std::max(wx, wy) / (float)grid_cells;
        occupancy_grid g(bound_rect, std::max(1, (int)std::ceil(wx / res)), std::max(1, (int)std::ceil(wy / res)));
        g.rasterize(obstacles);
        return g;
    }
    // Same role and result type as the reference's FMT* planner: start -> goal waypoint list, nullopt if
    // no path.  `n` and `rn` (sample count, connection radius) are accepted for source compatibility;
    // the grid resolution is `grid_cells`.
    std::optional<std::vector<Vector2f>> fast_marching_trees(const Vector2f& x_init, const Vector2f& x_goal, const int n = 0, const float rn = 0) {
        (void)n; (void)rn;
        auto r = plan_batch({x_init}, {x_goal});
        return r[0];
    }
    // The reference's own algorithm (FMT* over n Halton samples with connection radius rn, :1339-1407) for a batch of
    // queries, on the GPU (sc_fmt_star_batch): one sample set (drawn like the reference does inside the call, advancing
    // x_state / y_state once) shared by all queries.  Needs a free-space allocation, like sample_free.
    std::vector<std::optional<std::vector<Vector2f>>> fast_marching_trees_sampled(const std::vector<Vector2f>& starts,
                                                                                  const std::vector<Vector2f>& goals, const int n,
                                                                                  const float rn, gpu_context& ctx = default_context()) {
        const point_set ps = sample_free(n);
        std::vector<float> smp, lines, st, gl;
        smp.push_back(0.f); smp.push_back(0.f);                     // (0,0) first, as the reference inserts it (:1297)
        for (const auto& p : ps)
            if (!(p.x() == 0.f && p.y() == 0.f)) { smp.push_back(p.x()); smp.push_back(p.y()); }
        for (const auto& ob : obstacles)
            for (const auto& [a, b] : ob.lines) { lines.push_back(a.x()); lines.push_back(a.y()); lines.push_back(b.x()); lines.push_back(b.y()); }
        const int Q = (int)starts.size(), Lmax = (int)smp.size() / 2 + 2;
        for (int q = 0; q < Q; ++q) { st.push_back(starts[q].x()); st.push_back(starts[q].y()); gl.push_back(goals[q].x()); gl.push_back(goals[q].y()); }
        std::vector<float> path((size_t)Q * Lmax * 2), cost(Q);
        std::vector<int32_t> len(Q), status(Q);
        ctx.check(sc_fmt_star_batch_host(ctx.get(), smp.data(), (int)smp.size() / 2, st.data(), gl.data(), Q, rn, lines.empty() ? nullptr : lines.data(),
                                         (int)lines.size() / 4, Lmax, path.data(), len.data(), cost.data(), status.data()),
                  "sc_fmt_star_batch_host");
        std::vector<std::optional<std::vector<Vector2f>>> out(Q);
        for (int q = 0; q < Q; ++q) {
            if (status[q] != SC_Q_OK) continue;
            std::vector<Vector2f> wp;
            for (int i = 0; i < len[q]; ++i) wp.push_back(Vector2f(path[((size_t)q * Lmax + i) * 2], path[((size_t)q * Lmax + i) * 2 + 1]));
            out[q] = std::move(wp);
        }
        return out;
    }
    // batched form: one grid, one EDT, Q queries in one GPU launch
    std::vector<std::optional<std::vector<Vector2f>>> plan_batch(const std::vector<Vector2f>& starts, const std::vector<Vector2f>& goals,
                                                                 gpu_context& ctx = default_context()) {
        occupancy_grid g = make_grid();
        g.edt(ctx);
        std::vector<int32_t> s(starts.size()), t(goals.size());
        for (size_t i = 0; i < starts.size(); ++i) { s[i] = g.cell_of(starts[i]); t[i] = g.cell_of(goals[i]); }
        const float cc = clearance / g.resolution;
        auto br = g.astar_batch(s, t, (int32_t)std::ceil(cc * cc), 0, ctx);
        std::vector<std::optional<std::vector<Vector2f>>> out(starts.size());
        for (size_t q = 0; q < starts.size(); ++q) {
            if (br.status[q] != SC_Q_OK) continue;
            std::vector<Vector2f> wp;
            wp.reserve(br.len[q] + 2);
            wp.push_back(starts[q]);
            for (int i = 1; i + 1 < br.len[q]; ++i) wp.push_back(g.centre_of(br.path[(size_t)q * br.Lmax + i]));
            wp.push_back(goals[q]);
            out[q] = std::move(wp);
        }
        return out;
    }
};

// ---- path smoothing (sea_current.hpp:321-431, 522-1170) ---------------------------------------------
struct arclength_data {
    float arclength = 0;
    std::vector<VectorXf> segments;   // cumulative arclength of each segment at t = k * precision
    std::vector<VectorXf> positions;  // the parameters t of those table entries
};

// Chebyshev polynomial: coefficients of T_0 .. T_{degree-1} on [xmin, xmax] (:328-335)
struct chebpoly {
    VectorXf coeffs;
    float xmin;
    float xmax;
    chebpoly(VectorXf coeffs, const float xmin, const float xmax) : coeffs(std::move(coeffs)), xmin(xmin), xmax(xmax) {}
};

// Least-squares fit of y(x) by `degree` Chebyshev columns (:1109-1138; Householder least squares on the GPU,
// sc_chebfit_batch) and its evaluation (:1140-1170, sc_chebeval_batch).
inline chebpoly chebfit(const VectorXf& x, const VectorXf& y, const int degree, gpu_context& ctx = default_context()) {
    SC_ASSERT(degree >= 1, "degree must be a positive integer");
    SC_ASSERT(x.rows() == y.rows(), "x and y must have the same number of rows");
    const int m = (int)x.rows();
    SC_ASSERT(m > 0 && std::abs(x.maxCoeff() - x.minCoeff()) > 0.00001, "Error: vector x should not have all equal values");
    std::vector<float> xs(m), ys(m), coef(degree);
    for (int i = 0; i < m; ++i) { xs[i] = x(i); ys[i] = y(i); }
    const int32_t off[2] = {0, m};
    float xr[2] = {0, 0};
    ctx.check(sc_chebfit_batch_host(ctx.get(), xs.data(), ys.data(), off, 1, degree, coef.data(), xr), "sc_chebfit_batch_host");
    VectorXf c = VectorXf::Zero(degree);
    for (int k = 0; k < degree; ++k) c(k) = coef[k];
    return chebpoly(std::move(c), xr[0], xr[1]);
}
inline VectorXf chebeval(const VectorXf& x, const chebpoly& b, const int degree, gpu_context& ctx = default_context()) {
    SC_ASSERT(degree >= 1 && (int)b.coeffs.rows() >= degree, "degree must be a positive integer within the polynomial");
    const int m = (int)x.rows();
    VectorXf y = VectorXf::Zero(m);
    if (m == 0) return y;
    std::vector<float> xs(m), ys(m), coef(degree);
    for (int i = 0; i < m; ++i) xs[i] = x(i);
    for (int k = 0; k < degree; ++k) coef[k] = b.coeffs(k);
    const int32_t off[2] = {0, m};
    const float xr[2] = {b.xmin, b.xmax};
    ctx.check(sc_chebeval_batch_host(ctx.get(), xs.data(), off, 1, degree, coef.data(), xr, ys.data()), "sc_chebeval_batch_host");
    for (int i = 0; i < m; ++i) y(i) = ys[i];
    return y;
}

// Tangent heuristics of Lau, Sprunk, Burgard (IROS 2009) as the reference states them (:339-377): scalar host
// helpers for callers that build control points by hand (examples/test.cpp:88-104); from_path computes the same on the GPU.
inline float tangent_magnitude(const Vector2f& W_0, const Vector2f& W_1, const Vector2f& W_2) {
    return 0.5f * std::min(pt_dist(W_0, W_1), pt_dist(W_1, W_2));
}
// tangent at W_1: perpendicular to the bisector of the angle W_0 W_1 W_2, pointing on towards W_2
inline Vector2f calc_tangent(const Vector2f& W_0, const Vector2f& W_1, const Vector2f& W_2) {
    const Vector2f u = W_0 - W_1, v = W_2 - W_1;
    const float half = std::acos(u.dot(v) / (pt_dist(u) * pt_dist(v))) / 2;
    const float a_u = std::atan2(u.y(), u.x()), a_v = std::atan2(v.y(), v.x());
    const float ang = a_u + (a_v - a_u < 0 ? -half : half);
    Vector2f l90 = Vector2f(std::sin(ang), -std::cos(ang)).normalized();
    const float sign = pt_dist(W_1 + l90, W_2) < pt_dist(W_1 + (-l90), W_2) ? 1.0f : -1.0f;
    return tangent_magnitude(W_0, W_1, W_2) * (sign * l90);
}
inline Vector2f calc_start_tangent(const Vector2f& W_0, const Vector2f& W_1, const float theta) {
    return tangent_magnitude(W_0, W_1, W_0) * Vector2f(std::cos(theta), std::sin(theta));
}
inline Vector2f calc_end_tangent(const Vector2f& W_1, const Vector2f W_2) {
    return tangent_magnitude(W_1, W_2, W_1) * (W_2 - W_1).normalized();
}

struct velocity_profile;

class bezier_spline {
public:
    std::vector<std::vector<Vector2f>> ctrl_pts;  // control points per segment (degree + 1 each)
    points_matrix pts;                            // sampled points, one row each, in sample order
    std::vector<VectorXf> positions;              // per segment: the curve parameters of its samples (as :392)
    // per segment: the inverse discrete Fourier transform of its control points, the coefficients of the reference's
    // Bernstein-Fourier evaluation (:393, filled as :700-716 and :746 / :554-567 do).  Curves are evaluated from the control
    // points here (on the GPU); the member is kept, and kept filled, for callers that read it.
    std::vector<fourier_matrix> Q_cache;
    static fourier_matrix fourier_coefficients(const std::vector<Vector2f>& cp) {
        const size_t n = cp.size();
        fourier_matrix Q = fourier_matrix::Zero(n, 2);
        for (size_t k = 0; k < n; ++k) {
            std::complex<double> ax(0, 0), ay(0, 0);
            for (size_t j = 0; j < n; ++j) {
                const double ang = 2.0 * 3.14159265358979323846 * (double)((j * k) % n) / (double)n;      // inverse transform: e^{+2 pi i jk / n} / n
                const std::complex<double> w(std::cos(ang), std::sin(ang));
                ax += (double)cp[j].x() * w; ay += (double)cp[j].y() * w;
            }
            Q(k, 0) = std::complex<float>((float)(ax.real() / n), (float)(ax.imag() / n));
            Q(k, 1) = std::complex<float>((float)(ay.real() / n), (float)(ay.imag() / n));
        }
        return Q;
    }

    bezier_spline() = default;
    bezier_spline(const std::vector<std::vector<Vector2f>>& ctrl_pts, const points_matrix& pts, const std::vector<VectorXf>& positions)
        : ctrl_pts(ctrl_pts), pts(pts), positions(positions) {}
    int n_segments() const { return (int)ctrl_pts.size(); }
    int n_pts() const { return (int)pts.rows(); }
    int degree() const { return ctrl_pts.empty() ? 0 : (int)ctrl_pts[0].size() - 1; }

    // points of segments[i] (all of one degree) at parameters t[i], on the GPU: cubics through sc_bezier_eval_batch (the
    // arithmetic pinned to the recorded run), any other degree through sc_bezier_curve_batch
    static points_matrix evaluate(const std::vector<std::vector<Vector2f>>& cps, const std::vector<int32_t>& seg, const std::vector<float>& t,
                                  gpu_context& ctx = default_context()) {
        const int S = (int)cps.size(), M = (int)t.size(), deg = S ? (int)cps[0].size() - 1 : 0;
        points_matrix out(M, 2);
        if (M == 0) return out;
        SC_ASSERT(deg >= 1 && deg <= SC_BEZIER_MAX_DEGREE, "ctrl_pts must have at least 2 points");
        std::vector<float> c((size_t)S * (deg + 1) * 2), xy(2 * (size_t)M);
        for (int i = 0; i < S; ++i)
            for (int k = 0; k <= deg; ++k) { c[((size_t)i * (deg + 1) + k) * 2] = cps[i][k].x(); c[((size_t)i * (deg + 1) + k) * 2 + 1] = cps[i][k].y(); }
        if (deg == 3) ctx.check(sc_bezier_eval_batch_host(ctx.get(), c.data(), S, seg.data(), t.data(), M, 0, xy.data()), "sc_bezier_eval_batch_host");
        else ctx.check(sc_bezier_curve_batch_host(ctx.get(), c.data(), S, deg, seg.data(), t.data(), M, xy.data()), "sc_bezier_curve_batch_host");
        for (int i = 0; i < M; ++i) { out(i, 0) = xy[2 * i]; out(i, 1) = xy[2 * i + 1]; }
        return out;
    }

    // one curve from its control polygon, sampled at `positions` in [0, 1] (:684-752)
    static bezier_spline bezier_curve(const std::vector<Vector2f>& ctrl_pts, const VectorXf& positions, gpu_context& ctx = default_context()) {
        SC_ASSERT(ctrl_pts.size() >= 2, "ctrl_pts must have at least 2 points");
        const int M = (int)positions.rows();
        std::vector<float> t(M);
        for (int i = 0; i < M; ++i) t[i] = positions(i);
        bezier_spline bs({ctrl_pts}, evaluate({ctrl_pts}, std::vector<int32_t>(M, 0), t, ctx), {positions});
        bs.Q_cache = {fourier_coefficients(ctrl_pts)};
        return bs;
    }
    static bezier_spline bezier_curve(const std::vector<Vector2f>& ctrl_pts, const std::vector<float>& positions, gpu_context& ctx = default_context()) {
        VectorXf p = VectorXf::Zero(positions.size());
        for (size_t i = 0; i < positions.size(); ++i) p(i) = positions[i];
        return bezier_curve(ctrl_pts, p, ctx);
    }
    // ... sampled at k * precision, k = 0 .. 1/precision (:754-763)
    static bezier_spline bezier_curve(const std::vector<Vector2f>& ctrl_pts, const float precision, gpu_context& ctx = default_context()) {
        SC_ASSERT(precision < 1 && precision > 0, "spline percision must be in (0, 1)");
        const int n = (int)std::lround(1.0f / precision);
        VectorXf p = VectorXf::Zero(n + 1);
        for (int i = 0; i <= n; ++i) p(i) = std::max(0.0f, std::min(i * precision, 1.0f));
        return bezier_curve(ctrl_pts, p, ctx);
    }

    // the segments, sample positions and points of several splines, one after the other (:543-570)
    static bezier_spline join_splines(const std::vector<bezier_spline>& splines) {
        bezier_spline bs;
        size_t n = 0;
        for (const auto& sp : splines) n += (size_t)sp.n_pts();
        bs.pts = points_matrix::Zero(n, 2);
        size_t o = 0;
        for (const auto& sp : splines) {
            for (int i = 0; i < sp.n_segments(); ++i) {
                bs.ctrl_pts.push_back(sp.ctrl_pts[i]);
                if (i < (int)sp.positions.size()) bs.positions.push_back(sp.positions[i]);
                bs.Q_cache.push_back(i < (int)sp.Q_cache.size() ? sp.Q_cache[i] : fourier_coefficients(sp.ctrl_pts[i]));
            }
            for (int r = 0; r < sp.n_pts(); ++r, ++o) { bs.pts(o, 0) = sp.pts(r, 0); bs.pts(o, 1) = sp.pts(r, 1); }
        }
        return bs;
    }

    // k * T, cut where the stretch W +- k T crosses an obstacle edge of `ps` (same signature as :401 / :575-596; the edge
    // tests run on the GPU, sc_bezier_shrink_tangent_batch: what from_path applies to every waypoint's tangent)
    static Vector2f shrink_tangent(const Vector2f& T, const Vector2f& W, const float k, const planning_space& ps,
                                   gpu_context& ctx = default_context()) {
        std::vector<float> lines;
        for (const auto& ob : ps.obstacles)
            for (const auto& [a, b] : ob.lines) { lines.push_back(a.x()); lines.push_back(a.y()); lines.push_back(b.x()); lines.push_back(b.y()); }
        const float t[2] = {T.x(), T.y()}, w[2] = {W.x(), W.y()};
        float out[2] = {k * T.x(), k * T.y()};
        if (!lines.empty())
            ctx.check(sc_bezier_shrink_tangent_batch_host(ctx.get(), t, w, 1, k, lines.data(), (int)(lines.size() / 4), out), "sc_bezier_shrink_tangent_batch_host");
        return Vector2f(out[0], out[1]);
    }

    // the (degree + 1)-th roots of unity the reference's Bernstein-Fourier evaluation runs over (:422, :1096-1106): kept for
    // callers that use it; curves are evaluated from their control points here
    static inline std::vector<std::complex<float>> omega_table(const int degree) {
        std::vector<std::complex<float>> omegas((size_t)degree + 1);
        const double step = -2.0 * 3.14159265358979323846 / (double)(degree + 1);
        for (int i = 0; i <= degree; ++i) omegas[(size_t)i] = std::complex<float>((float)std::cos(step * i), (float)std::sin(step * i));
        return omegas;
    }

    // cubic Bezier spline through a piecewise-linear path; tangents by the Lau09 heuristics, shrunk against the
    // obstacle edges of `ps` (same signature as :599; tangents and control points on the GPU,
    // sc_bezier_from_path_batch), every leg sampled at 1e-4 like the reference's (:679)
    static bezier_spline from_path(const std::vector<Vector2f>& path, const planning_space& ps, float start_angle = NAN,
                                   gpu_context& ctx = default_context()) {
        SC_ASSERT(path.size() >= 2, "Not enough points for a path");
        const int n = (int)path.size();
        std::vector<float> xy(2 * (size_t)n), lines;
        for (int i = 0; i < n; ++i) { xy[2 * i] = path[i].x(); xy[2 * i + 1] = path[i].y(); }
        for (const auto& ob : ps.obstacles)
            for (const auto& [a, b] : ob.lines) { lines.push_back(a.x()); lines.push_back(a.y()); lines.push_back(b.x()); lines.push_back(b.y()); }
        const int32_t npts = n;
        std::vector<float> c(8 * (size_t)(n - 1));
        ctx.check(sc_bezier_from_path_batch_host(ctx.get(), xy.data(), &npts, 1, n, start_angle, lines.empty() ? nullptr : lines.data(),
                                                 (int)(lines.size() / 4), c.data()), "sc_bezier_from_path_batch_host");
        bezier_spline bs;
        bs.ctrl_pts.resize(n - 1);
        for (int i = 0; i < n - 1; ++i)
            for (int k = 0; k < 4; ++k) bs.ctrl_pts[i].push_back(Vector2f(c[8 * i + 2 * k], c[8 * i + 2 * k + 1]));
        constexpr int NS = 10000;   // 1 / 1e-4
        VectorXf p = VectorXf::Zero(NS + 1);
        for (int k = 0; k <= NS; ++k) p(k) = std::min(k * 1e-4f, 1.0f);
        std::vector<int32_t> seg((size_t)(n - 1) * (NS + 1));
        std::vector<float> t(seg.size());
        for (int i = 0; i < n - 1; ++i)
            for (int k = 0; k <= NS; ++k) { seg[(size_t)i * (NS + 1) + k] = i; t[(size_t)i * (NS + 1) + k] = p(k); }
        bs.pts = evaluate(bs.ctrl_pts, seg, t, ctx);
        bs.positions.assign(n - 1, p);
        for (const auto& cp : bs.ctrl_pts) bs.Q_cache.push_back(fourier_coefficients(cp));
        return bs;
    }

    // 32-point Gauss-Legendre arclength tables, 1/precision sub-intervals per segment (same as :765-896)
    arclength_data arclength(const float precision = 0.01f, gpu_context& ctx = default_context()) const {
        arclength_data ad;
        const int S = n_segments(), nsub = (int)std::lround(1.0f / precision);
        if (S == 0) return ad;
        SC_ASSERT(degree() == 3, "arclength tables are built for cubic segments");
        std::vector<float> c(8 * (size_t)S), cum((size_t)S * (nsub + 1)), len(S);
        for (int i = 0; i < S; ++i)
            for (int k = 0; k < 4; ++k) { c[8 * i + 2 * k] = ctrl_pts[i][k].x(); c[8 * i + 2 * k + 1] = ctrl_pts[i][k].y(); }
        ctx.check(sc_bezier_arclength_batch_host(ctx.get(), c.data(), S, nsub, cum.data(), len.data()), "sc_bezier_arclength_batch_host");
        ad.segments.assign(S, VectorXf::Zero(nsub + 1));
        ad.positions.assign(S, VectorXf::Zero(nsub + 1));
        for (int i = 0; i < S; ++i) {
            for (int k = 0; k <= nsub; ++k) { ad.segments[i](k) = cum[(size_t)i * (nsub + 1) + k]; ad.positions[i](k) = std::min(k * precision, 1.0f); }
            ad.arclength += len[i];
        }
        return ad;
    }

    // Map the arclength positions of a velocity profile back onto the curve (same signature as :898; profile_pos is
    // repaired in place when nudge_positions is set, as the reference's by-reference argument is).  Returns the spline
    // with one point per profile sample; throws if a segment receives no sample (the reference indexes out of range).
    bezier_spline resample(VectorXf& profile_pos, arclength_data ad, bool nudge_positions = false,
                           gpu_context& ctx = default_context()) const {
        SC_ASSERT(profile_pos.rows() > 0, "The vector of positions to be sampled must not be empty");
        const int S = n_segments(), n = (int)profile_pos.rows();
        SC_ASSERT((int)ad.segments.size() == S && S > 0, "arclength_data does not belong to this spline");
        const int nsub = (int)ad.segments[0].rows() - 1;
        std::vector<float> c(8 * (size_t)S), cum((size_t)S * (nsub + 1)), pp(n), out_pts(2 * (size_t)n), tp(n), cv(n);
        std::vector<int32_t> sg(n);
        for (int i = 0; i < S; ++i) {
            for (int k = 0; k < 4; ++k) { c[8 * i + 2 * k] = ctrl_pts[i][k].x(); c[8 * i + 2 * k + 1] = ctrl_pts[i][k].y(); }
            for (int k = 0; k <= nsub; ++k) cum[(size_t)i * (nsub + 1) + k] = ad.segments[i](k);
        }
        for (int i = 0; i < n; ++i) pp[i] = profile_pos(i);
        const int32_t seg_off[2] = {0, S}, prof_off[2] = {0, n};
        int32_t status = 0;
        ctx.check(sc_bezier_resample_batch_host(ctx.get(), c.data(), cum.data(), &ad.arclength, seg_off, 1, S, nsub, pp.data(), prof_off,
                                                nudge_positions ? 1 : 0, out_pts.data(), tp.data(), sg.data(), cv.data(), &status),
                  "sc_bezier_resample_batch_host");
        if (status != 0) throw std::runtime_error("bezier_spline::resample: a segment of the spline received no profile sample");
        for (int i = 0; i < n; ++i) profile_pos(i) = pp[i];
        bezier_spline re;
        re.ctrl_pts = ctrl_pts;
        re.pts = points_matrix::Zero(n, 2);
        std::vector<int> cnt(S, 0);
        for (int i = 0; i < n; ++i) { re.pts(i, 0) = out_pts[2 * i]; re.pts(i, 1) = out_pts[2 * i + 1]; ++cnt[sg[i]]; }
        re.positions.clear();
        for (int s = 0, o = 0; s < S; ++s) {
            VectorXf v = VectorXf::Zero(cnt[s]);
            for (int k = 0; k < cnt[s]; ++k) v(k) = tp[o + k];
            o += cnt[s];
            re.positions.push_back(std::move(v));
        }
        return re;
    }

    // first derivative as a spline of its own: control points degree * (P[j+1] - P[j]) per segment, sampled at the same
    // positions (:1041-1053)
    bezier_spline hodograph(gpu_context& ctx = default_context()) const {
        bezier_spline h;
        const int S = n_segments(), deg = degree();
        SC_ASSERT(deg >= 2, "the hodograph of a line has no control polygon");
        std::vector<int32_t> seg;
        std::vector<float> t;
        for (int i = 0; i < S; ++i) {
            std::vector<Vector2f> dc(deg);
            for (int j = 0; j < deg; ++j) dc[j] = (ctrl_pts[i][j + 1] - ctrl_pts[i][j]) * (float)deg;
            h.ctrl_pts.push_back(std::move(dc));
            if (i < (int)positions.size())
                for (size_t k = 0; k < (size_t)positions[i].rows(); ++k) { seg.push_back(i); t.push_back(positions[i](k)); }
        }
        h.positions = positions;
        h.pts = evaluate(h.ctrl_pts, seg, t, ctx);
        return h;
    }

    // signed curvature at every sample of `positions` (:1017-1039: hodograph and its hodograph)
    std::vector<float> curvature(gpu_context& ctx = default_context()) const {
        const bezier_spline d = hodograph(ctx), dd = d.hodograph(ctx);
        std::vector<float> res((size_t)d.n_pts());
        for (size_t i = 0; i < res.size(); ++i) {
            const float dx = d.pts(i, 0), dy = d.pts(i, 1), ddx = dd.pts(i, 0), ddy = dd.pts(i, 1);
            res[i] = (dx * ddy - dy * ddx) / std::pow(dx * dx + dy * dy, 1.5f);
        }
        return res;
    }

    inline std::vector<float> angular_velocity(const velocity_profile& vel_prof) const;    // :1055-1067
    inline std::vector<float> angular_velocity2(const velocity_profile& vel_prof) const;   // :1069-1094 (no stdout prints)
};

// ---- velocity profile (sea_current.hpp:379-386, 1172-1265) ------------------------------------------
struct velocity_profile {
    std::vector<VectorXf> pos;
    std::vector<VectorXf> vel;
    std::vector<VectorXf> acc;
    toppra_compat::Vector time;
    velocity_profile(std::vector<VectorXf> pos, std::vector<VectorXf> vel, std::vector<VectorXf> acc, toppra_compat::Vector time)
        : pos(std::move(pos)), vel(std::move(vel)), acc(std::move(acc)), time(std::move(time)) {}
};

inline std::vector<float> bezier_spline::angular_velocity(const velocity_profile& vel_prof) const {
    const std::vector<float> curv = curvature();
    SC_ASSERT(curv.size() == (size_t)vel_prof.vel[0].size(), "curvature and velocity vectors must be the same size");
    std::vector<float> w(curv.size());
    for (size_t i = 0; i < w.size(); ++i) w[i] = vel_prof.vel[0](i) * curv[i];
    return w;
}
// heading change of the tangent between consecutive samples over their time difference; zero at both ends
inline std::vector<float> bezier_spline::angular_velocity2(const velocity_profile& vel_prof) const {
    const bezier_spline d = hodograph();
    const size_t n = (size_t)d.n_pts();
    SC_ASSERT(n == (size_t)vel_prof.vel[0].size(), "hodograph and velocity vectors must be the same size");
    std::vector<float> rads(n), res(n, 0.0f);
    for (size_t i = 0; i < n; ++i) rads[i] = std::atan2(d.pts(i, 1), d.pts(i, 0));
    for (size_t i = 1; i + 1 < n; ++i) res[i] = (rads[i + 1] - rads[i]) / (float)(vel_prof.time(i + 1) - vel_prof.time(i));
    return res;
}

// limits as a function of the GRIDPOINT value s in [0,1] (the reference names the argument "time", :1175, :1185)
using vel_lim_func = std::function<std::tuple<toppra_compat::Vector, toppra_compat::Vector>(value_type time)>;   // = toppra::Vector

constexpr int SC_TOPPRA_GRID = 100;  // toppra's default number of grid intervals (confirmed by examples/output.json)

// P plans of `dof` joints in one GPU launch.  Arrays are [P][dof] row-major; limits per plan.
struct plan_request {
    std::vector<double> pos_end, pos_start, vel_end, vel_start, acc_min, acc_max;  // [dof] each
    vel_lim_func vel_lim;
};
inline std::vector<velocity_profile> gen_vel_prof_batch(const std::vector<plan_request>& plans, const float dt = 0.02f,
                                                        gpu_context& ctx = default_context()) {
    const int P = (int)plans.size();
    if (P == 0) return {};
    const int dof = (int)plans[0].pos_end.size(), N = SC_TOPPRA_GRID;
    std::vector<double> p0((size_t)P * dof), p1(p0), v0(p0), v1(p0), alo(p0), ahi(p0);
    std::vector<double> vlo((size_t)P * (N + 1) * dof), vhi(vlo);
    for (int p = 0; p < P; ++p) {
        const auto& r = plans[p];
        SC_ASSERT((int)r.pos_end.size() == dof, "all plans of a batch must have the same dof");
        for (int k = 0; k < dof; ++k) {
            const size_t o = (size_t)p * dof + k;
            p0[o] = r.pos_start[k]; p1[o] = r.pos_end[k]; v0[o] = r.vel_start[k]; v1[o] = r.vel_end[k];
            alo[o] = r.acc_min[k]; ahi[o] = r.acc_max[k];
        }
        for (int i = 0; i <= N; ++i) {
            auto [lo, hi] = r.vel_lim((double)i / N);
            for (int k = 0; k < dof; ++k) {
                // the reference's LinearJointVelocityVarying is built with size-1 limit vectors (:1179);
                // a 1-vector is broadcast over the joints
                vlo[((size_t)p * (N + 1) + i) * dof + k] = lo((size_t)lo.rows() == 1 ? 0 : k);
                vhi[((size_t)p * (N + 1) + i) * dof + k] = hi((size_t)hi.rows() == 1 ? 0 : k);
            }
        }
    }
    std::vector<double> K((size_t)P * (N + 1) * 2), x((size_t)P * (N + 1)), u((size_t)P * N), t((size_t)P * (N + 1));
    std::vector<int32_t> status(P);
    ctx.check(sc_toppra_hermite_batch_host(ctx.get(), P, dof, N, p0.data(), p1.data(), v0.data(), v1.data(), vlo.data(), vhi.data(), 1,
                                           alo.data(), ahi.data(), 0.0, 0.0, K.data(), x.data(), u.data(), t.data(), status.data()),
              "sc_toppra_hermite_batch_host");
    int max_len = 1;
    for (int p = 0; p < P; ++p) {
        SC_ASSERT(status[p] == 0, "TOPP-RA failed");  // the reference only SC_ASSERTs the return code (:1227)
        max_len = std::max(max_len, (int)std::ceil(t[(size_t)p * (N + 1) + N] / (double)dt) + 1);
    }
    std::vector<float> pos((size_t)P * dof * max_len), vel(pos.size()), acc(pos.size());
    std::vector<double> times((size_t)P * max_len);
    std::vector<int32_t> length(P);
    ctx.check(sc_toppra_sample_batch_host(ctx.get(), P, dof, N, p0.data(), p1.data(), v0.data(), v1.data(), x.data(), t.data(), (double)dt,
                                          max_len, pos.data(), vel.data(), acc.data(), times.data(), length.data()),
              "sc_toppra_sample_batch_host");
    std::vector<velocity_profile> out;
    out.reserve(P);
    for (int p = 0; p < P; ++p) {
        const int L = std::min(length[p], max_len);
        std::vector<VectorXf> ps(dof, VectorXf::Zero(L)), vs(dof, VectorXf::Zero(L)), as(dof, VectorXf::Zero(L));
        toppra_compat::Vector tm(L);
        for (int k = 0; k < dof; ++k)
            for (int j = 0; j < L; ++j) {
                const size_t o = ((size_t)p * dof + k) * max_len + j;
                ps[k](j) = pos[o]; vs[k](j) = vel[o]; as[k](j) = acc[o];
            }
        for (int j = 0; j < L; ++j) tm(j) = times[(size_t)p * max_len + j];
        out.emplace_back(std::move(ps), std::move(vs), std::move(as), std::move(tm));
    }
    return out;
}

// Same argument order as the reference: END before START (sea_current.hpp:1191-1199).
template <int N>
velocity_profile gen_vel_prof(const VectorNd<N>& pos_end, const VectorNd<N>& pos_start, const VectorNd<N>& vel_end,
                              const VectorNd<N>& vel_start, const vel_lim_func& vel_lim, const VectorNd<N>& acc_min,
                              const VectorNd<N>& acc_max, const float dt = 0.02f) {
    static_assert(N >= 1, "gen_vel_prof needs at least one degree of freedom");
    plan_request r;
    for (int k = 0; k < N; ++k) {
        r.pos_end.push_back(pos_end(k)); r.pos_start.push_back(pos_start(k));
        r.vel_end.push_back(vel_end(k)); r.vel_start.push_back(vel_start(k));
        r.acc_min.push_back(acc_min(k)); r.acc_max.push_back(acc_max(k));
    }
    r.vel_lim = vel_lim;
    return std::move(gen_vel_prof_batch({r}, dt)[0]);
}

// ---- wire formats of the example service ---------------------------------------------------
// Request text of examples/zmq_test.cpp:30-59: "acc_min acc_max vel_min vel_max" then one "x y" pair per waypoint.
struct path_request {
    float acc_min = 0, acc_max = 0, vel_min = 0, vel_max = 0;
    std::vector<Vector2f> path;
    float max_x = 0, max_y = 0;   // largest |x|, |y|: the service builds its bounding_rect from them (:61)
};
inline path_request parse_path_request(const std::string& req) {
    path_request r;
    const char* p = req.c_str();
    char* end = nullptr;
    auto next = [&](float& v) {
        v = std::strtof(p, &end);
        const bool ok = end != p;
        p = end;
        return ok;
    };
    if (!(next(r.acc_min) && next(r.acc_max) && next(r.vel_min) && next(r.vel_max))) throw std::invalid_argument("path request: four limits expected");
    float x, y;
    while (next(x) && next(y)) {
        r.max_x = std::max(r.max_x, std::fabs(x));
        r.max_y = std::max(r.max_y, std::fabs(y));
        r.path.push_back(Vector2f(x, y));
    }
    return r;
}

// Reply of the service: a JSON array with one state per sample, the keys of serialize_path_to_json (:1423-1457).  (The
// reference flattens vel / acc through format_vec_vecx, which repeats element [i] of DOF i for every sample (:1414);
// this writes the per-sample values.)  Floats are printed with 9 significant digits, which round-trips float32.
inline std::string serialize_path_to_json(const bezier_spline& spline, const velocity_profile& vel_prof, const arclength_data& arclens,
                                          const std::vector<float>& ang_vel) {
    (void)arclens;
    std::string out = "[";
    char buf[512];
    const size_t n = (size_t)spline.pts.rows();
    for (size_t i = 0; i < n; ++i) {
        std::snprintf(buf, sizeof(buf),
                      "%s{\"acceleration\":%.9g,\"angularVelocity\":%.9g,\"holonomicAngularVelocity\":0.0,\"holonomicRotation\":0.0,"
                      "\"pose\":{\"translation\":{\"x\":%.9g,\"y\":%.9g}},\"time\":%.9g,\"velocity\":%.9g}",
                      i ? "," : "", (double)vel_prof.acc[0](i), (double)ang_vel[i], (double)spline.pts(i, 0), (double)spline.pts(i, 1),
                      (double)(float)vel_prof.time(i), (double)vel_prof.vel[0](i));
        out += buf;
    }
    out += "]";
    return out;
}

}  // namespace turtle::sc
