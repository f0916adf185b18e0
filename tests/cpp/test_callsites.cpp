// Compile- and run-time compatibility of the successor header with the way the reference's programs spell things:
// toppra::Vector v{1} with v(0, 0) access inside a [](toppra::value_type) limit callback, Eigen::Vector<value_type, 1>{x}
// end points, gen_vel_prof<1>(end, start, ...), bezier_spline::bezier_curve(ctrl, step), join_splines, hodograph,
// arclength, chebfit / chebeval on pts.col(c), pts(i, c), from_path / resample / angular_velocity / serialize_path_to_json
// (the reference's examples/test.cpp and examples/zmq_test.cpp use these; only the spellings are taken from there).
// The curve, the limits, the order of operations and every check are this file's own: an S-shaped path of four waypoints
// with closed forms to compare against, and the one request the reference recorded (examples/output.json).
//   test_callsites              all checks, exit code 0 when they hold
//   test_callsites --serve      the reply to the recorded request, on stdout
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>

#include "../../sea-current_amd/sea_current.hpp"

using namespace turtle::sc;

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } \
    } while (0)

static bool near(float a, float b, float tol) { return std::fabs(a - b) <= tol; }

struct request {
    std::vector<Vector2f> waypoints;
    float half_width, half_height;       // the planning space is [-w, w] x [-h, h]
    float a_lo, a_hi, v_lo, v_hi;        // acceleration and velocity bounds along the path
};

// request -> reply, the way a service built on the header answers: smooth, measure, time-parametrise, resample, serialise
static std::string answer(const request& rq) {
    planning_space world(bounding_rect{rq.half_width, -rq.half_width, rq.half_height, -rq.half_height});
    bezier_spline smooth = bezier_spline::from_path(rq.waypoints, world);
    const arclength_data table = smooth.arclength();

    auto bounds = [&rq](toppra::value_type /*gridpoint*/) {
        toppra::Vector lo{1}, hi{1};
        hi(0, 0) = rq.v_hi;
        lo(0, 0) = rq.v_lo;
        return std::make_tuple(lo, hi);
    };
    const Eigen::Vector<value_type, 1> from{0}, to{table.arclength}, rest{0};
    const Eigen::Vector<value_type, 1> brake{rq.a_lo}, thrust{rq.a_hi};
    velocity_profile timing = gen_vel_prof<1>(to, from, rest, rest, bounds, brake, thrust);

    bezier_spline timed = smooth.resample(timing.pos[0], table, true);
    return serialize_path_to_json(timed, timing, table, timed.angular_velocity(timing));
}

static const request RECORDED = {{Vector2f(0, 0), Vector2f(10, 0), Vector2f(10, 10)}, 10, 10, -0.5f, 0.5f, -0.25f, 0.25f};

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "--serve") {
        std::fputs(answer(RECORDED).c_str(), stdout);
        return 0;
    }

    // ---- geometry helpers keep their reference meaning ----
    CHECK(pt_dist(Vector2f(-6, 8)) == 10.0f && pt_dist(Vector2f(2, -1), Vector2f(-1, 3)) == 5.0f);
    CHECK(near(dist_pt_line(Vector2f(-1, 1), Vector2f(3, 1), Vector2f(0.5f, -1.5f)), 2.5f, 1e-6f));

    // ---- an S through four waypoints: one cubic per leg, inner control points a third of a tangent away ----
    const Vector2f wp[4] = {Vector2f(0, 0), Vector2f(2, 1), Vector2f(4, -1), Vector2f(6, 0)};
    const Vector2f tan[4] = {calc_start_tangent(wp[0], wp[1], 0.0f), calc_tangent(wp[0], wp[1], wp[2]), calc_tangent(wp[1], wp[2], wp[3]),
                             calc_end_tangent(wp[2], wp[3])};
    // start tangent: along the given angle, half the first leg long; end tangent: along the last leg, half as long
    CHECK(near(tan[0].x(), 0.5f * std::sqrt(5.0f), 1e-5f) && near(tan[0].y(), 0.0f, 1e-6f));
    CHECK(near(tan[3].x(), 1.0f, 1e-6f) && near(tan[3].y(), 0.5f, 1e-6f));
    const float third = 1.0f / 3.0f;
    std::vector<bezier_spline> legs;
    std::vector<std::vector<Vector2f>> polygons;
    for (int leg = 0; leg < 3; ++leg) {
        std::vector<Vector2f> poly = {wp[leg], wp[leg] + third * tan[leg], wp[leg + 1] - third * tan[leg + 1], wp[leg + 1]};
        legs.push_back(bezier_spline::bezier_curve(poly, 0.001));
        polygons.push_back(poly);
        const bezier_spline& c = legs.back();
        CHECK(c.n_pts() == 1001 && c.n_segments() == 1 && c.degree() == 3);
        CHECK(c.pts(0, 0) == poly[0].x() && c.pts(0, 1) == poly[0].y());
        CHECK(near(c.pts(1000, 0), poly[3].x(), 1e-5f) && near(c.pts(1000, 1), poly[3].y(), 1e-5f));
        const Vector2f at_half = (poly[0] + 3.0f * poly[1] + 3.0f * poly[2] + poly[3]) / 8.0f;      // de Casteljau at 1/2
        CHECK(near(c.pts(500, 0), at_half.x(), 1e-5f) && near(c.pts(500, 1), at_half.y(), 1e-5f));
    }
    bezier_spline s_curve = bezier_spline::join_splines(legs);
    const int total = s_curve.n_pts();
    CHECK(total == 3003 && s_curve.n_segments() == 3 && s_curve.positions.size() == 3);
    CHECK(near(s_curve.pts(total - 1, 0), 6.0f, 1e-5f) && near(s_curve.pts(total - 1, 1), 0.0f, 1e-5f));

    // ---- derivative curve: end values 3 (P1 - P0) and 3 (P3 - P2) of every leg, a central difference inside ----
    bezier_spline velocity_curve = s_curve.hodograph();
    CHECK(velocity_curve.n_pts() == total && velocity_curve.degree() == 2 && velocity_curve.n_segments() == 3);
    for (int leg = 0; leg < 3; ++leg) {
        const Vector2f d0 = 3.0f * (polygons[leg][1] - polygons[leg][0]), d1 = 3.0f * (polygons[leg][3] - polygons[leg][2]);
        CHECK(near(velocity_curve.pts(1001 * leg, 0), d0.x(), 1e-4f) && near(velocity_curve.pts(1001 * leg, 1), d0.y(), 1e-4f));
        CHECK(near(velocity_curve.pts(1001 * leg + 1000, 0), d1.x(), 1e-4f) && near(velocity_curve.pts(1001 * leg + 1000, 1), d1.y(), 1e-4f));
        const int m = 1001 * leg + 300;
        CHECK(near(velocity_curve.pts(m, 1), (s_curve.pts(m + 1, 1) - s_curve.pts(m - 1, 1)) / 0.002f, 5e-3f));
    }

    // ---- arclength: the polyline through the samples bounds it from below and converges to it; tables are cumulative ----
    const arclength_data table = s_curve.arclength(0.01);
    double polyline = 0;
    for (int i = 1; i < total; ++i) polyline += std::hypot(s_curve.pts(i, 0) - s_curve.pts(i - 1, 0), s_curve.pts(i, 1) - s_curve.pts(i - 1, 1));
    CHECK(table.arclength >= polyline - 1e-4 && table.arclength - polyline < 1e-3);
    CHECK(table.segments.size() == 3);
    float legs_sum = 0;
    for (size_t k = 0; k < legs.size(); ++k) {
        const VectorXf cum = table.segments[k];
        for (int i = 1; i < (int)cum.rows(); ++i) CHECK(cum(i) > cum(i - 1));
        legs_sum += legs[k].arclength(0.01).arclength;
    }
    CHECK(near(legs_sum, table.arclength, 1e-4f));

    // ---- y over x is single-valued on this S (x grows all the way): a 14-column Chebyshev fit follows it ----
    constexpr int columns = 14;
    const chebpoly fit = chebfit(s_curve.pts.col(0), s_curve.pts.col(1), columns);
    const VectorXf refit = chebeval(s_curve.pts.col(0), fit, columns);
    CHECK(fit.coeffs.rows() == columns && fit.xmin == 0.0f && near(fit.xmax, 6.0f, 1e-5f) && (int)refit.rows() == total);
    float fit_err = 0;
    for (int i = 0; i < total; ++i) {
        if (i) CHECK(s_curve.pts(i, 0) >= s_curve.pts(i - 1, 0));
        fit_err = std::max(fit_err, std::fabs(refit(i) - s_curve.pts(i, 1)));
    }
    CHECK(fit_err < 0.05f && fit_err > 0.01f);      // the least-squares optimum for 14 columns is 0.032 (numpy), 0.053 for 10

    // ---- a profile under limits that depend on where along the path the robot is: cruise, a slow middle third, cruise ----
    const value_type cruise = 1.5, crawl = 0.6, push = 2.0;
    auto zone_limits = [cruise, crawl](toppra::value_type where) {     // `where` is the gridpoint in [0, 1]
        const value_type cap = (where > 1.0 / 3 && where < 2.0 / 3) ? crawl : cruise;
        toppra::Vector floor_{1}, ceil_{1};
        floor_(0, 0) = -cap;
        ceil_(0, 0) = cap;
        return std::make_tuple(floor_, ceil_);
    };
    const Eigen::Vector<value_type, 1> origin{0}, length{table.arclength}, standstill{0};
    const Eigen::Vector<value_type, 1> decel{-push}, accel{push};
    velocity_profile ride = gen_vel_prof<1>(length, origin, standstill, standstill, zone_limits, decel, accel);
    const VectorXf along = ride.pos[0], speed = ride.vel[0], change = ride.acc[0];
    const int samples = (int)along.rows();
    CHECK(samples > 20 && (int)ride.time.rows() == samples && (int)speed.rows() == samples && (int)change.rows() == samples);
    // the path is q(s) = L (3 s^2 - 2 s^3) (zero end tangents): the middle third of s is q / L in (7/27, 20/27)
    float top_outer = 0, top_middle = 0, top_change = 0;
    for (int i = 0; i < samples; ++i) {
        const float frac = along(i) / table.arclength;
        const bool strictly_middle = frac > 0.30f && frac < 0.70f;
        if (strictly_middle) top_middle = std::max(top_middle, speed(i));
        top_outer = std::max(top_outer, speed(i));
        top_change = std::max(top_change, std::fabs(change(i)));
        if (i) CHECK(ride.time(i) > ride.time(i - 1) && along(i) >= along(i - 1) - 1e-3f);   // the position spline may dip by a hair at standstill (what resample's nudge repairs)
    }
    CHECK(top_outer <= 1.03f * (float)cruise && top_outer > 0.9f * (float)cruise);      // the cruise bound is reached, with at most the spline's overshoot
    CHECK(top_middle <= 1.03f * (float)crawl && top_change <= 1.15f * (float)push);     // the time spline overshoots where the speed bound steps (oracle: 2.24 for 2)
    CHECK(near(along(samples - 1), table.arclength, 1e-3f) && near(speed(samples - 1), 0.0f, 2e-3f) && near(along(0), 0.0f, 1e-6f));

    // ---- back onto the curve at the profile's positions ----
    bezier_spline ridden = s_curve.resample(ride.pos[0], table, true);    // repairs the positions in place, as the reference does
    CHECK(ridden.n_pts() == samples);
    CHECK(near(ridden.pts(0, 0), 0.0f, 1e-3f) && near(ridden.pts(samples - 1, 0), 6.0f, 1e-3f) && near(ridden.pts(samples - 1, 1), 0.0f, 1e-3f));
    const std::vector<float> turn = ridden.angular_velocity(ride), turn2 = ridden.angular_velocity2(ride);
    CHECK((int)turn.size() == samples && (int)turn2.size() == samples && turn2.front() == 0.0f && turn2.back() == 0.0f);
    bool turns_both_ways = false;      // an S curves left, then right
    for (int i = 1; i < samples; ++i) turns_both_ways = turns_both_ways || turn[i] * turn[1 + samples / 8] < 0;
    CHECK(turns_both_ways);

    // ---- the recorded request: one state per sample of the recording ----
    const std::string reply = answer(RECORDED);
    size_t states = 0;
    for (size_t at = reply.find("\"time\""); at != std::string::npos; at = reply.find("\"time\"", at + 1)) ++states;
    CHECK(states == 4328);
    std::printf("call sites OK (%d curve samples, %d profile samples, %zu reply states)\n", total, samples, states);
    return 0;
}
