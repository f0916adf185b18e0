// The reference's callers, spelled their way, against the successor header: the vocabulary of
// examples/zmq_test.cpp:61-93 (service pipeline), examples/test.cpp:83-139 (control points by hand, bezier_curve,
// join_splines, hodograph, arclength, chebfit / chebeval) and examples/test.cpp:184-213 (position-dependent velocity
// limits through toppra::Vector / toppra::value_type and Eigen::Vector<value_type, 1>).  Own text and own checks: the
// examples assert nothing; the recorded run (examples/output.json) and closed forms are what is checked here.
//   test_callsites              all checks, exit code 0 when they hold
//   test_callsites --serve      the service's reply for the recorded request, on stdout
#include <cmath>
#include <cstdio>
#include <string>

#include "../../sea-current_amd/sea_current.hpp"

using namespace turtle::sc;

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } \
    } while (0)

// what the service does with one request (examples/zmq_test.cpp:61-95)
static std::string serve(const std::vector<Vector2f>& path, float max_x, float max_y, float acc_min_val, float acc_max_val,
                         float vel_min_val, float vel_max_val) {
    const bounding_rect br = {max_x, -max_x, max_y, -max_y};
    planning_space space(br);
    bezier_spline pad = bezier_spline::from_path(path, space);

    const arclength_data ad = pad.arclength();
    const Eigen::Vector<value_type, 1> pos_end{ad.arclength};
    const Eigen::Vector<value_type, 1> pos_start{0};
    const Eigen::Vector<value_type, 1> vel_end{0};
    const Eigen::Vector<value_type, 1> vel_start{0};
    const Eigen::Vector<value_type, 1> acc_min{acc_min_val};
    const Eigen::Vector<value_type, 1> acc_max{acc_max_val};

    auto vel_lim = [&](toppra::value_type) {
        toppra::Vector lower{1};
        toppra::Vector upper{1};
        lower(0, 0) = vel_min_val;
        upper(0, 0) = vel_max_val;
        return std::make_tuple(lower, upper);
    };

    velocity_profile prof = gen_vel_prof<1>(pos_end, pos_start, vel_end, vel_start, vel_lim, acc_min, acc_max);
    bezier_spline re = pad.resample(prof.pos[0], ad, true);
    const std::vector<float> ang_vel = re.angular_velocity(prof);
    return serialize_path_to_json(re, prof, ad, ang_vel);
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "--serve") {
        std::fputs(serve({Vector2f(0, 0), Vector2f(10, 0), Vector2f(10, 10)}, 10, 10, -0.5f, 0.5f, -0.25f, 0.25f).c_str(), stdout);
        return 0;
    }
    // --- two cubics built by hand from the tangent heuristics and joined (examples/test.cpp:83-120) ---
    const Vector2f W_0(0, 0), W_1(0.5, 0.5), W_2(1, 0);
    const Vector2f T_0 = calc_start_tangent(W_0, W_1, 0);
    const Vector2f T_1 = calc_tangent(W_0, W_1, W_2);
    const Vector2f T_2 = calc_end_tangent(W_1, W_2);
    const float half_leg = 0.5f * std::sqrt(0.5f);
    CHECK(std::fabs(T_0.x() - half_leg) < 1e-6f && std::fabs(T_0.y()) < 1e-6f);        // magnitude: half the leg, along theta = 0
    CHECK(std::fabs(T_1.x() - half_leg) < 1e-6f && std::fabs(T_1.y()) < 1e-6f);        // symmetric corner: tangent along +x
    CHECK(std::fabs(T_2.x() - 0.25f) < 1e-6f && std::fabs(T_2.y() + 0.25f) < 1e-6f);   // end tangent along the last leg
    constexpr float k = 0.2;
    std::vector<Vector2f> ctrl_pts, ctrl_pts2;
    ctrl_pts.push_back(W_0);
    ctrl_pts.push_back(W_0 + (k) * T_0);
    ctrl_pts.push_back(W_1 - (k) * T_1);
    ctrl_pts.push_back(W_1);
    ctrl_pts2.push_back(W_1);
    ctrl_pts2.push_back(W_1 + (k) * T_1);
    ctrl_pts2.push_back(W_2 - (k) * T_2);
    ctrl_pts2.push_back(W_2);

    bezier_spline bs = bezier_spline::bezier_curve(ctrl_pts, 0.0001);
    bezier_spline bs2 = bezier_spline::bezier_curve(ctrl_pts2, 0.0001);
    CHECK(bs.n_pts() == 10001 && bs.n_segments() == 1 && bs.degree() == 3);
    CHECK(bs.pts(0, 0) == W_0.x() && bs.pts(0, 1) == W_0.y());                          // a Bezier curve starts and ends on
    CHECK(std::fabs(bs.pts(10000, 0) - W_1.x()) < 1e-6f && std::fabs(bs.pts(10000, 1) - W_1.y()) < 1e-6f);   // its end points
    bs = bezier_spline::join_splines({bs, bs2});
    CHECK(bs.n_pts() == 20002 && bs.n_segments() == 2 && bs.positions.size() == 2);
    CHECK(std::fabs(bs.pts(20001, 0) - W_2.x()) < 1e-6f && std::fabs(bs.pts(20001, 1) - W_2.y()) < 1e-6f);
    {   // midpoint of the first cubic by hand: (P0 + 3 P1 + 3 P2 + P3) / 8
        const Vector2f mid = (ctrl_pts[0] + 3.0f * ctrl_pts[1] + 3.0f * ctrl_pts[2] + ctrl_pts[3]) / 8.0f;
        CHECK(std::fabs(bs.pts(5000, 0) - mid.x()) < 1e-6f && std::fabs(bs.pts(5000, 1) - mid.y()) < 1e-6f);
    }

    std::vector<float> x(bs.n_pts()), y(bs.n_pts());
    for (int i = 0; i < bs.n_pts(); ++i) {
        x[i] = bs.pts(i, 0);
        y[i] = bs.pts(i, 1);
    }

    bezier_spline deriv = bs.hodograph();
    CHECK(deriv.n_pts() == bs.n_pts() && deriv.degree() == 2 && deriv.n_segments() == 2);
    {   // B'(0) = 3 (P1 - P0), B'(1) = 3 (P3 - P2); and a central difference of the sampled curve in the middle
        CHECK(std::fabs(deriv.pts(0, 0) - 3 * k * T_0.x()) < 1e-5f && std::fabs(deriv.pts(0, 1) - 3 * k * T_0.y()) < 1e-5f);
        CHECK(std::fabs(deriv.pts(10000, 0) - 3 * k * T_1.x()) < 1e-5f);
        const float fd = (bs.pts(5001, 0) - bs.pts(4999, 0)) / 2e-4f;
        CHECK(std::fabs(deriv.pts(5000, 0) - fd) < 2e-3f);
    }

    float arclen = bs.arclength(0.01).arclength;
    float chord = 0;   // the polyline through the 20002 samples is a lower bound that converges to the arclength
    for (int i = 0; i + 1 < bs.n_pts(); ++i) chord += std::hypot(x[i + 1] - x[i], y[i + 1] - y[i]);
    CHECK(arclen >= chord - 1e-5f && arclen - chord < 1e-4f);
    CHECK(std::fabs(bs.arclength(0.01).arclength - (bs.arclength().segments[0](100) + bs2.arclength(0.01).arclength)) < 1e-5f);
    VectorXf arcs = bs.arclength().segments[0];
    float prev = -1;
    for (float arc : arcs) { CHECK(arc > prev); prev = arc; }                           // a cumulative table

    constexpr int degree = 10;
    chebpoly b = chebfit(bs.pts.col(0), bs.pts.col(1), degree);
    VectorXf y_hat = chebeval(bs.pts.col(0), b, degree);
    CHECK(b.coeffs.rows() == degree && b.xmin == 0.0f && std::fabs(b.xmax - 1.0f) < 1e-6f);
    CHECK(y_hat.rows() == (size_t)bs.n_pts() || (int)y_hat.rows() == bs.n_pts());
    {   // y(x) along this curve is smooth: a 10-column fit follows it closely
        double worst = 0;
        for (int i = 0; i < bs.n_pts(); ++i) worst = std::max(worst, (double)std::fabs(y_hat(i) - y[i]));
        CHECK(worst < 2e-2);
    }
    CHECK(std::fabs(dist_pt_line(Vector2f(0, 0), Vector2f(2, 0), Vector2f(1, 3)) - 3.0f) < 1e-6f);
    CHECK(pt_dist(Vector2f(3, 4)) == 5.0f && pt_dist(Vector2f(1, 1), Vector2f(4, 5)) == 5.0f);

    // --- position-dependent velocity limits (examples/test.cpp:184-213) ---
    Eigen::Vector<value_type, 1> pos_end{arclen};
    Eigen::Vector<value_type, 1> pos_start{0};
    Eigen::Vector<value_type, 1> vel_end{0};
    Eigen::Vector<value_type, 1> vel_start{0};
    Eigen::Vector<value_type, 1> acc_min{-40};
    Eigen::Vector<value_type, 1> acc_max{40};
    auto vel_lim = [](toppra::value_type time) {
        toppra::Vector lower{1};
        toppra::Vector upper{1};
        value_type slow = 4;
        value_type fast = 6;
        if (time > 0.5) {
            lower(0, 0) = -slow;
            upper(0, 0) = slow;
        } else {
            lower(0, 0) = -fast;
            upper(0, 0) = fast;
        }
        return std::make_tuple(lower, upper);
    };
    velocity_profile prof = gen_vel_prof<1>(pos_end, pos_start, vel_end, vel_start, vel_lim, acc_min, acc_max);
    VectorXf pos_plot = prof.pos[0];
    VectorXf vel_plot = prof.vel[0];
    VectorXf acc_plot = prof.acc[0];
    const int ns = (int)pos_plot.rows();
    CHECK(ns > 10 && (int)prof.time.rows() == ns);
    {
        float vmax_first = 0, vmax_second = 0, amax = 0;
        for (int i = 0; i < ns; ++i) {
            // the limit is a function of the gridpoint s = q / L of a path whose shape is 3 s^2 - 2 s^3 (zero end tangents)
            const bool second = pos_plot(i) > 0.5f * arclen;
            (second ? vmax_second : vmax_first) = std::max(second ? vmax_second : vmax_first, vel_plot(i));
            amax = std::max(amax, std::fabs(acc_plot(i)));
            if (i) CHECK(prof.time(i) > prof.time(i - 1) && pos_plot(i) >= pos_plot(i - 1) - 1e-5f);
        }
        CHECK(vmax_first <= 6.0f * 1.03f && vmax_second <= 4.0f * 1.03f);   // spline overshoot between knots stays within 3 %
        CHECK(amax <= 40.0f * 1.03f);
        CHECK(std::fabs(pos_plot(ns - 1) - arclen) < 1e-4f && std::fabs(vel_plot(ns - 1)) < 1e-3f);
    }
    arclength_data ad = bs.arclength();
    bezier_spline re = bs.resample(pos_plot, ad, true);
    CHECK(re.n_pts() == ns);
    CHECK(std::fabs(re.pts(ns - 1, 0) - W_2.x()) < 1e-3f && std::fabs(re.pts(ns - 1, 1) - W_2.y()) < 1e-3f);
    const std::vector<float> w = re.angular_velocity(prof), w2 = re.angular_velocity2(prof);
    CHECK((int)w.size() == ns && (int)w2.size() == ns && w2[0] == 0.0f && w2[ns - 1] == 0.0f);

    // --- the service pipeline on the recorded request: the reply has one state per sample of the recording ---
    const std::string reply = serve({Vector2f(0, 0), Vector2f(10, 0), Vector2f(10, 10)}, 10, 10, -0.5f, 0.5f, -0.25f, 0.25f);
    size_t states = 0;
    for (size_t p = reply.find("\"time\""); p != std::string::npos; p = reply.find("\"time\"", p + 1)) ++states;
    CHECK(states == 4328);
    std::printf("call sites OK (%d curve samples, %d profile samples, %zu reply states)\n", bs.n_pts(), ns, states);
    return 0;
}
