// Drives sea-current_amd/sea_current.hpp the way the reference's examples drive its header
// (inputs: examples/test.cpp:249-284 obstacles and FMT* query; examples/zmq_test.py:7-10 +
// examples/output.json for the 1-DOF velocity profile).  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>

#include "../../sea-current_amd/sea_current.hpp"

using namespace turtle::sc;

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } \
    } while (0)

// `test_header --serve-recorded-request`: what examples/zmq_test.cpp:25-101 does for the request of examples/zmq_test.py:7-10,
// reply on stdout.
static int serve_recorded_request() {
    const path_request rq = parse_path_request("-0.5 0.5 -0.25 0.25\n0 0\n10 0\n10 10");
    if (rq.path.size() != 3 || rq.acc_min != -0.5f || rq.vel_max != 0.25f || rq.max_x != 10.f || rq.max_y != 10.f) return 2;
    const bounding_rect brq = {rq.max_x, -rq.max_x, rq.max_y, -rq.max_y};
    planning_space sp(brq);
    bezier_spline pad = bezier_spline::from_path(rq.path, sp);
    const arclength_data ad = pad.arclength();
    auto lim = [&](value_type) {
        toppra_compat::Vector lo(1), hi(1);
        lo(0) = rq.vel_min; hi(0) = rq.vel_max;
        return std::make_tuple(lo, hi);
    };
    velocity_profile prof = gen_vel_prof<1>(VectorNd<1>{ad.arclength}, VectorNd<1>{0}, VectorNd<1>{0}, VectorNd<1>{0}, lim,
                                            VectorNd<1>{rq.acc_min}, VectorNd<1>{rq.acc_max});
    bezier_spline re = pad.resample(prof.pos[0], ad, true);
    const std::vector<float> w = re.angular_velocity(prof);
    std::fputs(serialize_path_to_json(re, prof, ad, w).c_str(), stdout);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "--serve-recorded-request") return serve_recorded_request();
    // --- planning path: examples/test.cpp:249-284 ---
    const bounding_rect br = {1, -1, 1, -1};
    planning_space space(br);
    obstacle ob({Vector2f(-0.5, 0), Vector2f(1, 0), Vector2f(1, 1), Vector2f(0, 1)});
    obstacle ob2({Vector2f(0, -0.5), Vector2f(1, 0), Vector2f(1, 1), Vector2f(0, 1)});
    obstacle ob3({Vector2f(-0.6, 0.148), Vector2f(-1, 0.148), Vector2f(-1, 0), Vector2f(-0.6, 0)});
    space.obstacles = {ob, ob2, ob3};
    CHECK(ob.contains(Vector2f(0.5, 0.5)));              // examples/test.cpp:272 prints 1
    CHECK(!ob.contains(Vector2f(-0.9, -0.9)));
    CHECK(std::get<0>(space.is_obstacle(Vector2f(0.5, 0.5))));
    CHECK(space.cost(Vector2f(-0.9, -0.9), Vector2f(-0.8, -0.9)) < 1.0f);
    CHECK(space.cost(Vector2f(-0.9, 0.5), Vector2f(0.9, 0.5)) == FLT_MAX);
    auto path = space.fast_marching_trees(Vector2f(-0.5, 1), Vector2f(1, -1), 200, 1);  // same call as :284
    CHECK(path.has_value());
    CHECK(path->size() > 2);
    CHECK(path->front() == Vector2f(-0.5, 1) && path->back() == Vector2f(1, -1));
    for (size_t i = 1; i + 2 < path->size(); ++i) {  // interior legs join free cell centres
        CHECK(!std::get<0>(space.is_obstacle((*path)[i])));
        CHECK(space.cost((*path)[i], (*path)[i + 1]) < FLT_MAX);
    }
    // a goal inside an obstacle has no path (nullopt, as :1383-1385)
    CHECK(!space.fast_marching_trees(Vector2f(-0.5, 1), Vector2f(0.5, 0.5)).has_value());
    // batched form agrees with the single call
    auto batch = space.plan_batch({Vector2f(-0.5, 1), Vector2f(-0.9, -0.9)}, {Vector2f(1, -1), Vector2f(0.9, -0.9)});
    CHECK(batch.size() == 2 && batch[0].has_value() && batch[1].has_value());
    CHECK(batch[0]->size() == path->size());

    // --- sampling helpers keep their signatures (sea_current.hpp:100-132, 1294-1337) ---
    {
        halton_state st2, st3;
        const std::vector<float> h2 = halton(2, 7, st2), h3 = halton(3, 4, st3);
        const float e2[7] = {0.5f, 0.25f, 0.75f, 0.125f, 0.625f, 0.375f, 0.875f}, e3[4] = {1.0f / 3, 2.0f / 3, 1.0f / 9, 4.0f / 9};
        for (int i = 0; i < 7; ++i) CHECK(h2[i] == e2[i]);
        for (int i = 0; i < 4; ++i) CHECK(h3[i] == e3[i]);
        const std::vector<float> more = halton(2, 2, st2);           // continues where the state left off
        CHECK(more[0] == 0.0625f && more[1] == 0.5625f);
        planning_space sp(br);
        sp.obstacles = space.obstacles;
        sp.free_space_allocations.push_back([](Vector2f) { return true; });
        const point_set pts = sp.sample_free(64);
        CHECK(pts.size() == 64 && pts.count(Vector2f(0, 0)) == 1);
        for (const auto& p : pts) CHECK(p.x() >= -1 && p.x() <= 1 && p.y() >= -1 && p.y() <= 1 && (!std::get<0>(sp.is_obstacle(p)) || (p.x() == 0 && p.y() == 0)));
        const point_set nb = sp.near(Vector2f(0, 0), pts, 0.7f);     // radius: distance <= 0.7^2
        for (const auto& p : nb) CHECK(p.norm() <= 0.49f + 1e-6f && !(p.x() == 0 && p.y() == 0));
        size_t cnt = 0;
        for (const auto& p : pts) cnt += (p.norm() <= 0.49f && !(p.x() == 0 && p.y() == 0));
        CHECK(cnt == nb.size());
        bool threw = false;
        try { planning_space none(br); none.sample_free(8); } catch (const std::logic_error&) { threw = true; }
        CHECK(threw);
    }
    // --- the reference's own FMT* (n = 200, rn = 1: the call of examples/test.cpp:284), batched on the GPU ---
    {
        planning_space sp(br);
        sp.obstacles = space.obstacles;
        sp.free_space_allocations.push_back([](Vector2f) { return true; });
        auto r = sp.fast_marching_trees_sampled({Vector2f(-0.5, 1), Vector2f(-0.9, -0.9)}, {Vector2f(1, -1), Vector2f(0.5, 0.5)}, 200, 1.0f);
        CHECK(r.size() == 2 && r[0].has_value() && !r[1].has_value());       // second goal lies inside an obstacle
        const auto& wp = *r[0];
        CHECK(wp.size() >= 3 && wp.front() == Vector2f(-0.5, 1) && wp.back() == Vector2f(1, -1));
        float total = 0;
        for (size_t i = 0; i + 1 < wp.size(); ++i) { CHECK(sp.cost(wp[i], wp[i + 1]) < FLT_MAX); total += sp.cost(wp[i], wp[i + 1]); }
        CHECK(total >= 2.5f && total < 3.75f);                               // straight-line distance 2.5
        CHECK(sp.x_state.i != 0 && sp.y_state.i != 0);                       // the Halton states advanced (:317-318)
    }
    // --- smoothing: examples/zmq_test.cpp:61-68 on the recorded request (0,0),(10,0),(10,10) ---
    {
        const bounding_rect br2 = {10, -10, 10, -10};
        planning_space free_space(br2);
        bezier_spline pad = bezier_spline::from_path({Vector2f(0, 0), Vector2f(10, 0), Vector2f(10, 10)}, free_space);
        CHECK(pad.n_segments() == 2 && pad.degree() == 3);
        CHECK(pad.n_pts() == 2 * 10001 && pad.pts(0, 0) == 0.0f && std::fabs(pad.pts(20001, 1) - 10.0f) < 1e-6f);   // every leg sampled at 1e-4 (:679)
        CHECK(std::fabs(pad.ctrl_pts[0][1].x() - 5.0f) < 1e-5f && std::fabs(pad.ctrl_pts[1][2].y() - 5.0f) < 1e-5f);
        const arclength_data ad = pad.arclength();
        CHECK(std::fabs(ad.arclength - 21.38861656f) < 2e-5f);                 // output.json: arclength.arclength
        CHECK(ad.segments.size() == 2 && ad.segments[0].rows() == 101);
        CHECK(std::fabs(ad.positions[0](100) - 1.0f) < 1e-6f);
    }
    // --- shrink_tangent on its own (sea_current.hpp:575-596): a wall across the tangent cuts it there; from_path applies the same ---
    {
        planning_space walled(bounding_rect{10, -10, 10, -10});
        walled.obstacles = {obstacle({Vector2f(2, -1), Vector2f(2, 1), Vector2f(3, 1), Vector2f(3, -1)})};
        const Vector2f cut = bezier_spline::shrink_tangent(Vector2f(10, 0), Vector2f(0, 0), 0.5f, walled);
        CHECK(std::fabs(cut.x() - 2.0f) < 1e-6f && std::fabs(cut.y()) < 1e-6f);
        const Vector2f kept = bezier_spline::shrink_tangent(Vector2f(0, 3), Vector2f(0, 0), 0.5f, walled);
        CHECK(kept.x() == 0.0f && kept.y() == 1.5f);
        bezier_spline pad = bezier_spline::from_path({Vector2f(0, 0), Vector2f(1.9f, 0), Vector2f(1.9f, 8)}, walled, 0.0f);
        CHECK(pad.ctrl_pts[0][1].x() <= 2.0f + 1e-6f);
        // Q_cache (:393): per segment the inverse DFT of the control points; with omega_table the reference's Bernstein-Fourier
        // sum  B(s) = sum_k Re(Q_k (1 + s (omega_k - 1))^degree)  (:736-743) must give the sampled points
        CHECK(pad.Q_cache.size() == 2 && pad.Q_cache[0].rows() == 4);
        {
            const auto om3 = bezier_spline::omega_table(3);
            for (int i : {0, 2500, 7000, 10000}) {
                const float sp = pad.positions[1](i);
                std::complex<float> bx(0, 0), by(0, 0);
                for (int k = 0; k < 4; ++k) {
                    const std::complex<float> b = std::complex<float>(1, 0) + sp * (om3[k] - std::complex<float>(1, 0));
                    bx += pad.Q_cache[1](k, 0) * (b * b * b); by += pad.Q_cache[1](k, 1) * (b * b * b);
                }
                CHECK(std::fabs(bx.real() - pad.pts(10001 + i, 0)) < 2e-5f && std::fabs(by.real() - pad.pts(10001 + i, 1)) < 2e-5f);
            }
        }
        const auto om = bezier_spline::omega_table(3);      // the 4th roots of unity, clockwise (:1096-1106)
        CHECK(om.size() == 4 && std::abs(om[0] - std::complex<float>(1, 0)) < 1e-6f && std::abs(om[1] - std::complex<float>(0, -1)) < 1e-6f &&
              std::abs(om[2] - std::complex<float>(-1, 0)) < 1e-6f);
    }
    // --- velocity profile: examples/zmq_test.cpp:69-88 with the recorded arclength of output.json ---
    const double L = 21.38861656188965;
    auto vel_lim = [&](value_type) {
        toppra_compat::Vector lo(1), hi(1);
        lo(0) = -0.25; hi(0) = 0.25;
        return std::make_tuple(lo, hi);
    };
    velocity_profile prof = gen_vel_prof<1>(VectorNd<1>{L}, VectorNd<1>{0}, VectorNd<1>{0}, VectorNd<1>{0}, vel_lim,
                                            VectorNd<1>{-0.5}, VectorNd<1>{0.5});
    CHECK(prof.pos.size() == 1 && prof.pos[0].rows() == 4328);      // output.json: 4328 samples
    CHECK(std::fabs(prof.time(4327) - 86.55526) < 1e-4);            // output.json: T = 86.55526
    float vmax = 0, amax = 0;
    for (int i = 0; i < 4328; ++i) { vmax = std::max(vmax, prof.vel[0](i)); amax = std::max(amax, std::fabs(prof.acc[0](i))); }
    CHECK(std::fabs(vmax - 0.253140f) < 2e-6f);                     // BASELINE.md extrema
    CHECK(std::fabs(amax - 0.497336f) < 2e-6f);
    // --- resample + angular velocity: examples/zmq_test.cpp:91-93 -------------------------------------
    {
        const bounding_rect br2 = {10, -10, 10, -10};
        planning_space free_space(br2);
        bezier_spline pad = bezier_spline::from_path({Vector2f(0, 0), Vector2f(10, 0), Vector2f(10, 10)}, free_space);
        const arclength_data ad = pad.arclength();
        velocity_profile prof2 = gen_vel_prof<1>(VectorNd<1>{ad.arclength}, VectorNd<1>{0}, VectorNd<1>{0}, VectorNd<1>{0}, vel_lim,
                                                 VectorNd<1>{-0.5}, VectorNd<1>{0.5});
        bezier_spline re = pad.resample(prof2.pos[0], ad, true);
        CHECK(re.n_pts() == (int)prof2.pos[0].rows() && re.n_pts() == 4328);
        CHECK(prof2.pos[0](0) == 0.0f && prof2.pos[0](4327) == ad.arclength);   // the nudge pins the ends (:903-904)
        float xmax = -1e9f, ymin = 1e9f;
        for (int i = 0; i < re.n_pts(); ++i) { xmax = std::max(xmax, re.pts(i, 0)); ymin = std::min(ymin, re.pts(i, 1)); }
        CHECK(std::fabs(re.pts(0, 0) - 0.00401974f) < 2e-5f && std::fabs(re.pts(4327, 0) - 9.999999f) < 2e-5f && std::fabs(re.pts(4327, 1) - 9.9983425f) < 2e-5f);   // output.json: pos_x/pos_y ends (the fit misses t = 0 by 2.7e-4)
        CHECK(std::fabs(xmax - 11.571348f) < 5e-5f && std::fabs(ymin + 1.5713487f) < 5e-5f);   // output.json: pos_x max, pos_y min
        const std::vector<float> w = re.angular_velocity(prof2);
        float wmax = 0;
        for (float v : w) wmax = std::max(wmax, std::fabs(v));
        CHECK(w.size() == 4328 && std::fabs(wmax - 0.08215085f) < 2e-6f);                      // output.json: max |ang_vel|
    }
    std::printf("sea_current.hpp: planning path + velocity profile OK (%zu waypoints, %d samples)\n", path->size(), 4328);
    return 0;
}
