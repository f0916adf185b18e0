// toppra.hip -- batched TOPP-RA over independent plans (gfx950, fp64).
//
// Takes over the arithmetic that gen_vel_prof<N> (sea_current.hpp:1191-1265) delegates to
// hungpham2511/toppra: LinearConstraint::computeParams of LinearJointVelocity(Varying) and
// LinearJointAcceleration (Interpolation discretisation), the backward controllable-set sweep and
// the forward greedy sweep of TOPPRA::computePathParametrization(0,0), the knot times of
// parametrizer::Spline, and (toppra_sample_kernel) its clamped cubic spline + uniform sampling.
// The maths is restated in oracle/toppra_oracle.c, which is pinned to the reference's recorded
// output; this file must agree with that oracle (built with -ffp-contract=off so it does so to
// round-off, far inside the 1e-5 relative bar).
//
// One wavefront per plan; computeParams is fused (nothing is read from HBM but the 4*dof Hermite coefficients and the
// limits).  The stage LPs are solved exactly by eliminating u over all (upper, lower) pairs of rows
// alpha u + beta x <= gamma; everything that does not depend on the previous stage is done for all stages at once
// before the sweeps (see toppra_kernel), which leaves a few dozen instructions per stage on the two dependent chains.
// This kernel is latency-bound, not HBM-bound.
#include "sc_internal.h"

#define TP_MAXSD 1e8
#define TP_NEARLY_ZERO 1e-8
#define TP_LP_TOL 1e-9
#define TP_MAXDOF 16
#define TP_MAXROWS (4 * TP_MAXDOF + 2)

struct toppra_args {
    int P, dof, N;
    const double *p0, *p1, *v0, *v1, *vlo, *vhi, *alo, *ahi;
    int vlim_per_stage;
    int lds_limits;   // 1: all velocity limits, the acceleration limits and K are staged in LDS (dynamic LDS sized by the host)
    double sd_start, sd_end;
    double *K, *x, *u, *t;
    int32_t* status;
};

// Wave-wide min / max of doubles through DPP lane moves (a few cycles each) instead of ds_bpermute shuffles (an LDS
// crossbar round trip each): a sweep is a chain of ~10 dependent reductions per stage, and the shuffle latency alone was
// most of a stage.  Butterfly inside each row of 16 lanes (quad_perm, half mirror, mirror), then row_bcast:15 / :31
// carry the row results upwards; lane 63 ends with the result, which is broadcast through an SGPR.  min / max are exact
// and order independent, so results are bit-identical to any other reduction order.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// v_min_f64 / v_max_f64 as they are: fmin / fmax compile to the same instruction behind a canonicalising v_max_f64 x, x
// per operand (signalling-NaN semantics), which tripled the instruction count of the reductions on the sweeps' chains.
// A quiet NaN operand loses against a number in both forms; there are no signalling NaNs here.
__device__ __forceinline__ double dmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double dmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <bool IS_MIN>
__device__ __forceinline__ double wave_minmax(double v) {
    auto op = [](double a, double b) { return IS_MIN ? dmin(a, b) : dmax(a, b); };
    v = op(v, dpp_f64<0xB1, 0xF>(v));    // quad_perm [1,0,3,2]
    v = op(v, dpp_f64<0x4E, 0xF>(v));    // quad_perm [2,3,0,1]
    v = op(v, dpp_f64<0x141, 0xF>(v));   // row_half_mirror
    v = op(v, dpp_f64<0x140, 0xF>(v));   // row_mirror: every lane of a row holds the row's result
    v = op(v, dpp_f64<0x142, 0xA>(v));   // row_bcast:15 -> rows 1 and 3
    v = op(v, dpp_f64<0x143, 0xC>(v));   // row_bcast:31 -> rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min(double v) { return wave_minmax<true>(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_minmax<false>(v); }

// min / max over the first 16 lanes only (the rows of a stage sit in lanes 0 .. 2 dof, dof <= 7): four DPP steps
template <bool IS_MIN>
__device__ __forceinline__ double row16_minmax(double v) {
    auto op = [](double a, double b) { return IS_MIN ? dmin(a, b) : dmax(a, b); };
    v = op(v, dpp_f64<0xB1, 0xF>(v));    // quad_perm [1,0,3,2]
    v = op(v, dpp_f64<0x4E, 0xF>(v));    // quad_perm [2,3,0,1]
    v = op(v, dpp_f64<0x141, 0xF>(v));   // row_half_mirror
    v = op(v, dpp_f64<0x140, 0xF>(v));   // row_mirror
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 0), hi = __builtin_amdgcn_readlane(__double2hiint(v), 0);
    return __hiloint2double(hi, lo);
}

// ---- the wide kernel (dof > 7: more rows than a row of 16 lanes holds) ----------------------------------------------
// The sweeps are chains of N dependent stages, and a lone wavefront pays every instruction of a stage in full
// (~5 cycles each), so what matters is how few instructions sit ON the chain.  Of a stage's rows only two -- the next
// controllable set [K_lo, K_hi] in the backward pass, the state x in the forward pass -- depend on the previous stage:
//   phase 0 (all stages at once, lanes over stages): the x-bounds of the velocity constraint, and the x-interval left
//           by every (upper, lower) pair of ACCELERATION rows -- 4 dof^2 pairs per stage, the bulk of the old stage --
//           with the alpha == 0 rows; results in LDS per stage;
//   backward chain, per stage: the 2 dof pairs that involve a K row (one lane per joint and discretisation slot:
//           two divisions), a 16-lane min / max, the clamp;
//   forward chain, per stage: one division per row, a 16-lane min / max, the step and its knot time.
// Every bound is computed by the same expression, on the same operands, as in oracle/toppra_oracle.c, and min / max
// do not care about order: K, x, u, t are bit-identical to the previous formulation's.
// A joint contributes two "slots" per stage (collocation at s_i; interpolation: the stage-(i+1) rows expressed at stage
// i), each a pair of rows  +(a u + b x) <= ahi,  -(a u + b x) <= -alo : for a > 0 the first is the upper row.
template <bool LDSLIM>
__global__ void __launch_bounds__(64) toppra_wide_kernel(toppra_args a) {
    __shared__ double s_c1[TP_MAXDOF], s_c2[TP_MAXDOF], s_c3[TP_MAXDOF];
    __shared__ double s_alo[TP_MAXDOF], s_ahi[TP_MAXDOF];
    extern __shared__ double s_dyn[];   // slo [N+1], shi [N+1], gridpoints [N+2], then K [N+1][2] sharing its place with the slots a / b [32][2 dof]
                                        // of phase 0 (K is first written after it), then (LDSLIM) vlo, vhi
    const int lane = threadIdx.x, p = blockIdx.x;
    const int dof = a.dof, N = a.N, n1 = N + 1, ns = 2 * dof;
    double* s_slo = s_dyn;               // x-interval of everything that does not depend on the next set: the velocity
    double* s_shi = s_slo + n1;          // bounds and the acceleration rows among themselves (slo > shi + tol: infeasible)
    double* s_tab = s_shi + n1;          // [N + 2] gridpoints i / N (an fp64 division each: off the chains)
    double* s_K = s_tab + n1 + 1;
    double* s_sa = s_K;                  // phase 0: slots (a, b) of a chunk of 32 stages, [32][ns]
    double* s_sb = s_sa + 32 * ns;
    const int kwords = 2 * n1 > 64 * ns ? 2 * n1 : 64 * ns;
    const int nv = (a.vlim_per_stage ? n1 : 1) * dof;
    double* s_vlo = s_K + kwords;
    double* s_vhi = s_vlo + nv;
    if (LDSLIM) {
        const size_t vo = a.vlim_per_stage ? (size_t)p * n1 * dof : (size_t)p * dof;
        for (int k = lane; k < nv; k += 64) { s_vlo[k] = a.vlo[vo + k]; s_vhi[k] = a.vhi[vo + k]; }
    }
    for (int i = lane; i <= N + 1; i += 64) s_tab[i] = (double)i / N;
    if (lane < dof) {
        const size_t o = (size_t)p * dof + lane;
        s_alo[lane] = a.alo[o]; s_ahi[lane] = a.ahi[o];
        const double d = a.p1[o] - a.p0[o], v0 = a.v0[o], v1 = a.v1[o];
        s_c1[lane] = v0;
        s_c2[lane] = 3.0 * d - 2.0 * v0 - v1;
        s_c3[lane] = -2.0 * d + v0 + v1;
    }
    __syncthreads();
    double* K = a.K + (size_t)p * n1 * 2;
    double* X = a.x + (size_t)p * n1;
    double* U = a.u + (size_t)p * N;
    double* T = a.t + (size_t)p * n1;

    // slot m of stage i: m = 2 k (collocation) or 2 k + 1 (interpolation) of joint k
    auto slot = [&](int i, int m, double& aa, double& bb) {
        const int k = m >> 1;
        const double s = s_tab[i], c1 = s_c1[k], c2 = s_c2[k], c3 = s_c3[k];
        aa = c1 + s * (2.0 * c2 + s * 3.0 * c3); bb = 2.0 * c2 + 6.0 * c3 * s;
        if ((m & 1) && i < N) {
            const double s1 = s_tab[i + 1], D = s1 - s;
            const double an = c1 + s1 * (2.0 * c2 + s1 * 3.0 * c3), bn = 2.0 * c2 + 6.0 * c3 * s1;
            aa = an + 2.0 * D * bn;
            bb = bn;
        }
    };

    // ---- phase 0a: velocity constraint (LinearJointVelocity::computeParams), one lane per stage ----
    for (int i = lane; i <= N; i += 64) {
        const double s = s_tab[i];
        double sdmin = -TP_MAXSD, sdmax = TP_MAXSD;
        for (int k = 0; k < dof; ++k) {
            const double v = s_c1[k] + s * (2.0 * s_c2[k] + s * 3.0 * s_c3[k]);
            double lo, hi;
            if (LDSLIM) {
                const int o = a.vlim_per_stage ? i * dof + k : k;
                lo = s_vlo[o]; hi = s_vhi[o];
            } else {
                const size_t o = a.vlim_per_stage ? ((size_t)p * n1 + i) * dof + k : (size_t)p * dof + k;
                lo = a.vlo[o]; hi = a.vhi[o];
            }
            if (v > 0) { sdmax = fmin(hi / v, sdmax); sdmin = fmax(lo / v, sdmin); }
            else if (v < 0) { sdmax = fmin(lo / v, sdmax); sdmin = fmax(hi / v, sdmin); }
        }
        s_slo[i] = sdmin > 0 ? sdmin * sdmin : 0.0;
        s_shi[i] = sdmax * sdmax;
    }
    // ---- phase 0b: acceleration rows against each other, chunks of 32 stages; lane (h, j): stage j, half h of the pairs ----
    for (int i0 = 0; i0 < N; i0 += 32) {
        __syncthreads();
        for (int w = lane; w < 32 * ns; w += 64) {
            const int j = w / ns, m = w - j * ns;
            double aa = 0, bb = 0;
            if (i0 + j < N) slot(i0 + j, m, aa, bb);
            s_sa[j * ns + m] = aa; s_sb[j * ns + m] = bb;
        }
        __syncthreads();
        const int j = lane & 31, h = lane >> 5, i = i0 + j;
        double lo = -INFINITY, hi = INFINITY;
        bool bad = false;
        if (i < N) {
            for (int mu = h; mu < ns; mu += 2) {          // the slot that supplies the upper row
                const double au = s_sa[j * ns + mu], bu = s_sb[j * ns + mu];
                const int ku = mu >> 1;
                if (au == 0.0) {
                    // alpha == 0: both rows of the slot bound x directly (handled once, by the half that owns the slot)
                    const double g0 = s_ahi[ku], g1 = -s_alo[ku];
                    if (bu > 0) { hi = fmin(hi, g0 / bu); lo = fmax(lo, g1 / -bu); }
                    else if (bu < 0) { lo = fmax(lo, g0 / bu); hi = fmin(hi, g1 / -bu); }
                    else if (g0 < -TP_LP_TOL || g1 < -TP_LP_TOL) bad = true;
                    continue;
                }
                // upper row (alpha > 0) of slot mu
                const double ali = au > 0 ? au : -au, bei = au > 0 ? bu : -bu, gai = au > 0 ? s_ahi[ku] : -s_alo[ku];
                for (int ml = 0; ml < ns; ++ml) {         // the slot that supplies the lower row
                    const double al_ = s_sa[j * ns + ml], bl_ = s_sb[j * ns + ml];
                    if (al_ == 0.0) continue;
                    const int kl = ml >> 1;
                    const double alj = al_ > 0 ? -al_ : al_, bej = al_ > 0 ? -bl_ : bl_, gaj = al_ > 0 ? -s_alo[kl] : s_ahi[kl];
                    const double cf = alj * bei - ali * bej, rhs = alj * gai - ali * gaj;
                    if (cf > 0) lo = fmax(lo, rhs / cf);
                    else if (cf < 0) hi = fmin(hi, rhs / cf);
                    else if (rhs > TP_LP_TOL) bad = true;
                }
            }
        }
        // the two halves of a stage
        lo = fmax(lo, __shfl_xor(lo, 32)); hi = fmin(hi, __shfl_xor(hi, 32));
        bad = bad || __shfl_xor((int)bad, 32);
        if (i < N && h == 0) { s_slo[i] = bad ? INFINITY : dmax(s_slo[i], lo); s_shi[i] = bad ? -INFINITY : dmin(s_shi[i], hi); }
    }
    __syncthreads();

    int status = 0;
    // ---- backward chain: controllable sets ----
    double klo = a.sd_end * a.sd_end, khi = klo;
    if (lane == 0) { K[2 * N] = klo; K[2 * N + 1] = khi; s_K[2 * N] = klo; s_K[2 * N + 1] = khi; }
    for (int i = N - 1; i >= 0; --i) {
        const double D = s_tab[i + 1] - s_tab[i], twoD = 2.0 * D;
        // lanes 0 .. ns-1: the two pairs of a slot's rows with the rows of the next set; lane ns: everything that does not
        // depend on the next set.  No branches: a bound that does not apply is replaced by -inf / +inf.
        double aa, bb;
        slot(i, lane < ns ? lane : 0, aa, bb);
        const int k = lane < ns ? lane >> 1 : 0;
        const bool pos = aa > 0, live = lane < ns && aa != 0.0;
        const double ali = pos ? aa : -aa, bei = pos ? bb : -bb, gai = pos ? s_ahi[k] : -s_alo[k];    // upper row of the slot
        const double alj = pos ? -aa : aa, bej = pos ? -bb : bb, gaj = pos ? -s_alo[k] : s_ahi[k];   // lower row
        // upper = (2D, 1, khi) of the next set, lower = this slot's
        const double cf1 = alj * 1.0 - twoD * bej, rhs1 = alj * khi - twoD * gaj, v1 = rhs1 / cf1;
        // upper = this slot's, lower = (-2D, -1, -klo) of the next set
        const double cf2 = (-twoD) * bei - ali * (-1.0), rhs2 = (-twoD) * gai - ali * (-klo), v2 = rhs2 / cf2;
        double lo = dmax(live && cf1 > 0 ? v1 : -INFINITY, live && cf2 > 0 ? v2 : -INFINITY);
        double hi = dmin(live && cf1 < 0 ? v1 : INFINITY, live && cf2 < 0 ? v2 : INFINITY);
        bool infeasible = live && ((cf1 == 0.0 && rhs1 > TP_LP_TOL) || (cf2 == 0.0 && rhs2 > TP_LP_TOL));
        // the velocity bounds, the acceleration rows among themselves, and the next set against itself (cf == 0)
        const double slo = s_slo[i], shi = s_shi[i];
        if (lane == ns) { lo = slo; hi = shi; infeasible = (-twoD) * khi - twoD * (-klo) > TP_LP_TOL; }
        if (ns < 16) { lo = row16_minmax<false>(lo); hi = row16_minmax<true>(hi); }
        else { lo = wave_max(lo); hi = wave_min(hi); }
        if (__ballot(infeasible) || lo > hi + TP_LP_TOL) { status = 1; break; }
        if (lo > hi) lo = hi;
        klo = lo > 0 ? lo : 0.0;
        khi = hi;
        if (lane == 0) { K[2 * i] = klo; K[2 * i + 1] = khi; s_K[2 * i] = klo; s_K[2 * i + 1] = khi; }
    }
    // ---- forward chain: greedy maximal u, knot times ----
    if (!status) {
        __syncthreads();
        double x = a.sd_start * a.sd_start, tt = 0.0;
        if (x < klo - TP_LP_TOL || x > khi + TP_LP_TOL) status = 2;
        if (lane == 0) { X[0] = x; T[0] = 0.0; }
        for (int i = 0; i < N && !status; ++i) {
            const double D = s_tab[i + 1] - s_tab[i], twoD = 2.0 * D;
            const double nlo = s_K[2 * (i + 1)], nhi = s_K[2 * (i + 1) + 1];
            double aa, bb;
            slot(i, lane < ns ? lane : 0, aa, bb);
            const int k = lane < ns ? lane >> 1 : 0;
            const bool live = lane < ns && aa != 0.0;
            // row (aa, bb, ahi) and row (-aa, -bb, -alo): the one with alpha > 0 bounds u from above
            const double q0 = (s_ahi[k] - bb * x) / aa, q1 = (-s_alo[k] - (-bb) * x) / -aa;
            double umax = live ? (aa > 0 ? q0 : q1) : INFINITY, umin = live ? (aa > 0 ? q1 : q0) : -INFINITY;
            if (lane == ns) {   // the rows of the next set
                umax = (nhi - 1.0 * x) / twoD;
                umin = (-nlo - (-1.0) * x) / -twoD;
            }
            if (ns < 16) { umax = row16_minmax<true>(umax); umin = row16_minmax<false>(umin); }
            else { umax = wave_min(umax); umin = wave_max(umin); }
            if (!(umax >= umin - 1e-6) || !isfinite(umax)) { status = 2; break; }
            double xn = x + twoD * umax;
            if (xn > nhi) xn = nhi;
            if (xn < nlo) xn = nlo;
            const double sda = 0.5 * (sqrt(fmax(x, 0.0)) + sqrt(fmax(xn, 0.0)));
            tt += sda > TP_NEARLY_ZERO ? D / sda : 5.0;
            if (lane == 0) { U[i] = umax; X[i + 1] = xn; T[i + 1] = tt; }
            x = xn;
        }
    }
    if (lane == 0) a.status[p] = status;
}

// ---- the fast kernel (dof <= 7) ---------------------------------------------------------------------------------------
// Three wavefronts per plan, in lock step of chunks of TF_CH stages (one barrier per chunk):
//   wave 0  the two dependent chains.  A stage's bounds are quotients (A k - T) / C with k the next set's K_hi or K_lo
//           (backward) or (G - B x) / A (forward): the coefficients are waiting in LDS with the reciprocal of the divisor,
//           so a quotient is two multiply-adds behind the product (the tail of the compiler's own fp64 division, without
//           its scaling steps).  Lanes 0..15 hold what bounds from above, lanes 16..31 the NEGATED lower bounds, so ONE
//           4-step DPP minimum inside the rows of 16 lanes yields both ends of the interval (lanes 0 and 16).
//   wave 1  backward: the x-interval the acceleration rows leave among themselves (4 dof^2 pairs per stage, one lane per
//           (stage, upper slot), a 16-lane reduction); forward: the stage coefficients;
//   wave 2  the Hermite slots (a, b) of a chunk two steps ahead, and the backward coefficients one step ahead.
// Bounds that do not apply are the constant 1e300 (every real bound is <= 1e16).  A divisor that is exactly zero (the
// row pair bounds nothing but can declare the stage infeasible) raises a per-chunk flag, and wave 0 then evaluates that
// rule as the wide kernel below does.  Quotients differ from IEEE division by at most one unit in the last place.
#define TF_CH 4
#define TF_THREADS 192
#define TF_BIG 1e300

__device__ __forceinline__ double fast_rcp(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return r;
}
// a / b given r = fast_rcp(b)
__device__ __forceinline__ double div_by(double a, double b, double r) {
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-q, b, a), r, q);
}
// minimum / maximum inside each row of 16 lanes, left in every lane of the row
template <int CTRL>
__device__ __forceinline__ double dpp_all_f64(double v) {     // every lane has a source lane: no previous value to keep
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <bool IS_MIN>
__device__ __forceinline__ double row16_all(double v) {
    auto op = [](double a, double b) { return IS_MIN ? dmin(a, b) : dmax(a, b); };
    v = op(v, dpp_all_f64<0xB1>(v));
    v = op(v, dpp_all_f64<0x4E>(v));
    v = op(v, dpp_all_f64<0x141>(v));
    v = op(v, dpp_all_f64<0x140>(v));
    return v;
}
__device__ __forceinline__ double lane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

template <int DOF>
__global__ void __launch_bounds__(TF_THREADS) toppra_fast_kernel(toppra_args a) {
    __shared__ double s_c1[8], s_c2[8], s_c3[8], s_alo[8], s_ahi[8];
    __shared__ int s_zf[2], s_out[3];
    extern __shared__ __align__(16) double s_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, p = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int dof = DOF, ns = 2 * DOF;
    const int N = a.N, n1 = N + 1;
    double* s_slo = s_dyn;                       // [n1] static x-interval of a stage; after the backward pass: x
    double* s_shi = s_slo + n1;                  // [n1]                                                        u
    double* s_tab = s_shi + n1;                  // [n1 + 2] gridpoints i / N (and one spare: what follows stays 16-byte aligned)
    double* s_K = s_tab + n1 + 2;                // [n1][2]
    double* s_term = s_K + 2 * n1;               // [n1] knot time increments
    double* s_slot = s_term + n1;                // [2][TF_CH][ns][6]   slots: a, b, and the slot's lower row (alpha, beta, gamma), one spare
    double2* s_ent = reinterpret_cast<double2*>(s_slot + 12 * TF_CH * ns);   // [2][TF_CH][4][32] coefficient pairs
    auto ent = [&](int b, int j, int f, int l) { return s_ent + (((b * TF_CH + j) * 4 + f) * 32 + l); };

    for (int i = tid; i <= N + 1; i += TF_THREADS) s_tab[i] = (double)i / N;
    if (tid < dof) {
        const size_t o = (size_t)p * dof + tid;
        s_alo[tid] = a.alo[o]; s_ahi[tid] = a.ahi[o];
        const double d = a.p1[o] - a.p0[o], v0 = a.v0[o], v1 = a.v1[o];
        s_c1[tid] = v0;
        s_c2[tid] = 3.0 * d - 2.0 * v0 - v1;
        s_c3[tid] = -2.0 * d + v0 + v1;
    }
    if (tid < 2) s_zf[tid] = 0;
    for (int w = tid; w < 2 * TF_CH * 4 * 32; w += TF_THREADS) s_ent[w] = ((w >> 5) & 1) ? make_double2(1.0, 1.0) : make_double2(0.0, -TF_BIG);
    __syncthreads();
    double* K = a.K + (size_t)p * n1 * 2;
    double* X = a.x + (size_t)p * n1;
    double* U = a.u + (size_t)p * N;
    double* T = a.t + (size_t)p * n1;

    auto slot = [&](int i, int m, double& aa, double& bb) {
        const int k = m >> 1;
        const double s = s_tab[i], c1 = s_c1[k], c2 = s_c2[k], c3 = s_c3[k];
        aa = c1 + s * (2.0 * c2 + s * 3.0 * c3); bb = 2.0 * c2 + 6.0 * c3 * s;
        if ((m & 1) && i < N) {
            const double s1 = s_tab[i + 1], D = s1 - s;
            const double an = c1 + s1 * (2.0 * c2 + s1 * 3.0 * c3), bn = 2.0 * c2 + 6.0 * c3 * s1;
            aa = an + 2.0 * D * bn;
            bb = bn;
        }
    };
    // slots of the stages i_of(j), j < TF_CH, into slot buffer b
    auto fill_slots = [&](int b, auto i_of) {
        for (int w = lane; w < TF_CH * ns; w += 64) {
            const int j = w / ns, m = w - j * ns, i = i_of(j);
            if (i < 0 || i >= N) continue;
            double aa, bb;
            slot(i, m, aa, bb);
            double* o = s_slot + ((b * TF_CH + j) * ns + m) * 6;
            const int k = m >> 1;
            const bool pos = aa > 0, live = aa != 0.0;
            o[0] = aa; o[1] = bb;
            // the slot's row with alpha < 0; all zero when alpha == 0 (such a row takes no part in the pair elimination:
            // a zero row pairs to cf = 0, rhs = 0, which bounds nothing)
            o[2] = live ? (pos ? -aa : aa) : 0.0; o[3] = live ? (pos ? -bb : bb) : 0.0; o[4] = live ? (pos ? -s_alo[k] : s_ahi[k]) : 0.0;
        }
    };

    // ---- velocity constraint (LinearJointVelocity::computeParams), one thread per stage ----
    for (int i = tid; i <= N; i += TF_THREADS) {
        const double s = s_tab[i];
        double sdmin = -TP_MAXSD, sdmax = TP_MAXSD;
        for (int k = 0; k < dof; ++k) {
            const double v = s_c1[k] + s * (2.0 * s_c2[k] + s * 3.0 * s_c3[k]);
            const size_t o = a.vlim_per_stage ? ((size_t)p * n1 + i) * dof + k : (size_t)p * dof + k;
            const double lo = a.vlo[o], hi = a.vhi[o];
            if (v > 0) { sdmax = fmin(hi / v, sdmax); sdmin = fmax(lo / v, sdmin); }
            else if (v < 0) { sdmax = fmin(lo / v, sdmax); sdmin = fmax(hi / v, sdmin); }
        }
        s_slo[i] = sdmin > 0 ? sdmin * sdmin : 0.0;
        s_shi[i] = sdmax * sdmax;
    }
    __syncthreads();

    const int nchunk = (N + TF_CH - 1) / TF_CH;
    int status = 0, kfirst = N, nx = 0;              // wave 0's
    double klo = a.sd_end * a.sd_end, khi = klo;
    if (tid == 0) { s_K[2 * N] = klo; s_K[2 * N + 1] = khi; }
    if (wave == 0) __builtin_amdgcn_s_setprio(3);

    // ================= backward: controllable sets, stages N-1 .. 0; chunk c holds stages N-1 - c TF_CH - j =================
    for (int st = 0; st < nchunk + 2; ++st) {
        if (wave == 2) {
            if (st < nchunk) fill_slots(st & 1, [&](int j) { return N - 1 - st * TF_CH - j; });
            if (st >= 1 && st <= nchunk) {
                const int c = st - 1, b = c & 1;
                bool zero = false;
                for (int w = lane; w < TF_CH * ns; w += 64) {
                    const int j = w / ns, m = w - j * ns, i = N - 1 - c * TF_CH - j;
                    if (i < 0) continue;
                    const double aa = s_slot[((b * TF_CH + j) * ns + m) * 6], bb = s_slot[((b * TF_CH + j) * ns + m) * 6 + 1];
                    const int k = m >> 1;
                    const double twoD = 2.0 * (s_tab[i + 1] - s_tab[i]);
                    const bool pos = aa > 0, live = aa != 0.0;
                    const double ali = pos ? aa : -aa, bei = pos ? bb : -bb, gai = pos ? s_ahi[k] : -s_alo[k];    // upper row of the slot
                    const double alj = pos ? -aa : aa, bej = pos ? -bb : bb, gaj = pos ? -s_alo[k] : s_ahi[k];   // lower row
                    // upper = (2D, 1, khi) of the next set, lower = this slot's:  v1 = (alj khi - 2D gaj) / cf1
                    const double cf1 = alj * 1.0 - twoD * bej, t1 = twoD * gaj, r1 = fast_rcp(cf1);
                    // upper = this slot's, lower = (-2D, -1, -klo) of the next set:  v2 = ((-2D) gai + ali klo) / cf2
                    const double cf2 = (-twoD) * bei - ali * (-1.0), t2 = (-twoD) * gai, r2 = fast_rcp(cf2);
                    zero = zero || (live && (cf1 == 0.0 || cf2 == 0.0));
                    const double2 nnum = make_double2(0.0, -TF_BIG), nden = make_double2(1.0, 1.0);
                    // cf < 0: the quotient bounds x from above (row 0); cf > 0: from below (row 1, negated numerator)
                    *ent(b, j, 0, m) = live && cf1 < 0 ? make_double2(alj, t1) : nnum;
                    *ent(b, j, 1, m) = live && cf1 < 0 ? make_double2(cf1, r1) : nden;
                    *ent(b, j, 0, 16 + m) = live && cf1 > 0 ? make_double2(-alj, -t1) : nnum;
                    *ent(b, j, 1, 16 + m) = live && cf1 > 0 ? make_double2(cf1, r1) : nden;
                    *ent(b, j, 2, m) = live && cf2 < 0 ? make_double2(ali, -t2) : nnum;
                    *ent(b, j, 3, m) = live && cf2 < 0 ? make_double2(cf2, r2) : nden;
                    *ent(b, j, 2, 16 + m) = live && cf2 > 0 ? make_double2(-ali, t2) : nnum;
                    *ent(b, j, 3, 16 + m) = live && cf2 > 0 ? make_double2(cf2, r2) : nden;
                }
                if (__ballot(zero) && lane == 0) s_zf[b] = 1;
            }
        } else if (wave == 1) {
            if (st >= 1 && st <= nchunk) {
                // the acceleration rows against each other: lane (j, mu) takes the upper row from slot mu, loops over the lower rows
                const int c = st - 1, b = c & 1, j = lane >> 4, mu = lane & 15, i = N - 1 - c * TF_CH - j;
                double lo = -INFINITY, hi = INFINITY;
                bool bad = false;
                if (mu < ns && i >= 0) {
                    const double* sl = s_slot + (size_t)(b * TF_CH + j) * ns * 6;
                    const double au = sl[6 * mu], bu = sl[6 * mu + 1];
                    const int ku = mu >> 1;
                    if (au == 0.0) {
                        // alpha == 0: both rows of the slot bound x directly
                        const double g0 = s_ahi[ku], g1 = -s_alo[ku];
                        if (bu > 0) { hi = fmin(hi, g0 / bu); lo = fmax(lo, g1 / -bu); }
                        else if (bu < 0) { lo = fmax(lo, g0 / bu); hi = fmin(hi, g1 / -bu); }
                        else if (g0 < -TP_LP_TOL || g1 < -TP_LP_TOL) bad = true;
                    } else {
                        const double ali = au > 0 ? au : -au, bei = au > 0 ? bu : -bu, gai = au > 0 ? s_ahi[ku] : -s_alo[ku];
#pragma unroll
                        for (int ml = 0; ml < ns; ++ml) {
                            const double alj = sl[6 * ml + 2], bej = sl[6 * ml + 3], gaj = sl[6 * ml + 4];
                            const double cf = alj * bei - ali * bej, rhs = alj * gai - ali * gaj;
                            const double v = div_by(rhs, cf, fast_rcp(cf));
                            lo = dmax(lo, cf > 0 ? v : -INFINITY);
                            hi = dmin(hi, cf < 0 ? v : INFINITY);
                            bad = bad || (cf == 0.0 && rhs > TP_LP_TOL);
                        }
                    }
                }
                lo = row16_all<false>(lo); hi = row16_all<true>(hi);
                const unsigned long long bm = __ballot(bad);
                if (mu == 0 && i >= 0) {
                    const bool anybad = (bm >> (16 * j)) & 0xFFFFull;
                    s_slo[i] = anybad ? INFINITY : dmax(s_slo[i], lo);
                    s_shi[i] = anybad ? -INFINITY : dmin(s_shi[i], hi);
                }
            }
        } else if (st >= 2 && !status) {
            const int c = st - 2, b = c & 1, i0 = N - 1 - c * TF_CH, l32 = lane & 31;
            // the static interval of each stage of the chunk, as the constant "quotients" (0 k + shi) / 1 and (0 k - slo) / 1
            if (lane < 2 * TF_CH) {
                const int j = lane >> 1, r = lane & 1, i = i0 - j;
                if (i >= 0) *ent(b, j, 0, 16 * r + ns) = make_double2(0.0, r ? s_slo[i] : -s_shi[i]);
            }
            const int zf = s_zf[b];
            if (zf && lane == 0) s_zf[b] = 0;
            double2 f[TF_CH][4];
            double twoD[TF_CH];
#pragma unroll
            for (int j = 0; j < TF_CH; ++j) {
                const int i = i0 - j > 0 ? i0 - j : 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) f[j][q] = *ent(b, j, q, l32);
                twoD[j] = 2.0 * (s_tab[i + 1] - s_tab[i]);
            }
#pragma unroll
            for (int j = 0; j < TF_CH; ++j) {
                const int i = i0 - j;
                if (i < 0 || status) continue;
                const double v1 = div_by(f[j][0].x * khi - f[j][0].y, f[j][1].x, f[j][1].y);
                const double v2 = div_by(f[j][2].x * klo - f[j][2].y, f[j][3].x, f[j][3].y);
                const double w = row16_all<true>(dmin(v1, v2));
                double hi = lane_f64(w, 0), lo = -lane_f64(w, 16);
                bool infeasible = (-twoD[j]) * khi - twoD[j] * (-klo) > TP_LP_TOL;      // the next set against itself
                if (zf) {
                    // a row pair with a zero divisor somewhere in this chunk: the rule of the wide kernel, lanes over slots
                    double aa, bb;
                    slot(i, lane < ns ? lane : 0, aa, bb);
                    const int k = lane < ns ? lane >> 1 : 0;
                    const bool pos = aa > 0, live = lane < ns && aa != 0.0;
                    const double ali = pos ? aa : -aa, bei = pos ? bb : -bb, gai = pos ? s_ahi[k] : -s_alo[k];
                    const double alj = pos ? -aa : aa, bej = pos ? -bb : bb, gaj = pos ? -s_alo[k] : s_ahi[k];
                    const double cf1 = alj * 1.0 - twoD[j] * bej, rhs1 = alj * khi - twoD[j] * gaj;
                    const double cf2 = (-twoD[j]) * bei - ali * (-1.0), rhs2 = (-twoD[j]) * gai - ali * (-klo);
                    if (__ballot(live && ((cf1 == 0.0 && rhs1 > TP_LP_TOL) || (cf2 == 0.0 && rhs2 > TP_LP_TOL)))) infeasible = true;
                }
                if (infeasible || lo > hi + TP_LP_TOL) { status = 1; continue; }
                if (lo > hi) lo = hi;
                klo = lo > 0 ? lo : 0.0;
                khi = hi;
                kfirst = i;
                if (lane == 0) { s_K[2 * i] = klo; s_K[2 * i + 1] = khi; }
            }
        }
        __syncthreads();
    }

    // ================= forward: greedy maximal u; chunk c holds stages c TF_CH + j =================
    double x = a.sd_start * a.sd_start;
    if (wave == 0) {
        if (!status) {
            if (x < klo - TP_LP_TOL || x > khi + TP_LP_TOL) status = 2;
            nx = 1;
        }
        if (lane == 0) s_out[0] = status;
    }
    // bounds that do not apply, forward form: (1e300 - 0 x) / 1
    for (int w = tid; w < 2 * TF_CH * 4 * 32; w += TF_THREADS)
        if (((w >> 5) & 3) < 2) s_ent[w] = ((w >> 5) & 1) ? make_double2(1.0, 1.0) : make_double2(TF_BIG, 0.0);
    __syncthreads();
    const bool run_forward = s_out[0] == 0;           // the same for every wave
    if (wave == 0 && lane == 0) s_slo[0] = x;
    for (int st = 0; run_forward && st < nchunk + 2; ++st) {
        if (wave == 2) {
            if (st < nchunk) fill_slots(st & 1, [&](int j) { return st * TF_CH + j; });
        } else if (wave == 1) {
            if (st >= 1 && st <= nchunk) {
                const int c = st - 1, b = c & 1;
                for (int w = lane; w < TF_CH * (ns + 1); w += 64) {
                    const int j = w / (ns + 1), m = w - j * (ns + 1), i = c * TF_CH + j;
                    if (i >= N) continue;
                    double2 g0 = make_double2(TF_BIG, 0.0), g1 = g0, d0 = make_double2(1.0, 1.0), d1 = d0;
                    if (m == ns) {
                        // the rows of the next set:  u <= (nhi - x) / 2D,  -u <= (nlo - x) / (-2D)
                        const double twoD = 2.0 * (s_tab[i + 1] - s_tab[i]);
                        const double nlo = s_K[2 * (i + 1)], nhi = s_K[2 * (i + 1) + 1];
                        g0 = make_double2(nhi, 1.0); d0 = make_double2(twoD, fast_rcp(twoD));
                        g1 = make_double2(nlo, 1.0); d1 = make_double2(-twoD, fast_rcp(-twoD));
                    } else {
                        const double aa = s_slot[((b * TF_CH + j) * ns + m) * 6], bb = s_slot[((b * TF_CH + j) * ns + m) * 6 + 1];
                        const int k = m >> 1;
                        if (aa != 0.0) {
                            // rows (aa, bb, ahi) and (-aa, -bb, -alo): the one with alpha > 0 bounds u from above, the other from below
                            const double rp = fast_rcp(aa), rn = fast_rcp(-aa), ahi = s_ahi[k], alo = s_alo[k];
                            if (aa > 0) { g0 = make_double2(ahi, bb); d0 = make_double2(aa, rp); g1 = make_double2(alo, bb); d1 = make_double2(-aa, rn); }
                            else { g0 = make_double2(-alo, -bb); d0 = make_double2(-aa, rn); g1 = make_double2(-ahi, -bb); d1 = make_double2(aa, rp); }
                        }
                    }
                    *ent(b, j, 0, m) = g0; *ent(b, j, 1, m) = d0;
                    *ent(b, j, 0, 16 + m) = g1; *ent(b, j, 1, 16 + m) = d1;
                }
            }
        } else if (st >= 2 && !status) {
            const int c = st - 2, b = c & 1, i0 = c * TF_CH, l32 = lane & 31;
            double2 f[TF_CH][2];
            double twoD[TF_CH], nlo[TF_CH], nhi[TF_CH];
#pragma unroll
            for (int j = 0; j < TF_CH; ++j) {
                const int i = i0 + j < N ? i0 + j : N - 1;
                f[j][0] = *ent(b, j, 0, l32); f[j][1] = *ent(b, j, 1, l32);
                twoD[j] = 2.0 * (s_tab[i + 1] - s_tab[i]);
                nlo[j] = s_K[2 * (i + 1)]; nhi[j] = s_K[2 * (i + 1) + 1];
            }
#pragma unroll
            for (int j = 0; j < TF_CH; ++j) {
                const int i = i0 + j;
                if (i >= N || status) continue;
                const double w = row16_all<true>(div_by(f[j][0].x - f[j][0].y * x, f[j][1].x, f[j][1].y));
                const double umax = lane_f64(w, 0), umin = -lane_f64(w, 16);
                if (!(umax >= umin - 1e-6) || !isfinite(umax)) { status = 2; continue; }
                double xn = x + twoD[j] * umax;
                if (xn > nhi[j]) xn = nhi[j];
                if (xn < nlo[j]) xn = nlo[j];
                if (lane == 0) { s_shi[i] = umax; s_slo[i + 1] = xn; }
                x = xn;
                nx = i + 2;
            }
        }
        __syncthreads();
    }
    if (wave == 0 && lane == 0) { s_out[0] = status; s_out[1] = kfirst; s_out[2] = nx; }
    __syncthreads();
    // ================= results: K, x, u from LDS; knot times (a sequential sum, in the oracle's order) =================
    status = s_out[0]; kfirst = s_out[1]; nx = s_out[2];
    for (int i = kfirst + tid; i <= N; i += TF_THREADS) { K[2 * i] = s_K[2 * i]; K[2 * i + 1] = s_K[2 * i + 1]; }
    for (int i = tid; i < nx; i += TF_THREADS) X[i] = s_slo[i];
    for (int i = tid; i < nx - 1; i += TF_THREADS) {
        U[i] = s_shi[i];
        const double D = s_tab[i + 1] - s_tab[i];
        const double sda = 0.5 * (sqrt(fmax(s_slo[i], 0.0)) + sqrt(fmax(s_slo[i + 1], 0.0)));
        s_term[i] = sda > TP_NEARLY_ZERO ? D / sda : 5.0;
    }
    __syncthreads();
    if (tid == 0) {
        a.status[p] = status;
        if (nx > 0) {
            double tt = 0.0;
            T[0] = 0.0;
#pragma unroll 8
            for (int i = 0; i < nx - 1; ++i) { tt += s_term[i]; T[i + 1] = tt; }
        }
    }
}

extern "C" int sc_toppra_hermite_batch(sc_ctx* ctx, int P, int dof, int N,
                                       const double* p0, const double* p1, const double* v0, const double* v1,
                                       const double* vlim_lo, const double* vlim_hi, int vlim_per_stage,
                                       const double* alim_lo, const double* alim_hi,
                                       double sd_start, double sd_end,
                                       double* K, double* x, double* u, double* t, int32_t* status) {
    if (!ctx || P <= 0 || dof <= 0 || dof > TP_MAXDOF || N <= 0 || !p0 || !p1 || !v0 || !v1 || !vlim_lo || !vlim_hi ||
        !alim_lo || !alim_hi || !K || !x || !u || !t || !status)
        return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    if (dof <= 7) {
        const size_t lds = ((size_t)6 * (N + 1) + 2 + (size_t)24 * TF_CH * dof) * sizeof(double) + (size_t)2 * TF_CH * 4 * 32 * sizeof(double2);
        if (lds > 150 * 1024) { snprintf(ctx->err, sizeof(ctx->err), "sc_toppra_hermite_batch: N = %d stages do not fit the LDS of a CU", N); return SC_ERR_INVALID; }
        toppra_args a{P, dof, N, p0, p1, v0, v1, vlim_lo, vlim_hi, alim_lo, alim_hi, vlim_per_stage, 0, sd_start, sd_end, K, x, u, t, status};
        void (*kern)(toppra_args) = dof == 1 ? toppra_fast_kernel<1> : dof == 2 ? toppra_fast_kernel<2> : dof == 3 ? toppra_fast_kernel<3> :
                                    dof == 4 ? toppra_fast_kernel<4> : dof == 5 ? toppra_fast_kernel<5> : dof == 6 ? toppra_fast_kernel<6> : toppra_fast_kernel<7>;
        int r_ = sc_allow_big_lds(ctx, reinterpret_cast<const void*>(kern), 150 * 1024);
        if (r_ != SC_OK) return r_;
        int tk = sc_time_begin(ctx, SC_K_TOPPRA);
        hipLaunchKernelGGL(kern, dim3(P), dim3(TF_THREADS), lds, ctx->stream, a);
        sc_time_end(ctx, tk);
        SC_HIP(ctx, hipGetLastError());
        return SC_OK;
    }
    const size_t kwords = (size_t)2 * (N + 1) > (size_t)128 * dof ? (size_t)2 * (N + 1) : (size_t)128 * dof;
    const size_t base = ((size_t)3 * (N + 1) + 1 + kwords) * sizeof(double);                          // static interval, gridpoints, K | slot chunk
    const size_t lim = (size_t)2 * (vlim_per_stage ? (N + 1) : 1) * dof * sizeof(double);
    const int lds_limits = base + lim <= 64 * 1024 ? 1 : 0;   // otherwise phase 0 reads the limits from global memory
    const size_t lds = base + (lds_limits ? lim : 0);
    if (lds > 150 * 1024) { snprintf(ctx->err, sizeof(ctx->err), "sc_toppra_hermite_batch: N = %d stages do not fit the LDS of a CU", N); return SC_ERR_INVALID; }
    toppra_args a{P, dof, N, p0, p1, v0, v1, vlim_lo, vlim_hi, alim_lo, alim_hi, vlim_per_stage, lds_limits,
                  sd_start, sd_end, K, x, u, t, status};
    {
        int r_ = sc_allow_big_lds(ctx, lds_limits ? reinterpret_cast<const void*>(toppra_wide_kernel<true>) : reinterpret_cast<const void*>(toppra_wide_kernel<false>), 150 * 1024);
        if (r_ != SC_OK) return r_;
    }
    int tk = sc_time_begin(ctx, SC_K_TOPPRA);
    if (lds_limits) hipLaunchKernelGGL(toppra_wide_kernel<true>, dim3(P), dim3(64), lds, ctx->stream, a);
    else hipLaunchKernelGGL(toppra_wide_kernel<false>, dim3(P), dim3(64), lds, ctx->stream, a);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// ---- parametrizer::Spline + uniform sampling --------------------------------------------------
struct sample_args {
    int P, dof, N, max_len;
    const double *p0, *p1, *v0, *v1, *x, *t;
    double dt;
    float *pos, *vel, *acc;
    double* times;
    int32_t* length;
};

// One block (256 threads) per plan; dynamic LDS: (2 + 3 dof)(N + 1) doubles + (N + 1) 16-bit knot indices.  All threads
// pick the knots (ballot compaction), evaluate the joints' knot positions and chord slopes; then lane k < dof of the first
// wavefront solves joint k's clamped-spline system (N dependent eliminations, the joints side by side; the pivots depend
// on the knot times only and are shared); then a thread takes a sample, finds its knot interval once and evaluates every
// joint there.  The eliminations and the per-sample cubic multiply by reciprocals (one division per pivot, one per
// sample) where the oracle divides: results agree to a few units in the last place.
#define TS_THREADS 256
__global__ void __launch_bounds__(TS_THREADS) toppra_sample_kernel(sample_args a) {
    extern __shared__ __align__(16) double sm[];
    const int N = a.N, n1 = N + 1, dof = a.dof;
    double* tk = sm;                       // [n1] knot times (zero increments dropped)
    double* cs = tk + n1;                  // [n1] Thomas algorithm: upper / pivot (the same for every joint)
    double* yk = cs + n1;                  // [dof][n1] joint positions at the knots
    double* M = yk + (size_t)dof * n1;     // [dof][n1] chord slopes, then second derivatives
    double* dp = M + (size_t)dof * n1;     // [dof][n1] Thomas algorithm: right-hand sides
    unsigned short* idx = reinterpret_cast<unsigned short*>(dp + (size_t)dof * n1);   // [n1] stage of each knot
    __shared__ int s_cnt[TS_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = blockIdx.x;
    const double* t = a.t + (size_t)p * n1;
    const double* x = a.x + (size_t)p * n1;
    // knots with a (nearly) zero time increment are dropped, as parametrizer::Spline does
    int n = 0;
    for (int base = 0; base <= N; base += TS_THREADS) {
        const int i = base + tid;
        double ti = 0;
        bool keep = false;
        if (i <= N) { ti = t[i]; keep = i == 0 || ti - t[i - 1] >= TP_NEARLY_ZERO; }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = n, tot = 0;
        for (int w = 0; w < TS_THREADS / 64; ++w) { const int c = s_cnt[w]; if (w < wave) off += c; tot += c; }
        if (keep) { const int r = off + __popcll(m & ((1ull << lane) - 1)); tk[r] = ti; idx[r] = (unsigned short)i; }
        n += tot;
        __syncthreads();
    }
    // knot positions of every joint (each an fp64 division for s: kept off the serial solves)
    for (int w = tid; w < dof * n; w += TS_THREADS) {
        const int k = w / n, j = w - k * n;
        const size_t o = (size_t)p * dof + k;
        const double q0 = a.p0[o], c1 = a.v0[o], dd = a.p1[o] - q0;
        const double c2 = 3.0 * dd - 2.0 * c1 - a.v1[o], c3 = -2.0 * dd + c1 + a.v1[o];
        const double s = (double)idx[j] / N;
        yk[(size_t)k * n1 + j] = q0 + s * (c1 + s * (c2 + s * c3));
    }
    __syncthreads();
    // chord slopes (y[j+1] - y[j]) / (tk[j+1] - tk[j]), parked in M, which the solves only write in their back substitution
    for (int w = tid; w < dof * (n - 1); w += TS_THREADS) {
        const int k = w / (n - 1), j = w - k * (n - 1);
        const double* y = yk + (size_t)k * n1;
        M[(size_t)k * n1 + j] = (y[j + 1] - y[j]) / (tk[j + 1] - tk[j]);
    }
    __syncthreads();
    if (tid < dof) {
        const int k = tid;
        const size_t o = (size_t)p * dof + k;
        const double q0 = a.p0[o], c1 = a.v0[o], dd = a.p1[o] - q0;
        const double c2 = 3.0 * dd - 2.0 * c1 - a.v1[o], c3 = -2.0 * dd + c1 + a.v1[o];
        double* Mk = M + (size_t)k * n1;
        double* d = dp + (size_t)k * n1;
        const double d0 = c1 * sqrt(fmax(x[0], 0.0));
        const double d1 = (c1 + 2.0 * c2 + 3.0 * c3) * sqrt(fmax(x[N], 0.0));
        if (n == 1) Mk[0] = 0.0;
        else {
            double cprev = 0, dprev = 0;
            for (int j = 0; j < n; ++j) {
                double lo, di, up, rhs;
                if (j == 0) {
                    const double h = tk[1] - tk[0];
                    lo = 0; di = 2 * h; up = h; rhs = 6 * (Mk[0] - d0);
                } else if (j == n - 1) {
                    const double h = tk[j] - tk[j - 1];
                    lo = h; di = 2 * h; up = 0; rhs = 6 * (d1 - Mk[j - 1]);
                } else {
                    const double h0 = tk[j] - tk[j - 1], h1 = tk[j + 1] - tk[j];
                    lo = h0; di = 2 * (h0 + h1); up = h1;
                    rhs = 6 * (Mk[j] - Mk[j - 1]);
                }
                const double im = fast_rcp(di - lo * cprev);
                cprev = up * im;
                dprev = (rhs - lo * dprev) * im;
                if (k == 0) cs[j] = cprev;
                d[j] = dprev;
            }
            double mn = dprev;
            Mk[n - 1] = mn;
            for (int j = n - 2; j >= 0; --j) { mn = d[j] - cs[j] * mn; Mk[j] = mn; }
        }
    }
    __syncthreads();
    const double T = tk[n - 1];
    const int length = (int)ceil(T / a.dt);
    const int wl = min(length, a.max_len);
    if (tid == 0) a.length[p] = length;
    for (int j = tid; j < wl; j += TS_THREADS) {
        const double tt = length > 1 ? (j == length - 1 ? T : (T * j) / (length - 1)) : 0.0;
        a.times[(size_t)p * a.max_len + j] = tt;
        int seg = 0;
        if (n > 1) {
            // largest seg with tk[seg] < tt (seg = 0 if none), capped at n-2
            int lo = 0, hi = n - 2;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (tk[mid] < tt) lo = mid; else hi = mid - 1;
            }
            seg = lo;
        }
        double h = 1, ih = 1, aa = 0, bb = 0;
        if (n > 1) { h = tk[seg + 1] - tk[seg]; ih = 1.0 / h; aa = tk[seg + 1] - tt; bb = tt - tk[seg]; }
        const double h6 = h * (1.0 / 6.0), i6h = ih * (1.0 / 6.0), i2h = 0.5 * ih;
        for (int k = 0; k < dof; ++k) {
            const double* y = yk + (size_t)k * n1;
            const double* Mk = M + (size_t)k * n1;
            double P_, V_, A_;
            if (n == 1) { P_ = y[0]; V_ = 0; A_ = 0; }
            else {
                const double m0 = Mk[seg], m1 = Mk[seg + 1];
                const double ca = y[seg] * ih - m0 * h6, cb = y[seg + 1] * ih - m1 * h6;
                P_ = (m0 * aa * aa * aa + m1 * bb * bb * bb) * i6h + ca * aa + cb * bb;
                V_ = (m1 * bb * bb - m0 * aa * aa) * i2h - ca + cb;
                A_ = (m0 * aa + m1 * bb) * ih;
            }
            const size_t o = ((size_t)p * dof + k) * a.max_len + j;
            a.pos[o] = (float)P_; a.vel[o] = (float)V_; a.acc[o] = (float)A_;
        }
    }
}

extern "C" int sc_toppra_sample_batch(sc_ctx* ctx, int P, int dof, int N,
                                      const double* p0, const double* p1, const double* v0, const double* v1,
                                      const double* x, const double* t, double dt, int max_len,
                                      float* pos, float* vel, float* acc, double* times, int32_t* length) {
    if (!ctx || P <= 0 || dof <= 0 || N <= 0 || N > 65534 || max_len <= 0 || !(dt > 0) || !p0 || !p1 || !v0 || !v1 ||
        !x || !t || !pos || !vel || !acc || !times || !length)
        return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    sample_args a{P, dof, N, max_len, p0, p1, v0, v1, x, t, dt, pos, vel, acc, times, length};
    const size_t lds = (size_t)(2 + 3 * dof) * (N + 1) * sizeof(double) + (((size_t)(N + 1) * 2 + 15) & ~(size_t)15);
    if (lds > 140 * 1024) { snprintf(ctx->err, sizeof(ctx->err), "sc_toppra_sample_batch: (2 + 3 dof)(N + 1) doubles exceed the LDS of a CU"); return SC_ERR_INVALID; }
    {
        int r_ = sc_allow_big_lds(ctx, reinterpret_cast<const void*>(toppra_sample_kernel), 140 * 1024);
        if (r_ != SC_OK) return r_;
    }
    int tk = sc_time_begin(ctx, SC_K_TOPPRA_SAMPLE);
    hipLaunchKernelGGL(toppra_sample_kernel, dim3((unsigned)P), dim3(TS_THREADS), lds, ctx->stream, a);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}
