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
__global__ void __launch_bounds__(64) toppra_kernel(toppra_args a) {
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
    const size_t kwords = (size_t)2 * (N + 1) > (size_t)128 * dof ? (size_t)2 * (N + 1) : (size_t)128 * dof;
    const size_t base = ((size_t)3 * (N + 1) + 1 + kwords) * sizeof(double);                          // static interval, gridpoints, K | slot chunk
    const size_t lim = (size_t)2 * (vlim_per_stage ? (N + 1) : 1) * dof * sizeof(double);
    const int lds_limits = base + lim <= 64 * 1024 ? 1 : 0;   // otherwise phase 0 reads the limits from global memory
    const size_t lds = base + (lds_limits ? lim : 0);
    if (lds > 150 * 1024) { snprintf(ctx->err, sizeof(ctx->err), "sc_toppra_hermite_batch: N = %d stages do not fit the LDS of a CU", N); return SC_ERR_INVALID; }
    toppra_args a{P, dof, N, p0, p1, v0, v1, vlim_lo, vlim_hi, alim_lo, alim_hi, vlim_per_stage, lds_limits,
                  sd_start, sd_end, K, x, u, t, status};
    {
        int r_ = sc_allow_big_lds(ctx, lds_limits ? reinterpret_cast<const void*>(toppra_kernel<true>) : reinterpret_cast<const void*>(toppra_kernel<false>), 150 * 1024);
        if (r_ != SC_OK) return r_;
    }
    int tk = sc_time_begin(ctx, SC_K_TOPPRA);
    if (lds_limits) hipLaunchKernelGGL(toppra_kernel<true>, dim3(P), dim3(64), lds, ctx->stream, a);
    else hipLaunchKernelGGL(toppra_kernel<false>, dim3(P), dim3(64), lds, ctx->stream, a);
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

// One block (64 threads) per plan; dynamic LDS: (1 + 4 dof) * (N+1) doubles.  Lane k < dof solves joint k's clamped-spline
// system (a chain of N dependent eliminations each, the joints side by side; one block per (plan, joint) with a single
// busy lane each took 4x as long for the batch); then a lane takes a sample, finds its knot interval once and evaluates
// every joint there.
__global__ void __launch_bounds__(64) toppra_sample_kernel(sample_args a) {
    extern __shared__ double sm[];
    const int N = a.N, n1 = N + 1, dof = a.dof;
    double* tk = sm;                       // [n1] knot times (zero increments dropped)
    double* yk = tk + n1;                  // [dof][n1] joint positions at the knots
    double* M = yk + (size_t)dof * n1;     // [dof][n1] second derivatives
    double* cp = M + (size_t)dof * n1;     // [dof][n1] Thomas algorithm
    double* dp = cp + (size_t)dof * n1;
    __shared__ int s_n;
    __shared__ int s_idx[4001];
    const int lane = threadIdx.x, p = blockIdx.x;
    const double* t = a.t + (size_t)p * n1;
    const double* x = a.x + (size_t)p * n1;
    if (lane == 0) {
        // knots with a (nearly) zero time increment are dropped, as parametrizer::Spline does
        int n = 0;
        for (int i = 0; i <= N; ++i)
            if (i == 0 || t[i] - t[i - 1] >= TP_NEARLY_ZERO) { tk[n] = t[i]; s_idx[n] = i; ++n; }
        s_n = n;
    }
    __syncthreads();
    const int n = s_n;
    // knot positions of every joint: independent, all lanes (each an fp64 division for s: kept off the serial solves)
    for (int w = lane; w < dof * n; w += 64) {
        const int k = w / n, j = w - k * n;
        const size_t o = (size_t)p * dof + k;
        const double q0 = a.p0[o], c1 = a.v0[o], dd = a.p1[o] - q0;
        const double c2 = 3.0 * dd - 2.0 * c1 - a.v1[o], c3 = -2.0 * dd + c1 + a.v1[o];
        const double s = (double)s_idx[j] / N;
        yk[(size_t)k * n1 + j] = q0 + s * (c1 + s * (c2 + s * c3));
    }
    __syncthreads();
    // chord slopes (y[j+1] - y[j]) / (tk[j+1] - tk[j]): independent divisions, all lanes; parked in M, which the solves
    // only write once their forward eliminations are through
    for (int w = lane; w < dof * (n - 1); w += 64) {
        const int k = w / (n - 1), j = w - k * (n - 1);
        const double* y = yk + (size_t)k * n1;
        M[(size_t)k * n1 + j] = (y[j + 1] - y[j]) / (tk[j + 1] - tk[j]);
    }
    __syncthreads();
    if (lane < dof) {
        const int k = lane;
        const size_t o = (size_t)p * dof + k;
        const double q0 = a.p0[o], c1 = a.v0[o], dd = a.p1[o] - q0;
        const double c2 = 3.0 * dd - 2.0 * c1 - a.v1[o], c3 = -2.0 * dd + c1 + a.v1[o];
        double* Mk = M + (size_t)k * n1;
        double* c = cp + (size_t)k * n1;
        double* d = dp + (size_t)k * n1;
        const double d0 = c1 * sqrt(fmax(x[0], 0.0));
        const double d1 = (c1 + 2.0 * c2 + 3.0 * c3) * sqrt(fmax(x[N], 0.0));
        if (n == 1) Mk[0] = 0.0;
        else {
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
                if (j == 0) { c[0] = up / di; d[0] = rhs / di; }
                else {
                    const double m = di - lo * c[j - 1];
                    c[j] = up / m;
                    d[j] = (rhs - lo * d[j - 1]) / m;
                }
            }
            Mk[n - 1] = d[n - 1];
            for (int j = n - 2; j >= 0; --j) Mk[j] = d[j] - c[j] * Mk[j + 1];
        }
    }
    __syncthreads();
    const double T = tk[n - 1];
    const int length = (int)ceil(T / a.dt);
    const int wl = min(length, a.max_len);
    if (lane == 0) a.length[p] = length;
    for (int j = lane; j < wl; j += 64) {
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
        for (int k = 0; k < dof; ++k) {
            const double* y = yk + (size_t)k * n1;
            const double* Mk = M + (size_t)k * n1;
            double P_, V_, A_;
            if (n == 1) { P_ = y[0]; V_ = 0; A_ = 0; }
            else {
                const double h = tk[seg + 1] - tk[seg], aa = tk[seg + 1] - tt, bb = tt - tk[seg];
                const double ca = y[seg] / h - Mk[seg] * h / 6, cb = y[seg + 1] / h - Mk[seg + 1] * h / 6;
                P_ = Mk[seg] * aa * aa * aa / (6 * h) + Mk[seg + 1] * bb * bb * bb / (6 * h) + ca * aa + cb * bb;
                V_ = -Mk[seg] * aa * aa / (2 * h) + Mk[seg + 1] * bb * bb / (2 * h) - ca + cb;
                A_ = Mk[seg] * aa / h + Mk[seg + 1] * bb / h;
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
    if (!ctx || P <= 0 || dof <= 0 || N <= 0 || N > 4000 || max_len <= 0 || !(dt > 0) || !p0 || !p1 || !v0 || !v1 ||
        !x || !t || !pos || !vel || !acc || !times || !length)
        return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    sample_args a{P, dof, N, max_len, p0, p1, v0, v1, x, t, dt, pos, vel, acc, times, length};
    const size_t lds = (size_t)(1 + 4 * dof) * (N + 1) * sizeof(double);
    if (lds > 140 * 1024) { snprintf(ctx->err, sizeof(ctx->err), "sc_toppra_sample_batch: (1 + 4 dof)(N + 1) doubles exceed the LDS of a CU"); return SC_ERR_INVALID; }
    {
        int r_ = sc_allow_big_lds(ctx, reinterpret_cast<const void*>(toppra_sample_kernel), 140 * 1024);
        if (r_ != SC_OK) return r_;
    }
    int tk = sc_time_begin(ctx, SC_K_TOPPRA_SAMPLE);
    hipLaunchKernelGGL(toppra_sample_kernel, dim3((unsigned)P), dim3(64), lds, ctx->stream, a);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}
