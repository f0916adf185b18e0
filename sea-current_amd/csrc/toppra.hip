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
// One wavefront per plan.  Per stage the wave builds the 4*dof + 2 rows  alpha u + beta x <= gamma
// in LDS (one lane per row: computeParams fused, nothing is read from HBM but the 4*dof Hermite
// coefficients), splits them by the sign of alpha with a ballot, and eliminates u over all
// (upper, lower) pairs in parallel -- the exact solution of both stage LPs at once -- followed by a
// shuffle min/max reduction.  The sweeps are sequential in the stage index; throughput comes from
// running every plan of the batch concurrently.  This kernel is latency-bound, not HBM-bound.
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
template <bool IS_MIN>
__device__ __forceinline__ double wave_minmax(double v) {
    auto op = [](double a, double b) { return IS_MIN ? fmin(a, b) : fmax(a, b); };
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

template <bool LDSLIM>
__global__ void __launch_bounds__(64) toppra_kernel(toppra_args a) {
    __shared__ double s_al[TP_MAXROWS], s_be[TP_MAXROWS], s_ga[TP_MAXROWS];
    __shared__ double s_c1[TP_MAXDOF], s_c2[TP_MAXDOF], s_c3[TP_MAXDOF];
    __shared__ int s_up[TP_MAXROWS], s_lw[TP_MAXROWS];
    __shared__ double s_alo[TP_MAXDOF], s_ahi[TP_MAXDOF];
    extern __shared__ double s_dyn[];   // lds_limits: vlo [(N+1) or 1][dof], vhi [same], K [N+1][2]
    const int lane = threadIdx.x, p = blockIdx.x;
    const int dof = a.dof, N = a.N;
    const int nr = 4 * dof + 2;
    // A sweep is a chain of N dependent stages; a stage that waits for limits from HBM (a fresh line per stage) or for
    // the previous sweep's K costs more than its arithmetic.  Everything a stage reads is therefore staged in LDS once,
    // with coalesced loads, before the sweeps start.
    const int nv = (a.vlim_per_stage ? (N + 1) : 1) * dof;
    double* s_vlo = s_dyn;
    double* s_vhi = s_dyn + nv;
    double* s_K = s_dyn + 2 * nv;
    if (LDSLIM) {
        const size_t vo = a.vlim_per_stage ? (size_t)p * (N + 1) * dof : (size_t)p * dof;
        for (int k = lane; k < nv; k += 64) { s_vlo[k] = a.vlo[vo + k]; s_vhi[k] = a.vhi[vo + k]; }
    }
    if (lane < dof) { s_alo[lane] = a.alo[(size_t)p * dof + lane]; s_ahi[lane] = a.ahi[(size_t)p * dof + lane]; }
    if (lane < dof) {
        const size_t o = (size_t)p * dof + lane;
        const double d = a.p1[o] - a.p0[o], v0 = a.v0[o], v1 = a.v1[o];
        s_c1[lane] = v0;
        s_c2[lane] = 3.0 * d - 2.0 * v0 - v1;
        s_c3[lane] = -2.0 * d + v0 + v1;
    }
    __syncthreads();
    double* K = a.K + (size_t)p * (N + 1) * 2;
    double* X = a.x + (size_t)p * (N + 1);
    double* U = a.u + (size_t)p * N;
    double* T = a.t + (size_t)p * (N + 1);

    // rows of stage i with next-stage set [klo, khi]; returns x-bounds from the velocity constraint
    auto build = [&](int i, double klo, double khi, double& xlo, double& xhi) {
        const double s = (double)i / N, s1 = (double)(i + 1) / N, D = s1 - s;
        // velocity constraint (LinearJointVelocity::computeParams): lanes over dof
        double sdmin = -TP_MAXSD, sdmax = TP_MAXSD;
        if (lane < dof) {
            const double v = s_c1[lane] + s * (2.0 * s_c2[lane] + s * 3.0 * s_c3[lane]);
            double lo, hi;
            if (LDSLIM) {
                const int o = a.vlim_per_stage ? i * dof + lane : lane;
                lo = s_vlo[o]; hi = s_vhi[o];
            } else {
                const size_t o = a.vlim_per_stage ? ((size_t)p * (N + 1) + i) * dof + lane : (size_t)p * dof + lane;
                lo = a.vlo[o]; hi = a.vhi[o];
            }
            if (v > 0) { sdmax = fmin(hi / v, sdmax); sdmin = fmax(lo / v, sdmin); }
            else if (v < 0) { sdmax = fmin(lo / v, sdmax); sdmin = fmax(hi / v, sdmin); }
        }
        sdmin = wave_max(sdmin);
        sdmax = wave_min(sdmax);
        xlo = sdmin > 0 ? sdmin * sdmin : 0.0;
        xhi = sdmax * sdmax;
        // acceleration rows (collocation + interpolation) and the next-set rows: lanes over rows
        for (int r = lane; r < nr; r += 64) {
            double al, be, ga;
            if (r < 4 * dof) {
                const int k = r >> 2, var = r & 3;
                const double c1 = s_c1[k], c2 = s_c2[k], c3 = s_c3[k];
                double aa = c1 + s * (2.0 * c2 + s * 3.0 * c3), bb = 2.0 * c2 + 6.0 * c3 * s;
                if ((var & 2) && i < N) {
                    const double an = c1 + s1 * (2.0 * c2 + s1 * 3.0 * c3), bn = 2.0 * c2 + 6.0 * c3 * s1;
                    aa = an + 2.0 * D * bn;
                    bb = bn;
                }
                if (var & 1) { al = -aa; be = -bb; ga = -s_alo[k]; }
                else { al = aa; be = bb; ga = s_ahi[k]; }
            } else if (r == 4 * dof) { al = 2.0 * D; be = 1.0; ga = khi; }
            else { al = -2.0 * D; be = -1.0; ga = -klo; }
            s_al[r] = al; s_be[r] = be; s_ga[r] = ga;
        }
        __syncthreads();
    };

    int status = 0;
    // ---- backward pass: controllable sets ----
    double klo = a.sd_end * a.sd_end, khi = klo;
    if (lane == 0) { K[2 * N] = klo; K[2 * N + 1] = khi; }
    if (LDSLIM && lane == 0) { s_K[2 * N] = klo; s_K[2 * N + 1] = khi; }
    for (int i = N - 1; i >= 0; --i) {
        double lo, hi;
        build(i, klo, khi, lo, hi);
        bool infeasible = false;
        // classify rows by sign(alpha); alpha == 0 rows bound x directly
        int nu = 0, nl = 0;
        for (int r0 = 0; r0 < nr; r0 += 64) {
            const int r = r0 + lane;
            const double al = r < nr ? s_al[r] : 0.0;
            const bool up = r < nr && al > 0, lw = r < nr && al < 0;
            const unsigned long long mu = __ballot(up), ml = __ballot(lw);
            const unsigned long long lt = (1ull << lane) - 1ull;
            if (up) s_up[nu + __popcll(mu & lt)] = r;
            if (lw) s_lw[nl + __popcll(ml & lt)] = r;
            nu += __popcll(mu); nl += __popcll(ml);
            if (r < nr && al == 0.0) {
                const double be = s_be[r], ga = s_ga[r];
                if (be > 0) hi = fmin(hi, ga / be);
                else if (be < 0) lo = fmax(lo, ga / be);
                else if (ga < -TP_LP_TOL) infeasible = true;
            }
        }
        __syncthreads();
        const float inv_nl = nl > 0 ? 1.0f / (float)nl : 0.f;
        for (int pi = lane; pi < nu * nl; pi += 64) {
            // pi / nl without the integer-division expansion: pi < 66 * 66, so the float quotient of (pi + 0.5) is
            // at least 0.5 / 66 away from an integer and truncates to the exact result
            const int qi = (int)(((float)pi + 0.5f) * inv_nl);
            const int ri = s_up[qi], rj = s_lw[pi - qi * nl];
            const double ali = s_al[ri], bei = s_be[ri], gai = s_ga[ri];
            const double alj = s_al[rj], bej = s_be[rj], gaj = s_ga[rj];
            const double cf = alj * bei - ali * bej, rhs = alj * gai - ali * gaj;
            if (cf > 0) lo = fmax(lo, rhs / cf);
            else if (cf < 0) hi = fmin(hi, rhs / cf);
            else if (rhs > TP_LP_TOL) infeasible = true;
        }
        lo = wave_max(lo);
        hi = wave_min(hi);
        if (__ballot(infeasible) || lo > hi + TP_LP_TOL) { status = 1; break; }
        if (lo > hi) lo = hi;
        klo = lo > 0 ? lo : 0.0;
        khi = hi;
        if (lane == 0) { K[2 * i] = klo; K[2 * i + 1] = khi; }
        if (LDSLIM && lane == 0) { s_K[2 * i] = klo; s_K[2 * i + 1] = khi; }
        __syncthreads();
    }
    // ---- forward pass: greedy maximal u, knot times ----
    if (!status) {
        __threadfence_block();
        double x = a.sd_start * a.sd_start, tt = 0.0;
        if (x < klo - TP_LP_TOL || x > khi + TP_LP_TOL) status = 2;
        if (lane == 0) { X[0] = x; T[0] = 0.0; }
        for (int i = 0; i < N && !status; ++i) {
            double nlo, nhi;
            if (LDSLIM) { nlo = s_K[2 * (i + 1)]; nhi = s_K[2 * (i + 1) + 1]; }
            else { nlo = K[2 * (i + 1)]; nhi = K[2 * (i + 1) + 1]; }
            double lo, hi;
            build(i, nlo, nhi, lo, hi);
            double umax = INFINITY, umin = -INFINITY;
            for (int r = lane; r < nr; r += 64) {
                const double al = s_al[r], num = s_ga[r] - s_be[r] * x;
                if (al > 0) umax = fmin(umax, num / al);
                else if (al < 0) umin = fmax(umin, num / al);
            }
            umax = wave_min(umax);
            umin = wave_max(umin);
            if (!(umax >= umin - 1e-6) || !isfinite(umax)) { status = 2; break; }
            const double D = (double)(i + 1) / N - (double)i / N;
            double xn = x + 2.0 * D * umax;
            if (xn > nhi) xn = nhi;
            if (xn < nlo) xn = nlo;
            const double sda = 0.5 * (sqrt(fmax(x, 0.0)) + sqrt(fmax(xn, 0.0)));
            tt += sda > TP_NEARLY_ZERO ? D / sda : 5.0;
            if (lane == 0) { U[i] = umax; X[i + 1] = xn; T[i + 1] = tt; }
            x = xn;
            __syncthreads();
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
    const size_t lds = ((size_t)2 * (vlim_per_stage ? (N + 1) : 1) * dof + (size_t)2 * (N + 1)) * sizeof(double);
    const int lds_limits = lds <= 56 * 1024 ? 1 : 0;   // otherwise the sweeps read limits and K from global memory
    toppra_args a{P, dof, N, p0, p1, v0, v1, vlim_lo, vlim_hi, alim_lo, alim_hi, vlim_per_stage, lds_limits,
                  sd_start, sd_end, K, x, u, t, status};
    int tk = sc_time_begin(ctx, SC_K_TOPPRA);
    if (lds_limits) hipLaunchKernelGGL(toppra_kernel<true>, dim3(P), dim3(64), lds, ctx->stream, a);
    else hipLaunchKernelGGL(toppra_kernel<false>, dim3(P), dim3(64), 0, ctx->stream, a);
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

// one block (64 threads) per (plan, dof); dynamic LDS: 5 * (N+1) doubles
__global__ void __launch_bounds__(64) toppra_sample_kernel(sample_args a) {
    extern __shared__ double sm[];
    const int N = a.N, n1 = N + 1;
    double* tk = sm;
    double* yk = tk + n1;
    double* M = yk + n1;
    double* cp = M + n1;
    double* dp = cp + n1;
    __shared__ int s_n;
    const int lane = threadIdx.x;
    const int p = blockIdx.x / a.dof, k = blockIdx.x % a.dof;
    const size_t o = (size_t)p * a.dof + k;
    const double q0 = a.p0[o], c1 = a.v0[o], dd = a.p1[o] - q0;
    const double c2 = 3.0 * dd - 2.0 * c1 - a.v1[o], c3 = -2.0 * dd + c1 + a.v1[o];
    const double* t = a.t + (size_t)p * n1;
    const double* x = a.x + (size_t)p * n1;
    if (lane == 0) {
        // knots with a (nearly) zero time increment are dropped, as parametrizer::Spline does
        int n = 0;
        for (int i = 0; i <= N; ++i)
            if (i == 0 || t[i] - t[i - 1] >= TP_NEARLY_ZERO) {
                const double s = (double)i / N;
                tk[n] = t[i];
                yk[n] = q0 + s * (c1 + s * (c2 + s * c3));
                ++n;
            }
        s_n = n;
        const double d0 = c1 * sqrt(fmax(x[0], 0.0));
        const double d1 = (c1 + 2.0 * c2 + 3.0 * c3) * sqrt(fmax(x[N], 0.0));
        if (n == 1) M[0] = 0.0;
        else {
            for (int j = 0; j < n; ++j) {
                double lo, di, up, rhs;
                if (j == 0) {
                    const double h = tk[1] - tk[0];
                    lo = 0; di = 2 * h; up = h; rhs = 6 * ((yk[1] - yk[0]) / h - d0);
                } else if (j == n - 1) {
                    const double h = tk[j] - tk[j - 1];
                    lo = h; di = 2 * h; up = 0; rhs = 6 * (d1 - (yk[j] - yk[j - 1]) / h);
                } else {
                    const double h0 = tk[j] - tk[j - 1], h1 = tk[j + 1] - tk[j];
                    lo = h0; di = 2 * (h0 + h1); up = h1;
                    rhs = 6 * ((yk[j + 1] - yk[j]) / h1 - (yk[j] - yk[j - 1]) / h0);
                }
                if (j == 0) { cp[0] = up / di; dp[0] = rhs / di; }
                else {
                    const double m = di - lo * cp[j - 1];
                    cp[j] = up / m;
                    dp[j] = (rhs - lo * dp[j - 1]) / m;
                }
            }
            M[n - 1] = dp[n - 1];
            for (int j = n - 2; j >= 0; --j) M[j] = dp[j] - cp[j] * M[j + 1];
        }
    }
    __syncthreads();
    const int n = s_n;
    const double T = tk[n - 1];
    const int length = (int)ceil(T / a.dt);
    const int wl = min(length, a.max_len);
    if (k == 0 && lane == 0) a.length[p] = length;
    float* pos = a.pos + o * a.max_len;
    float* vel = a.vel + o * a.max_len;
    float* acc = a.acc + o * a.max_len;
    for (int j = lane; j < wl; j += 64) {
        const double tt = length > 1 ? (j == length - 1 ? T : (T * j) / (length - 1)) : 0.0;
        double P_, V_, A_;
        if (n == 1) { P_ = yk[0]; V_ = 0; A_ = 0; }
        else {
            // largest seg with tk[seg] < tt (seg = 0 if none), capped at n-2
            int lo = 0, hi = n - 2;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (tk[mid] < tt) lo = mid; else hi = mid - 1;
            }
            const int seg = lo;
            const double h = tk[seg + 1] - tk[seg], aa = tk[seg + 1] - tt, bb = tt - tk[seg];
            const double ca = yk[seg] / h - M[seg] * h / 6, cb = yk[seg + 1] / h - M[seg + 1] * h / 6;
            P_ = M[seg] * aa * aa * aa / (6 * h) + M[seg + 1] * bb * bb * bb / (6 * h) + ca * aa + cb * bb;
            V_ = -M[seg] * aa * aa / (2 * h) + M[seg + 1] * bb * bb / (2 * h) - ca + cb;
            A_ = M[seg] * aa / h + M[seg + 1] * bb / h;
        }
        pos[j] = (float)P_; vel[j] = (float)V_; acc[j] = (float)A_;
        if (k == 0) a.times[(size_t)p * a.max_len + j] = tt;
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
    const size_t lds = (size_t)5 * (N + 1) * sizeof(double);
    {
        int r_ = sc_allow_big_lds(ctx, reinterpret_cast<const void*>(toppra_sample_kernel), 160 * 1024 - 64);
        if (r_ != SC_OK) return r_;
    }
    int tk = sc_time_begin(ctx, SC_K_TOPPRA_SAMPLE);
    hipLaunchKernelGGL(toppra_sample_kernel, dim3((unsigned)(P * dof)), dim3(64), lds, ctx->stream, a);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}
