// bezier.hip -- batched path smoothing and arclength (gfx950).  SURVEY.md 8f rank 1-2 ("next" rows).
//
// Takes over bezier_spline::from_path (sea_current.hpp:599-683: Lau09 tangent heuristics :343-377,
// shrink_tangent :575-596), the cubic Bezier / hodograph evaluation (:700-763, :1041-1053) and
// bezier_spline::arclength (:767-896: 32-point Gauss-Legendre on 1/precision sub-intervals per segment).
// The maths is restated in oracle/bezier_oracle.c, which is pinned to examples/output.json (total
// arclength 5e-8 relative, cumulative tables 2e-6 absolute).  Curves are evaluated in Bernstein form
// (the reference evaluates the same polynomial through its Bernstein-Fourier form) in fp64.
//
// Parallelism: one thread per (path, waypoint) for the tangents; one wavefront per segment for the
// arclength (lanes over sub-intervals, 32 quadrature points each, cumulative table by a wave scan);
// one thread per sample for evaluation.  All three are tiny next to EDT and A*.
#include "sc_internal.h"

#include <cmath>
#include <utility>
#include <stdlib.h>

struct seg4 { float lx0, ly0, lx1, ly1; };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}


__device__ __forceinline__ bool seg_hit(double p1x, double p1y, double p2x, double p2y, const float* l, double* ix, double* iy) {
    const double rx = p2x - p1x, ry = p2y - p1y, sx = (double)l[2] - l[0], sy = (double)l[3] - l[1];
    const double den = rx * sy - ry * sx;
    if (den == 0.0) return false;
    const double qx = l[0] - p1x, qy = l[1] - p1y;
    const double t = (qx * sy - qy * sx) / den, u = (qx * ry - qy * rx) / den;
    if (t < 0 || t > 1 || u < 0 || u > 1) return false;
    *ix = p1x + t * rx; *iy = p1y + t * ry;
    return true;
}

__device__ __forceinline__ void shrink(double& tx, double& ty, double wx, double wy, const float* lines, int nlines) {
    for (int i = 0; i < nlines; ++i) {
        double ix, iy;
        if (seg_hit(wx + tx, wy + ty, wx, wy, lines + 4 * i, &ix, &iy)) { tx = ix - wx; ty = iy - wy; }
        if (seg_hit(wx - tx, wy - ty, wx, wy, lines + 4 * i, &ix, &iy)) { tx = wx - ix; ty = wy - iy; }
    }
}

// tangent of waypoint i of path p -> tang [P][n_max][2]
__global__ void __launch_bounds__(256)
bezier_tangent_kernel(const float* __restrict__ path, const int32_t* __restrict__ npts, int P, int n_max, float start_angle,
                      const float* __restrict__ lines, int nlines, double* __restrict__ tang) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= P * n_max) return;
    const int p = gid / n_max, i = gid % n_max;
    const int n = npts[p];
    if (i >= n || n < 2) return;
    const float* w = path + (size_t)p * n_max * 2;
    auto X = [&](int k) { return (double)w[2 * k]; };
    auto Y = [&](int k) { return (double)w[2 * k + 1]; };
    double tx, ty;
    if (i == 0) {
        double th = start_angle;
        if (isnan(start_angle)) th = atan2(Y(1) - Y(0), X(1) - X(0));
        const double m = 0.5 * hypot(X(1) - X(0), Y(1) - Y(0));
        tx = m * cos(th); ty = m * sin(th);
    } else if (i == n - 1) {
        const double ex = X(n - 1) - X(n - 2), ey = Y(n - 1) - Y(n - 2), el = hypot(ex, ey);
        tx = 0.5 * ex; ty = 0.5 * ey;   // 0.5 * |e| * e / |e|
        (void)el;
    } else {
        const double ux = X(i - 1) - X(i), uy = Y(i - 1) - Y(i), vx = X(i + 1) - X(i), vy = Y(i + 1) - Y(i);
        const double theta = acos((ux * vx + uy * vy) / (hypot(ux, uy) * hypot(vx, vy))) / 2;
        const double off = atan2(uy, ux), toff = atan2(vy, vx);
        const int mult = (toff - off < 0) ? -1 : 1;
        double lx = sin(off + mult * theta), ly = -cos(off + mult * theta);
        const double ll = hypot(lx, ly);
        lx /= ll; ly /= ll;
        const int mult2 = hypot(X(i) + lx - X(i + 1), Y(i) + ly - Y(i + 1)) < hypot(X(i) - lx - X(i + 1), Y(i) - ly - Y(i + 1)) ? 1 : -1;
        const double mag = 0.5 * fmin(hypot(ux, uy), hypot(vx, vy));
        tx = mag * mult2 * lx; ty = mag * mult2 * ly;
    }
    shrink(tx, ty, X(i), Y(i), lines, nlines);
    tang[((size_t)p * n_max + i) * 2] = tx;
    tang[((size_t)p * n_max + i) * 2 + 1] = ty;
}

// control points of leg i of path p -> ctrl [P][n_max-1][4][2]
__global__ void __launch_bounds__(256)
bezier_ctrl_kernel(const float* __restrict__ path, const int32_t* __restrict__ npts, int P, int n_max,
                   const double* __restrict__ tang, float* __restrict__ ctrl) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= P * (n_max - 1)) return;
    const int p = gid / (n_max - 1), i = gid % (n_max - 1);
    float* c = ctrl + ((size_t)p * (n_max - 1) + i) * 8;
    if (i >= npts[p] - 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = 0.f;
        return;
    }
    const float* w = path + ((size_t)p * n_max + i) * 2;
    const double* t = tang + ((size_t)p * n_max + i) * 2;
    c[0] = w[0]; c[1] = w[1];
    c[2] = (float)((double)w[0] + t[0]); c[3] = (float)((double)w[1] + t[1]);
    c[4] = (float)((double)w[2] - t[2]); c[5] = (float)((double)w[3] - t[3]);
    c[6] = w[2]; c[7] = w[3];
}

__device__ __forceinline__ void bez_eval(const float* c, double s, int order, double& ox, double& oy) {
    const double r = 1 - s;
    double o[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const double p0 = c[a], p1 = c[2 + a], p2 = c[4 + a], p3 = c[6 + a];
        if (order == 0) o[a] = r * r * r * p0 + 3 * r * r * s * p1 + 3 * r * s * s * p2 + s * s * s * p3;
        else if (order == 1) o[a] = 3 * (r * r * (p1 - p0) + 2 * r * s * (p2 - p1) + s * s * (p3 - p2));
        else o[a] = 6 * (r * (p2 - 2 * p1 + p0) + s * (p3 - 2 * p2 + p1));
    }
    ox = o[0]; oy = o[1];
}

__global__ void __launch_bounds__(256)
bezier_eval_kernel(const float* __restrict__ ctrl, const int32_t* __restrict__ seg, const float* __restrict__ t, int M, int order,
                   float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    double x, y;
    bez_eval(ctrl + (size_t)seg[i] * 8, (double)t[i], order, x, y);
    out[2 * i] = (float)x; out[2 * i + 1] = (float)y;
}

// one wavefront per segment; gl = 32 nodes then 32 weights; cum [S][nsub+1], seg_len [S]
__global__ void __launch_bounds__(64)
bezier_arclength_kernel(const float* __restrict__ ctrl, int S, int nsub, const double* __restrict__ gl, float* __restrict__ cum,
                        float* __restrict__ seg_len) {
    extern __shared__ double sub[];   // [nsub] arclength of each sub-interval
    const int s = blockIdx.x, lane = threadIdx.x;
    const float* c = ctrl + (size_t)s * 8;
    for (int k = lane; k < nsub; k += 64) {
        const double a = (double)k / nsub, b = (double)(k + 1) / nsub;
        double sum = 0;
        for (int i = 0; i < 32; ++i) {
            double dx, dy;
            bez_eval(c, 0.5 * (b - a) * gl[i] + 0.5 * (b + a), 1, dx, dy);
            sum += gl[32 + i] * sqrt(dx * dx + dy * dy);
        }
        sub[k] = sum * 0.5 * (b - a);
    }
    __syncthreads();
    // cumulative table: each lane owns a contiguous chunk, wave scan of the chunk sums
    const int chunk = (nsub + 63) / 64;
    const int k0 = lane * chunk, k1 = min(nsub, k0 + chunk);
    double local = 0;
    for (int k = k0; k < k1; ++k) local += sub[k];
    double incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    double run = incl - local;
    float* out = cum + (size_t)s * (nsub + 1);
    if (lane == 0) out[0] = 0.f;
    for (int k = k0; k < k1; ++k) { run += sub[k]; out[k + 1] = (float)run; }
    if (lane == 63) seg_len[s] = (float)incl;
}

// ---- resample (sea_current.hpp:898-1005, chebfit / chebeval :1109-1170, curvature :1017-1039) ------------------
// Kernel 1, one wavefront per spline: the reference's sequential "nudge" of the profile positions (chunks of 64
// samples are checked in parallel; a chunk with a sample to fix is replayed in order by lane 0, which keeps the
// sequential semantics exactly) and the split of the samples over the segments (first sample past the end of a
// segment's table, found by ballot).  seginfo [S] = (first sample, last sample, spline, segment within spline).
#define RESAMPLE_STAGE_MAX 12288      // profile samples of a spline staged in LDS (48 KiB); longer profiles are scanned in HBM
__global__ void __launch_bounds__(64)
resample_prepare_kernel(const float* __restrict__ cum, const int32_t* __restrict__ seg_off, int nsub, const float* __restrict__ arclen,
                        float* pp_all, const int32_t* __restrict__ prof_off, int nudge, int4* __restrict__ seginfo,
                        int32_t* __restrict__ status) {
    extern __shared__ float spp[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int s0 = seg_off[b], nseg = seg_off[b + 1] - s0;
    float* gpp = pp_all + prof_off[b];
    const int n = prof_off[b + 1] - prof_off[b];
    if (n <= 0 || nseg <= 0) {
        if (lane == 0) status[b] = 1;
        for (int i = lane; i < nseg; i += 64) seginfo[s0 + i] = make_int4(0, -1, b, i);
        return;
    }
    const float AL = arclen[b];
    // The scans below touch every sample a few times, 64 at a time, each time waiting for the loads: from HBM that was a
    // microsecond per step (116 us for the bench's 1500-sample profiles).  The profile is staged in LDS once, coalesced;
    // the nudge's corrections go to both copies.
    const bool staged = n <= RESAMPLE_STAGE_MAX;
    if (staged) {
        for (int i = lane; i < n; i += 64) spp[i] = gpp[i];
        __syncthreads();
    }
    // the body once per address space (a pointer that may be either compiles to flat loads that wait for every store)
    // One wavefront per spline: lane 0's corrections have to be ordered before the other lanes' later reads, nothing more.
    // (__threadfence() is an agent-scope fence: on gfx950 it writes the L2 back and invalidates it -- 40 us a time here.)
    auto wave_order = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); };
    auto body = [&](float* const pp, const bool mirror) {
        if (nudge) {
            if (lane == 0) { pp[0] = 0.f; pp[n - 1] = AL; if (mirror) { gpp[0] = 0.f; gpp[n - 1] = AL; } }
            wave_order();
            for (int base = 1; base < n - 1; base += 64) {
                const int i = base + lane;
                bool fix = false;
                if (i < n - 1) {
                    const float a = pp[i - 1], v = pp[i], c = pp[i + 1];
                    fix = v < a || v > c || v < 0.f || v > AL;
                }
                if (__ballot(fix) == 0ull) continue;
                if (lane == 0) {
                    // in order, as the reference does it: sample k against its already corrected predecessor (carried in a
                    // register) and its not yet visited successor
                    const int e = min(base + 64, n - 1);
                    float prev = pp[base - 1], cur = pp[base];
                    for (int k = base; k < e; ++k) {
                        const float nxt = pp[k + 1];
                        float v = cur;
                        if (v < prev || v > nxt) v = (prev + nxt) / 2;
                        if (v < 0.f) v = 0.f;
                        if (v > AL) v = AL;
                        pp[k] = v;
                        if (mirror) gpp[k] = v;
                        prev = v; cur = nxt;
                    }
                }
                wave_order();
            }
        }
        const int m = nsub + 1;
        int j = 0, st = 0;
        float offset = 0.f;
        for (int i = 0; i < nseg; ++i) {
            const float last = cum[(size_t)(s0 + i) * m + m - 1];
            const int start = j;
            while (j < n) {
                const int k = j + lane;
                const bool past = k >= n || !(pp[k] - offset <= last);
                const unsigned long long bal = __ballot(past);
                if (bal) { j += __ffsll((long long)bal) - 1; break; }
                j += 64;
            }
            if (j > n) j = n;
            if (i + 1 == nseg && i == 0) j = n;
            else if (i + 1 == nseg) j = n - 1;
            j -= 1;
            if (j < start) { st = 1; j = start; }
            offset = pp[j];
            if (lane == 0) seginfo[s0 + i] = make_int4(start, j, b, i);
        }
        if (lane == 0) status[b] = st;
    };
    if (staged) body(spp, true);
    else body(gpp, false);
}


// Kernel 2, one workgroup per segment: wave 0 fits the Chebyshev polynomial arclength -> parameter by Householder
// least squares in fp64 ([T | y] of (nsub+1) x (deg+1) in LDS, rows over lanes), then all threads evaluate the
// segment's block of samples: parameter, point, curvature.
__global__ void __launch_bounds__(256)
resample_eval_kernel(const float* __restrict__ ctrl, const float* __restrict__ cum, int nsub, const float* __restrict__ pp_all,
                     const int32_t* __restrict__ prof_off, const int4* __restrict__ seginfo, float* __restrict__ pts,
                     float* __restrict__ tpar, int32_t* __restrict__ seg, float* __restrict__ curv) {
    extern __shared__ double A[];                 // [m][nc]
    __shared__ double coef[10];
    __shared__ double xr[2];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int4 info = seginfo[s];
    const int m = nsub + 1, deg = m < 10 ? m : 10, nc = deg + 1;
    const float* tab = cum + (size_t)s * m;
    if (tid < 64) {
        const int lane = tid;
        float mn = INFINITY, mx = -INFINITY;
        for (int r = lane; r < m; r += 64) { mn = fminf(mn, tab[r]); mx = fmaxf(mx, tab[r]); }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
        const double xmin = mn, xmax = mx;
        const float prec = 1.0f / (float)nsub;
        for (int r = lane; r < m; r += 64) {
            const double xn = (2 * (double)tab[r] - (xmax + xmin)) / (xmax - xmin);
            double* a = A + (size_t)r * nc;
            a[0] = 1;
            if (deg > 1) a[1] = xn;
            for (int j = 2; j < deg; ++j) a[j] = 2 * xn * a[j - 1] - a[j - 2];
            const float v = (float)r * prec;
            a[deg] = v < 1.0f ? v : 1.0f;
        }
        wave_lds_sync();
        for (int k = 0; k < deg && k < m; ++k) {
            double part = 0;
            for (int r = k + lane; r < m; r += 64) part += A[(size_t)r * nc + k] * A[(size_t)r * nc + k];
            const double n2 = wave_sum(part), nrm = sqrt(n2);
            if (nrm == 0) continue;
            const double akk = A[(size_t)k * nc + k], alpha = akk > 0 ? -nrm : nrm;
            const double vk = akk - alpha, vtv = nrm * nrm - akk * akk + vk * vk;
            wave_lds_sync();                      // everyone has read a_kk before column k's head is touched
            for (int j = k + 1; j < nc; ++j) {
                double d = 0;
                for (int r = k + 1 + lane; r < m; r += 64) d += A[(size_t)r * nc + k] * A[(size_t)r * nc + j];
                d = wave_sum(d) + vk * A[(size_t)k * nc + j];
                const double f = 2 * d / vtv;
                wave_lds_sync();
                for (int r = k + 1 + lane; r < m; r += 64) A[(size_t)r * nc + j] -= f * A[(size_t)r * nc + k];
                if (lane == 0) A[(size_t)k * nc + j] -= f * vk;
                wave_lds_sync();
            }
            if (lane == 0) A[(size_t)k * nc + k] = alpha;
            wave_lds_sync();
        }
        if (lane == 0) {
            for (int k = deg - 1; k >= 0; --k) {
                double v = A[(size_t)k * nc + deg];
                for (int j = k + 1; j < deg; ++j) v -= A[(size_t)k * nc + j] * coef[j];
                coef[k] = v / A[(size_t)k * nc + k];
            }
            xr[0] = xmin; xr[1] = xmax;
        }
    }
    __syncthreads();
    const int start = info.x, end = info.y, b = info.z, il = info.w;
    const int p0 = prof_off[b], n = prof_off[b + 1] - p0;
    const float* pp = pp_all + p0;
    const double xmin = xr[0], xmax = xr[1];
    const float* c = ctrl + (size_t)s * 8;
    for (int k = start + tid; k <= end; k += 256) {
        const int o = k + il;
        if (o >= n) break;
        const float xb = pp[k] - pp[start];
        const double xn = (2 * (double)xb - (xmax + xmin)) / (xmax - xmin);
        double t0 = 1, t1 = xn, y = coef[0];
        if (deg > 1) y += coef[1] * t1;
        for (int j = 2; j < deg; ++j) { const double t2 = 2 * xn * t1 - t0; y += coef[j] * t2; t0 = t1; t1 = t2; }
        float t = (float)y;
        if (t < 0.f) t = 0.f;
        if (t > 1.f) t = 1.f;
        const size_t og = (size_t)p0 + o;
        double px, py;
        bez_eval(c, (double)t, 0, px, py);
        if (pts) { pts[2 * og] = (float)px; pts[2 * og + 1] = (float)py; }
        if (tpar) tpar[og] = t;
        if (seg) seg[og] = il;
        if (curv) {
            double ax, ay, bx, by;
            bez_eval(c, (double)t, 1, ax, ay);
            bez_eval(c, (double)t, 2, bx, by);
            curv[og] = (float)((ax * by - ay * bx) / pow(ax * ax + ay * ay, 1.5));
        }
    }
}

// Kernel 2 in registers, one WAVEFRONT per segment, for tables of 10 .. 64 RPL rows (nsub = 100: RPL 2): the [T | y] matrix
// lives in the lanes' registers (row r in lane r % 64, slot r / 64; rows past the table are zero and stay zero), a dot product
// over the rows is a DPP reduction (quad_perm, row_half_mirror, row_mirror, row_bcast 15 / 31: no LDS, no barrier), and the
// pivot row k is lane k's slot 0, read with v_readlane.  The Householder steps of the LDS form above, column k against all the
// columns to its right at once (their dot products are reduced together; 2 / v'v is formed once per column) --
// which spent 86 us per segment on 55 (k, j) pairs of LDS passes and shuffles by one wavefront of four (630 us for the
// bench's 15 k segments); the sums over rows are grouped differently (fp64: 1e-16 relative).
template <int CTRL, int ROW_MASK, bool ALL>
__device__ __forceinline__ double dpp_f64(double v) {
    // ALL: every lane has a source lane (the permutations inside a row); otherwise lanes the row mask leaves out read 0
    const long long b = __double_as_longlong(v);
    int lo, hi;
    if (ALL) {
        lo = __builtin_amdgcn_mov_dpp((int)(b & 0xFFFFFFFFll), CTRL, ROW_MASK, 0xF, true);
        hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, ROW_MASK, 0xF, true);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, ROW_MASK, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    }
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// the sums over the 64 lanes of N values at once (step by step over all of them: the N chains are independent, so no
// DPP wait states and no dependent adds back to back); the same totals in every lane
template <int N>
__device__ __forceinline__ void wave_sum_dpp(double (&v)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] += dpp_f64<0xB1, 0xF, true>(v[j]);     // quad_perm [1,0,3,2]
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] += dpp_f64<0x4E, 0xF, true>(v[j]);     // quad_perm [2,3,0,1]
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] += dpp_f64<0x141, 0xF, true>(v[j]);    // row_half_mirror
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] += dpp_f64<0x140, 0xF, true>(v[j]);    // row_mirror: every lane of a row of 16 holds the row's sum
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] += dpp_f64<0x142, 0xA, false>(v[j]);   // row_bcast 15 into rows 1 and 3
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] += dpp_f64<0x143, 0xC, false>(v[j]);   // row_bcast 31 into rows 2 and 3: lane 63 holds the total
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = readlane_f64(v[j], 63);
}

// Householder step of column K of the [T | y] matrix in registers (rows: lane + 64 slot), against all the columns to its right
template <int K, int RPL>
__device__ __forceinline__ void hh_col(double (&a)[RPL][11], const int lane) {
    constexpr int NC = 11;
    double part[1] = {lane >= K ? a[0][K] * a[0][K] : 0.0};
#pragma unroll
    for (int i = 1; i < RPL; ++i) part[0] += a[i][K] * a[i][K];
    wave_sum_dpp<1>(part);
    const double nrm = sqrt(part[0]);
    if (nrm == 0) return;                                     // wave-uniform
    const double akk = readlane_f64(a[0][K], K), alpha = akk > 0 ? -nrm : nrm;
    const double vk = akk - alpha, vtv = nrm * nrm - akk * akk + vk * vk;
    const double tvv = 2 / vtv;
    double d[NC - 1 - K];                                     // the dot products of column K with every column to its right
#pragma unroll
    for (int j = K + 1; j < NC; ++j) {
        double t = lane > K ? a[0][K] * a[0][j] : 0.0;
#pragma unroll
        for (int i = 1; i < RPL; ++i) t += a[i][K] * a[i][j];
        d[j - K - 1] = t;
    }
    wave_sum_dpp<NC - 1 - K>(d);                              // reduced together
#pragma unroll
    for (int j = K + 1; j < NC; ++j) {
        const double f = (d[j - K - 1] + vk * readlane_f64(a[0][j], K)) * tvv;
        if (lane > K) a[0][j] -= f * a[0][K];
        else if (lane == K) a[0][j] -= f * vk;
#pragma unroll
        for (int i = 1; i < RPL; ++i) a[i][j] -= f * a[i][K];
    }
    if (lane == K) a[0][K] = alpha;
}
template <int RPL, int... Ks>
__device__ __forceinline__ void hh_all(double (&a)[RPL][11], const int lane, std::integer_sequence<int, Ks...>) {
    (hh_col<Ks, RPL>(a, lane), ...);
}

template <int RPL>
__global__ void __launch_bounds__(256)
resample_eval_reg_kernel(const float* __restrict__ ctrl, const float* __restrict__ cum, int nsub, int S, const float* __restrict__ pp_all,
                         const int32_t* __restrict__ prof_off, const int4* __restrict__ seginfo, float* __restrict__ pts,
                         float* __restrict__ tpar, int32_t* __restrict__ seg, float* __restrict__ curv) {
    constexpr int DEG = 10, NC = DEG + 1;
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    const int4 info = seginfo[s];
    const int m = nsub + 1;
    const float* tab = cum + (size_t)s * m;
    float mn = INFINITY, mx = -INFINITY;
    float tv[RPL];
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int r = lane + 64 * i;
        tv[i] = r < m ? tab[r] : 0.f;
        if (r < m) { mn = fminf(mn, tv[i]); mx = fmaxf(mx, tv[i]); }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    const double xmin = mn, xmax = mx;
    const float prec = 1.0f / (float)nsub;
    double a[RPL][NC];
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int r = lane + 64 * i;
        const double xn = (2 * (double)tv[i] - (xmax + xmin)) / (xmax - xmin);
        a[i][0] = 1;
        a[i][1] = xn;
#pragma unroll
        for (int j = 2; j < DEG; ++j) a[i][j] = 2 * xn * a[i][j - 1] - a[i][j - 2];
        const float v = (float)r * prec;
        a[i][DEG] = v < 1.0f ? v : 1.0f;
        if (r >= m) {
#pragma unroll
            for (int j = 0; j < NC; ++j) a[i][j] = 0;
        }
    }
    hh_all<RPL>(a, lane, std::make_integer_sequence<int, DEG>{});
    // back substitution on rows 0 .. 9 (lanes 0 .. 9, slot 0)
    double coef[DEG];
#pragma unroll
    for (int k = DEG - 1; k >= 0; --k) {
        double v = a[0][DEG];
#pragma unroll
        for (int j = k + 1; j < DEG; ++j) v -= a[0][j] * coef[j];
        coef[k] = readlane_f64(v / a[0][k], k);
    }
    const int start = info.x, end = info.y, b = info.z, il = info.w;
    const int p0 = prof_off[b], n = prof_off[b + 1] - p0;
    const float* pp = pp_all + p0;
    const float* c = ctrl + (size_t)s * 8;
    const float pstart = end >= start ? pp[start] : 0.f;
    for (int k = start + lane; k <= end; k += 64) {
        const int o = k + il;
        if (o >= n) break;
        const float xb = pp[k] - pstart;
        const double xn = (2 * (double)xb - (xmax + xmin)) / (xmax - xmin);
        double t0 = 1, t1 = xn, y = coef[0] + coef[1] * t1;
#pragma unroll
        for (int j = 2; j < DEG; ++j) { const double t2 = 2 * xn * t1 - t0; y += coef[j] * t2; t0 = t1; t1 = t2; }
        float t = (float)y;
        if (t < 0.f) t = 0.f;
        if (t > 1.f) t = 1.f;
        const size_t og = (size_t)p0 + o;
        double px, py;
        bez_eval(c, (double)t, 0, px, py);
        if (pts) { pts[2 * og] = (float)px; pts[2 * og + 1] = (float)py; }
        if (tpar) tpar[og] = t;
        if (seg) seg[og] = il;
        if (curv) {
            double ax, ay, bx, by;
            bez_eval(c, (double)t, 1, ax, ay);
            bez_eval(c, (double)t, 2, bx, by);
            curv[og] = (float)((ax * by - ay * bx) / pow(ax * ax + ay * ay, 1.5));
        }
    }
}

// ---- general-degree curve (bezier_spline::bezier_curve :700-763, hodograph control points :1041-1053) -------------
// de Casteljau in fp64 on degree + 1 control points; the reference evaluates the same polynomial through its
// Bernstein-Fourier form.  Cubics keep bez_eval above (what the recorded run is pinned against).
__global__ void __launch_bounds__(256)
bezier_curve_kernel(const float* __restrict__ ctrl, int deg, const int32_t* __restrict__ seg, const float* __restrict__ t, int M,
                    float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float* c = ctrl + (size_t)seg[i] * (deg + 1) * 2;
    double bx[SC_BEZIER_MAX_DEGREE + 1], by[SC_BEZIER_MAX_DEGREE + 1];
#pragma unroll
    for (int j = 0; j <= SC_BEZIER_MAX_DEGREE; ++j) { bx[j] = j <= deg ? (double)c[2 * j] : 0.0; by[j] = j <= deg ? (double)c[2 * j + 1] : 0.0; }
    const double s = t[i], r = 1.0 - s;
#pragma unroll
    for (int lvl = 1; lvl <= SC_BEZIER_MAX_DEGREE; ++lvl)
#pragma unroll
        for (int j = 0; j + lvl <= SC_BEZIER_MAX_DEGREE; ++j)
            if (lvl <= deg && j + lvl <= deg) { bx[j] = r * bx[j] + s * bx[j + 1]; by[j] = r * by[j] + s * by[j + 1]; }
    out[2 * i] = (float)bx[0]; out[2 * i + 1] = (float)by[0];
}

// ---- free chebfit / chebeval (sea_current.hpp:1109-1170) ------------------------------------------------------
// Least squares over the columns T_0 .. T_{degree-1} of the abscissa normalised to [-1, 1] (the reference builds
// `degree` columns, :1122-1128), by Householder reflections in fp64 (Eigen's HouseholderQR in float32 there).  One
// workgroup per problem; [T | y] lives in global scratch (a problem may have tens of thousands of rows:
// examples/test.cpp:168 fits 20 002 curve points), column operations are block reductions.
__device__ __forceinline__ double block_sum256(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256)
cheb_fit_kernel(const float* __restrict__ x, const float* __restrict__ y, const int32_t* __restrict__ off, int degree, double* __restrict__ Aall,
                float* __restrict__ coef_out, float* __restrict__ xrange) {
    __shared__ double red[4];
    __shared__ double coef[SC_CHEB_MAX_DEGREE];
    __shared__ float mm[2][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int r0 = off[b], m = off[b + 1] - r0, nc = degree + 1;
    double* A = Aall + (size_t)r0 * nc;
    const float* xb = x + r0;
    const float* yb = y + r0;
    float mn = INFINITY, mx = -INFINITY;
    for (int r = tid; r < m; r += 256) { mn = fminf(mn, xb[r]); mx = fmaxf(mx, xb[r]); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if ((tid & 63) == 0) { mm[0][tid >> 6] = mn; mm[1][tid >> 6] = mx; }
    __syncthreads();
    const double xmin = fminf(fminf(mm[0][0], mm[0][1]), fminf(mm[0][2], mm[0][3]));
    const double xmax = fmaxf(fmaxf(mm[1][0], mm[1][1]), fmaxf(mm[1][2], mm[1][3]));
    for (int r = tid; r < m; r += 256) {
        const double xn = (2 * (double)xb[r] - (xmax + xmin)) / (xmax - xmin);
        double* a = A + (size_t)r * nc;
        a[0] = 1;
        if (degree > 1) a[1] = xn;
        for (int j = 2; j < degree; ++j) a[j] = 2 * xn * a[j - 1] - a[j - 2];
        a[degree] = yb[r];
    }
    __syncthreads();
    for (int k = 0; k < degree && k < m; ++k) {
        double part = 0;
        for (int r = k + tid; r < m; r += 256) part += A[(size_t)r * nc + k] * A[(size_t)r * nc + k];
        const double n2 = block_sum256(part, red), nrm = sqrt(n2);
        if (nrm == 0) continue;
        const double akk = A[(size_t)k * nc + k], alpha = akk > 0 ? -nrm : nrm;
        const double vk = akk - alpha, vtv = nrm * nrm - akk * akk + vk * vk;
        __syncthreads();                      // everyone has read a_kk before column k's head is touched
        for (int j = k + 1; j < nc; ++j) {
            double dpart = 0;
            for (int r = k + 1 + tid; r < m; r += 256) dpart += A[(size_t)r * nc + k] * A[(size_t)r * nc + j];
            const double dd = block_sum256(dpart, red) + vk * A[(size_t)k * nc + j];
            const double f = 2 * dd / vtv;
            __syncthreads();
            for (int r = k + 1 + tid; r < m; r += 256) A[(size_t)r * nc + j] -= f * A[(size_t)r * nc + k];
            if (tid == 0) A[(size_t)k * nc + j] -= f * vk;
            __syncthreads();
        }
        if (tid == 0) A[(size_t)k * nc + k] = alpha;
        __syncthreads();
    }
    if (tid == 0) {
        for (int k = degree - 1; k >= 0; --k) {
            double v = k < m ? A[(size_t)k * nc + degree] : 0.0;
            for (int j = k + 1; j < degree; ++j) v -= (k < m ? A[(size_t)k * nc + j] : 0.0) * coef[j];
            const double dg = k < m ? A[(size_t)k * nc + k] : 0.0;
            coef[k] = dg != 0 ? v / dg : 0.0;   // rank-deficient column (the reference asserts full rank in DEBUG, :1134)
        }
        for (int k = 0; k < degree; ++k) coef_out[(size_t)b * degree + k] = (float)coef[k];
        xrange[2 * b] = (float)xmin; xrange[2 * b + 1] = (float)xmax;
    }
}

__global__ void __launch_bounds__(256)
cheb_eval_kernel(const float* __restrict__ x, const int32_t* __restrict__ off, int degree, const float* __restrict__ coef,
                 const float* __restrict__ xrange, float* __restrict__ y) {
    const int b = blockIdx.x;
    const int r0 = off[b], m = off[b + 1] - r0;
    const double xmin = xrange[2 * b], xmax = xrange[2 * b + 1];
    const float* c = coef + (size_t)b * degree;
    for (int r = threadIdx.x; r < m; r += 256) {
        const double xn = (2 * (double)x[r0 + r] - (xmax + xmin)) / (xmax - xmin);
        double t0 = 1, t1 = xn, v = c[0];
        if (degree > 1) v += (double)c[1] * t1;
        for (int j = 2; j < degree; ++j) { const double t2 = 2 * xn * t1 - t0; v += (double)c[j] * t2; t0 = t1; t1 = t2; }
        y[r0 + r] = (float)v;
    }
}

static void gl32_host(double* x, double* w) {
    const int N = 32;
    for (int i = 0; i < N; ++i) {
        double z = std::cos(std::acos(-1.0) * (i + 0.75) / (N + 0.5)), pp = 0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1, p2 = 0;
            for (int j = 1; j <= N; ++j) { const double p3 = p2; p2 = p1; p1 = ((2.0 * j - 1) * z * p2 - (j - 1.0) * p3) / j; }
            pp = N * (z * p1 - p2) / (z * z - 1);
            const double z1 = z;
            z = z1 - p1 / pp;
            if (std::fabs(z - z1) < 1e-16) break;
        }
        x[i] = z; w[i] = 2 / ((1 - z * z) * pp * pp);
    }
}

extern "C" int sc_bezier_from_path_batch(sc_ctx* ctx, const float* path, const int32_t* npts, int P, int n_max, float start_angle,
                                         const float* lines, int nlines, float* ctrl) {
    if (!ctx || !path || !npts || !ctrl || P <= 0 || n_max < 2 || nlines < 0 || (nlines > 0 && !lines)) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int r = sc_scratch_reserve(ctx, &ctx->bez_tang, (size_t)P * n_max * 2 * sizeof(double));
    if (r != SC_OK) return r;
    double* tang = (double*)ctx->bez_tang.p;
    int tk = sc_time_begin(ctx, SC_K_BEZIER);
    hipLaunchKernelGGL(bezier_tangent_kernel, dim3((P * n_max + 255) / 256), dim3(256), 0, ctx->stream, path, npts, P, n_max,
                       start_angle, lines, nlines, tang);
    hipLaunchKernelGGL(bezier_ctrl_kernel, dim3((P * (n_max - 1) + 255) / 256), dim3(256), 0, ctx->stream, path, npts, P, n_max,
                       tang, ctrl);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// bezier_spline::shrink_tangent (sea_current.hpp:575-596) on its own: M tangents k * T at waypoints Wp against the edges
__global__ void __launch_bounds__(256)
bezier_shrink_kernel(const float* __restrict__ T, const float* __restrict__ Wp, int M, float k, const float* __restrict__ lines, int nlines,
                     float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    double tx = (double)k * (double)T[2 * i], ty = (double)k * (double)T[2 * i + 1];
    shrink(tx, ty, (double)Wp[2 * i], (double)Wp[2 * i + 1], lines, nlines);
    out[2 * i] = (float)tx; out[2 * i + 1] = (float)ty;
}

extern "C" int sc_bezier_shrink_tangent_batch(sc_ctx* ctx, const float* T, const float* Wp, int M, float k, const float* lines, int nlines,
                                              float* out) {
    if (!ctx || !T || !Wp || !out || M <= 0 || nlines < 0 || (nlines > 0 && !lines)) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int tk = sc_time_begin(ctx, SC_K_BEZIER);
    hipLaunchKernelGGL(bezier_shrink_kernel, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, T, Wp, M, k, lines, nlines, out);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

extern "C" int sc_bezier_eval_batch(sc_ctx* ctx, const float* ctrl, const int32_t* seg, const float* t, int M, int order, float* out) {
    if (!ctx || !ctrl || !seg || !t || !out || M <= 0 || order < 0 || order > 2) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int tk = sc_time_begin(ctx, SC_K_BEZIER);
    hipLaunchKernelGGL(bezier_eval_kernel, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, ctrl, seg, t, M, order, out);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

extern "C" int sc_bezier_curve_batch(sc_ctx* ctx, const float* ctrl, int degree, const int32_t* seg, const float* t, int M, float* out) {
    if (!ctx || !ctrl || !seg || !t || !out || M <= 0 || degree < 1 || degree > SC_BEZIER_MAX_DEGREE) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int tk = sc_time_begin(ctx, SC_K_BEZIER);
    hipLaunchKernelGGL(bezier_curve_kernel, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, ctrl, degree, seg, t, M, out);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

extern "C" int sc_chebfit_batch(sc_ctx* ctx, const float* x, const float* y, const int32_t* off, int B, int total, int degree, float* coef,
                                float* xrange) {
    if (!ctx || !x || !y || !off || !coef || !xrange || B <= 0 || total <= 0 || degree < 1 || degree > SC_CHEB_MAX_DEGREE) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int r = sc_scratch_reserve(ctx, &ctx->cheb_a, (size_t)total * (degree + 1) * sizeof(double));
    if (r != SC_OK) return r;
    int tk = sc_time_begin(ctx, SC_K_RESAMPLE);
    hipLaunchKernelGGL(cheb_fit_kernel, dim3(B), dim3(256), 0, ctx->stream, x, y, off, degree, (double*)ctx->cheb_a.p, coef, xrange);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

extern "C" int sc_chebeval_batch(sc_ctx* ctx, const float* x, const int32_t* off, int B, int degree, const float* coef, const float* xrange,
                                 float* y) {
    if (!ctx || !x || !off || !coef || !xrange || !y || B <= 0 || degree < 1 || degree > SC_CHEB_MAX_DEGREE) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int tk = sc_time_begin(ctx, SC_K_RESAMPLE);
    hipLaunchKernelGGL(cheb_eval_kernel, dim3(B), dim3(256), 0, ctx->stream, x, off, degree, coef, xrange, y);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

extern "C" int sc_bezier_arclength_batch(sc_ctx* ctx, const float* ctrl, int S, int nsub, float* cum, float* seg_len) {
    if (!ctx || !ctrl || !cum || !seg_len || S <= 0 || nsub <= 0 || nsub > 4096) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->bez_gl.p) {
        int r = sc_scratch_reserve(ctx, &ctx->bez_gl, 64 * sizeof(double));
        if (r != SC_OK) return r;
        double h[64];
        gl32_host(h, h + 32);
        SC_HIP(ctx, hipMemcpy(ctx->bez_gl.p, h, sizeof(h), hipMemcpyHostToDevice));
    }
    int tk = sc_time_begin(ctx, SC_K_ARCLENGTH);
    hipLaunchKernelGGL(bezier_arclength_kernel, dim3(S), dim3(64), (size_t)nsub * sizeof(double), ctx->stream, ctrl, S, nsub,
                       (const double*)ctx->bez_gl.p, cum, seg_len);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

extern "C" int sc_bezier_resample_batch(sc_ctx* ctx, const float* ctrl, const float* cum, const float* arclength, const int32_t* seg_off,
                                        int B, int S, int nsub, float* profile_pos, const int32_t* prof_off, int nudge, float* pts,
                                        float* tpar, int32_t* seg, float* curvature, int32_t* status) {
    if (!ctx || !ctrl || !cum || !arclength || !seg_off || !profile_pos || !prof_off || !status || B <= 0 || S <= 0 || nsub <= 0 ||
        nsub > SC_RESAMPLE_MAX_NSUB)
        return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int r = sc_scratch_reserve(ctx, &ctx->bez_seginfo, (size_t)S * sizeof(int4));
    if (r != SC_OK) return r;
    int4* seginfo = (int4*)ctx->bez_seginfo.p;
    const int m = nsub + 1, nc = (m < 10 ? m : 10) + 1;
    int tk = sc_time_begin(ctx, SC_K_RESAMPLE);
    hipLaunchKernelGGL(resample_prepare_kernel, dim3(B), dim3(64), (size_t)RESAMPLE_STAGE_MAX * sizeof(float), ctx->stream, cum, seg_off, nsub, arclength, profile_pos, prof_off,
                       nudge, seginfo, status);
    // tables of 10 .. 256 rows: the fit in registers, one wavefront per segment; others: the LDS form, one workgroup per segment
#define SC_RESAMPLE_REG(RPL)                                                                                                         \
    hipLaunchKernelGGL(resample_eval_reg_kernel<RPL>, dim3((S + 3) / 4), dim3(256), 0, ctx->stream, ctrl, cum, nsub, S,               \
                       (const float*)profile_pos, prof_off, (const int4*)seginfo, pts, tpar, seg, curvature)
    static const bool lds_form = getenv("SC_RESAMPLE_LDS") != nullptr;   // A/B switch: the LDS form for every table size
    const bool reg = !lds_form && m >= 10;
    if (reg && m <= 64) SC_RESAMPLE_REG(1);
    else if (reg && m <= 128) SC_RESAMPLE_REG(2);
    else if (reg && m <= 192) SC_RESAMPLE_REG(3);
    else if (reg && m <= 256) SC_RESAMPLE_REG(4);
    else
        hipLaunchKernelGGL(resample_eval_kernel, dim3(S), dim3(256), (size_t)m * nc * sizeof(double), ctx->stream, ctrl, cum, nsub,
                           (const float*)profile_pos, prof_off, (const int4*)seginfo, pts, tpar, seg, curvature);
#undef SC_RESAMPLE_REG
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}
