// ctx.hip -- context, stream, scratch, launch timing and the `_host` convenience wrappers.
#include "sc_internal.h"
#include <stdlib.h>

extern "C" int sc_abi_version(void) { return SC_ABI_VERSION; }

extern "C" const char* sc_status_string(int s) {
    switch (s) {
        case SC_OK: return "ok";
        case SC_ERR_INVALID: return "invalid argument";
        case SC_ERR_HIP: return "HIP runtime error";
        case SC_ERR_NOMEM: return "out of device memory";
        case SC_ERR_NO_DEVICE: return "no usable device";
        default: return "unknown status";
    }
}

extern "C" const char* sc_last_error(const sc_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int sc_ctx_create(int device, sc_ctx** out) {
    if (!out) return SC_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return SC_ERR_NO_DEVICE;
    sc_ctx* c = new sc_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return SC_ERR_HIP;
    }
    c->stream = c->own_stream;
    // scratch the A* slots of this context may take, in GiB (a slot = one resident wavefront's g array, closed bitmap and rings)
    if (const char* e = getenv("SC_ASTAR_SLOT_GB")) { const long v = atol(e); if (v >= 1 && v <= 256) c->astar_slot_budget = (size_t)v << 30; }
    *out = c;
    return SC_OK;
}

static void free_scratch(sc_scratch* s) {
    if (s->p) (void)hipFree(s->p);
    s->p = nullptr;
    s->bytes = 0;
}

extern "C" int sc_ctx_destroy(sc_ctx* ctx) {
    if (!ctx) return SC_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)sc_comm_destroy(ctx);
    for (auto& p : ctx->pending) { if (!p.shared_a) (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->wait_ev) (void)hipEventDestroy(ctx->wait_ev);
    free_scratch(&ctx->colbits); free_scratch(&ctx->updown); free_scratch(&ctx->edt_fault); free_scratch(&ctx->fmt_nbr); free_scratch(&ctx->edt_flags); free_scratch(&ctx->moves); free_scratch(&ctx->gslots);
    free_scratch(&ctx->buckets); free_scratch(&ctx->qstats); free_scratch(&ctx->closed); free_scratch(&ctx->actr); free_scratch(&ctx->bez_tang); free_scratch(&ctx->bez_gl); free_scratch(&ctx->bez_seginfo); free_scratch(&ctx->cheb_a); free_scratch(&ctx->gather_msg);
    for (auto& s : ctx->staging) free_scratch(&s);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return SC_OK;
}

extern "C" int sc_ctx_set_stream(sc_ctx* ctx, void* s) {
    if (!ctx) return SC_ERR_INVALID;
    ctx->stream = (hipStream_t)s;
    return SC_OK;
}

extern "C" int sc_ctx_use_own_stream(sc_ctx* ctx) {
    if (!ctx) return SC_ERR_INVALID;
    ctx->stream = ctx->own_stream;
    return SC_OK;
}

int sc_allow_big_lds(sc_ctx* ctx, const void* kernel, int bytes) {
    if (ctx->big_lds_done.count(kernel)) return SC_OK;
    SC_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    ctx->big_lds_done.insert(kernel);
    return SC_OK;
}

int sc_stream_wait(sc_ctx* ctx) {
    if (!ctx->wait_ev) SC_HIP(ctx, hipEventCreateWithFlags(&ctx->wait_ev, hipEventBlockingSync | hipEventDisableTiming));
    SC_HIP(ctx, hipEventRecord(ctx->wait_ev, ctx->stream));
    SC_HIP(ctx, hipEventSynchronize(ctx->wait_ev));
    return SC_OK;
}

extern "C" int sc_ctx_synchronize(sc_ctx* ctx) {
    if (!ctx) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int r = sc_stream_wait(ctx);
    if (r != SC_OK) return r;
    // A* bucket rings overflowed in some launch since the last synchronisation (the overflowed queries were rerun
    // with 16x the space on the device): start later launches with larger rings
    if (ctx->actr.p) {
        int32_t sticky = 0;
        SC_HIP(ctx, hipMemcpy(&sticky, (const int32_t*)ctx->actr.p + 4, sizeof(sticky), hipMemcpyDeviceToHost));
        if (sticky) {
            SC_HIP(ctx, hipMemset((int32_t*)ctx->actr.p + 4, 0, sizeof(int32_t)));
            if (ctx->astar_cap < (1 << 22)) ctx->astar_cap *= 4;
        }
    }
    r = sc_edt_open_mode_update(ctx);
    if (r != SC_OK) return r;
    // the wide-row EDT kernel's wavefronts wait for one another with bounded spins: one that ran out left rows unwritten
    if (ctx->edt_fault.p) {
        int32_t fault = 0;
        SC_HIP(ctx, hipMemcpy(&fault, ctx->edt_fault.p, sizeof(fault), hipMemcpyDeviceToHost));
        if (fault) {
            SC_HIP(ctx, hipMemset(ctx->edt_fault.p, 0, sizeof(int32_t)));
            snprintf(ctx->err, sizeof(ctx->err), "edt_band_wide_kernel: a wait between the wavefronts of a workgroup ran out; the distances of that call are incomplete");
            return SC_ERR_HIP;
        }
    }
    return SC_OK;
}

int sc_scratch_reserve(sc_ctx* ctx, sc_scratch* s, size_t bytes) {
    if (bytes <= s->bytes) return SC_OK;
    if (s->p) {
        SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SC_HIP(ctx, hipFree(s->p));
        s->p = nullptr;
        s->bytes = 0;
    }
    SC_HIP(ctx, hipMalloc(&s->p, bytes));
    s->bytes = bytes;
    return SC_OK;
}

extern "C" int sc_ctx_scratch_bytes(sc_ctx* ctx, int64_t* bytes) {
    if (!ctx || !bytes) return SC_ERR_INVALID;
    size_t b = ctx->colbits.bytes + ctx->updown.bytes + ctx->edt_fault.bytes + ctx->edt_flags.bytes + ctx->moves.bytes + ctx->gslots.bytes + ctx->closed.bytes + ctx->buckets.bytes +
               ctx->qstats.bytes + ctx->actr.bytes + ctx->bez_tang.bytes + ctx->bez_gl.bytes + ctx->bez_seginfo.bytes + ctx->cheb_a.bytes + ctx->gather_msg.bytes + ctx->fmt_nbr.bytes;
    for (auto& s : ctx->staging) b += s.bytes;
    *bytes = (int64_t)b;
    return SC_OK;
}

// ---- timing ---------------------------------------------------------------
static hipEvent_t get_event(sc_ctx* ctx) {
    if (!ctx->ev_pool.empty()) {
        hipEvent_t e = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    return e;
}

int sc_time_begin(sc_ctx* ctx, int kid) {
    if (!ctx->timing) return -1;
    sc_ctx::pending_ev p{kid, get_event(ctx), get_event(ctx)};
    if (!p.a || !p.b) {   // no event to be had: this launch goes untimed
        if (p.a) ctx->ev_pool.push_back(p.a);
        if (p.b) ctx->ev_pool.push_back(p.b);
        return -1;
    }
    (void)hipEventRecord(p.a, ctx->stream);
    ctx->pending.push_back(p);
    return (int)ctx->pending.size() - 1;
}

void sc_time_end(sc_ctx* ctx, int token) {
    if (token < 0) return;
    (void)hipEventRecord(ctx->pending[token].b, ctx->stream);
}

// end of bracket `token` and begin of a bracket for `kid` with ONE event record (back-to-back kernels)
int sc_time_chain(sc_ctx* ctx, int token, int kid) {
    if (token < 0) return -1;
    (void)hipEventRecord(ctx->pending[token].b, ctx->stream);
    sc_ctx::pending_ev p{kid, ctx->pending[token].b, get_event(ctx)};
    if (!p.b) return -1;
    p.shared_a = true;
    ctx->pending.push_back(p);
    return (int)ctx->pending.size() - 1;
}

static void drain_timing(sc_ctx* ctx) {
    for (auto& p : ctx->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->t_ms[p.kid] += ms;
            ctx->t_n[p.kid] += 1;
        }
        if (!p.shared_a) ctx->ev_pool.push_back(p.a);
        ctx->ev_pool.push_back(p.b);
    }
    ctx->pending.clear();
}

extern "C" int sc_ctx_set_timing(sc_ctx* ctx, int enable) {
    if (!ctx) return SC_ERR_INVALID;
    drain_timing(ctx);
    ctx->timing = enable ? 1 : 0;
    return SC_OK;
}

extern "C" int sc_ctx_reset_timing(sc_ctx* ctx) {
    if (!ctx) return SC_ERR_INVALID;
    drain_timing(ctx);
    for (int i = 0; i < SC_K_COUNT; ++i) { ctx->t_ms[i] = 0; ctx->t_n[i] = 0; }
    return SC_OK;
}

extern "C" int sc_ctx_get_timing(sc_ctx* ctx, int kid, double* total_ms, int64_t* launches) {
    if (!ctx || kid < 0 || kid >= SC_K_COUNT) return SC_ERR_INVALID;
    drain_timing(ctx);
    if (total_ms) *total_ms = ctx->t_ms[kid];
    if (launches) *launches = ctx->t_n[kid];
    return SC_OK;
}

// ---- host wrappers ---------------------------------------------------------
#define STAGE(i, bytes)                                                        \
    do {                                                                       \
        int r_ = sc_scratch_reserve(ctx, &ctx->staging[i], (bytes));           \
        if (r_ != SC_OK) return r_;                                            \
    } while (0)
#define H2D(i, src, bytes) SC_HIP(ctx, hipMemcpyAsync(ctx->staging[i].p, (src), (bytes), hipMemcpyHostToDevice, ctx->stream))
#define D2H(dst, i, bytes) SC_HIP(ctx, hipMemcpyAsync((dst), ctx->staging[i].p, (bytes), hipMemcpyDeviceToHost, ctx->stream))

extern "C" int sc_edt_u8_i32_host(sc_ctx* ctx, const uint8_t* occ, int W, int H, int batch, int32_t* d2) {
    if (!ctx || !occ || !d2 || W <= 0 || H <= 0 || batch <= 0) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    size_t n = (size_t)W * H * batch;
    STAGE(0, n);
    STAGE(1, n * 4);
    H2D(0, occ, n);
    int r = sc_edt_u8_i32(ctx, (const uint8_t*)ctx->staging[0].p, W, H, batch, (int32_t*)ctx->staging[1].p);
    if (r != SC_OK) return r;
    D2H(d2, 1, n * 4);
    return sc_ctx_synchronize(ctx);
}

extern "C" int sc_astar_batch_host(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2,
                                   const int32_t* start, const int32_t* goal, int Q, int Lmax,
                                   int32_t* path, int32_t* len, int32_t* cost, int32_t* status) {
    if (!ctx || !d2 || !start || !goal || !path || !len || !cost || !status || W <= 0 || H <= 0 || Q < 0 || Lmax <= 0)
        return SC_ERR_INVALID;
    if (Q == 0) return SC_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    size_t n = (size_t)W * H;
    STAGE(0, n * 4); STAGE(1, (size_t)Q * 4); STAGE(2, (size_t)Q * 4);
    STAGE(3, (size_t)Q * Lmax * 4); STAGE(4, (size_t)Q * 4); STAGE(5, (size_t)Q * 4); STAGE(6, (size_t)Q * 4);
    H2D(0, d2, n * 4); H2D(1, start, (size_t)Q * 4); H2D(2, goal, (size_t)Q * 4);
    int r = sc_astar_batch(ctx, (const int32_t*)ctx->staging[0].p, W, H, r2, (const int32_t*)ctx->staging[1].p,
                           (const int32_t*)ctx->staging[2].p, Q, Lmax, (int32_t*)ctx->staging[3].p,
                           (int32_t*)ctx->staging[4].p, (int32_t*)ctx->staging[5].p, (int32_t*)ctx->staging[6].p);
    if (r != SC_OK) return r;
    D2H(path, 3, (size_t)Q * Lmax * 4); D2H(len, 4, (size_t)Q * 4); D2H(cost, 5, (size_t)Q * 4); D2H(status, 6, (size_t)Q * 4);
    return sc_ctx_synchronize(ctx);
}

extern "C" int sc_toppra_hermite_batch_host(sc_ctx* ctx, int P, int dof, int N,
                                            const double* p0, const double* p1, const double* v0, const double* v1,
                                            const double* vlim_lo, const double* vlim_hi, int vlim_per_stage,
                                            const double* alim_lo, const double* alim_hi,
                                            double sd_start, double sd_end,
                                            double* K, double* x, double* u, double* t, int32_t* status) {
    if (!ctx || P <= 0 || dof <= 0 || N <= 0 || !p0 || !p1 || !v0 || !v1 || !vlim_lo || !vlim_hi || !alim_lo ||
        !alim_hi || !K || !x || !u || !t || !status)
        return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    size_t pd = (size_t)P * dof * 8, vl = vlim_per_stage ? pd * (N + 1) : pd;
    // one staging block: inputs then outputs
    size_t off_in[8] = {0, pd, 2 * pd, 3 * pd, 4 * pd, 4 * pd + vl, 4 * pd + 2 * vl, 5 * pd + 2 * vl};
    size_t in_bytes = 6 * pd + 2 * vl;
    size_t oK = in_bytes, ox = oK + (size_t)P * (N + 1) * 16, ou = ox + (size_t)P * (N + 1) * 8,
           ot = ou + (size_t)P * N * 8, os = ot + (size_t)P * (N + 1) * 8, total = os + (size_t)P * 4;
    STAGE(0, total);
    char* b = (char*)ctx->staging[0].p;
    const void* src[8] = {p0, p1, v0, v1, vlim_lo, vlim_hi, alim_lo, alim_hi};
    size_t sz[8] = {pd, pd, pd, pd, vl, vl, pd, pd};
    for (int i = 0; i < 8; ++i) SC_HIP(ctx, hipMemcpyAsync(b + off_in[i], src[i], sz[i], hipMemcpyHostToDevice, ctx->stream));
    int r = sc_toppra_hermite_batch(ctx, P, dof, N, (double*)(b + off_in[0]), (double*)(b + off_in[1]),
                                    (double*)(b + off_in[2]), (double*)(b + off_in[3]), (double*)(b + off_in[4]),
                                    (double*)(b + off_in[5]), vlim_per_stage, (double*)(b + off_in[6]),
                                    (double*)(b + off_in[7]), sd_start, sd_end, (double*)(b + oK), (double*)(b + ox),
                                    (double*)(b + ou), (double*)(b + ot), (int32_t*)(b + os));
    if (r != SC_OK) return r;
    SC_HIP(ctx, hipMemcpyAsync(K, b + oK, (size_t)P * (N + 1) * 16, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(x, b + ox, (size_t)P * (N + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(u, b + ou, (size_t)P * N * 8, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(t, b + ot, (size_t)P * (N + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(status, b + os, (size_t)P * 4, hipMemcpyDeviceToHost, ctx->stream));
    return sc_ctx_synchronize(ctx);
}

extern "C" int sc_toppra_sample_batch_host(sc_ctx* ctx, int P, int dof, int N,
                                           const double* p0, const double* p1, const double* v0, const double* v1,
                                           const double* x, const double* t, double dt, int max_len,
                                           float* pos, float* vel, float* acc, double* times, int32_t* length) {
    if (!ctx || P <= 0 || dof <= 0 || N <= 0 || max_len <= 0 || !p0 || !p1 || !v0 || !v1 || !x || !t || !pos ||
        !vel || !acc || !times || !length)
        return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    size_t pd = (size_t)P * dof * 8, xs = (size_t)P * (N + 1) * 8;
    size_t o[6] = {0, pd, 2 * pd, 3 * pd, 4 * pd, 4 * pd + xs};
    size_t in_bytes = 4 * pd + 2 * xs;
    size_t fb = (size_t)P * dof * max_len * 4;
    size_t opos = in_bytes, ovel = opos + fb, oacc = ovel + fb, otim = oacc + fb, olen = otim + (size_t)P * max_len * 8,
           total = olen + (size_t)P * 4;
    STAGE(1, total);
    char* b = (char*)ctx->staging[1].p;
    const void* src[6] = {p0, p1, v0, v1, x, t};
    size_t sz[6] = {pd, pd, pd, pd, xs, xs};
    for (int i = 0; i < 6; ++i) SC_HIP(ctx, hipMemcpyAsync(b + o[i], src[i], sz[i], hipMemcpyHostToDevice, ctx->stream));
    int r = sc_toppra_sample_batch(ctx, P, dof, N, (double*)(b + o[0]), (double*)(b + o[1]), (double*)(b + o[2]),
                                   (double*)(b + o[3]), (double*)(b + o[4]), (double*)(b + o[5]), dt, max_len,
                                   (float*)(b + opos), (float*)(b + ovel), (float*)(b + oacc), (double*)(b + otim),
                                   (int32_t*)(b + olen));
    if (r != SC_OK) return r;
    SC_HIP(ctx, hipMemcpyAsync(pos, b + opos, fb, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(vel, b + ovel, fb, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(acc, b + oacc, fb, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(times, b + otim, (size_t)P * max_len * 8, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(length, b + olen, (size_t)P * 4, hipMemcpyDeviceToHost, ctx->stream));
    return sc_ctx_synchronize(ctx);
}

extern "C" int sc_bezier_from_path_batch_host(sc_ctx* ctx, const float* path, const int32_t* npts, int P, int n_max, float start_angle,
                                              const float* lines, int nlines, float* ctrl) {
    if (!ctx || !path || !npts || !ctrl || P <= 0 || n_max < 2 || nlines < 0 || (nlines > 0 && !lines)) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t pb = (size_t)P * n_max * 2 * 4, nb = (size_t)P * 4, lb = (size_t)nlines * 16, cb = (size_t)P * (n_max - 1) * 32;
    STAGE(2, pb + nb + lb + cb + 64);
    char* b = (char*)ctx->staging[2].p;
    SC_HIP(ctx, hipMemcpyAsync(b, path, pb, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + pb, npts, nb, hipMemcpyHostToDevice, ctx->stream));
    if (lb) SC_HIP(ctx, hipMemcpyAsync(b + pb + nb, lines, lb, hipMemcpyHostToDevice, ctx->stream));
    int r = sc_bezier_from_path_batch(ctx, (const float*)b, (const int32_t*)(b + pb), P, n_max, start_angle,
                                      lb ? (const float*)(b + pb + nb) : nullptr, nlines, (float*)(b + pb + nb + lb));
    if (r != SC_OK) return r;
    SC_HIP(ctx, hipMemcpyAsync(ctrl, b + pb + nb + lb, cb, hipMemcpyDeviceToHost, ctx->stream));
    return sc_ctx_synchronize(ctx);
}

extern "C" int sc_bezier_arclength_batch_host(sc_ctx* ctx, const float* ctrl, int S, int nsub, float* cum, float* seg_len) {
    if (!ctx || !ctrl || !cum || !seg_len || S <= 0 || nsub <= 0) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cb = (size_t)S * 32, ub = (size_t)S * (nsub + 1) * 4, sb = (size_t)S * 4;
    STAGE(3, cb + ub + sb);
    char* b = (char*)ctx->staging[3].p;
    SC_HIP(ctx, hipMemcpyAsync(b, ctrl, cb, hipMemcpyHostToDevice, ctx->stream));
    int r = sc_bezier_arclength_batch(ctx, (const float*)b, S, nsub, (float*)(b + cb), (float*)(b + cb + ub));
    if (r != SC_OK) return r;
    SC_HIP(ctx, hipMemcpyAsync(cum, b + cb, ub, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(seg_len, b + cb + ub, sb, hipMemcpyDeviceToHost, ctx->stream));
    return sc_ctx_synchronize(ctx);
}

extern "C" int sc_bezier_resample_batch_host(sc_ctx* ctx, const float* ctrl, const float* cum, const float* arclength,
                                             const int32_t* seg_off, int B, int S, int nsub, float* profile_pos, const int32_t* prof_off,
                                             int nudge, float* pts, float* tpar, int32_t* seg, float* curvature, int32_t* status) {
    if (!ctx || !ctrl || !cum || !arclength || !seg_off || !profile_pos || !prof_off || !status || B <= 0 || S <= 0 || nsub <= 0)
        return SC_ERR_INVALID;
    if (seg_off[B] != S || seg_off[0] != 0 || prof_off[0] != 0 || prof_off[B] < 0) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t M = (size_t)prof_off[B];
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_ctrl = 0, o_cum = o_ctrl + al((size_t)S * 32), o_al = o_cum + al((size_t)S * (nsub + 1) * 4), o_so = o_al + al((size_t)B * 4),
                 o_po = o_so + al((size_t)(B + 1) * 4), o_pp = o_po + al((size_t)(B + 1) * 4), o_pts = o_pp + al(M * 4), o_t = o_pts + al(M * 8),
                 o_sg = o_t + al(M * 4), o_cv = o_sg + al(M * 4), o_st = o_cv + al(M * 4), total = o_st + al((size_t)B * 4);
    STAGE(4, total);
    char* b = (char*)ctx->staging[4].p;
    SC_HIP(ctx, hipMemcpyAsync(b + o_ctrl, ctrl, (size_t)S * 32, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + o_cum, cum, (size_t)S * (nsub + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + o_al, arclength, (size_t)B * 4, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + o_so, seg_off, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + o_po, prof_off, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    if (M) SC_HIP(ctx, hipMemcpyAsync(b + o_pp, profile_pos, M * 4, hipMemcpyHostToDevice, ctx->stream));
    int r = sc_bezier_resample_batch(ctx, (const float*)(b + o_ctrl), (const float*)(b + o_cum), (const float*)(b + o_al),
                                     (const int32_t*)(b + o_so), B, S, nsub, (float*)(b + o_pp), (const int32_t*)(b + o_po), nudge,
                                     pts ? (float*)(b + o_pts) : nullptr, tpar ? (float*)(b + o_t) : nullptr,
                                     seg ? (int32_t*)(b + o_sg) : nullptr, curvature ? (float*)(b + o_cv) : nullptr, (int32_t*)(b + o_st));
    if (r != SC_OK) return r;
    if (M) {
        SC_HIP(ctx, hipMemcpyAsync(profile_pos, b + o_pp, M * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (pts) SC_HIP(ctx, hipMemcpyAsync(pts, b + o_pts, M * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (tpar) SC_HIP(ctx, hipMemcpyAsync(tpar, b + o_t, M * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (seg) SC_HIP(ctx, hipMemcpyAsync(seg, b + o_sg, M * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (curvature) SC_HIP(ctx, hipMemcpyAsync(curvature, b + o_cv, M * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    SC_HIP(ctx, hipMemcpyAsync(status, b + o_st, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
    return sc_ctx_synchronize(ctx);
}

extern "C" int sc_bezier_eval_batch_host(sc_ctx* ctx, const float* ctrl, int S, const int32_t* seg, const float* t, int M, int order,
                                         float* out) {
    if (!ctx || !ctrl || !seg || !t || !out || S <= 0 || M <= 0) return SC_ERR_INVALID;
    for (int i = 0; i < M; ++i)
        if (seg[i] < 0 || seg[i] >= S) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cb = ((size_t)S * 32 + 255) & ~(size_t)255, mb = ((size_t)M * 4 + 255) & ~(size_t)255;
    STAGE(5, cb + 2 * mb + (size_t)M * 8);
    char* b = (char*)ctx->staging[5].p;
    SC_HIP(ctx, hipMemcpyAsync(b, ctrl, (size_t)S * 32, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + cb, seg, (size_t)M * 4, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + cb + mb, t, (size_t)M * 4, hipMemcpyHostToDevice, ctx->stream));
    int r = sc_bezier_eval_batch(ctx, (const float*)b, (const int32_t*)(b + cb), (const float*)(b + cb + mb), M, order,
                                 (float*)(b + cb + 2 * mb));
    if (r != SC_OK) return r;
    SC_HIP(ctx, hipMemcpyAsync(out, b + cb + 2 * mb, (size_t)M * 8, hipMemcpyDeviceToHost, ctx->stream));
    return sc_ctx_synchronize(ctx);
}

extern "C" int sc_bezier_shrink_tangent_batch_host(sc_ctx* ctx, const float* T, const float* Wp, int M, float k, const float* lines, int nlines,
                                                   float* out) {
    if (!ctx || !T || !Wp || !out || M <= 0 || nlines < 0 || (nlines > 0 && !lines)) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t mb = ((size_t)M * 8 + 255) & ~(size_t)255, lb = ((size_t)nlines * 16 + 255) & ~(size_t)255;
    STAGE(5, 3 * mb + lb);
    char* b = (char*)ctx->staging[5].p;
    SC_HIP(ctx, hipMemcpyAsync(b, T, (size_t)M * 8, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + mb, Wp, (size_t)M * 8, hipMemcpyHostToDevice, ctx->stream));
    if (nlines) SC_HIP(ctx, hipMemcpyAsync(b + 2 * mb, lines, (size_t)nlines * 16, hipMemcpyHostToDevice, ctx->stream));
    int r = sc_bezier_shrink_tangent_batch(ctx, (const float*)b, (const float*)(b + mb), M, k, nlines ? (const float*)(b + 2 * mb) : nullptr, nlines,
                                           (float*)(b + 2 * mb + lb));
    if (r != SC_OK) return r;
    SC_HIP(ctx, hipMemcpyAsync(out, b + 2 * mb + lb, (size_t)M * 8, hipMemcpyDeviceToHost, ctx->stream));
    return sc_ctx_synchronize(ctx);
}

extern "C" int sc_fmt_star_batch_host(sc_ctx* ctx, const float* samples, int n, const float* starts, const float* goals, int Q, float rn,
                                      const float* lines, int E, int Lmax, float* path, int32_t* len, float* cost, int32_t* status) {
    if (!ctx || !samples || !starts || !goals || !path || !len || !cost || !status || n < 0 || Q < 0 || E < 0 || Lmax <= 0) return SC_ERR_INVALID;
    if (Q == 0) return SC_OK;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_s = 0, o_st = o_s + al((size_t)(n > 0 ? n : 1) * 8), o_g = o_st + al((size_t)Q * 8), o_l = o_g + al((size_t)Q * 8),
                 o_p = o_l + al((size_t)(E > 0 ? E : 1) * 16), o_len = o_p + al((size_t)Q * Lmax * 8), o_c = o_len + al((size_t)Q * 4),
                 o_status = o_c + al((size_t)Q * 4), total = o_status + al((size_t)Q * 4);
    STAGE(6, total);
    char* b = (char*)ctx->staging[6].p;
    if (n) SC_HIP(ctx, hipMemcpyAsync(b + o_s, samples, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + o_st, starts, (size_t)Q * 8, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + o_g, goals, (size_t)Q * 8, hipMemcpyHostToDevice, ctx->stream));
    if (E) SC_HIP(ctx, hipMemcpyAsync(b + o_l, lines, (size_t)E * 16, hipMemcpyHostToDevice, ctx->stream));
    int r = sc_fmt_star_batch(ctx, (const float*)(b + o_s), n, (const float*)(b + o_st), (const float*)(b + o_g), Q, rn, (const float*)(b + o_l), E,
                              Lmax, (float*)(b + o_p), (int32_t*)(b + o_len), (float*)(b + o_c), (int32_t*)(b + o_status));
    if (r != SC_OK) return r;
    SC_HIP(ctx, hipMemcpyAsync(path, b + o_p, (size_t)Q * Lmax * 8, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(len, b + o_len, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(cost, b + o_c, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(status, b + o_status, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream));
    return sc_ctx_synchronize(ctx);
}

extern "C" int sc_bezier_curve_batch_host(sc_ctx* ctx, const float* ctrl, int S, int degree, const int32_t* seg, const float* t, int M, float* out) {
    if (!ctx || !ctrl || !seg || !t || !out || S <= 0 || M <= 0 || degree < 1 || degree > SC_BEZIER_MAX_DEGREE) return SC_ERR_INVALID;
    for (int i = 0; i < M; ++i)
        if (seg[i] < 0 || seg[i] >= S) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cbytes = (size_t)S * (degree + 1) * 8;
    const size_t cb = (cbytes + 255) & ~(size_t)255, mb = ((size_t)M * 4 + 255) & ~(size_t)255;
    STAGE(5, cb + 2 * mb + (size_t)M * 8);
    char* b = (char*)ctx->staging[5].p;
    SC_HIP(ctx, hipMemcpyAsync(b, ctrl, cbytes, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + cb, seg, (size_t)M * 4, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + cb + mb, t, (size_t)M * 4, hipMemcpyHostToDevice, ctx->stream));
    int r = sc_bezier_curve_batch(ctx, (const float*)b, degree, (const int32_t*)(b + cb), (const float*)(b + cb + mb), M, (float*)(b + cb + 2 * mb));
    if (r != SC_OK) return r;
    SC_HIP(ctx, hipMemcpyAsync(out, b + cb + 2 * mb, (size_t)M * 8, hipMemcpyDeviceToHost, ctx->stream));
    return sc_ctx_synchronize(ctx);
}

extern "C" int sc_chebfit_batch_host(sc_ctx* ctx, const float* x, const float* y, const int32_t* off, int B, int degree, float* coef, float* xrange) {
    if (!ctx || !x || !y || !off || !coef || !xrange || B <= 0 || degree < 1 || degree > SC_CHEB_MAX_DEGREE || off[0] != 0 || off[B] <= 0) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)off[B];
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_x = 0, o_y = al(n * 4), o_off = o_y + al(n * 4), o_c = o_off + al((size_t)(B + 1) * 4), o_r = o_c + al((size_t)B * degree * 4),
                 total = o_r + al((size_t)B * 8);
    STAGE(6, total);
    char* b = (char*)ctx->staging[6].p;
    SC_HIP(ctx, hipMemcpyAsync(b + o_x, x, n * 4, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + o_y, y, n * 4, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + o_off, off, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    int r = sc_chebfit_batch(ctx, (const float*)(b + o_x), (const float*)(b + o_y), (const int32_t*)(b + o_off), B, (int)n, degree, (float*)(b + o_c),
                             (float*)(b + o_r));
    if (r != SC_OK) return r;
    SC_HIP(ctx, hipMemcpyAsync(coef, b + o_c, (size_t)B * degree * 4, hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(xrange, b + o_r, (size_t)B * 8, hipMemcpyDeviceToHost, ctx->stream));
    return sc_ctx_synchronize(ctx);
}

extern "C" int sc_chebeval_batch_host(sc_ctx* ctx, const float* x, const int32_t* off, int B, int degree, const float* coef, const float* xrange,
                                      float* y) {
    if (!ctx || !x || !off || !coef || !xrange || !y || B <= 0 || degree < 1 || degree > SC_CHEB_MAX_DEGREE || off[0] != 0 || off[B] <= 0) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)off[B];
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_x = 0, o_y = al(n * 4), o_off = o_y + al(n * 4), o_c = o_off + al((size_t)(B + 1) * 4), o_r = o_c + al((size_t)B * degree * 4),
                 total = o_r + al((size_t)B * 8);
    STAGE(6, total);
    char* b = (char*)ctx->staging[6].p;
    SC_HIP(ctx, hipMemcpyAsync(b + o_x, x, n * 4, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + o_off, off, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + o_c, coef, (size_t)B * degree * 4, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(b + o_r, xrange, (size_t)B * 8, hipMemcpyHostToDevice, ctx->stream));
    int r = sc_chebeval_batch(ctx, (const float*)(b + o_x), (const int32_t*)(b + o_off), B, degree, (const float*)(b + o_c), (const float*)(b + o_r),
                              (float*)(b + o_y));
    if (r != SC_OK) return r;
    SC_HIP(ctx, hipMemcpyAsync(y, b + o_y, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    return sc_ctx_synchronize(ctx);
}
