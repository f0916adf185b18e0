// grid.hip -- occupancy-grid upkeep for the dynamic-obstacle replan loop (BASELINE.json configs[4]).
//
// The reference keeps obstacles as polygons (planning_space::obstacles, sea_current.hpp:314) and answers every
// collision question from them directly (:201-251, :1274-1326); a moving obstacle is just an edited polygon.  Here a
// frame's obstacle set arrives as a short list of cell rectangles (the per-frame delta the ranks exchange is this list,
// KBs -- SURVEY.md 8e) and is painted over a static base layer.  The EDT is then recomputed in full: one 1024^2 EDT is
// two launches of ~10 us, below the cost of any bookkeeping an incremental brushfire would need, and it is exact.
#include "sc_internal.h"

// one workgroup per rectangle; rect = (x0, y0, x1, y1), x1/y1 exclusive, clipped here
__global__ void __launch_bounds__(256)
occ_paint_rects_kernel(const int32_t* __restrict__ rects, int W, int H, int free_border, uint8_t* __restrict__ occ) {
    const int32_t* r = rects + 4 * (size_t)blockIdx.x;
    const int lo = free_border ? 1 : 0;
    const int x0 = max(r[0], lo), y0 = max(r[1], lo), x1 = min(r[2], W - lo), y1 = min(r[3], H - lo);
    const int w = x1 - x0, h = y1 - y0;
    if (w <= 0 || h <= 0) return;
    for (int i = threadIdx.x; i < w * h; i += 256) occ[(size_t)(y0 + i / w) * W + x0 + i % w] = 1;
}

__global__ void __launch_bounds__(256)
occ_clear_border_kernel(int W, int H, uint8_t* __restrict__ occ) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < W) { occ[i] = 0; occ[(size_t)(H - 1) * W + i] = 0; }
    if (i < H) { occ[(size_t)i * W] = 0; occ[(size_t)i * W + W - 1] = 0; }
}

extern "C" int sc_occ_from_rects(sc_ctx* ctx, const uint8_t* base, const int32_t* rects, int R, int W, int H, int free_border,
                                 uint8_t* occ) {
    if (!ctx || !occ || R < 0 || (R > 0 && !rects) || W <= 0 || H <= 0 || W > SC_MAX_DIM || H > SC_MAX_DIM) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)W * H;
    int tk = sc_time_begin(ctx, SC_K_OCC);
    if (base) {
        if (base != occ) SC_HIP(ctx, hipMemcpyAsync(occ, base, cells, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        SC_HIP(ctx, hipMemsetAsync(occ, 0, cells, ctx->stream));
    }
    if (R > 0) hipLaunchKernelGGL(occ_paint_rects_kernel, dim3(R), dim3(256), 0, ctx->stream, rects, W, H, free_border, occ);
    if (free_border && base)
        hipLaunchKernelGGL(occ_clear_border_kernel, dim3((max(W, H) + 255) / 256), dim3(256), 0, ctx->stream, W, H, occ);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// ---- nearest obstacle cell (the optional second output of the EDT) --------------------------------
// The nearest occupied cell of c lies on the circle dx^2 + dy^2 = d2[c]: one thread per cell tests the integer points
// of that circle (O(sqrt d2) candidates, occupancy bytes from L2) and keeps the smallest linear index.  Exact; 4 more
// bytes written per cell (9 B/cell with d2).  Not part of the planning hot path -- the planner only needs d2.
__global__ void __launch_bounds__(256)
edt_nearest_kernel(const uint8_t* __restrict__ occ, const int32_t* __restrict__ d2, int W, int H, int32_t* __restrict__ nearest) {
    const size_t cells = (size_t)W * H;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cells) return;
    const size_t gi = (size_t)blockIdx.y * cells + i;
    const uint8_t* o = occ + (size_t)blockIdx.y * cells;
    const int x = (int)(i % W), y = (int)(i / W);
    const int32_t D = d2[gi];
    int32_t arg = -1;
    if (D == 0) arg = (int32_t)i;
    else if (D != INT32_MAX) {
        int best = INT32_MAX;
        for (int dx = 0; dx * dx <= D; ++dx) {
            const int rem = D - dx * dx;
            int dy = (int)sqrtf((float)rem);
            while (dy * dy > rem) --dy;
            while ((dy + 1) * (dy + 1) <= rem) ++dy;
            if (dy * dy != rem) continue;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int xx = x + ((s & 1) ? dx : -dx), yy = y + ((s & 2) ? dy : -dy);
                if (xx >= 0 && yy >= 0 && xx < W && yy < H && o[(size_t)yy * W + xx]) best = min(best, yy * W + xx);
            }
        }
        arg = best;
    }
    nearest[gi] = arg;
}

extern "C" int sc_edt_nearest_i32(sc_ctx* ctx, const uint8_t* occ, const int32_t* d2, int W, int H, int batch, int32_t* nearest) {
    if (!ctx || !occ || !d2 || !nearest || W <= 0 || H <= 0 || batch <= 0 || batch > 65535 || W > SC_MAX_DIM || H > SC_MAX_DIM)
        return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)W * H;
    int tk = sc_time_begin(ctx, SC_K_NEAREST);
    hipLaunchKernelGGL(edt_nearest_kernel, dim3((unsigned)((cells + 255) / 256), (unsigned)batch), dim3(256), 0, ctx->stream, occ, d2, W, H,
                       nearest);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}
