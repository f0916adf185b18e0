// edt.hip -- exact squared Euclidean distance transform of occupancy grids (gfx950).
//
// Takes over the clearance/collision queries of the reference's polygon world model
// (obstacle::contains sea_current.hpp:201-251, planning_space::is_obstacle :1274-1280,
// ::cost :1315-1326).  The result is defined mathematically (oracle/sc_oracle.h).
//
// Data flow (HBM bytes per cell):
//   occ  u8 [batch][H][W]   --edt_colbits-->  colbits u32 [batch][nb][W]   (1 read, 1/8 write)
//   colbits                 --edt_band---->   d2 i32 [batch][H][W]         (1/8 + look-back read, 4 write)
// colbits[b][x] bit i = occ[32 b + i][x] != 0: a bit-transposed copy of the grid, one word per
// column per 32-row band.  It is at once the compressed occupancy and the band summary that lets a
// band find the nearest obstacle above/below it (clz/ffs on neighbouring bands' words) without
// touching rows outside the band.
//
// edt_band: one 256-thread workgroup per (grid, band); each wave owns whole rows.
// Row pass = exact cascade of 3-point parabolic erosions.  Because k^2 = 1 + 3 + ... + (2k-1),
//     F_j(x) = min(F_{j-1}(x), min(F_{j-1}(x-1), F_{j-1}(x+1)) + (2j-1)),   F_0(x) = g(x)^2
// equals  min_{|k|<=j} g(x+k)^2 + k^2  exactly (g = vertical distance), and F_j is final once
// (j+1)^2 >= max_x F_j(x).  Lane l keeps PPL consecutive pixels of the row in registers, packed two
// per VGPR as u16 pairs (pixel j with pixel j + PPL/2, so the left/right neighbour vectors of
// register j are simply registers j-1 / j+1); only the two pixels at the lane's ends come from the
// adjacent lanes (one wave_shr and one wave_shl DPP move per iteration).  An iteration over a whole
// row costs 3 packed VALU ops per two pixels and touches no memory.  Saturating u16 adds commute
// with clipping at 65535, so the packed F_j equals min(65535, exact F_j): every pixel that ends
// below 65535 is exact; a row with a pixel still at 65535 (d2 >= 65535: a very sparse grid) is redone
// with 32-bit registers.  Rows are contiguous in HBM: occupancy reads and d2 writes are coalesced.
#include "sc_internal.h"

#define EDT_G_INF 0x7FFF                       // "no obstacle in this column"
#define EDT_F_INF (EDT_G_INF * EDT_G_INF)      // > any real d2 for dims <= 8192

__device__ __forceinline__ uint32_t nonzero_bytes_hi(uint32_t v) {
    // bit 7 of every non-zero byte
    return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
}

// One thread = 4*D columns of one band (D dwords per row); W % (4*D) == 0, rows 4*D-byte aligned.
template <int D>
__global__ void __launch_bounds__(256)
edt_colbits_kernel(const uint8_t* __restrict__ occ, int W, int H, int nb, int batch, uint32_t* __restrict__ colbits) {
    const int per_row = W / (4 * D);
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)batch * nb * per_row;
    if (t >= total) return;
    const int xg = (int)(t % per_row);
    const int b = (int)((t / per_row) % nb);
    const int g = (int)(t / ((long long)per_row * nb));
    const int x0 = xg * 4 * D;
    const uint8_t* base = occ + ((size_t)g * H + (size_t)b * 32) * W + x0;
    const int rows = min(32, H - b * 32);
    uint32_t Q[D][4];
#pragma unroll
    for (int dd = 0; dd < D; ++dd)
#pragma unroll
        for (int j = 0; j < 4; ++j) Q[dd][j] = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        uint32_t v[D];
#pragma unroll
        for (int dd = 0; dd < D; ++dd) v[dd] = 0;
        if (i < rows) {
            if constexpr (D == 4) {
                uint4 q = *reinterpret_cast<const uint4*>(base + (size_t)i * W);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
                v[0] = *reinterpret_cast<const uint32_t*>(base + (size_t)i * W);
            }
        }
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            uint32_t m = nonzero_bytes_hi(v[dd]);  // byte k bit 7 = column k occupied in row i
            Q[dd][i >> 3] |= (m >> 7) << (i & 7);  // byte k of Q[.][j] = rows 8j..8j+7 of column k
        }
    }
    uint32_t* out = colbits + ((size_t)g * nb + b) * W + x0;
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = ((Q[dd][0] >> (8 * k)) & 0xFF) | (((Q[dd][1] >> (8 * k)) & 0xFF) << 8) |
                   (((Q[dd][2] >> (8 * k)) & 0xFF) << 16) | (((Q[dd][3] >> (8 * k)) & 0xFF) << 24);
        *reinterpret_cast<uint4*>(out + 4 * dd) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// Any W: one column per thread, byte loads.
__global__ void __launch_bounds__(256)
edt_colbits_generic_kernel(const uint8_t* __restrict__ occ, int W, int H, int nb, uint32_t* __restrict__ colbits) {
    const int b = blockIdx.y, g = blockIdx.z;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const uint8_t* base = occ + ((size_t)g * H + (size_t)b * 32) * W + x;
    const int rows = min(32, H - b * 32);
    uint32_t w = 0;
    for (int i = 0; i < rows; ++i) w |= (uint32_t)(base[(size_t)i * W] != 0) << i;
    colbits[((size_t)g * nb + b) * W + x] = w;
}

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in issue order; only the compiler must not reorder them.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b)));
}
__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b)));
}
// lane l <- lane l-1 (lane 0 keeps `fill`) / lane l <- lane l+1 (lane 63 keeps `fill`)
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t from_lane_above(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130 /*wave_shl:1*/, 0xF, 0xF, false);
}

template <int PPL, bool FULL>
__global__ void __launch_bounds__(256)
edt_band_kernel(const uint32_t* __restrict__ colbits, int W, int H, int nb, int32_t* __restrict__ d2) {
    constexpr int WAVES = 4;
    constexpr int HP = PPL / 2;                    // packed registers per lane
    constexpr int G = PPL < 16 ? PPL : 16;         // pixels per lane per transpose pass
    constexpr int WP = 64 * PPL;                   // padded row width
    constexpr int COLN = WP + 64;                  // pitch PPL+1 per PPL columns: conflict-free per-lane reads
    constexpr int TRN = (64 * (G + 1) > WP / 2 ? 64 * (G + 1) : WP / 2);  // dwords per wave
    constexpr bool REG32 = PPL <= 32;              // 32-bit fallback in registers (else LDS scan)
    extern __shared__ uint32_t smem[];
    uint32_t* cw = smem;              // [COLN] band's own column words
    uint32_t* cud = cw + COLN;        // [COLN] up | dn << 16: rows to the nearest obstacle above the band top / below its bottom
    uint32_t* trs = cud + COLN;       // [WAVES][TRN]
    const int b = blockIdx.x % nb, g = blockIdx.x / nb;
    const uint32_t* cb = colbits + (size_t)g * nb * W;

    // ---- phase 1: per column, nearest obstacle above / below the band (look-back over band words) ----
    for (int x = threadIdx.x; x < WP; x += WAVES * 64) {
        uint32_t w = 0;
        int up = EDT_G_INF, dn = EDT_G_INF;
        if (x < W) {
            w = cb[(size_t)b * W + x];
            for (int base = b - 1; base >= 0 && up == EDT_G_INF; base -= 4) {
                uint32_t ww[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) ww[t] = base - t >= 0 ? cb[(size_t)(base - t) * W + x] : 0u;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (up == EDT_G_INF && ww[t]) up = (b - (base - t)) * 32 - (31 - __clz((int)ww[t]));
            }
            for (int base = b + 1; base < nb && dn == EDT_G_INF; base += 4) {
                uint32_t ww[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) ww[t] = base + t < nb ? cb[(size_t)(base + t) * W + x] : 0u;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (dn == EDT_G_INF && ww[t]) dn = ((base + t) - b) * 32 + (__ffs((int)ww[t]) - 1) - 31;
            }
        }
        const int pos = x + x / PPL;
        cw[pos] = w;
        cud[pos] = (uint32_t)up | ((uint32_t)dn << 16);
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* tr = trs + (size_t)wave * TRN;
    const int y0 = b * 32;
    const int nrows = min(32, H - y0);
    const int nvalid = W - PPL * lane;  // pixels j < nvalid of this lane are inside the row

    // vertical distance of the lane's pixel j in band row i (EDT_G_INF: none in this column)
    auto gdist = [&](int i, int j) -> uint32_t {
        const int pos = (PPL + 1) * lane + j;
        const uint32_t w = cw[pos], ud = cud[pos];
        const uint32_t wl = w >> i, wh = w << (31 - i);
        uint32_t gg = min(i + (ud & 0xFFFFu), (31 - i) + (ud >> 16));
        gg = min(gg, (uint32_t)(__ffs((int)wl) - 1));              // 0xFFFFFFFF when no bit at/below row i
        gg = min(gg, wh ? (uint32_t)__clz((int)wh) : 0xFFFFFFFFu);
        return min(gg, (uint32_t)EDT_G_INF);
    };

    for (int i = wave; i < nrows; i += WAVES) {
        int32_t* out = d2 + ((size_t)g * H + y0 + i) * W;
        uint32_t V[PPL];  // final values of the lane's pixels (32-bit)
        bool done = false;
        // ---- packed u16 cascade ----
        {
            uint32_t P[HP];
#pragma unroll
            for (int j = 0; j < HP; ++j) {
                const uint32_t glo = min(gdist(i, j), 256u), ghi = min(gdist(i, j + HP), 256u);
                P[j] = min(glo * glo, 0xFFFFu) | (min(ghi * ghi, 0xFFFFu) << 16);
                if (!FULL) {  // pixels past the row end are "no obstacle"
                    if (j >= nvalid) P[j] |= 0x0000FFFFu;
                    if (j + HP >= nvalid) P[j] |= 0xFFFF0000u;
                }
            }
            bool saturated = false;
            for (int it = 1; it <= 256; ++it) {
                const uint32_t c = (uint32_t)(2 * it - 1) * 0x00010001u;
                const uint32_t T = P[HP - 1], S = P[0];
                const uint32_t below = from_lane_below(T, 0xFFFFFFFFu);
                const uint32_t above = from_lane_above(S, 0xFFFFFFFFu);
                // left neighbours of (pixel 0, pixel HP) and right neighbours of (pixel HP-1, pixel PPL-1)
                const uint32_t L0 = __builtin_amdgcn_alignbit(T, below, 16);
                const uint32_t RL = __builtin_amdgcn_alignbit(above, S, 16);
                uint32_t prev = L0;
#pragma unroll
                for (int j = 0; j < HP; ++j) {
                    const uint32_t cur = P[j];
                    const uint32_t nxt = j < HP - 1 ? P[j + 1] : RL;
                    P[j] = pk_min(cur, pk_add_sat(pk_min(prev, nxt), c));
                    prev = cur;
                }
                if ((it & 3) == 0 || it <= 2) {
                    uint32_t m;
                    if (FULL) {
                        m = P[0];
#pragma unroll
                        for (int j = 1; j < HP; ++j) m = pk_max(m, P[j]);
                        m = max(m & 0xFFFFu, m >> 16);
                    } else {
                        m = 0;
#pragma unroll
                        for (int j = 0; j < HP; ++j) {
                            if (j < nvalid) m = max(m, P[j] & 0xFFFFu);
                            if (j + HP < nvalid) m = max(m, P[j] >> 16);
                        }
                    }
                    const uint32_t thr = (uint32_t)(it + 1) * (uint32_t)(it + 1);
                    if (__ballot(m > thr) == 0) break;
                    // Still "no obstacle" somewhere after 64 columns: a sparse row.  Packed values
                    // cannot reach it before iteration 256; go to the 32-bit path now.
                    if ((it == 64 || it == 256) && __ballot(m == 0xFFFFu) != 0) { saturated = true; break; }
                }
            }
            if (!saturated) {
#pragma unroll
                for (int j = 0; j < HP; ++j) { V[j] = P[j] & 0xFFFFu; V[j + HP] = P[j] >> 16; }
                done = true;
            }
        }
        // ---- 32-bit cascade (rows of very sparse grids) ----
        if (REG32 && !done) {
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const uint32_t gg = gdist(i, j);
                V[j] = (FULL || j < nvalid) ? gg * gg : (uint32_t)EDT_F_INF;
            }
            for (int it = 1; it < WP; ++it) {
                const uint32_t c = (uint32_t)(2 * it - 1);
                const uint32_t below = from_lane_below(V[PPL - 1], (uint32_t)EDT_F_INF);
                const uint32_t above = from_lane_above(V[0], (uint32_t)EDT_F_INF);
                uint32_t prev = below;
#pragma unroll
                for (int j = 0; j < PPL; ++j) {
                    const uint32_t cur = V[j];
                    const uint32_t nxt = j < PPL - 1 ? V[j + 1] : above;
                    V[j] = min(cur, min(prev, nxt) + c);
                    prev = cur;
                }
                if ((it & 7) == 0) {
                    uint32_t m = 0;
#pragma unroll
                    for (int j = 0; j < PPL; ++j)
                        if (FULL || j < nvalid) m = max(m, V[j]);
                    // finite values are final once (it+1)^2 covers them; EDT_F_INF needs the whole row
                    const uint32_t thr = (uint32_t)(it + 1) * (uint32_t)(it + 1);
                    if (__ballot(m > thr && (m < (uint32_t)EDT_F_INF || it + 1 < W)) == 0) break;
                }
            }
#pragma unroll
            for (int j = 0; j < PPL; ++j) V[j] = V[j] >= (uint32_t)EDT_F_INF ? (uint32_t)INT32_MAX : V[j];
            done = true;
        }
#ifdef EDT_DIRECT_STORE
        if (done && FULL && PPL % 4 == 0) {
            // each lane stores its own PPL consecutive pixels as 16-byte pieces
#pragma unroll
            for (int j = 0; j < PPL; j += 4)
                *reinterpret_cast<int4*>(out + PPL * lane + j) = make_int4((int)V[j], (int)V[j + 1], (int)V[j + 2], (int)V[j + 3]);
        } else
#endif
        if (done) {
            // ---- transpose through LDS so that global stores are lane-contiguous ----
#pragma unroll
            for (int p = 0; p < PPL / G; ++p) {
#pragma unroll
                for (int jj = 0; jj < G; ++jj) tr[(G + 1) * lane + jj] = V[G * p + jj];
                wave_lds_sync();
                if constexpr (G % 4 == 0) {
#pragma unroll
                    for (int k = 0; k < G / 4; ++k) {
                        const int e = 4 * (64 * k + lane);          // element index in [0, 64 G)
                        const int x = PPL * (e / G) + G * p + e % G;
                        const int a = e + e / G;
                        int4 v = make_int4((int)tr[a], (int)tr[a + 1], (int)tr[a + 2], (int)tr[a + 3]);
                        if (FULL || x + 3 < W) {
                            if (FULL || (((uintptr_t)(out + x)) & 15) == 0) *reinterpret_cast<int4*>(out + x) = v;
                            else { out[x] = v.x; out[x + 1] = v.y; out[x + 2] = v.z; out[x + 3] = v.w; }
                        } else {
                            if (x < W) out[x] = v.x;
                            if (x + 1 < W) out[x + 1] = v.y;
                            if (x + 2 < W) out[x + 2] = v.z;
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < G; ++k) {
                        const int e = 64 * k + lane;
                        const int x = PPL * (e / G) + G * p + e % G;
                        if (x < W) out[x] = (int)tr[e + e / G];
                    }
                }
                wave_lds_sync();
            }
        } else {
            // ---- wide rows of very sparse grids: LDS outward scan with 32-bit values ----
            uint16_t* row = reinterpret_cast<uint16_t*>(tr);
#pragma unroll
            for (int j = 0; j < PPL; ++j) row[PPL * lane + j] = (uint16_t)gdist(i, j);
            wave_lds_sync();
            for (int x = lane; x < W; x += 64) {
                const int g0 = row[x];
                int best = g0 * g0;
                for (int k = 1; k < W && k * k < best; ++k) {
                    const int xl = x - k, xr = x + k;
                    const int gl = xl >= 0 ? (int)row[xl] : EDT_G_INF;
                    const int gr = xr < W ? (int)row[xr] : EDT_G_INF;
                    const int gm = min(gl, gr);
                    best = min(best, gm * gm + k * k);
                }
                out[x] = best >= EDT_F_INF ? INT32_MAX : best;
            }
            wave_lds_sync();
        }
    }
}

template <int PPL, bool FULL>
static int launch_band(sc_ctx* ctx, const uint32_t* colbits, int W, int H, int nb, int batch, int32_t* d2) {
    constexpr int G = PPL < 16 ? PPL : 16;
    constexpr int WP = 64 * PPL;
    constexpr int TRN = (64 * (G + 1) > WP / 2 ? 64 * (G + 1) : WP / 2);
    const size_t lds = (size_t)(2 * (WP + 64) + 4 * TRN) * sizeof(uint32_t);
    static bool attr_set = false;
    if (!attr_set) {
        SC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(edt_band_kernel<PPL, FULL>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    int tk = sc_time_begin(ctx, SC_K_EDT_BAND);
    hipLaunchKernelGGL((edt_band_kernel<PPL, FULL>), dim3((unsigned)(nb * batch)), dim3(256), lds, ctx->stream,
                       colbits, W, H, nb, d2);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

template <int PPL>
static int launch_band_ppl(sc_ctx* ctx, const uint32_t* colbits, int W, int H, int nb, int batch, int32_t* d2) {
    return W == 64 * PPL ? launch_band<PPL, true>(ctx, colbits, W, H, nb, batch, d2)
                         : launch_band<PPL, false>(ctx, colbits, W, H, nb, batch, d2);
}

int sc_launch_edt(sc_ctx* ctx, const uint8_t* occ, int W, int H, int batch, int32_t* d2) {
    const int nb = (H + 31) / 32;
    int r = sc_scratch_reserve(ctx, &ctx->colbits, (size_t)batch * nb * W * sizeof(uint32_t));
    if (r != SC_OK) return r;
    uint32_t* colbits = (uint32_t*)ctx->colbits.p;
    if (batch > 65535 || nb > 65535) return SC_ERR_INVALID;

    int tk = sc_time_begin(ctx, SC_K_EDT_COLBITS);
#ifndef EDT_COLBITS_D
#define EDT_COLBITS_D 4
#endif
    if (EDT_COLBITS_D == 4 && W % 16 == 0 && ((uintptr_t)occ & 15) == 0) {
        const long long total = (long long)batch * nb * (W / 16);
        hipLaunchKernelGGL(edt_colbits_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           occ, W, H, nb, batch, colbits);
    } else if (W % 4 == 0 && ((uintptr_t)occ & 3) == 0) {
        const long long total = (long long)batch * nb * (W / 4);
        hipLaunchKernelGGL(edt_colbits_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           occ, W, H, nb, batch, colbits);
    } else {
        dim3 grid((W + 255) / 256, nb, batch);
        hipLaunchKernelGGL(edt_colbits_generic_kernel, grid, dim3(256), 0, ctx->stream, occ, W, H, nb, colbits);
    }
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());

    // pixels per lane: smallest power of two with 64 * PPL >= W
    if (W <= 128) return launch_band_ppl<2>(ctx, colbits, W, H, nb, batch, d2);
    if (W <= 256) return launch_band_ppl<4>(ctx, colbits, W, H, nb, batch, d2);
    if (W <= 512) return launch_band_ppl<8>(ctx, colbits, W, H, nb, batch, d2);
    if (W <= 1024) return launch_band_ppl<16>(ctx, colbits, W, H, nb, batch, d2);
    if (W <= 2048) return launch_band_ppl<32>(ctx, colbits, W, H, nb, batch, d2);
    if (W <= 4096) return launch_band_ppl<64>(ctx, colbits, W, H, nb, batch, d2);
    return launch_band_ppl<128>(ctx, colbits, W, H, nb, batch, d2);
}

extern "C" int sc_edt_u8_i32(sc_ctx* ctx, const uint8_t* occ, int W, int H, int batch, int32_t* d2) {
    if (!ctx || !occ || !d2 || W <= 0 || H <= 0 || batch <= 0 || W > SC_MAX_DIM || H > SC_MAX_DIM) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    return sc_launch_edt(ctx, occ, W, H, batch, d2);
}

// ---- legal-move mask --------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
moves_kernel(const int32_t* __restrict__ d2, int W, int H, int32_t rmin, uint8_t* __restrict__ moves) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const int32_t* r1 = d2 + (size_t)y * W;
    bool t[3][3];
#pragma unroll
    for (int j = -1; j <= 1; ++j)
#pragma unroll
        for (int i = -1; i <= 1; ++i) {
            int xx = x + i, yy = y + j;
            t[j + 1][i + 1] = xx >= 0 && xx < W && yy >= 0 && yy < H && r1[(ptrdiff_t)j * W + xx] >= rmin;
        }
    uint32_t m = 0;
    if (t[1][1]) {
        // d: dx = {1,-1,0,0,1,-1,1,-1}, dy = {0,0,1,-1,1,1,-1,-1}
        m |= (uint32_t)t[1][2] << 0;
        m |= (uint32_t)t[1][0] << 1;
        m |= (uint32_t)t[2][1] << 2;
        m |= (uint32_t)t[0][1] << 3;
        m |= (uint32_t)(t[2][2] && t[1][2] && t[2][1]) << 4;
        m |= (uint32_t)(t[2][0] && t[1][0] && t[2][1]) << 5;
        m |= (uint32_t)(t[0][2] && t[1][2] && t[0][1]) << 6;
        m |= (uint32_t)(t[0][0] && t[1][0] && t[0][1]) << 7;
    }
    moves[(size_t)y * W + x] = (uint8_t)m;
}

int sc_launch_moves(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2, uint8_t* moves) {
    int32_t rmin = r2 > 1 ? r2 : 1;
    int tk = sc_time_begin(ctx, SC_K_MOVES);
    hipLaunchKernelGGL(moves_kernel, dim3((W + 255) / 256, H), dim3(256), 0, ctx->stream, d2, W, H, rmin, moves);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

extern "C" int sc_moves_i32_u8(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2_clear, uint8_t* moves) {
    if (!ctx || !d2 || !moves || W <= 0 || H <= 0 || W > SC_MAX_DIM || H > SC_MAX_DIM) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    return sc_launch_moves(ctx, d2, W, H, r2_clear, moves);
}
