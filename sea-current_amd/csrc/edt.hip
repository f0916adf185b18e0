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
#include <stdlib.h>

#define EDT_G_INF 0x7FFF                       // "no obstacle in this column"
#define EDT_F_INF (EDT_G_INF * EDT_G_INF)      // > any real d2 for dims <= 8192

__device__ __forceinline__ uint32_t nonzero_bytes_hi(uint32_t v) {
    // bit 7 of every non-zero byte
    return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
}

// Fast path (W % 16 == 0 with 16-byte loads, W % 4 == 0 with dword loads): block = 64 column groups x 4 row groups of ONE band.  Thread (xg, rg)
// loads rows 8 rg .. 8 rg + 7 of 16 adjacent columns (8 independent 16-byte loads, issued back to
// back: the block's requests are one contiguous 32 KiB burst, which keeps HBM row-buffer locality --
// one thread walking all 32 rows gives 2048 concurrent 1-KiB streams 32 KiB apart and reads at
// under half the rate), reduces them to 8 row bits per column, and the block transposes through
// LDS so that every thread assembles and stores the 32-bit words of 4 adjacent columns.
template <bool ALIGNED16>
__global__ void __launch_bounds__(256)
edt_colbits_kernel(const uint8_t* __restrict__ occ, int W, int H, int nb, uint32_t* __restrict__ colbits) {
    __shared__ uint4 sq[4][64];
    const int xg = threadIdx.x, rg = threadIdx.y;
    const int b = blockIdx.y, g = blockIdx.z;
    const int x0 = (blockIdx.x * 64 + xg) * 16;
    uint32_t Q[4] = {0, 0, 0, 0};   // byte k of Q[dd] = rows 8 rg .. 8 rg + 7 of column x0 + 4 dd + k
    if (x0 < W) {
        const int row0 = b * 32 + rg * 8;
        const uint8_t* base = occ + ((size_t)g * H + row0) * W + x0;
        uint4 v[8];
        if (ALIGNED16) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = row0 + i < H ? *reinterpret_cast<const uint4*>(base + (size_t)i * W) : make_uint4(0, 0, 0, 0);
        } else {
            // W % 4 == 0 only: rows are dword-aligned; the last thread of a row may hold fewer than 16 columns
            const int nd = min(4, (W - x0) >> 2);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t* r4 = reinterpret_cast<const uint32_t*>(base + (size_t)i * W);
                const bool in = row0 + i < H;
                v[i] = make_uint4(in && nd > 0 ? r4[0] : 0u, in && nd > 1 ? r4[1] : 0u, in && nd > 2 ? r4[2] : 0u, in && nd > 3 ? r4[3] : 0u);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            Q[0] |= (nonzero_bytes_hi(v[i].x) >> 7) << i;
            Q[1] |= (nonzero_bytes_hi(v[i].y) >> 7) << i;
            Q[2] |= (nonzero_bytes_hi(v[i].z) >> 7) << i;
            Q[3] |= (nonzero_bytes_hi(v[i].w) >> 7) << i;
        }
    }
    sq[rg][xg] = make_uint4(Q[0], Q[1], Q[2], Q[3]);
    __syncthreads();
    // thread t assembles columns 4 t .. 4 t + 3 of the block: dword (t % 4) of column group t / 4
    const int t = rg * 64 + xg;
    const int xo = blockIdx.x * 1024 + 4 * t;
    if (xo < W) {
        const uint32_t* sw = reinterpret_cast<const uint32_t*>(&sq[0][0]);
        uint32_t q[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = sw[(r * 64 + t / 4) * 4 + (t & 3)];
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = ((q[0] >> (8 * k)) & 0xFF) | (((q[1] >> (8 * k)) & 0xFF) << 8) |
                   (((q[2] >> (8 * k)) & 0xFF) << 16) | (((q[3] >> (8 * k)) & 0xFF) << 24);
        *reinterpret_cast<uint4*>(colbits + ((size_t)g * nb + b) * W + xo) = make_uint4(w[0], w[1], w[2], w[3]);
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

// d2 is written once and not read again by this library: non-temporal stores keep the 256 MiB of output
// from lingering as dirty lines whose write-back would otherwise slow whatever runs next (measured:
// the following colbits launch drops from ~25 us to ~10 us, the read floor for 64 MiB).
typedef int v4i_t __attribute__((ext_vector_type(4)));
#define EDT_STORE4(p, v) __builtin_nontemporal_store(v4i_t{(v).x, (v).y, (v).z, (v).w}, reinterpret_cast<v4i_t*>(p))

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
__device__ __forceinline__ uint32_t pk_mul_lo(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2_t, a) * __builtin_bit_cast(us2_t, b));
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
edt_band_kernel(const uint32_t* __restrict__ colbits, int W, int H, int nb, int32_t* __restrict__ d2,
                const int32_t* __restrict__ flags) {
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
    // Workgroups are dealt to the 8 XCDs round-robin, each XCD with its own L2.  Remap so that an XCD works through a
    // contiguous range of (grid, band) pairs: the column words of neighbouring bands that the look-back re-reads are
    // then found in the same L2 instead of being fetched once per XCD.
    const unsigned nwg = gridDim.x, per = nwg >> 3;
    const unsigned vid = blockIdx.x < (per << 3) ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
    const int b = (int)(vid % (unsigned)nb), g = (int)(vid / (unsigned)nb);
    if (flags && flags[(size_t)g * nb + b] == 0) return;   // second pass after the windowed kernel: only the bands it gave up on
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
                // in chunks of up to 8 registers: all neighbour minima of a chunk from the old values first, then the
                // updates (no dependent packed pairs, no register copies); `left_old` carries the old value across chunks
                constexpr int CH = HP < 8 ? HP : 8;
                uint32_t left_old = L0;
#pragma unroll
                for (int j0 = 0; j0 < HP; j0 += CH) {
                    uint32_t m[CH];
#pragma unroll
                    for (int t = 0; t < CH; ++t) {
                        const int j = j0 + t;
                        m[t] = pk_min(t ? P[j - 1] : left_old, j < HP - 1 ? P[j + 1] : RL);
                    }
                    left_old = P[j0 + CH - 1];
#pragma unroll
                    for (int t = 0; t < CH; ++t) m[t] = pk_add_sat(m[t], c);
#pragma unroll
                    for (int t = 0; t < CH; ++t) P[j0 + t] = pk_min(P[j0 + t], m[t]);
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
                    // still "no obstacle" (d2 >= 65535) somewhere after 256 columns: 32-bit path
                    if (it == 256 && __ballot(m == 0xFFFFu) != 0) { saturated = true; break; }
                    const uint32_t thr = (uint32_t)(it + 1) * (uint32_t)(it + 1);
                    if (__ballot(m > thr) == 0) break;
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
                            if (FULL || (((uintptr_t)(out + x)) & 15) == 0) EDT_STORE4(out + x, v);
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

// ---- edt_band_g8_kernel: the W <= 1024 fast path -------------------------------------------------
// Same cascade, leaner set-up.  Phase 1 turns the band's column words into ALL 32 x W vertical
// distances with a packed-u16 recurrence (two columns per VGPR:  gu_i = bit_i ? 0 : gu_{i-1} + 1
// top-down, gd bottom-up, g = min(gu, gd, 255)), ~5 VALU ops per pixel instead of ~16 for the
// closed form, and stores them as bytes g8[row][x] in LDS (32 KiB at W = 1024).  A row is then one
// 16-byte LDS read per lane plus perm + pk_mul per pixel pair.  255 stands for ">= 255", so a packed
// result is trusted only below 255^2 = 65025; rows that end above it (very sparse grids) are redone
// by the 32-bit cascade with exact distances recomputed from the column words in global memory.
__device__ __noinline__ uint32_t edt_gdist_global(const uint32_t* cb, int W, int nb, int b, int x, int i) {
    const uint32_t w = cb[(size_t)b * W + x];
    const uint32_t wl = w >> i, wh = w << (31 - i);
    if (wl & 1u) return 0;
    uint32_t gg = EDT_G_INF;
    if (wl) gg = (uint32_t)(__ffs((int)wl) - 1);
    else
        for (int bb = b + 1; bb < nb; ++bb) {
            const uint32_t ww = cb[(size_t)bb * W + x];
            if (ww) { gg = (uint32_t)((bb - b) * 32 + (__ffs((int)ww) - 1) - i); break; }
        }
    if (wh) gg = min(gg, (uint32_t)__clz((int)wh));
    else
        for (int bb = b - 1; bb >= 0; --bb) {
            const uint32_t ww = cb[(size_t)bb * W + x];
            if (ww) { gg = min(gg, (uint32_t)((b - bb) * 32 + i - (31 - __clz((int)ww)))); break; }
        }
    return min(gg, (uint32_t)EDT_G_INF);
}

// TILED (rows wider than 1024, PPL == 16): a workgroup handles a window of 1024 columns of its band -- a core of
// 1024 - 2 halo columns plus `halo` on either side.  After `it` cascade steps a core pixel has seen every site
// within `it` columns, all of them inside the window, so the usual stopping rule holds as long as it <= halo;
// only the core is tested and stored.  A row that needs more steps (or leaves the packed range) raises its band's flag and
// the whole-row kernel redoes that band afterwards.
#define EDT_TILE_HALO_MIN 32   // the halo is a multiple of 16 (whole lanes) chosen per width: see launch_band_g8_tiled
template <int PPL, bool FULL, bool TILED = false>
__global__ void __launch_bounds__(512, 8)
edt_band_g8_kernel(const uint32_t* __restrict__ colbits, int W, int H, int nb, int32_t* __restrict__ d2, int tiles, int halo,
                   int32_t* __restrict__ flags, const int32_t* __restrict__ only) {
    static_assert(PPL == 8 || PPL == 16, "g8 path: 8 or 16 pixels per lane");
    static_assert(!TILED || (PPL == 16 && FULL), "tiled windows are full 1024-column rows");
    constexpr int WAVES = 8;
    constexpr int HP = PPL / 2;
    constexpr int WP = 64 * PPL;
    constexpr int TRN = 64 * (HP + 1);             // packed transpose buffer, dwords per wave
    constexpr uint32_t TRUST = 255u * 255u;        // packed values below this are exact
    extern __shared__ uint32_t smem[];
    uint8_t* g8 = reinterpret_cast<uint8_t*>(smem);            // [32][WP]
    uint32_t* trs = smem + 32 * WP / 4;                        // [WAVES][TRN]
    // Workgroups are dealt to the 8 XCDs round-robin, each XCD with its own L2.  Remap so that an XCD works through a
    // contiguous range of (grid, band) pairs: the column words of neighbouring bands that the look-back re-reads are
    // then found in the same L2 instead of being fetched once per XCD.
    const unsigned nwg = gridDim.x, per = nwg >> 3;
    const unsigned vid = blockIdx.x < (per << 3) ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
    const int tile = TILED ? (int)(vid % (unsigned)tiles) : 0;
    const unsigned bg = TILED ? vid / (unsigned)tiles : vid;
    const int b = (int)(bg % (unsigned)nb), g = (int)(bg / (unsigned)nb);
    if (TILED && only && only[(size_t)g * nb + b] == 0) return;        // a later pass: only the bands an earlier one gave up on
    const int tcore = 1024 - 2 * halo;                                  // columns a window stores
    const int xw0 = TILED ? tile * tcore - halo : 0;                    // global column of the window's first pixel
    const uint32_t* cb = colbits + (size_t)g * nb * W;

#ifdef EDT_ABLATE_PHASE1   // timing-only: no look-back / recurrence, constant distances
    for (int q = threadIdx.x; q < 32 * WP / 4; q += WAVES * 64) smem[q] = 0x03020302u + (cb[(size_t)b * W + (q & (W - 1))] & 1u);
    if (false)
#endif
    // ---- phase 1: two adjacent columns per thread -> 32 rows of 2 distance bytes ----
    for (int q = threadIdx.x; q < WP / 2; q += WAVES * 64) {
        uint32_t nw[2];
        int up[2], dn[2];
        {
            // first round: the band's own word and the 4 bands above / below, for both columns, as 18
            // independent loads (one memory round trip); further rounds only for columns still unresolved
            uint32_t w[2], wu[2][4], wd[2][4];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int x = xw0 + 2 * q + c;
                const bool in = x < W && (!TILED || x >= 0);
                w[c] = in ? cb[(size_t)b * W + x] : 0u;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    wu[c][t] = (in && b - 1 - t >= 0) ? cb[(size_t)(b - 1 - t) * W + x] : 0u;
                    wd[c][t] = (in && b + 1 + t < nb) ? cb[(size_t)(b + 1 + t) * W + x] : 0u;
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int x = xw0 + 2 * q + c;
                up[c] = EDT_G_INF; dn[c] = EDT_G_INF;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (up[c] == EDT_G_INF && wu[c][t]) up[c] = (t + 1) * 32 - (31 - __clz((int)wu[c][t]));
                    if (dn[c] == EDT_G_INF && wd[c][t]) dn[c] = (t + 1) * 32 + (__ffs((int)wd[c][t]) - 1) - 31;
                }
                if (x < W && (!TILED || x >= 0)) {
                    for (int base = b - 5; base >= 0 && up[c] == EDT_G_INF; base -= 4) {
                        uint32_t ww[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) ww[t] = base - t >= 0 ? cb[(size_t)(base - t) * W + x] : 0u;
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (up[c] == EDT_G_INF && ww[t]) up[c] = (b - (base - t)) * 32 - (31 - __clz((int)ww[t]));
                    }
                    for (int base = b + 5; base < nb && dn[c] == EDT_G_INF; base += 4) {
                        uint32_t ww[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) ww[t] = base + t < nb ? cb[(size_t)(base + t) * W + x] : 0u;
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (dn[c] == EDT_G_INF && ww[t]) dn[c] = ((base + t) - b) * 32 + (__ffs((int)ww[t]) - 1) - 31;
                    }
                }
                nw[c] = ~w[c];
            }
        }
        // top-down: gu = rows to the nearest obstacle at or above, clamped at 255 and kept two rows per
        // VGPR (bytes: row 2m in [7:0] / [23:16], row 2m+1 in [15:8] / [31:24]) to stay within 64 VGPRs
        uint32_t GU2[16];
        const uint32_t nwA = (nw[0] & 0xFFFFu) | (nw[1] << 16), nwB = (nw[0] >> 16) | (nw[1] & 0xFFFF0000u);
        uint32_t gu = (uint32_t)(up[0] - 1) | ((uint32_t)(up[1] - 1) << 16);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            // 1 per half whose cell is free in row i (rows 0..15 from nwA, 16..31 from nwB: both columns' bits of a
            // row sit 16 apart, so one shift and one mask fetch them together); multiplying resets the others to 0
            const uint32_t free01 = ((i < 16 ? nwA : nwB) >> (i & 15)) & 0x00010001u;
            gu = pk_mul_lo(pk_add_sat(gu, 0x00010001u), free01);
            const uint32_t guc = pk_min(gu, 0x00FF00FFu);
            if (i & 1) GU2[i >> 1] |= guc << 8; else GU2[i >> 1] = guc;
        }
        // bottom-up: gd restarts at 0 wherever gu == 0 (an obstacle); values past 255 may be clipped
        // at >= 256 because only min(g, 255) is kept
        uint32_t gd = (uint32_t)(dn[0] - 1) | ((uint32_t)(dn[1] - 1) << 16);
#pragma unroll
        for (int i = 31; i >= 0; --i) {
            const uint32_t guc = ((i & 1) ? (GU2[i >> 1] >> 8) : GU2[i >> 1]) & 0x00FF00FFu;
            const uint32_t cap = guc << 8;   // <= 0xFF00 per half
            gd = pk_min(pk_add_sat(gd, 0x00010001u), cap);
            const uint32_t gg = pk_min(guc, gd);
            // bytes 0 and 2 of gg (the two columns' distances) -> one halfword, with a single byte permute
            *reinterpret_cast<uint16_t*>(g8 + (size_t)i * WP + 2 * q) = (uint16_t)__builtin_amdgcn_perm(0u, gg, 0x0C0C0200u);
        }
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int y0 = b * 32;
    const int nrows = min(32, H - y0);
    const int nvalid = W - PPL * lane;
    // PPL == 16: a wave's transposition buffer is 1 KiB of its own (lanes 0..31) plus the g8 row it
    // processes first (row `wave`, lanes 32..63), which nobody else reads and which is dead once the wave
    // has pulled it into registers.  The kernel then needs 40 KiB of LDS and four workgroups (32 waves) fit.
    uint32_t* tr = trs + (size_t)wave * TRN;
    uint32_t* trb = tr;   // buffer half of lanes 32..63
    if constexpr (PPL == 16) {
        tr = trs + (size_t)wave * 256;
        trb = smem + (size_t)wave * (WP / 4) - 256;   // so that trb[8 * lane + r] with lane >= 32 lands in row `wave`
    }
#ifdef EDT_ABLATE_ROWS   // timing-only: phase 1 alone (one store keeps it alive)
    if (threadIdx.x == 0) d2[((size_t)g * H + y0) * W] = g8[lane];
    if (true) return;
#endif
    int chk_from = 2;   // first cascade step after which this wave tests for convergence (adapted row by row)
    for (int i = wave; i < nrows; i += WAVES) {
        int32_t* out = d2 + ((size_t)g * H + y0 + i) * W;
        uint32_t P[HP];
        {
            uint32_t dw[PPL / 4];
            if constexpr (PPL == 16) {
                const uint4 v = *reinterpret_cast<const uint4*>(g8 + (size_t)i * WP + 16 * lane);
                dw[0] = v.x; dw[1] = v.y; dw[2] = v.z; dw[3] = v.w;
            } else {
                const uint2 v = *reinterpret_cast<const uint2*>(g8 + (size_t)i * WP + 8 * lane);
                dw[0] = v.x; dw[1] = v.y;
            }
#pragma unroll
            for (int j = 0; j < HP; ++j) {
                // (byte j) | (byte j + HP) << 16, then square both halves
                const uint32_t sel = 0x0C000C00u | (uint32_t)(j % 4) | ((uint32_t)(4 + j % 4) << 16);
                const uint32_t t = __builtin_amdgcn_perm(dw[(j + HP) / 4], dw[j / 4], sel);
                P[j] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2_t, t) * __builtin_bit_cast(us2_t, t));
            }
        }
        bool saturated = false;
        // One cascade step.  All eight neighbour minima are formed from the old values before any register is updated:
        // no dependent packed instruction follows its producer directly (each such pair costs a wait state) and no
        // register has to be copied to keep the previous value alive.
        auto cascade_step = [&](int it) {
            const uint32_t c = (uint32_t)(2 * it - 1) * 0x00010001u;
            const uint32_t T = P[HP - 1], S = P[0];
            const uint32_t below = from_lane_below(T, 0xFFFFFFFFu);
            const uint32_t above = from_lane_above(S, 0xFFFFFFFFu);
            const uint32_t L0 = __builtin_amdgcn_alignbit(T, below, 16);
            const uint32_t RL = __builtin_amdgcn_alignbit(above, S, 16);
            uint32_t m[HP];
#pragma unroll
            for (int j = 0; j < HP; ++j) m[j] = pk_min(j ? P[j - 1] : L0, j < HP - 1 ? P[j + 1] : RL);
#pragma unroll
            for (int j = 0; j < HP; ++j) m[j] = pk_add_sat(m[j], c);
#pragma unroll
            for (int j = 0; j < HP; ++j) P[j] = pk_min(P[j], m[j]);
        };
        // The convergence test costs a third of a step, so it only starts one step before the count the wave's previous
        // row needed (rows of a band are alike), then runs every step for a while and every second / fourth step later.
        // When to test only affects how many surplus steps run, never the result.
        int it = 1;
        const int IT_MAX = TILED ? halo : 256;
        for (; it <= IT_MAX; ++it) {
            cascade_step(it);
            if (it < chk_from && !(TILED && it == IT_MAX)) continue;
            if (it >= chk_from + 4 && ((it & 1) || (it > 16 && (it & 3)))) continue;
            uint32_t m;
            if (TILED) {
                // only core pixels that exist in the row count: lanes 2 .. 61, global column < W
                m = 0;
                const int nv = W - (xw0 + PPL * lane);
                if (lane >= halo / PPL && lane < 64 - halo / PPL) {
#pragma unroll
                    for (int j = 0; j < HP; ++j) {
                        if (j < nv) m = max(m, P[j] & 0xFFFFu);
                        if (j + HP < nv) m = max(m, P[j] >> 16);
                    }
                }
            } else if (FULL) {
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
            if (it >= 252 && __ballot(m >= TRUST) != 0) { saturated = true; break; }   // beyond what the clamped bytes can represent
        }
        if (TILED && it > IT_MAX) saturated = true;   // not settled within the halo: leave the grid to the whole-row kernel
        chk_from = max(2, min(it, TILED ? IT_MAX - 1 : 250) - 1);
        if (!saturated) {
            // packed transpose: lane writes its HP packed registers and reads back the halves it needs so that
            // each lane then holds 4 consecutive pixels for one 16-byte store
            if constexpr (PPL == 16) {
                // two 16-byte chunks per lane (registers 0-3 and 4-7), swapped in every other group of four lanes: eight
                // consecutive lanes then cover all 32 banks exactly once, for the writes and for the reads below
                uint32_t* wb = (lane < 32 ? tr : trb) + 8 * lane;
                const int s = (lane >> 2) & 1;
                *reinterpret_cast<uint4*>(wb + 4 * s) = make_uint4(P[0], P[1], P[2], P[3]);
                *reinterpret_cast<uint4*>(wb + 4 * (1 - s)) = make_uint4(P[4], P[5], P[6], P[7]);
            } else {
#pragma unroll
                for (int j = 0; j < HP; ++j) tr[(HP + 1) * lane + j] = P[j];
            }
            wave_lds_sync();
            const uint16_t* trh = reinterpret_cast<const uint16_t*>(tr);
#pragma unroll
            for (int k = 0; k < PPL / 4; ++k) {
                const int x = 4 * (64 * k + lane);
                const int l2 = x / PPL, j2 = x % PPL;                    // owner lane, pixel index there
                int4 v;
                if constexpr (PPL == 16) {
                    // owner lane l2 = 16 k + lane / 4; its chunk (lane & 1) sits at 4 * ((lane & 1) ^ ((l2 >> 2) & 1)) and
                    // (l2 >> 2) & 1 == (lane >> 4) & 1 for every k: one base address per lane, k only adds 512 bytes
                    const uint32_t* rb = (k < 2 ? tr : trb) + 8 * l2 + 4 * ((lane & 1) ^ ((lane >> 4) & 1));
                    const uint4 q = *reinterpret_cast<const uint4*>(rb);
                    const uint32_t sh = (lane & 2) ? 16u : 0u;   // pixels 8..15 are the high halves
                    v = make_int4((int)__builtin_amdgcn_ubfe(q.x, sh, 16), (int)__builtin_amdgcn_ubfe(q.y, sh, 16),
                                  (int)__builtin_amdgcn_ubfe(q.z, sh, 16), (int)__builtin_amdgcn_ubfe(q.w, sh, 16));
                } else {
                    const int hidx = 2 * ((HP + 1) * l2 + (j2 % HP)) + (j2 >= HP ? 1 : 0);
                    v = make_int4(trh[hidx], trh[hidx + 2], trh[hidx + 4], trh[hidx + 6]);
                }
                if constexpr (TILED) {
                    const int xg = xw0 + x;   // global column; only the core of the window is stored
                    if (x >= halo && x < halo + tcore) {
                        if (xg + 3 < W && (((uintptr_t)(out + xg)) & 15) == 0) EDT_STORE4(out + xg, v);
                        else {
                            if (xg < W) out[xg] = v.x;
                            if (xg + 1 < W) out[xg + 1] = v.y;
                            if (xg + 2 < W) out[xg + 2] = v.z;
                            if (xg + 3 < W) out[xg + 3] = v.w;
                        }
                    }
                } else if (FULL || x + 3 < W) {
                    if (FULL || (((uintptr_t)(out + x)) & 15) == 0) EDT_STORE4(out + x, v);
                    else { out[x] = v.x; out[x + 1] = v.y; out[x + 2] = v.z; out[x + 3] = v.w; }
                } else {
                    if (x < W) out[x] = v.x;
                    if (x + 1 < W) out[x + 1] = v.y;
                    if (x + 2 < W) out[x + 2] = v.z;
                }
            }
            wave_lds_sync();
        } else if constexpr (TILED) {
            if (lane == 0) flags[(size_t)g * nb + b] = 1;
        } else {
            // ---- 32-bit cascade with exact distances (very sparse rows) ----
            uint32_t V[PPL];
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const int x = PPL * lane + j;
                uint32_t gg = EDT_G_INF;
                if (x < W) gg = edt_gdist_global(cb, W, nb, b, x, i);
                V[j] = gg * gg;
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
                    const uint32_t thr = (uint32_t)(it + 1) * (uint32_t)(it + 1);
                    if (__ballot(m > thr && (m < (uint32_t)EDT_F_INF || it + 1 < W)) == 0) break;
                }
            }
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const int x = PPL * lane + j;
                if (x < W) out[x] = V[j] >= (uint32_t)EDT_F_INF ? INT32_MAX : (int)V[j];
            }
        }
    }
}

// ---- packed helpers shared by the 1024-wide and the wide-row kernels ----
// v_pk_minimum3_f16 on bit patterns 0 .. 0x7C00 (non-negative f16, denormals included): the f16 order is the integer order
// (tools/microbench/min3_mb.hip checks the instruction on 32 M triples), so the cascade step is two packed instructions per
// register -- T = P + (2 it - 1), P = min3(P, T[left], T[right]) -- instead of three.  To stay below the NaN patterns the
// distance bytes are clamped at 177 (packed values below 177^2 = 31329 are exact) and the cascade stops after 175 steps
// (T <= 31329 + 349 < 0x7C00); rows that need more (very sparse grids) are redone in 32 bits.
#define EDT_W_GCAP 177u
#define EDT_W_ITMAX 175
__device__ __forceinline__ uint32_t pk_min3_f16bits(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_add_wrap(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2_t, a) + __builtin_bit_cast(us2_t, b));
}
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_bit_cast(uint32_t, (us2_t)(__builtin_bit_cast(us2_t, a) * __builtin_bit_cast(us2_t, b) + __builtin_bit_cast(us2_t, c)));
}
// lane l <- lane l-1, lane 0 <- lane 63 / lane l <- lane l+1, lane 63 <- lane 0
__device__ __forceinline__ uint32_t wave_ror1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x13C /*wave_ror:1*/, 0xF, 0xF, true);   // every lane is written: no old value
}
__device__ __forceinline__ uint32_t wave_rol1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x134 /*wave_rol:1*/, 0xF, 0xF, true);
}

// ---- edt_updown_kernel: per (band, column), rows to the nearest obstacle in the bands above / below -----------
// ud[b][x] = up | dn << 16: up = rows from the band's FIRST row up to the nearest obstacle of an earlier band, dn = rows
// from its LAST row down to the nearest obstacle of a later band (EDT_G_INF: none).  With it a band kernel needs two
// words per column -- its own bits and this -- whatever the map looks like, instead of walking neighbouring bands' words
// until it finds an obstacle (up to 18 loads per column pair and round, and a data-dependent loop on open maps).
// Block = 64 columns x 16 band groups: every thread scans its PER <= 16 consecutive bands (words stay in registers), the
// groups exchange their highest / lowest obstacle row through LDS, and each thread finishes its own bands.
template <int PER>
__global__ void __launch_bounds__(1024)
edt_updown_kernel(const uint32_t* __restrict__ colbits, int W, int nb, uint32_t* __restrict__ ud) {
    __shared__ int2 s_lf[16][64];                   // (last, first) obstacle row of every band group, per column
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + tx, g = blockIdx.y;
    const int b0 = ty * PER;                        // PER = ceil(nb / 16) rounded up to a power of two
    const uint32_t* cb = colbits + (size_t)g * nb * W;
    uint32_t wv[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) wv[k] = (b0 + k < nb && x < W) ? cb[(size_t)(b0 + k) * W + x] : 0u;
    int last = -1, first = 0x7FFFFFFF;              // global rows of the group's last / first obstacle
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (wv[k]) {
            const int base = (b0 + k) * 32;
            if (first == 0x7FFFFFFF) first = base + __ffs((int)wv[k]) - 1;
            last = base + 31 - __clz((int)wv[k]);
        }
    s_lf[ty][tx] = make_int2(last, first);
    __syncthreads();
    int run_up = -1, run_dn = 0x7FFFFFFF;
    for (int t = 0; t < ty; ++t) run_up = max(run_up, s_lf[t][tx].x);
    for (int t = ty + 1; t < 16; ++t) run_dn = min(run_dn, s_lf[t][tx].y);
    if (x >= W) return;
    uint32_t upv[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int base = (b0 + k) * 32;
        upv[k] = run_up >= 0 ? (uint32_t)min(base - run_up, EDT_G_INF) : (uint32_t)EDT_G_INF;
        if (wv[k]) run_up = base + 31 - __clz((int)wv[k]);
    }
#pragma unroll
    for (int k = PER - 1; k >= 0; --k) {
        const int b = b0 + k, base = b * 32;
        const uint32_t dn = run_dn != 0x7FFFFFFF ? (uint32_t)min(run_dn - (base + 31), EDT_G_INF) : (uint32_t)EDT_G_INF;
        if (wv[k]) run_dn = base + __ffs((int)wv[k]) - 1;
        if (b < nb) ud[((size_t)g * nb + b) * W + x] = upv[k] | (dn << 16);
    }
}

static void launch_updown(sc_ctx* ctx, const uint32_t* colbits, int W, int nb, int batch, uint32_t* ud) {
    const dim3 grid((W + 63) / 64, batch), block(1024);
    const int per = (nb + 15) / 16;
    if (per <= 1) hipLaunchKernelGGL(edt_updown_kernel<1>, grid, block, 0, ctx->stream, colbits, W, nb, ud);
    else if (per <= 2) hipLaunchKernelGGL(edt_updown_kernel<2>, grid, block, 0, ctx->stream, colbits, W, nb, ud);
    else if (per <= 4) hipLaunchKernelGGL(edt_updown_kernel<4>, grid, block, 0, ctx->stream, colbits, W, nb, ud);
    else if (per <= 8) hipLaunchKernelGGL(edt_updown_kernel<8>, grid, block, 0, ctx->stream, colbits, W, nb, ud);
    else hipLaunchKernelGGL(edt_updown_kernel<16>, grid, block, 0, ctx->stream, colbits, W, nb, ud);   // nb <= 256 (H <= 8192)
}

// ---- edt_band_wide_kernel: rows of 1025 .. 4096 pixels, whole rows in registers -------------------------
// A row is TILES stretches of 1024 pixels; lane l keeps pixels 1024 t + 16 l .. + 15 of every stretch t in 8 packed registers
// (the W <= 1024 layout, TILES times), so a cascade step is the same register work per pixel and the two end pixels of a
// lane come from the adjacent lanes by one wave rotation per stretch and side -- lane 0 / 63 take the rotated register of
// the neighbouring stretch instead.  No windows, no halo columns, no flags and no second launch: a 4096-pixel row costs four
// times a 1024-pixel one.  A workgroup owns 16 rows (half a band of column words: 64 KiB of distance bytes at W = 4096,
// two workgroups per CU); each wave transposes through the LDS row it has already pulled into registers.
//
// The cascade step, the clamp of the distance bytes at 177 and the 175-step limit are those of edt_band_k16_kernel (pk_min3_f16bits).
// One workgroup per CU, 16 wavefronts, persistent over a strided set of row groups (16 rows each), and NO workgroup
// barrier in the loop.  The distance bytes of a group live in one of two LDS slots.  Wavefronts 0-7 are PRODUCERS: each
// owns an eighth of the columns and turns their column words into the distance bytes of the next group (vertical pass,
// pure VALU + LDS) as soon as the slot's previous group has been consumed.  Rows of a finished group are claimed one at a
// time (compare-and-swap on an LDS counter) by every wavefront that has nothing to produce: the eight CONSUMERS always,
// the producers in between.  Three monotonic LDS counters per slot carry the protocol:
//     p1_cnt   producer parts written, ever      group n (use k = n / 2 of its slot) is readable at 8 (k + 1)
//     row_next rows claimed, ever                claims of use k are RB k .. RB (k + 1) - 1
//     row_done rows finished, ever               the slot may be refilled for use k + 1 at RB (k + 1)
// Vertical pass, cascade and stores of one CU then overlap by construction and no wavefront waits for the slowest row of
// a group -- as two phases of one set of wavefronts, separated by a barrier, all CUs computed and then all CUs stored
// (71 us for four 4096^2 grids, of which 18 were the vertical pass alone).  Every wait is bounded (EDT_W_SPIN_LIMIT).
#define EDT_W_SPIN_LIMIT (1 << 22)
// Ordering between the wavefronts of the workgroup goes through LDS only, and the LDS operations of a wave execute in issue
// order: what is needed is that the compiler keeps the order and that earlier LDS results have arrived -- NOT a
// workgroup-scope fence, whose s_waitcnt vmcnt(0) would make every row wait for its own HBM stores to be acknowledged.
#define EDT_LDS_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int TILES, bool FULL>
__global__ void __launch_bounds__(1024, 4)
edt_band_wide_kernel(const uint32_t* __restrict__ colbits, const uint32_t* __restrict__ updown, int W, int H, int nb, int nsb, int ngroups,
                     int32_t* __restrict__ d2, int32_t* __restrict__ fault) {
    static_assert(TILES >= 2 && TILES <= 4, "rows of 1025 .. 4096 pixels");
    constexpr int RB = 16;                          // rows per group
    constexpr int NP = 8;                           // producer wavefronts
    constexpr int WP = 1024 * TILES;
    constexpr uint32_t GC2 = EDT_W_GCAP | (EDT_W_GCAP << 16);
    constexpr uint32_t EDGE = 0x7BFF7BFFu;          // beyond the row ends: above every value, below the f16 NaN patterns
    constexpr uint32_t UDCAP = 0x70007000u;         // "no obstacle" for up / dn: still a non-negative finite f16 pattern after + 32
    extern __shared__ uint32_t smem[];              // [2][RB][WP] distance bytes, clamped at EDT_W_GCAP
    __shared__ uint32_t p1_cnt[2], row_next[2], row_done[2];
    // XCD-aware order (see edt_band_g8_kernel): an XCD works through a contiguous run of (grid, row group) pairs
    const unsigned nwg = gridDim.x, per = nwg >> 3;
    const unsigned vid = blockIdx.x < (per << 3) ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
    // groups vid, vid + nwg, vid + 2 nwg, ...: neighbouring groups cost about the same (on block-type maps up to 10x the
    // mean), so a contiguous run per workgroup leaves the launch waiting for its dearest run; a strided set samples the maps
    const int G = (int)vid < ngroups ? (ngroups - 1 - (int)vid) / (int)nwg + 1 : 0;
    const bool producer = threadIdx.x < NP * 64;
    const int tid = threadIdx.x & (NP * 64 - 1);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 2) { p1_cnt[threadIdx.x] = 0; row_next[threadIdx.x] = 0; row_done[threadIdx.x] = 0; }
    __syncthreads();

    // ---- vertical pass of group n (producers): two adjacent columns per thread and round -> RB rows of 2 distance bytes.
    // All words of the thread's rounds are requested first (one memory round trip), then the recurrences run.
    auto produce = [&](int n) {
        const int s = (int)vid + n * (int)nwg;
        const int sb = s % nsb, g = s / nsb;
        const int y0 = sb * RB;
        const int b = y0 >> 5, r0 = y0 & 31;        // band of column words, first bit of this row group in them
        uint8_t* g8 = reinterpret_cast<uint8_t*>(smem) + (size_t)(n & 1) * RB * WP;
        const uint32_t* cw = colbits + ((size_t)g * nb + b) * W;
        const uint32_t* ub = updown + ((size_t)g * nb + b) * W;
        uint32_t wq[TILES][2], uq[TILES][2];
        if (FULL || !(W & 1)) {
#pragma unroll
            for (int r = 0; r < TILES; ++r) {
                const int x = 2 * (tid + NP * 64 * r);
                uint2 a = make_uint2(0u, 0u), u = make_uint2(UDCAP, UDCAP);
                if (FULL || x < W) { a = *reinterpret_cast<const uint2*>(cw + x); u = *reinterpret_cast<const uint2*>(ub + x); }
                wq[r][0] = a.x; wq[r][1] = a.y; uq[r][0] = u.x; uq[r][1] = u.y;
            }
        } else {
#pragma unroll
            for (int r = 0; r < TILES; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int x = 2 * (tid + NP * 64 * r) + c;
                    wq[r][c] = x < W ? cw[x] : 0u;
                    uq[r][c] = x < W ? ub[x] : UDCAP;
                }
        }
#pragma unroll
        for (int r = 0; r < TILES; ++r) {
            const int q = tid + NP * 64 * r;
            uint32_t nw[2], up[2], dn[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint32_t w = wq[r][c];
                const uint32_t u = pk_min(uq[r][c], UDCAP);
                up[c] = u & 0xFFFFu; dn[c] = u >> 16;           // from the band's first / last row
                // rows of the band's own word outside this group decide first; then measured from the group's first / last row
                const uint32_t wlow = r0 ? (w & ((1u << r0) - 1u)) : 0u;
                const uint32_t whigh = r0 + RB < 32 ? (w >> (r0 + RB)) : 0u;
                up[c] = wlow ? (uint32_t)(r0 - (31 - __clz((int)wlow))) : up[c] + (uint32_t)r0;
                dn[c] = whigh ? (uint32_t)__ffs((int)whigh) : dn[c] + (uint32_t)(32 - (r0 + RB));
                nw[c] = ~(w >> r0);
            }
            // top-down: gu_i = free_i ? gu_{i-1} + 1 : 0 as one packed multiply-add; bottom-up gd likewise; g = min(gu, gd, cap)
            // is one min3 on the f16 order (all values are below 0x7C00)
            const uint32_t nwA = (nw[0] & 0xFFFFu) | (nw[1] << 16);
            uint32_t GU[RB], FR[RB];
            uint32_t gu = (up[0] - 1u) | ((up[1] - 1u) << 16);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                FR[i] = (nwA >> i) & 0x00010001u;
                gu = pk_mad_u16(gu, FR[i], FR[i]);
                GU[i] = gu;
            }
            // Two rows at a time: the thread's 2 x 2 distance bytes go to its partner lane (lane ^ 1) and each lane of the pair
            // stores ONE dword -- the even lane the four columns of the upper row, the odd lane those of the lower row --
            // instead of two 2-byte pieces each (LDS store instructions cost the same whatever their width).
            uint32_t gd = (dn[0] - 1u) | ((dn[1] - 1u) << 16);
            const uint32_t psel = (lane & 1) ? 0x03020706u : 0x05040100u;
            uint8_t* dst = g8 + (size_t)(lane & 1) * WP + 2 * (q & ~1);
#pragma unroll
            for (int i = RB - 2; i >= 0; i -= 2) {
                gd = pk_mad_u16(gd, FR[i + 1], FR[i + 1]);
                const uint32_t g1 = pk_min3_f16bits(GU[i + 1], gd, GC2);
                gd = pk_mad_u16(gd, FR[i], FR[i]);
                const uint32_t g0 = pk_min3_f16bits(GU[i], gd, GC2);
                const uint32_t own = __builtin_amdgcn_perm(g1, g0, 0x06040200u);   // bytes: row i (col 0, col 1), row i + 1 (col 0, col 1)
                const uint32_t oth = (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xF, 0xF, true);
                *reinterpret_cast<uint32_t*>(dst + (size_t)i * WP) = __builtin_amdgcn_perm(oth, own, psel);
            }
        }
    };

    int hint = 4;   // cascade steps the wave's previous row needed (rows claimed in a row are often neighbours)
    int open_rows = 0;   // rows in a row that this wave could not settle by the packed cascade (open space)
    int open_sites = 0;  // ... and the number of sites of the last of them
    // ---- row i of group n: cascade, transposition, stores ----
    auto consume_row = [&](int n, int i) {
        const int s = (int)vid + n * (int)nwg;
        const int sb = s % nsb, g = s / nsb;
        const int y0 = sb * RB;
        if (y0 + i >= H) return;                    // rows past the grid's last one (the group still counts RB claims)
        const int b = y0 >> 5, r0 = y0 & 31;
        const uint32_t* cb = colbits + (size_t)g * nb * W;
        uint32_t* gbuf = smem + (size_t)(n & 1) * (RB * WP / 4);
        const uint8_t* g8 = reinterpret_cast<const uint8_t*>(gbuf);
        int32_t* out = d2 + ((size_t)g * H + y0 + i) * W;
        uint32_t* tr = gbuf + (size_t)i * (WP / 4);   // transposition buffer: the row's own bytes, dead once in registers
        uint32_t P[TILES][8];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const uint4 v = *reinterpret_cast<const uint4*>(g8 + (size_t)i * WP + 1024 * t + 16 * lane);
            const uint32_t dw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t sel = 0x0C000C00u | (uint32_t)(j % 4) | ((uint32_t)(4 + j % 4) << 16);
                const uint32_t tt = __builtin_amdgcn_perm(dw[(j + 8) / 4], dw[j / 4], sel);
                P[t][j] = pk_mul_lo(tt, tt);      // <= 177^2: pixels past the row end carry 177 ("no obstacle")
            }
        }
        auto cascade_step = [&](int it) {
            const uint32_t c = (uint32_t)(2 * it - 1) * 0x00010001u;
            uint32_t Tl[TILES], Tf[TILES], lo[TILES], hi[TILES];
#pragma unroll
            for (int t = 0; t < TILES; ++t) { Tl[t] = pk_add_wrap(P[t][7], c); Tf[t] = pk_add_wrap(P[t][0], c); }
#pragma unroll
            for (int t = 0; t < TILES; ++t) { lo[t] = wave_ror1(Tl[t]); hi[t] = wave_rol1(Tf[t]); }
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const uint32_t below = lane == 0 ? (t ? lo[t ? t - 1 : 0] : EDGE) : lo[t];
                const uint32_t above = lane == 63 ? (t < TILES - 1 ? hi[t < TILES - 1 ? t + 1 : t] : EDGE) : hi[t];
                const uint32_t L0 = __builtin_amdgcn_alignbit(Tl[t], below, 16);   // T of the left neighbours of pixels (0, 8)
                const uint32_t RL = __builtin_amdgcn_alignbit(above, Tf[t], 16);   // T of the right neighbours of pixels (7, 15)
                uint32_t T[8];
                T[0] = Tf[t]; T[7] = Tl[t];
#pragma unroll
                for (int j = 1; j < 7; ++j) T[j] = pk_add_wrap(P[t][j], c);
#pragma unroll
                for (int j = 0; j < 8; ++j) P[t][j] = pk_min3_f16bits(P[t][j], j ? T[j - 1] : L0, j < 7 ? T[j + 1] : RL);
            }
        };
        // The convergence test costs about half a step.  It starts one step before the count the wave's previous row needed
        // (at most at step 8: the previous row may have been another map's), runs every step for a while and then at steps
        // an eighth apart.  When to test only affects how many surplus steps run, never the result.
        bool saturated = false;
        // Open space shows before the first step: 23 adjacent lanes of one stretch whose 16 pixels all carry the clamped
        // distance (368 columns without an obstacle within 176 rows) hold a pixel that no 175 steps can settle -- straight
        // to the site search below instead of 175 steps first.  (Sufficient, not necessary: other rows find out at the end.)
        int nopen = 0;                              // lanes (of 64 TILES) whose 16 pixels all carry the clamped distance
        {
            constexpr uint32_t OPEN2 = (EDT_W_GCAP * EDT_W_GCAP) * 0x00010001u;
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                bool open16 = FULL || W - (1024 * t + 16 * lane) >= 16;
#pragma unroll
                for (int j = 0; j < 8; ++j) open16 = open16 && P[t][j] == OPEN2;
                unsigned long long r = __ballot(open16);
                nopen += __popcll(r);
                r &= r >> 1; r &= r >> 2; r &= r >> 4; r &= r >> 8; r &= r >> 7;      // runs of 2, 4, 8, 16, 23 lanes
                if (r) saturated = true;
            }
        }
        // ... and a row three quarters of which has no obstacle within 176 rows gets 48 steps, not 175, before it changes over
        const int it_cap = 4 * nopen >= 3 * 64 * TILES ? 48 : EDT_W_ITMAX;
        // After a row of open space with few sites (the search then costs less than the 175 steps that settled nothing) the
        // wave's next rows go to the site search at once, except every eighth, which tries the cascade again.  (A row that
        // would have settled costs at most about twice as much that way; maps without open space never get here.)
        if ((open_rows & 7) != 0 && open_sites <= 512) saturated = true;
        int it = 1, next_chk = max(2, min(hint - 1, 8));
        for (; !saturated && it <= it_cap; ++it) {
            cascade_step(it);
            const bool last = it == it_cap;
            if (!last && it < next_chk) continue;
            next_chk = it + 1 + (it >= 12 ? (it >> 3) : 0);
            uint32_t m = 0;
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                uint32_t mt = P[t][0];
#pragma unroll
                for (int j = 1; j < 8; ++j) mt = pk_max(mt, P[t][j]);
                mt = max(mt & 0xFFFFu, mt >> 16);
                if (!FULL) {
                    const int nv = W - (1024 * t + 16 * lane);   // pixels of this lane inside the row
                    if (nv <= 0) mt = 0;
                    else if (nv < 16) {
                        mt = 0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if (j < nv) mt = max(mt, P[t][j] & 0xFFFFu);
                            if (j + 8 < nv) mt = max(mt, P[t][j] >> 16);
                        }
                    }
                }
                m = max(m, mt);
            }
            const uint32_t thr = (uint32_t)(it + 1) * (uint32_t)(it + 1);
            if (__ballot(m > thr) == 0) break;
            if (last) { saturated = true; break; }
        }
        if (!saturated) { hint = it; open_rows = 0; }
        if (saturated) {
            ++open_rows;
            // ---- the packed cascade cannot settle this row (distances beyond 175 columns: open space) ----
            // The row's SITES -- columns with an obstacle anywhere, with their exact vertical distance from the column word
            // and the up / down word: O(1) per column -- are compacted into the row's own LDS bytes (dead by now), and the
            // nearest site of a pixel is found through the monotonicity of the argmin (the leftmost nearest site never
            // moves left as the pixel moves right): one lane per 64-pixel block finds the block's first pixel's site among
            // all K, then every block's pixels look only between their block's site and the next block's.  K + W / 64
            // broadcast reads per lane instead of a scan over the whole row per pixel.  Rows with more sites than the row's
            // bytes hold (WP / 4: a dense region and open space in one row) take several passes, each over the next WP / 4
            // sites, and keep the minimum.
            {
                const uint32_t* ubb = updown + ((size_t)g * nb + b) * W;
                const uint32_t* cbb = cb + (size_t)b * W;
                const uint32_t ri = (uint32_t)(r0 + i);
                constexpr int KMAX = WP / 4;                           // sites per pass: the row's own LDS bytes
                constexpr uint32_t NOSITE = (uint32_t)EDT_G_INF << 16; // padding: farther than any real site, no overflow
                const uint4* tr4 = reinterpret_cast<const uint4*>(tr);
                const int nblk = (W + 63) >> 6;                        // <= 64: one lane per block
                auto cand = [](const uint32_t sv, const int x) {
                    const int dx = x - (int)(sv & 0xFFFFu), gq = (int)(sv >> 16);
                    return gq * gq + dx * dx;
                };
                int K = KMAX;                                          // known after the first pass
                for (int k0 = 0; k0 < K; k0 += KMAX) {                 // KMAX sites at a time (a pass per LDS-full of them)
                    int seen = 0;
                    for (int x0 = 0; x0 < W; x0 += 256) {              // four chunks of columns per round trip
                        uint32_t wv[4], uv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int x = x0 + 64 * u + lane;
                            wv[u] = x < W ? cbb[x] : 0u;
                            uv[u] = x < W ? ubb[x] : 0x7FFF7FFFu;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int x = x0 + 64 * u + lane;
                            const uint32_t wl = wv[u] >> ri, wh = wv[u] << (31u - ri);
                            uint32_t gg = min(ri + (uv[u] & 0xFFFFu), (31u - ri) + (uv[u] >> 16));
                            gg = min(gg, (uint32_t)(__ffs((int)wl) - 1));          // 0xFFFFFFFF when no bit at / below the row
                            gg = min(gg, wh ? (uint32_t)__clz((int)wh) : 0xFFFFFFFFu);
                            const bool has = x < W && gg < (uint32_t)EDT_G_INF;
                            const unsigned long long hm = __ballot(has);
                            const int si = seen + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u)) - k0;
                            if (has && si >= 0 && si < KMAX) tr[si] = (uint32_t)x | gg << 16;
                            seen += __popcll(hm);
                        }
                    }
                    K = seen;
                    open_sites = K;
                    const int Kp = min(K - k0, KMAX);                  // sites of this pass (0: a row without any)
                    if (lane < 4 && Kp + lane < KMAX) tr[Kp + lane] = NOSITE;       // whole quads of sites are read below
                    wave_lds_sync();
                    const int nq = (Kp + 3) >> 2;
                    int arg = 0;
                    {
                        const int xb = 64 * lane;
                        int best = INT32_MAX;
                        for (int q = 0; q < nq; ++q) {
                            const uint4 sv = tr4[q];
                            const int v0 = cand(sv.x, xb), v1 = cand(sv.y, xb), v2 = cand(sv.z, xb), v3 = cand(sv.w, xb);
                            if (v0 < best) { best = v0; arg = 4 * q; }          // strictly smaller, left to right: the leftmost
                            if (v1 < best) { best = v1; arg = 4 * q + 1; }      // of equally near sites
                            if (v2 < best) { best = v2; arg = 4 * q + 2; }
                            if (v3 < best) { best = v3; arg = 4 * q + 3; }
                        }
                    }
                    if (k0 > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the earlier pass's stores, read back below
                    for (int bl = 0; bl < nblk; ++bl) {
                        const int qlo = __builtin_amdgcn_readlane(arg, bl) >> 2;
                        const int qhi = bl + 1 < nblk ? __builtin_amdgcn_readlane(arg, bl + 1) >> 2 : nq - 1;
                        const int x = 64 * bl + lane;
                        int best = INT32_MAX;
                        for (int q = qlo; q <= qhi && q < nq; ++q) {   // (no site at all: nothing to read)
                            const uint4 sv = tr4[q];
                            best = min(min(best, cand(sv.x, x)), min(cand(sv.y, x), min(cand(sv.z, x), cand(sv.w, x))));
                        }
                        if (x < W) {
                            if (k0 > 0) best = min(best, (int)__hip_atomic_load(out + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            out[x] = best >= EDT_F_INF ? INT32_MAX : best;
                        }
                    }
                    wave_lds_sync();
                }
                return;
            }
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            // Transposition of one stretch through LDS.  The registers are first re-paired so that a dword holds two ADJACENT
            // pixels (pixel 2m | pixel 2m + 1 << 16): the lane's 16 pixels are then 32 contiguous bytes (two 16-byte writes,
            // swapped in every other group of four lanes so that eight consecutive lanes cover all banks once) and the four
            // pixels of a 16-byte global store are ONE 8-byte read -- half the LDS read traffic of reading packed registers
            // back as 16 bytes and keeping one half of every dword.
            uint32_t R[8];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                R[m] = __builtin_amdgcn_perm(P[t][2 * m + 1], P[t][2 * m], 0x05040100u);       // pixels 2m, 2m + 1
                R[m + 4] = __builtin_amdgcn_perm(P[t][2 * m + 1], P[t][2 * m], 0x07060302u);   // pixels 8 + 2m, 9 + 2m
            }
            uint32_t* wb = tr + 8 * lane;
            const int sw = (lane >> 2) & 1;
            *reinterpret_cast<uint4*>(wb + 4 * sw) = make_uint4(R[0], R[1], R[2], R[3]);
            *reinterpret_cast<uint4*>(wb + 4 * (1 - sw)) = make_uint4(R[4], R[5], R[6], R[7]);
            wave_lds_sync();
            // piece 64 k + lane = pixels 4 (64 k + lane) ..: owner lane 16 k + lane / 4, quarter lane % 4 of its 32 bytes
            const uint32_t* rb = tr + 8 * (lane >> 2) + 4 * (((lane >> 1) & 1) ^ ((lane >> 4) & 1)) + 2 * (lane & 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint2 qv = *reinterpret_cast<const uint2*>(rb + 128 * k);
                const int4 v = make_int4((int)(qv.x & 0xFFFFu), (int)(qv.x >> 16), (int)(qv.y & 0xFFFFu), (int)(qv.y >> 16));
                const int xg = 1024 * t + 4 * (64 * k + lane);
                if (FULL || xg + 3 < W) {
                    if (FULL || (((uintptr_t)(out + xg)) & 15) == 0) EDT_STORE4(out + xg, v);
                    else { out[xg] = v.x; out[xg + 1] = v.y; out[xg + 2] = v.z; out[xg + 3] = v.w; }
                } else {
                    if (xg < W) out[xg] = v.x;
                    if (xg + 1 < W) out[xg + 1] = v.y;
                    if (xg + 2 < W) out[xg + 2] = v.z;
                }
            }
            wave_lds_sync();
        }
    };

    // ---- the scheduler of one wavefront: produce when the slot is free (producers), else claim a row, else wait ----
    volatile uint32_t* v_p1 = p1_cnt;
    volatile uint32_t* v_next = row_next;
    volatile uint32_t* v_done = row_done;
    int pn = 0, cn = 0, spin = 0;
    while (cn < G || (producer && pn < G)) {
        bool did = false;
        if (producer && pn < G) {
            const int slot = pn & 1, k = pn >> 1;
            if (k == 0 || v_done[slot] >= (uint32_t)(RB * k)) {       // every row of the slot's previous group has been consumed
                EDT_LDS_ORDER();
                produce(pn);
                EDT_LDS_ORDER();
                if (lane == 0) atomicAdd(&p1_cnt[slot], 1u);
                ++pn; did = true;
            }
        }
        if (!did && cn < G) {
            const int slot = cn & 1, k = cn >> 1;
            if (v_p1[slot] >= (uint32_t)(NP * (k + 1))) {               // all eight producer parts of group cn are written
                int i = -1;
                if (lane == 0) {
                    uint32_t v = v_next[slot];
                    while (v < (uint32_t)(RB * (k + 1))) {
                        const uint32_t old = atomicCAS(&row_next[slot], v, v + 1u);
                        if (old == v) { i = (int)(v - (uint32_t)(RB * k)); break; }
                        v = old;
                    }
                }
                i = __builtin_amdgcn_readfirstlane(i);
                if (i < 0) ++cn;                                        // all rows of this group are claimed: on to the next
                else {
                    EDT_LDS_ORDER();
                    consume_row(cn, i);
                    EDT_LDS_ORDER();
                    if (lane == 0) atomicAdd(&row_done[slot], 1u);
                }
                did = true;
            }
        }
        if (did) spin = 0;
        else {
            __builtin_amdgcn_s_sleep(8);
            if (++spin > EDT_W_SPIN_LIMIT) {                            // never expected; a bounded wait cannot hang the GPU,
                if (lane == 0) *fault = 1;                              // and sc_ctx_synchronize reports the call as failed
                break;
            }
        }
    }
}

template <int TILES, bool FULL>
static int launch_band_wide(sc_ctx* ctx, const uint32_t* colbits, int W, int H, int nb, int batch, int32_t* d2) {
    const int nsb = (H + 15) / 16;
    const int ngroups = nsb * batch;
    const size_t lds = (size_t)2 * 16 * 1024 * TILES;
    {
        int r_ = sc_allow_big_lds(ctx, reinterpret_cast<const void*>(edt_band_wide_kernel<TILES, FULL>), (int)lds);
        if (r_ != SC_OK) return r_;
    }
    if (ctx->cu_count <= 0) {
        hipDeviceProp_t prop;
        ctx->cu_count = (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    // one workgroup (16 wavefronts) per CU, each with a strided set of row groups
    const int nwg = min(ngroups, ctx->cu_count);
    int tk = ctx->edt_open_token;   // opened in front of the updown launch (-1: timing off)
    ctx->edt_open_token = -1;
    if (!ctx->edt_fault.p) {
        int r_ = sc_scratch_reserve(ctx, &ctx->edt_fault, sizeof(int32_t));
        if (r_ != SC_OK) return r_;
        SC_HIP(ctx, hipMemsetAsync(ctx->edt_fault.p, 0, sizeof(int32_t), ctx->stream));
    }
    hipLaunchKernelGGL((edt_band_wide_kernel<TILES, FULL>), dim3((unsigned)nwg), dim3(1024), lds, ctx->stream, colbits,
                       (const uint32_t*)ctx->updown.p, W, H, nb, nsb, ngroups, d2, (int32_t*)ctx->edt_fault.p);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

template <int TILES>
static int launch_band_wide_t(sc_ctx* ctx, const uint32_t* colbits, int W, int H, int nb, int batch, int32_t* d2) {
    return W == 1024 * TILES ? launch_band_wide<TILES, true>(ctx, colbits, W, H, nb, batch, d2)
                             : launch_band_wide<TILES, false>(ctx, colbits, W, H, nb, batch, d2);
}

// Open space at widths up to 1024 (a hint between launches, per device): [0] a row of edt_band_k16_kernel<.., false> took the
// 32-bit fallback; [1] a row of its OPEN build took the site search.  sc_ctx_synchronize reads and clears them: a context
// that has seen [0] runs the OPEN build (with edt_updown_kernel in front) until a launch of it leaves [1] clear.
__device__ int32_t g_edt_open[2];

// One row of open space by the site search, for edt_band_k16_kernel's OPEN build: the row's sites -- obstacle columns with
// their exact vertical distance, from the band's column word and its up / down word -- compacted in column order into 512
// words of LDS (SA: the wave's transposition buffer, SB: the first KiB of its first row's bytes), then the monotone
// nearest-site rule (see edt_band_wide_kernel) with a lane per 16 pixels: lane l finds the nearest site of its first pixel
// among all K, and its 16 pixels look only between that site and lane l + 1's.  Returns K; writes nothing when K > 508.
template <bool FULL>
__device__ __forceinline__ int edt_k16_site_row(const uint32_t* __restrict__ cbrow, const uint32_t* __restrict__ ubrow, const int W, const int i,
                                                const int lane, uint32_t* const SA, uint32_t* const SB, int32_t* __restrict__ out) {
    constexpr uint32_t NOSITE = (uint32_t)EDT_G_INF << 16;
    const int nvalid = W - 16 * lane;
    const uint32_t* ubb = ubrow + 16 * lane;
    const uint32_t* cbb = cbrow + 16 * lane;
    const uint32_t ri = (uint32_t)i;
    auto site = [&](const int s) -> uint32_t& { return s < 256 ? SA[s] : SB[s - 256]; };
    auto gdist16 = [&](const int j) -> uint32_t {              // exact vertical distance of pixel j of this lane (INF: none)
        if (!FULL && j >= nvalid) return EDT_G_INF;
        const uint32_t w = cbb[j], ud = ubb[j];
        const uint32_t wl = w >> ri, wh = w << (31u - ri);
        uint32_t gg = min(ri + (ud & 0xFFFFu), (31u - ri) + (ud >> 16));
        gg = min(gg, (uint32_t)(__ffs((int)wl) - 1));
        gg = min(gg, wh ? (uint32_t)__clz((int)wh) : 0xFFFFFFFFu);
        return min(gg, (uint32_t)EDT_G_INF);
    };
    int cnt = 0;
    for (int j = 0; j < 16; ++j) cnt += gdist16(j) < (uint32_t)EDT_G_INF;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    const int K = __builtin_amdgcn_readlane(incl, 63);
    if (K > 508) return K;
    int pos = incl - cnt;
    for (int j = 0; j < 16; ++j) {
        const uint32_t gg = gdist16(j);
        if (gg < (uint32_t)EDT_G_INF) site(pos++) = (uint32_t)(16 * lane + j) | gg << 16;
    }
    if (lane < 4) site(K + lane) = NOSITE;                     // whole quads of sites are read below
    wave_lds_sync();
    auto cand = [](const uint32_t sv, const int x) {
        const int dx = x - (int)(sv & 0xFFFFu), gq = (int)(sv >> 16);
        return gq * gq + dx * dx;
    };
    const int nq = (K + 3) >> 2;
    int arg = 0;
    {
        const int xb = 16 * lane;
        int best = INT32_MAX;
        for (int q = 0; q < nq; ++q) {
            const uint4 sv = *reinterpret_cast<const uint4*>(q < 64 ? SA + 4 * q : SB + 4 * (q - 64));
            const int v0 = cand(sv.x, xb), v1 = cand(sv.y, xb), v2 = cand(sv.z, xb), v3 = cand(sv.w, xb);
            if (v0 < best) { best = v0; arg = 4 * q; }          // strictly smaller, left to right: the leftmost of
            if (v1 < best) { best = v1; arg = 4 * q + 1; }      // equally near sites
            if (v2 < best) { best = v2; arg = 4 * q + 2; }
            if (v3 < best) { best = v3; arg = 4 * q + 3; }
        }
    }
    const int nxt = (int)wave_rol1((uint32_t)arg);             // lane l + 1's site (lane 63: lane 0's, not used)
    const int lo = arg, hi = lane == 63 ? K - 1 : nxt;
    int trip = hi - lo + 1;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) trip = max(trip, __shfl_xor(trip, o));
    int best[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) best[j] = INT32_MAX;
    if (K > 0)
        for (int t = 0; t < trip; ++t) {
            const uint32_t sv = site(min(lo + t, hi));         // past its own range a lane repeats its last site
#pragma unroll
            for (int j = 0; j < 16; ++j) best[j] = min(best[j], cand(sv, 16 * lane + j));
        }
#pragma unroll
    for (int j = 0; j < 16; ++j) best[j] = best[j] >= EDT_F_INF ? INT32_MAX : best[j];
    // out through LDS (the sites are no longer needed), half the row at a time: a lane's 16 values are 64 bytes, the row's
    // lines want 16 bytes per lane from consecutive lanes -- 4 full-line stores instead of 16 that touch a dword of 64 lines each
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        wave_lds_sync();
        if ((lane >> 5) == p) {
            uint4* d = reinterpret_cast<uint4*>(((lane & 31) < 16 ? SA : SB) + (lane & 15) * 16);
            d[0] = make_uint4(best[0], best[1], best[2], best[3]);
            d[1] = make_uint4(best[4], best[5], best[6], best[7]);
            d[2] = make_uint4(best[8], best[9], best[10], best[11]);
            d[3] = make_uint4(best[12], best[13], best[14], best[15]);
        }
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = 4 * lane + 256 * k;                       // word of this half row
            const uint4 v = *reinterpret_cast<const uint4*>((k == 0 ? SA : SB) + 4 * lane);
            const int x = 512 * p + idx;
            if (FULL || x + 3 < W) {
                if (FULL || (((uintptr_t)(out + x)) & 15) == 0) EDT_STORE4(out + x, make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w));
                else { out[x] = (int)v.x; out[x + 1] = (int)v.y; out[x + 2] = (int)v.z; out[x + 3] = (int)v.w; }
            } else {
                if (x < W) out[x] = (int)v.x;
                if (x + 1 < W) out[x + 1] = (int)v.y;
                if (x + 2 < W) out[x + 2] = (int)v.z;
            }
        }
    }
    wave_lds_sync();
    return K;
}

// ---- edt_band_k16_kernel: rows of 513 .. 1024 pixels (the headline width) -------------------------------------
// edt_band_g8_kernel<16> with three changes (measured on the wide-row kernel first, whose LDS traffic was its bound):
//  * the cascade step is add + min3 (two packed instructions per register instead of three, see pk_min3_f16bits);
//  * the vertical pass is one packed multiply-add per row and direction (gu = free ? gu + 1 : 0 is (gu + 1) * free), the
//    32 rows in two halves so that the running values stay in registers, and pairs of lanes exchange their 2 x 2 bytes so
//    that every lane stores ONE dword per two rows instead of two 2-byte pieces each;
//  * the transposition re-pairs the registers so that a dword holds two ADJACENT pixels: a 16-byte global store is then ONE
//    8-byte LDS read (half the read traffic of reading packed registers back and keeping one half of every dword).
template <bool FULL, bool OPEN>
__global__ void __launch_bounds__(512, 8)
edt_band_k16_kernel(const uint32_t* __restrict__ colbits, int W, int H, int nb, int32_t* __restrict__ d2, const uint32_t* __restrict__ udg) {
    constexpr int WAVES = 8, WP = 1024;
    constexpr uint32_t GC2 = EDT_W_GCAP | (EDT_W_GCAP << 16);
    constexpr uint32_t EDGE = 0x7BFF7BFFu;          // beyond the row ends: above every value, below the f16 NaN patterns
    constexpr int UDCAP = 0x7000;                   // "no obstacle" for up / dn: still a non-negative finite f16 pattern after + 32
    extern __shared__ uint32_t smem[];
    uint8_t* g8 = reinterpret_cast<uint8_t*>(smem);            // [32][WP] vertical distances, clamped at EDT_W_GCAP
    uint32_t* trs = smem + 32 * WP / 4;                        // [WAVES][256] transposition buffers (half of what a wave needs)
    const unsigned nwg = gridDim.x, per = nwg >> 3;
    const unsigned vid = blockIdx.x < (per << 3) ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
    const int b = (int)(vid % (unsigned)nb), g = (int)(vid / (unsigned)nb);
    const uint32_t* cb = colbits + (size_t)g * nb * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // ---- phase 1: two adjacent columns per thread -> 32 rows of 2 distance bytes ----
    {
        const int q = threadIdx.x;
        uint32_t w[2], wu[2][4], wd[2][4];
        int up[2], dn[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int x = 2 * q + c;
            const bool in = FULL || x < W;
            w[c] = in ? cb[(size_t)b * W + x] : 0u;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                wu[c][t] = (in && b - 1 - t >= 0) ? cb[(size_t)(b - 1 - t) * W + x] : 0u;
                wd[c][t] = (in && b + 1 + t < nb) ? cb[(size_t)(b + 1 + t) * W + x] : 0u;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int x = 2 * q + c;
            up[c] = EDT_G_INF; dn[c] = EDT_G_INF;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (up[c] == EDT_G_INF && wu[c][t]) up[c] = (t + 1) * 32 - (31 - __clz((int)wu[c][t]));
                if (dn[c] == EDT_G_INF && wd[c][t]) dn[c] = (t + 1) * 32 + (__ffs((int)wd[c][t]) - 1) - 31;
            }
            if (FULL || x < W) {
                for (int base = b - 5; base >= 0 && up[c] == EDT_G_INF; base -= 4) {
                    uint32_t ww[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) ww[t] = base - t >= 0 ? cb[(size_t)(base - t) * W + x] : 0u;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (up[c] == EDT_G_INF && ww[t]) up[c] = (b - (base - t)) * 32 - (31 - __clz((int)ww[t]));
                }
                for (int base = b + 5; base < nb && dn[c] == EDT_G_INF; base += 4) {
                    uint32_t ww[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) ww[t] = base + t < nb ? cb[(size_t)(base + t) * W + x] : 0u;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (dn[c] == EDT_G_INF && ww[t]) dn[c] = ((base + t) - b) * 32 + (__ffs((int)ww[t]) - 1) - 31;
                }
            }
            up[c] = min(up[c], UDCAP); dn[c] = min(dn[c], UDCAP);
        }
        // free01 of row i: 1 per half whose cell is free (rows 0..15 of both columns from nwA, 16..31 from nwB)
        const uint32_t nw0 = ~w[0], nw1 = ~w[1];
        const uint32_t nwA = (nw0 & 0xFFFFu) | (nw1 << 16), nwB = (nw0 >> 16) | (nw1 & 0xFFFF0000u);
        const uint32_t psel = (lane & 1) ? 0x03020706u : 0x05040100u;
        uint8_t* dst = g8 + (size_t)(lane & 1) * WP + 2 * (q & ~1);
        uint32_t GU[16];
        // top-down over all 32 rows, keeping rows 16..31; bottom-up over them; the same for rows 0..15 (their top-down run
        // again from the band's top: 16 multiply-adds instead of 16 more registers; the free bits are extracted again on the
        // way up for the same reason: the kernel has to stay within 64 VGPRs for its eight wavefronts per SIMD)
        uint32_t gu = (uint32_t)(up[0] - 1) | ((uint32_t)(up[1] - 1) << 16);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t f = (nwA >> i) & 0x00010001u;
            gu = pk_mad_u16(gu, f, f);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t f = (nwB >> i) & 0x00010001u;
            gu = pk_mad_u16(gu, f, f);
            GU[i] = gu;
        }
        uint32_t gd = (uint32_t)(dn[0] - 1) | ((uint32_t)(dn[1] - 1) << 16);
#pragma unroll
        for (int half = 1; half >= 0; --half) {
            uint32_t nwH = half ? nwB : nwA, nwT = nwA;
            asm("" : "+v"(nwH), "+v"(nwT));   // fresh values to the compiler: it would otherwise keep all 32 extracted words alive (spills)
            if (half == 0) {
                gu = (uint32_t)(up[0] - 1) | ((uint32_t)(up[1] - 1) << 16);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t f = (nwT >> i) & 0x00010001u;
                    gu = pk_mad_u16(gu, f, f);
                    GU[i] = gu;
                }
            }
#pragma unroll
            for (int i = 14; i >= 0; i -= 2) {
                const uint32_t f1 = (nwH >> (i + 1)) & 0x00010001u, f0 = (nwH >> i) & 0x00010001u;
                gd = pk_mad_u16(gd, f1, f1);
                const uint32_t g1 = pk_min3_f16bits(GU[i + 1], gd, GC2);
                gd = pk_mad_u16(gd, f0, f0);
                const uint32_t g0 = pk_min3_f16bits(GU[i], gd, GC2);
                const uint32_t own = __builtin_amdgcn_perm(g1, g0, 0x06040200u);   // bytes: row i (col 0, col 1), row i + 1 (col 0, col 1)
                const uint32_t oth = (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xF, 0xF, true);
                *reinterpret_cast<uint32_t*>(dst + (size_t)(16 * half + i) * WP) = __builtin_amdgcn_perm(oth, own, psel);
            }
        }
    }
    __syncthreads();

    const int y0 = b * 32;
    const int nrows = min(32, H - y0);
    const int nvalid = W - 16 * lane;
    // a wave's transposition buffer: 1 KiB of its own (lanes 0..31) plus the first KiB of the g8 row it processes first
    // (lanes 32..63), dead once the wave has pulled that row into registers
    uint32_t* tr = trs + (size_t)wave * 256;
    uint32_t* trb = smem + (size_t)wave * (WP / 4) - 256;      // trb[8 * lane + r] with lane >= 32 lands in row `wave`
    int hint = 4;
    int open_rows = 0, open_sites = 0;   // OPEN: rows in a row that ended in the site search, and the sites of the last one
    if (OPEN) {
        // the band's obstacle columns (an obstacle anywhere in the column: the same for all its rows), counted once: what
        // the site search would cost decides how many steps the cascade gets first (48 + 3 K / 4: from timings of maps of
        // 2e-5 .. 1e-3 obstacle density -- fewer steps help the sparsest and cost the ones in between)
        const uint32_t* ubb = udg + ((size_t)g * nb + b) * W + 16 * lane;
        const uint32_t* cbb = cb + (size_t)b * W + 16 * lane;
        int cnt = 0;
        for (int j = 0; j < 16; ++j)
            if (FULL || 16 * lane + j < W) {
                const uint32_t ud = ubb[j];
                cnt += cbb[j] != 0u || (ud & 0xFFFFu) != (uint32_t)EDT_G_INF || (ud >> 16) != (uint32_t)EDT_G_INF;
            }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
        open_sites = cnt;
    }
    for (int i = wave; i < nrows; i += WAVES) {
        int32_t* out = d2 + ((size_t)g * H + y0 + i) * W;
        uint32_t P[8];
        {
            const uint4 v = *reinterpret_cast<const uint4*>(g8 + (size_t)i * WP + 16 * lane);
            const uint32_t dw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t sel = 0x0C000C00u | (uint32_t)(j % 4) | ((uint32_t)(4 + j % 4) << 16);
                const uint32_t tt = __builtin_amdgcn_perm(dw[(j + 8) / 4], dw[j / 4], sel);
                P[j] = pk_mul_lo(tt, tt);      // <= 177^2: pixels past the row end carry 177 ("no obstacle")
            }
        }
        auto cascade_step = [&](int it) {
            const uint32_t c = (uint32_t)(2 * it - 1) * 0x00010001u;
            uint32_t T[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) T[j] = pk_add_wrap(P[j], c);
            const uint32_t lo = wave_ror1(T[7]), hi = wave_rol1(T[0]);
            const uint32_t below = lane == 0 ? EDGE : lo, above = lane == 63 ? EDGE : hi;
            const uint32_t L0 = __builtin_amdgcn_alignbit(T[7], below, 16);   // T of the left neighbours of pixels (0, 8)
            const uint32_t RL = __builtin_amdgcn_alignbit(above, T[0], 16);   // T of the right neighbours of pixels (7, 15)
#pragma unroll
            for (int j = 0; j < 8; ++j) P[j] = pk_min3_f16bits(P[j], j ? T[j - 1] : L0, j < 7 ? T[j + 1] : RL);
        };
        bool saturated = false;
        // OPEN: after a row of open space with few sites the wave's next rows go to the site search at once (every eighth
        // tries the cascade again), as in edt_band_wide_kernel; and the cascade's first try is as long as the search would
        // cost -- a row with more sites than the search holds comes back and runs the remaining steps
        bool direct = OPEN && (open_rows & 7) != 0 && open_sites <= 128;
        if (direct) saturated = true;
        int it = 1, next_chk = max(2, hint - 1);
        int it_lim = OPEN ? min(EDT_W_ITMAX, 48 + ((3 * open_sites) >> 2)) : EDT_W_ITMAX;
        bool row_done = false;
        for (int attempt = 0;; ++attempt) {
        for (; !saturated && it <= it_lim; ++it) {
            cascade_step(it);
            const bool last = it == it_lim;
            if (!last && it < next_chk) continue;
            next_chk = it + 1 + (it >= 12 ? (it >> 3) : 0);
            uint32_t m;
            if (FULL) {
                m = P[0];
#pragma unroll
                for (int j = 1; j < 8; ++j) m = pk_max(m, P[j]);
                m = max(m & 0xFFFFu, m >> 16);
            } else {
                m = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < nvalid) m = max(m, P[j] & 0xFFFFu);
                    if (j + 8 < nvalid) m = max(m, P[j] >> 16);
                }
            }
            const uint32_t thr = (uint32_t)(it + 1) * (uint32_t)(it + 1);
            if (__ballot(m > thr) == 0) break;
            if (last) { saturated = true; break; }
        }
        if (!OPEN || !saturated || attempt == 1) break;
        // OPEN, not settled (or not tried): the site search
        ++open_rows;
        open_sites = edt_k16_site_row<FULL>(cb + (size_t)b * W, udg + ((size_t)g * nb + b) * W, W, i, lane, tr, smem + (size_t)wave * (WP / 4), out);
        if (lane == 0) g_edt_open[1] = 1;
        if (open_sites <= 508) { row_done = true; break; }
        // more sites than the search's LDS holds: the rest of the 175 steps after all
        it = direct ? 1 : it + 1;
        next_chk = it;
        it_lim = EDT_W_ITMAX;
        saturated = false;
        direct = false;
        }
        if (OPEN && row_done) continue;
        if (!saturated) { hint = it; open_rows = 0; }
        if (!saturated) {
            uint32_t R[8];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                R[m] = __builtin_amdgcn_perm(P[2 * m + 1], P[2 * m], 0x05040100u);       // pixels 2m, 2m + 1
                R[m + 4] = __builtin_amdgcn_perm(P[2 * m + 1], P[2 * m], 0x07060302u);   // pixels 8 + 2m, 9 + 2m
            }
            uint32_t* wb = (lane < 32 ? tr : trb) + 8 * lane;
            const int sw = (lane >> 2) & 1;
            *reinterpret_cast<uint4*>(wb + 4 * sw) = make_uint4(R[0], R[1], R[2], R[3]);
            *reinterpret_cast<uint4*>(wb + 4 * (1 - sw)) = make_uint4(R[4], R[5], R[6], R[7]);
            wave_lds_sync();
            // piece 64 k + lane = pixels 4 (64 k + lane) ..: owner lane 16 k + lane / 4, quarter lane % 4 of its 32 bytes
            const int roff = 8 * (lane >> 2) + 4 * (((lane >> 1) & 1) ^ ((lane >> 4) & 1)) + 2 * (lane & 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint2 qv = *reinterpret_cast<const uint2*>((k < 2 ? tr : trb) + roff + 128 * k);
                const int4 v = make_int4((int)(qv.x & 0xFFFFu), (int)(qv.x >> 16), (int)(qv.y & 0xFFFFu), (int)(qv.y >> 16));
                const int x = 4 * (64 * k + lane);
                if (FULL || x + 3 < W) {
                    if (FULL || (((uintptr_t)(out + x)) & 15) == 0) EDT_STORE4(out + x, v);
                    else { out[x] = v.x; out[x + 1] = v.y; out[x + 2] = v.z; out[x + 3] = v.w; }
                } else {
                    if (x < W) out[x] = v.x;
                    if (x + 1 < W) out[x + 1] = v.y;
                    if (x + 2 < W) out[x + 2] = v.z;
                }
            }
            wave_lds_sync();
        } else {
            // ---- the packed cascade cannot settle this row (open space: some distance beyond 175 columns) ----
            if (!OPEN && lane == 0) g_edt_open[0] = 1;             // the site search is in this kernel's OPEN twin: tell the host
            // 32-bit cascade with exact distances (OPEN: rows with more sites than the search's LDS holds)
            uint32_t V[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int x = 16 * lane + j;
                uint32_t gg = EDT_G_INF;
                if (x < W) gg = edt_gdist_global(cb, W, nb, b, x, i);
                V[j] = gg * gg;
            }
            for (int it2 = 1; it2 < WP; ++it2) {
                const uint32_t c = (uint32_t)(2 * it2 - 1);
                const uint32_t below = from_lane_below(V[15], (uint32_t)EDT_F_INF);
                const uint32_t above = from_lane_above(V[0], (uint32_t)EDT_F_INF);
                uint32_t prev = below;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t cur = V[j];
                    const uint32_t nxt = j < 15 ? V[j + 1] : above;
                    V[j] = min(cur, min(prev, nxt) + c);
                    prev = cur;
                }
                if ((it2 & 7) == 0) {
                    uint32_t m = 0;
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (FULL || j < nvalid) m = max(m, V[j]);
                    const uint32_t thr = (uint32_t)(it2 + 1) * (uint32_t)(it2 + 1);
                    if (__ballot(m > thr && (m < (uint32_t)EDT_F_INF || it2 + 1 < W)) == 0) break;
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int x = 16 * lane + j;
                if (x < W) out[x] = V[j] >= (uint32_t)EDT_F_INF ? INT32_MAX : (int)V[j];
            }
        }
    }
}

template <bool FULL>
static int launch_band_k16(sc_ctx* ctx, const uint32_t* colbits, int W, int H, int nb, int batch, int32_t* d2) {
    const size_t lds = (size_t)32 * 1024 + 8 * 1024;
    ctx->edt_k16_launched = true;
    if (ctx->edt_open_mode) {
        // open space seen on this context (sc_ctx_synchronize): the up / down words first, then the build with the site search
        int r_ = sc_scratch_reserve(ctx, &ctx->updown, (size_t)batch * nb * W * sizeof(uint32_t));
        if (r_ != SC_OK) return r_;
        int tk = ctx->edt_chain_token >= 0 ? sc_time_chain(ctx, ctx->edt_chain_token, SC_K_EDT_BAND) : sc_time_begin(ctx, SC_K_EDT_BAND);
        ctx->edt_chain_token = -1;
        launch_updown(ctx, colbits, W, nb, batch, (uint32_t*)ctx->updown.p);
        hipLaunchKernelGGL((edt_band_k16_kernel<FULL, true>), dim3((unsigned)(nb * batch)), dim3(512), lds, ctx->stream, colbits, W, H, nb, d2,
                           (const uint32_t*)ctx->updown.p);
        ctx->edt_open_launched = true;
        sc_time_end(ctx, tk);
        SC_HIP(ctx, hipGetLastError());
        return SC_OK;
    }
    int tk = ctx->edt_chain_token >= 0 ? sc_time_chain(ctx, ctx->edt_chain_token, SC_K_EDT_BAND) : sc_time_begin(ctx, SC_K_EDT_BAND);
    ctx->edt_chain_token = -1;
    hipLaunchKernelGGL((edt_band_k16_kernel<FULL, false>), dim3((unsigned)(nb * batch)), dim3(512), lds, ctx->stream, colbits, W, H, nb, d2,
                       (const uint32_t*)nullptr);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

// sc_ctx_synchronize: open space seen / still there?  (Only after a launch of edt_band_k16_kernel on this context.)
int sc_edt_open_mode_update(sc_ctx* ctx) {
    if (!ctx->edt_k16_launched) return SC_OK;
    int32_t h[2] = {0, 0};
    SC_HIP(ctx, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_edt_open), sizeof(h)));
    if (h[0] || h[1]) {
        const int32_t z[2] = {0, 0};
        SC_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_edt_open), z, sizeof(z)));
    }
    if (h[0]) ctx->edt_open_mode = true;
    else if (ctx->edt_open_mode && ctx->edt_open_launched && !h[1]) ctx->edt_open_mode = false;
    ctx->edt_k16_launched = false;
    ctx->edt_open_launched = false;
    return SC_OK;
}

template <int PPL, bool FULL>
static int launch_band_g8(sc_ctx* ctx, const uint32_t* colbits, int W, int H, int nb, int batch, int32_t* d2) {
    constexpr int WP = 64 * PPL;
    const size_t lds = PPL == 16 ? (size_t)32 * WP + 8 * 1024 : (size_t)32 * WP + (size_t)8 * 64 * (PPL / 2 + 1) * sizeof(uint32_t);
    {
        int r_ = sc_allow_big_lds(ctx, reinterpret_cast<const void*>(edt_band_g8_kernel<PPL, FULL>), 160 * 1024);
        if (r_ != SC_OK) return r_;
    }
    int tk = ctx->edt_chain_token >= 0 ? sc_time_chain(ctx, ctx->edt_chain_token, SC_K_EDT_BAND) : sc_time_begin(ctx, SC_K_EDT_BAND);
    ctx->edt_chain_token = -1;
    hipLaunchKernelGGL((edt_band_g8_kernel<PPL, FULL>), dim3((unsigned)(nb * batch)), dim3(512), lds, ctx->stream,
                       colbits, W, H, nb, d2, 1, 0, (int32_t*)nullptr, (const int32_t*)nullptr);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

template <int PPL>
static int launch_band_g8_ppl(sc_ctx* ctx, const uint32_t* colbits, int W, int H, int nb, int batch, int32_t* d2) {
    return W == 64 * PPL ? launch_band_g8<PPL, true>(ctx, colbits, W, H, nb, batch, d2)
                         : launch_band_g8<PPL, false>(ctx, colbits, W, H, nb, batch, d2);
}

template <int PPL, bool FULL>
static int launch_band(sc_ctx* ctx, const uint32_t* colbits, int W, int H, int nb, int batch, int32_t* d2, const int32_t* flags = nullptr) {
    constexpr int G = PPL < 16 ? PPL : 16;
    constexpr int WP = 64 * PPL;
    constexpr int TRN = (64 * (G + 1) > WP / 2 ? 64 * (G + 1) : WP / 2);
    const size_t lds = (size_t)(2 * (WP + 64) + 4 * TRN) * sizeof(uint32_t);
    {
        int r_ = sc_allow_big_lds(ctx, reinterpret_cast<const void*>(edt_band_kernel<PPL, FULL>), 160 * 1024);
        if (r_ != SC_OK) return r_;
    }
    int tk = ctx->edt_chain_token >= 0 ? sc_time_chain(ctx, ctx->edt_chain_token, SC_K_EDT_BAND) : sc_time_begin(ctx, SC_K_EDT_BAND);
    ctx->edt_chain_token = -1;
    hipLaunchKernelGGL((edt_band_kernel<PPL, FULL>), dim3((unsigned)(nb * batch)), dim3(256), lds, ctx->stream,
                       colbits, W, H, nb, d2, flags);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

template <int PPL>
static int launch_band_ppl(sc_ctx* ctx, const uint32_t* colbits, int W, int H, int nb, int batch, int32_t* d2, const int32_t* flags = nullptr) {
    return W == 64 * PPL ? launch_band<PPL, true>(ctx, colbits, W, H, nb, batch, d2, flags)
                         : launch_band<PPL, false>(ctx, colbits, W, H, nb, batch, d2, flags);
}

// Rows wider than 1024: windows of 1024 columns through the fast kernel; flags[grid][band] != 0 where it gave up.
// `halo` columns on either side of a window's core bound the cascade steps (= the largest distance) a row may need;
// `only`: process just the bands an earlier pass flagged.
static int launch_band_g8_tiled(sc_ctx* ctx, const uint32_t* colbits, int W, int H, int nb, int batch, int32_t* d2, int halo, int32_t* flags,
                                const int32_t* only, bool first) {
    const int tiles = (W + (1024 - 2 * halo) - 1) / (1024 - 2 * halo);
    const size_t lds = (size_t)32 * 1024 + 8 * 1024;
    {
        int r_ = sc_allow_big_lds(ctx, reinterpret_cast<const void*>(edt_band_g8_kernel<16, true, true>), 160 * 1024);
        if (r_ != SC_OK) return r_;
    }
    SC_HIP(ctx, hipMemsetAsync(flags, 0, (size_t)batch * nb * sizeof(int32_t), ctx->stream));
    int tk = -1;
    if (first) {
        tk = ctx->edt_chain_token >= 0 ? sc_time_chain(ctx, ctx->edt_chain_token, SC_K_EDT_BAND) : sc_time_begin(ctx, SC_K_EDT_BAND);
        ctx->edt_chain_token = -1;
    } else tk = sc_time_begin(ctx, SC_K_EDT_BAND);
    hipLaunchKernelGGL((edt_band_g8_kernel<16, true, true>), dim3((unsigned)((size_t)nb * batch * tiles)), dim3(512), lds, ctx->stream,
                       colbits, W, H, nb, d2, tiles, halo, flags, only);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

int sc_launch_edt(sc_ctx* ctx, const uint8_t* occ, int W, int H, int batch, int32_t* d2) {
    const int nb = (H + 31) / 32;
    int r = sc_scratch_reserve(ctx, &ctx->colbits, (size_t)batch * nb * W * sizeof(uint32_t));
    if (r != SC_OK) return r;
    uint32_t* colbits = (uint32_t*)ctx->colbits.p;
    if (batch > 65535 || nb > 65535) return SC_ERR_INVALID;

    static const char* only = getenv("SC_EDT_ONLY");  // debug/profiling: run a single kernel of the pair
    const bool skip_colbits = only && only[0] == 'b', skip_band = only && only[0] == 'c';
    int tk = sc_time_begin(ctx, SC_K_EDT_COLBITS);
    if (skip_colbits) {
    } else if (W % 16 == 0 && ((uintptr_t)occ & 15) == 0) {
        hipLaunchKernelGGL(edt_colbits_kernel<true>, dim3((W + 1023) / 1024, nb, batch), dim3(64, 4), 0, ctx->stream,
                           occ, W, H, nb, colbits);
    } else if (W % 4 == 0 && ((uintptr_t)occ & 3) == 0) {
        hipLaunchKernelGGL(edt_colbits_kernel<false>, dim3((W + 1023) / 1024, nb, batch), dim3(64, 4), 0, ctx->stream,
                           occ, W, H, nb, colbits);
    } else {
        dim3 grid((W + 255) / 256, nb, batch);
        hipLaunchKernelGGL(edt_colbits_generic_kernel, grid, dim3(256), 0, ctx->stream, occ, W, H, nb, colbits);
    }
    SC_HIP(ctx, hipGetLastError());
    if (skip_band) { sc_time_end(ctx, tk); return SC_OK; }
    ctx->edt_chain_token = tk;
    // pixels per lane: smallest power of two with 64 * PPL >= W
    if (W <= 128) return launch_band_ppl<2>(ctx, colbits, W, H, nb, batch, d2);
    if (W <= 256) return launch_band_ppl<4>(ctx, colbits, W, H, nb, batch, d2);
#ifndef EDT_NO_G8
    if (W > 256 && W <= 512) return launch_band_g8_ppl<8>(ctx, colbits, W, H, nb, batch, d2);
    if (W > 512 && W <= 1024) {
        static const bool old16 = getenv("SC_EDT_OLD16") && atoi(getenv("SC_EDT_OLD16"));   // A/B against edt_band_g8_kernel<16>
        if (old16) return launch_band_g8_ppl<16>(ctx, colbits, W, H, nb, batch, d2);
        return W == 1024 ? launch_band_k16<true>(ctx, colbits, W, H, nb, batch, d2) : launch_band_k16<false>(ctx, colbits, W, H, nb, batch, d2);
    }
#endif
    if (W <= 512) return launch_band_ppl<8>(ctx, colbits, W, H, nb, batch, d2);
    if (W <= 1024) return launch_band_ppl<16>(ctx, colbits, W, H, nb, batch, d2);
#ifndef EDT_NO_WIDE
    // rows of up to 4096 pixels: whole rows in registers, 16 rows per workgroup
    if ((size_t)((H + 15) / 16) * batch <= 0x7FFFFFFFu) {
        r = sc_scratch_reserve(ctx, &ctx->updown, (size_t)batch * nb * W * sizeof(uint32_t));
        if (r != SC_OK) return r;
        colbits = (uint32_t*)ctx->colbits.p;
        // timing: the look-up table of the band kernel counts as band time (the colbits bracket ends here)
        if (ctx->edt_chain_token >= 0) ctx->edt_open_token = sc_time_chain(ctx, ctx->edt_chain_token, SC_K_EDT_BAND);
        ctx->edt_chain_token = -1;
        launch_updown(ctx, colbits, W, nb, batch, (uint32_t*)ctx->updown.p);
        if (W <= 2048) return launch_band_wide_t<2>(ctx, colbits, W, H, nb, batch, d2);
        if (W <= 3072) return launch_band_wide_t<3>(ctx, colbits, W, H, nb, batch, d2);
        if (W <= 4096) return launch_band_wide_t<4>(ctx, colbits, W, H, nb, batch, d2);
    }
#endif
    // wider rows still: 1024-column windows through the fast kernel first, then the whole-row kernel for the grids (if
    // any) in which some row did not settle within the window's halo
    const int32_t* flags = nullptr;
#ifndef EDT_NO_G8
    {
        // Pass 1: as few windows as the minimal halo allows, with the halo as wide as that number of windows leaves room
        // for (whole lanes of 16 pixels): at W = 4096 five windows either way, with a halo of 96 columns instead of 32.
        // Pass 2, only for the bands in which some row needed more steps than that (block-type maps: a tenth of the bands
        // at 4096^2): windows with a halo of 256 columns, which covers every distance the packed cascade can represent.
        // Pass 3: the whole-row kernel for what is still open (grids so sparse that distances exceed 255).
        r = sc_scratch_reserve(ctx, &ctx->edt_flags, (size_t)2 * batch * nb * sizeof(int32_t));
        if (r != SC_OK) return r;
        int32_t* f1 = (int32_t*)ctx->edt_flags.p;
        int32_t* f2 = f1 + (size_t)batch * nb;
        const int tiles = (W + (1024 - 2 * EDT_TILE_HALO_MIN) - 1) / (1024 - 2 * EDT_TILE_HALO_MIN);
        int halo = ((1024 - (W + tiles - 1) / tiles) / 2) / 16 * 16;
        if (halo < EDT_TILE_HALO_MIN) halo = EDT_TILE_HALO_MIN;
        r = launch_band_g8_tiled(ctx, colbits, W, H, nb, batch, d2, halo, f1, nullptr, true);
        if (r != SC_OK) return r;
        flags = f1;
        if (halo < 256) {
            r = launch_band_g8_tiled(ctx, colbits, W, H, nb, batch, d2, 256, f2, f1, false);
            if (r != SC_OK) return r;
            flags = f2;
        }
    }
#endif
    if (W <= 2048) return launch_band_ppl<32>(ctx, colbits, W, H, nb, batch, d2, flags);
    if (W <= 4096) return launch_band_ppl<64>(ctx, colbits, W, H, nb, batch, d2, flags);
    return launch_band_ppl<128>(ctx, colbits, W, H, nb, batch, d2, flags);
}

extern "C" int sc_edt_u8_i32(sc_ctx* ctx, const uint8_t* occ, int W, int H, int batch, int32_t* d2) {
    if (!ctx || !occ || !d2 || W <= 0 || H <= 0 || batch <= 0 || W > SC_MAX_DIM || H > SC_MAX_DIM) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    return sc_launch_edt(ctx, occ, W, H, batch, d2);
}

// ---- legal-move mask --------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
moves_kernel(const int32_t* __restrict__ d2, int W, int Hall, int H, int32_t rmin, uint8_t* __restrict__ moves) {
    // Hall = G * H rows: G grids of H rows stacked; a cell's neighbours are looked up inside its own grid only
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int yall = blockIdx.y;
    if (x >= W) return;
    const int y = yall % H;
    const int32_t* r1 = d2 + (size_t)yall * W;
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
    moves[(size_t)yall * W + x] = (uint8_t)m;
}

// The same, four cells per thread (W a multiple of 4, rows 16-byte aligned): the three rows come in as 16-byte loads plus
// the two cells beside them, the four masks leave as one dword -- the one-cell form moves 5 B/cell at 0.8 TB/s (a byte per
// thread and store), this one is bound by the d2 reads.
__global__ void __launch_bounds__(256)
moves4_kernel(const int32_t* __restrict__ d2, int W, int Hall, int H, int32_t rmin, uint8_t* __restrict__ moves) {
    const int x = 4 * (blockIdx.x * 256 + threadIdx.x);
    const int yall = blockIdx.y;
    if (x >= W) return;
    const int y = yall % H;
    const int32_t* r1 = d2 + (size_t)yall * W;
    uint32_t t[3] = {0u, 0u, 0u};          // bit i + 1 of t[j + 1]: cell (x + i, y + j) is traversable, i = -1 .. 4
#pragma unroll
    for (int j = -1; j <= 1; ++j) {
        const int yy = y + j;
        if (yy < 0 || yy >= H) continue;
        const int32_t* r = r1 + (ptrdiff_t)j * W;
        const int4 v = *reinterpret_cast<const int4*>(r + x);
        const int32_t l = x > 0 ? r[x - 1] : 0, q = x + 4 < W ? r[x + 4] : 0;      // 0 < rmin: outside the grid is blocked
        t[j + 1] = (uint32_t)(l >= rmin) | (uint32_t)(v.x >= rmin) << 1 | (uint32_t)(v.y >= rmin) << 2 | (uint32_t)(v.z >= rmin) << 3 |
                   (uint32_t)(v.w >= rmin) << 4 | (uint32_t)(q >= rmin) << 5;
    }
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t a = t[0] >> i, c = t[1] >> i, b = t[2] >> i;   // rows y - 1, y, y + 1; bit 0 = x - 1, bit 1 = x, bit 2 = x + 1
        uint32_t m = 0;
        if (c & 2u) {
            // d: dx = {1,-1,0,0,1,-1,1,-1}, dy = {0,0,1,-1,1,1,-1,-1}
            const uint32_t E = (c >> 2) & 1u, Wt = c & 1u, S = (b >> 1) & 1u, N = (a >> 1) & 1u;
            m = E | Wt << 1 | S << 2 | N << 3 | (((b >> 2) & 1u) & E & S) << 4 | ((b & 1u) & Wt & S) << 5 | (((a >> 2) & 1u) & E & N) << 6 |
                ((a & 1u) & Wt & N) << 7;
        }
        out |= m << (8 * i);
    }
    *reinterpret_cast<uint32_t*>(moves + (size_t)yall * W + x) = out;
}

// ... and four rows per thread as well (H a multiple of 4: a block of rows stays inside its grid): six rows read for four
// written instead of twelve.
__global__ void __launch_bounds__(256)
moves4x4_kernel(const int32_t* __restrict__ d2, int W, int Hall, int H, int32_t rmin, uint8_t* __restrict__ moves) {
    const int x = 4 * (blockIdx.x * 256 + threadIdx.x);
    const int yall0 = 4 * blockIdx.y;
    if (x >= W) return;
    const int y0 = yall0 % H;
    const int32_t* r1 = d2 + (size_t)yall0 * W;
    uint32_t t[6];                          // rows y0 - 1 .. y0 + 4; bit i + 1: cell x + i is traversable, i = -1 .. 4
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int yy = y0 + j - 1;
        t[j] = 0u;
        if (yy < 0 || yy >= H) continue;
        const int32_t* r = r1 + (ptrdiff_t)(j - 1) * W;
        const int4 v = *reinterpret_cast<const int4*>(r + x);
        const int32_t l = x > 0 ? r[x - 1] : 0, q = x + 4 < W ? r[x + 4] : 0;
        t[j] = (uint32_t)(l >= rmin) | (uint32_t)(v.x >= rmin) << 1 | (uint32_t)(v.y >= rmin) << 2 | (uint32_t)(v.z >= rmin) << 3 |
               (uint32_t)(v.w >= rmin) << 4 | (uint32_t)(q >= rmin) << 5;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t a = t[k] >> i, c = t[k + 1] >> i, b = t[k + 2] >> i;
            uint32_t m = 0;
            if (c & 2u) {
                const uint32_t E = (c >> 2) & 1u, Wt = c & 1u, S = (b >> 1) & 1u, N = (a >> 1) & 1u;
                m = E | Wt << 1 | S << 2 | N << 3 | (((b >> 2) & 1u) & E & S) << 4 | ((b & 1u) & Wt & S) << 5 | (((a >> 2) & 1u) & E & N) << 6 |
                    ((a & 1u) & Wt & N) << 7;
            }
            out |= m << (8 * i);
        }
        *reinterpret_cast<uint32_t*>(moves + (size_t)(yall0 + k) * W + x) = out;
    }
}

int sc_launch_moves(sc_ctx* ctx, const int32_t* d2, int W, int Hall, int H, int32_t r2, uint8_t* moves) {
    int32_t rmin = r2 > 1 ? r2 : 1;
    int tk = sc_time_begin(ctx, SC_K_MOVES);
    if (W % 4 == 0 && H % 4 == 0 && ((uintptr_t)d2 & 15) == 0 && ((uintptr_t)moves & 3) == 0)
        hipLaunchKernelGGL(moves4x4_kernel, dim3((W / 4 + 255) / 256, Hall / 4), dim3(256), 0, ctx->stream, d2, W, Hall, H, rmin, moves);
    else if (W % 4 == 0 && ((uintptr_t)d2 & 15) == 0 && ((uintptr_t)moves & 3) == 0)
        hipLaunchKernelGGL(moves4_kernel, dim3((W / 4 + 255) / 256, Hall), dim3(256), 0, ctx->stream, d2, W, Hall, H, rmin, moves);
    else
    hipLaunchKernelGGL(moves_kernel, dim3((W + 255) / 256, Hall), dim3(256), 0, ctx->stream, d2, W, Hall, H, rmin, moves);
    sc_time_end(ctx, tk);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

extern "C" int sc_moves_i32_u8(sc_ctx* ctx, const int32_t* d2, int W, int H, int32_t r2_clear, uint8_t* moves) {
    if (!ctx || !d2 || !moves || W <= 0 || H <= 0 || W > SC_MAX_DIM || H > SC_MAX_DIM) return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    return sc_launch_moves(ctx, d2, W, H, H, r2_clear, moves);
}
