// sc_internal.h -- context, scratch and launch-timing plumbing shared by the HIP sources.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <set>
#include <vector>

#include "../../include/sea_current_hip.h"

struct sc_scratch {
    void* p = nullptr;
    size_t bytes = 0;
};

struct sc_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t wait_ev = nullptr;   // blocking-sync event: host waits sleep instead of spinning (one host thread per context in flight)
    char err[256] = {0};
    // timing
    int timing = 0;
    struct pending_ev { int kid; hipEvent_t a, b; bool shared_a = false; };
    std::vector<pending_ev> pending;
    std::vector<hipEvent_t> ev_pool;
    double t_ms[SC_K_COUNT] = {0};
    int64_t t_n[SC_K_COUNT] = {0};
    // scratch (grow-only)
    sc_scratch colbits;     // EDT: uint32 [batch][nb][W]
    sc_scratch updown;      // EDT, rows wider than 1024: uint32 [batch][nb][W], rows to the nearest obstacle in the bands above / below
    sc_scratch edt_fault;   // EDT, rows wider than 1024: int32 [1], set by a wavefront whose bounded wait ran out (read by sc_ctx_synchronize)
    sc_scratch edt_flags;   // EDT, rows wider than 1024: int32 [2][batch][bands], != 0 where a windowed pass gave a band up
    sc_scratch moves;       // A*: uint8 [H][W]
    sc_scratch gslots;      // A*: uint32 [S][g cells] (4 x 4-cell tiles)
    sc_scratch closed;      // A*: uint32 [S][bitmap words] closed set, one bit per cell (32 x 16-cell tiles)
    sc_scratch buckets;     // A*: uint32 [S][32][cap]
    sc_scratch qstats;      // A*: int32 expanded[Q] | queue order[Q] | overflow list[Q]
    sc_scratch actr;        // A*: int32 [8] queue / overflow counters of a launch, [4] = sticky overflow flag
    sc_scratch bez_tang;    // Bezier: double [P][n_max][2] tangents
    sc_scratch bez_gl;      // Bezier: 32 Gauss-Legendre nodes + 32 weights
    sc_scratch bez_seginfo; // resample: int4 [S] (first sample, last sample, spline, segment in spline)
    sc_scratch cheb_a;      // chebfit: double [rows][degree + 1], the [T | y] matrices of a batch
    sc_scratch fmt_nbr;     // FMT*: uint16 [n][256] samples in range of every sample | int32 count [n] | int32 overflow
    sc_scratch gather_msg;  // gather: this rank's message, every rank's messages, local offsets
    sc_scratch staging[8];  // _host wrappers
    int astar_cap = 1 << 16;          // ring entries per bucket (power of two)
    size_t astar_slot_budget = (size_t)96 << 30;  // bytes of g + bitmap + ring scratch this context may take (SC_ASTAR_SLOT_GB), further bounded by what the device has free; 4096^2: 96 GiB = 1966 slots measured best (48: -34 %, 160: -17 %)
    int last_Q = 0;
    int edt_chain_token = -1;       // timing: colbits' end event doubles as band's start event
    int edt_open_token = -1;        // timing: band bracket already opened (wide rows: in front of the updown launch)
    void* comm = nullptr;           // ncclComm_t of sc_allgather_paths
    bool comm_owned = false;
    int comm_ranks = 0, comm_rank = 0;
    int64_t gather_bytes = 0;       // bytes every rank received in the last gather
    std::set<const void*> big_lds_done;   // kernels whose dynamic-LDS limit this context has raised on its device
    int cu_count = 0;               // compute units of the device (0: not asked yet)
    bool edt_open_mode = false;     // EDT, rows of 513 .. 1024 pixels: open space seen -> the band kernel's build with the site search
    bool edt_k16_launched = false, edt_open_launched = false;   // since the last sc_ctx_synchronize
    int astar_waves = 0;            // wavefronts an A* launch keeps resident (0: not yet determined)
    int astar_dual = -1;   // queries the two-wavefront A* kernel keeps resident (-1: not asked yet, 0: off)
    int astar_dual_lat = -1;   // the same for its latency build (larger LDS ring: fewer per CU)
};

#define SC_HIP(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call, \
                     hipGetErrorString(e_));                                               \
            return e_ == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_HIP;                  \
        }                                                                                  \
    } while (0)

int sc_scratch_reserve(sc_ctx* ctx, sc_scratch* s, size_t bytes);

// RAII-free timing bracket: t = sc_time_begin(ctx, kid); launch...; sc_time_end(ctx, t)
int sc_time_begin(sc_ctx* ctx, int kid);
void sc_time_end(sc_ctx* ctx, int token);
int sc_time_chain(sc_ctx* ctx, int token, int kid);

// kernels' host launchers (defined in the respective .hip files)
int sc_launch_edt(sc_ctx* ctx, const uint8_t* occ, int W, int H, int batch, int32_t* d2);
int sc_launch_moves(sc_ctx* ctx, const int32_t* d2, int W, int Hall, int H, int32_t r2, uint8_t* moves);   // Hall / H grids of H rows, stacked

// Raise a kernel's dynamic-LDS limit (> 64 KiB needs hipFuncSetAttribute, which is per DEVICE): once per context, i.e.
// once per device and host thread -- a process-wide flag would leave a second GPU's copy of the kernel at the default
// and be written by several threads at once.
int sc_allow_big_lds(sc_ctx* ctx, const void* kernel, int bytes);

// wait for everything enqueued on the context's stream without burning a host core
int sc_stream_wait(sc_ctx* ctx);
int sc_edt_open_mode_update(sc_ctx* ctx);

#ifdef __HIPCC__
__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in issue order; only the compiler must not reorder them.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
#endif
