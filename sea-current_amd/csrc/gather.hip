// gather.hip -- all-gather of result paths across the GPUs of a node over RCCL (xGMI), in the C ABI.
//
// SURVEY.md 8e / BASELINE.json: independent start-goal queries shard over the ranks in contiguous blocks (rank r owns
// queries [r Q / R, (r + 1) Q / R) up to rounding, sc_rank_range), every rank plans its block on its own replica of the
// grid, and the only exchange is the gather of the results.  The reference has no counterpart (its only transport is
// the ZMQ REP loop of examples/zmq_test.cpp:18-22).
//
// Only sum(len) cells travel, not Q x Lmax: a pack kernel turns a rank's fixed-stride results into one message
//   [ cells used | truncated | len[Qmax] | cost[Qmax] | status[Qmax] | cells back to back, cap_cells ]
// (lengths -> exclusive scan -> compact copy, one wavefront per path), ONE ncclAllGather moves every rank's message to
// every rank over the direct xGMI links, and an unpack kernel rebuilds, in query order, len / cost / status, the CSR
// offsets, the compact cells and -- if asked for -- the fixed-stride [Q][Lmax] parity layout.  Everything is enqueued
// on the context's stream: no host synchronisation, no host-side sizes (the message has a fixed capacity; a rank whose
// paths do not fit says so in the message and the flag reaches the caller's device word).
//
// RCCL is bound at run time (dlopen): a process that never gathers does not need it, and a process that already holds an
// RCCL (PyTorch loads its own copy) shares that one instead of mapping a second.
#include "sc_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct rccl_api {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};
rccl_api* rccl() {
    static rccl_api api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;          // one that is already mapped (e.g. PyTorch's)
        for (const char* n : names)
            if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) {
            api.get_unique_id = (decltype(api.get_unique_id))dlsym(api.lib, "ncclGetUniqueId");
            api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(api.lib, "ncclCommInitRank");
            api.comm_destroy = (decltype(api.comm_destroy))dlsym(api.lib, "ncclCommDestroy");
            api.all_gather = (decltype(api.all_gather))dlsym(api.lib, "ncclAllGather");
            api.error_string = (decltype(api.error_string))dlsym(api.lib, "ncclGetErrorString");
        }
    }
    return api.lib && api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_gather ? &api : nullptr;
}
int rccl_fail(sc_ctx* ctx, const char* what, ncclResult_t r) {
    rccl_api* a = rccl();
    snprintf(ctx->err, sizeof(ctx->err), "%s -> %s", what, a && a->error_string ? a->error_string(r) : "RCCL error");
    return SC_ERR_HIP;
}
}  // namespace

extern "C" void sc_rank_range(int Q, int world, int rank, int* q0, int* q1) {
    const int base = Q / world, rem = Q % world;
    const int a = rank * base + (rank < rem ? rank : rem);
    if (q0) *q0 = a;
    if (q1) *q1 = a + base + (rank < rem ? 1 : 0);
}

extern "C" int sc_comm_unique_id(void* id128) {
    rccl_api* a = rccl();
    if (!a || !id128) return a ? SC_ERR_INVALID : SC_ERR_NO_DEVICE;
    ncclUniqueId id;
    if (a->get_unique_id(&id) != ncclSuccess) return SC_ERR_HIP;
    memcpy(id128, &id, sizeof(id));
    return SC_OK;
}

extern "C" int sc_comm_init(sc_ctx* ctx, const void* id128, int nranks, int rank) {
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return SC_ERR_INVALID;
    rccl_api* a = rccl();
    if (!a) { snprintf(ctx->err, sizeof(ctx->err), "librccl.so could not be loaded"); return SC_ERR_NO_DEVICE; }
    SC_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->comm && ctx->comm_owned) (void)a->comm_destroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr; ctx->comm_owned = false; ctx->comm_ranks = 0; ctx->comm_rank = 0;   // nothing dangles if the init below fails
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t r = a->comm_init_rank(&c, nranks, id, rank);
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclCommInitRank", r);
    ctx->comm = c; ctx->comm_owned = true; ctx->comm_ranks = nranks; ctx->comm_rank = rank;
    return SC_OK;
}

extern "C" int sc_comm_adopt(sc_ctx* ctx, void* nccl_comm, int nranks, int rank) {
    if (!ctx || !nccl_comm || nranks < 1 || rank < 0 || rank >= nranks) return SC_ERR_INVALID;
    if (!rccl()) { snprintf(ctx->err, sizeof(ctx->err), "librccl.so could not be loaded"); return SC_ERR_NO_DEVICE; }
    if (ctx->comm && ctx->comm_owned) (void)rccl()->comm_destroy((ncclComm_t)ctx->comm);
    ctx->comm = nccl_comm; ctx->comm_owned = false; ctx->comm_ranks = nranks; ctx->comm_rank = rank;
    return SC_OK;
}

extern "C" int sc_comm_destroy(sc_ctx* ctx) {
    if (!ctx) return SC_ERR_INVALID;
    if (ctx->comm && ctx->comm_owned && rccl()) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        (void)rccl()->comm_destroy((ncclComm_t)ctx->comm);
    }
    ctx->comm = nullptr; ctx->comm_owned = false; ctx->comm_ranks = 0; ctx->comm_rank = 0;
    return SC_OK;
}

// message layout (int32 words)
#define MSG_USED 0
#define MSG_TRUNC 1
#define MSG_META 2

// Exclusive scan of the effective lengths (len where status == SC_Q_OK, else 0), one workgroup: eff_off [n + 1] (int64).
// Source of (len, status): local arrays (rank = -1) or the gathered messages (query q of rank r at message word
// MSG_META + (q - q0(r))).  Also the length / cost / status in query order when gathering.
__global__ void __launch_bounds__(1024)
gather_scan_kernel(const int32_t* __restrict__ len, const int32_t* __restrict__ status, const int32_t* __restrict__ msgs, int stride, int qmax,
                   int world, int Q, int Lmax, int64_t* __restrict__ off, int32_t* __restrict__ len_all, int32_t* __restrict__ cost_all,
                   int32_t* __restrict__ status_all) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int chunk = (Q + 1023) / 1024;
    const int q_lo = min(Q, tid * chunk), q_hi = min(Q, q_lo + chunk);
    const int base = Q / world, rem = Q % world;
    auto fetch = [&](int q, int& ln, int& cs, int& st) {
        if (!msgs) { ln = len[q]; st = status[q]; cs = 0; return; }
        // rank of query q under sc_rank_range: the first `rem` ranks own base + 1 queries
        const int big = rem * (base + 1);
        const int r = q < big ? q / (base + 1) : rem + (base ? (q - big) / base : 0);
        const int q0 = r * base + min(r, rem);
        const int32_t* m = msgs + (size_t)r * stride + MSG_META + (q - q0);
        ln = m[0]; cs = m[qmax]; st = m[2 * qmax];
    };
    long long sum = 0;
    for (int q = q_lo; q < q_hi; ++q) {
        int ln, cs, st;
        fetch(q, ln, cs, st);
        sum += st == SC_Q_OK ? min(max(ln, 0), Lmax) : 0;
    }
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const long long v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    long long run = part[tid] - sum;
    for (int q = q_lo; q < q_hi; ++q) {
        int ln, cs, st;
        fetch(q, ln, cs, st);
        off[q] = run;
        run += st == SC_Q_OK ? min(max(ln, 0), Lmax) : 0;
        if (msgs) { len_all[q] = ln; cost_all[q] = cs; status_all[q] = st; }
    }
    if (tid == 1023) off[Q] = part[1023];
}

// one wavefront per local query: its cells to the message (if they fit), the meta words, the header
__global__ void __launch_bounds__(256)
gather_pack_kernel(const int32_t* __restrict__ path, const int32_t* __restrict__ len, const int32_t* __restrict__ cost,
                   const int32_t* __restrict__ status, int Ql, int qmax, int Lmax, int cap_cells, const int64_t* __restrict__ off,
                   int32_t* __restrict__ msg) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= qmax) return;
    int32_t* meta = msg + MSG_META;
    if (q == 0 && lane == 0) {   // header, also of a rank that owns no query at all (Q_total < world)
        const long long total = off[Ql];
        msg[MSG_USED] = (int32_t)(total < cap_cells ? total : cap_cells);
        msg[MSG_TRUNC] = total > cap_cells ? 1 : 0;
    }
    if (q >= Ql) {   // padding of a rank with one query fewer
        if (lane == 0) { meta[q] = 0; meta[qmax + q] = -1; meta[2 * qmax + q] = SC_Q_NO_PATH; }
        return;
    }
    const int ln = len[q], st = status[q];
    const long long o = off[q];
    if (lane == 0) { meta[q] = ln; meta[qmax + q] = cost[q]; meta[2 * qmax + q] = st; }
    const int n = st == SC_Q_OK ? min(max(ln, 0), Lmax) : 0;
    int32_t* dst = msg + MSG_META + 3 * (size_t)qmax;
    const int32_t* src = path + (size_t)q * Lmax;
    for (int i = lane; i < n; i += 64)
        if (o + i < cap_cells) dst[o + i] = src[i];
}

// one wavefront per global query: its cells from its rank's message to the compact array and / or the fixed-stride rows
__global__ void __launch_bounds__(256)
gather_unpack_kernel(const int32_t* __restrict__ msgs, int stride, int qmax, int world, int Q, int Lmax, int cap_cells,
                     const int64_t* __restrict__ off_all, const int32_t* __restrict__ len_all, const int32_t* __restrict__ status_all,
                     int32_t* __restrict__ cells_all, long long cells_capacity, int32_t* __restrict__ path_all, int32_t* __restrict__ flag) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= Q) return;
    const int base = Q / world, rem = Q % world, big = rem * (base + 1);
    const int r = q < big ? q / (base + 1) : rem + (base ? (q - big) / base : 0);
    const int q0 = r * base + min(r, rem);
    const int32_t* m = msgs + (size_t)r * stride;
    if (q == q0 && lane == 0 && m[MSG_TRUNC]) atomicOr(flag, 1);
    if (q == 0 && lane == 0 && cells_all && off_all[Q] > cells_capacity) atomicOr(flag, 2);   // the caller's compact array is too small
    const int n = status_all[q] == SC_Q_OK ? min(max(len_all[q], 0), Lmax) : 0;
    const long long o_local = off_all[q] - off_all[q0];          // cells of this rank in front of query q
    const long long o = off_all[q];
    const int32_t* src = m + MSG_META + 3 * (size_t)qmax + o_local;
    for (int i = lane; i < n; i += 64) {
        const bool have = o_local + i < cap_cells;               // beyond the message's capacity: never sent
        const int32_t v = have ? src[i] : -1;
        if (cells_all && o + i < cells_capacity) cells_all[o + i] = v;
        if (path_all) path_all[(size_t)q * Lmax + i] = v;
    }
}

// Words of one rank's message.  The cell area starts after an even number of words and the whole message is an even
// number of words, so messages laid back to back keep 8-byte alignment for whatever follows them.
static size_t gather_stride(int Q_total, int world, int cap_cells) {
    const size_t qmax = (size_t)((Q_total + world - 1) / world);
    size_t w = (size_t)MSG_META + 3 * qmax + (size_t)cap_cells;
    return w + (w & 1);
}

extern "C" int64_t sc_gather_msg_words(int Q_total, int world, int cap_cells) {
    if (Q_total <= 0 || world < 1 || cap_cells <= 0) return 0;
    return (int64_t)gather_stride(Q_total, world, cap_cells);
}

// scratch of the gather: [ local offsets int64 [qmax + 1] | this rank's message | every rank's messages ]
static int gather_scratch(sc_ctx* ctx, int Q_total, int world, int cap_cells, bool with_msgs, int64_t** off_local, int32_t** msg, int32_t** msgs) {
    const size_t qmax = (size_t)((Q_total + world - 1) / world);
    const size_t stride = gather_stride(Q_total, world, cap_cells);
    const size_t off_bytes = (qmax + 2) / 2 * 2 * sizeof(int64_t);
    int r = sc_scratch_reserve(ctx, &ctx->gather_msg, off_bytes + stride * (size_t)(with_msgs ? world + 1 : 1) * sizeof(int32_t));
    if (r != SC_OK) return r;
    *off_local = (int64_t*)ctx->gather_msg.p;                   // 8-byte data first: aligned whatever the sizes
    *msg = (int32_t*)((char*)ctx->gather_msg.p + off_bytes);
    if (msgs) *msgs = *msg + stride;
    return SC_OK;
}

static int gather_pack_launch(sc_ctx* ctx, const int32_t* path, const int32_t* len, const int32_t* cost, const int32_t* status, int Q_local,
                              int qmax, int Lmax, int cap_cells, int64_t* off_local, int32_t* msg) {
    hipLaunchKernelGGL(gather_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, len, status, (const int32_t*)nullptr, 0, qmax, 1, Q_local > 0 ? Q_local : 0,
                       Lmax, off_local, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
    hipLaunchKernelGGL(gather_pack_kernel, dim3((qmax + 3) / 4), dim3(256), 0, ctx->stream, path, len, cost, status, Q_local, qmax, Lmax, cap_cells,
                       (const int64_t*)off_local, msg);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

static int gather_unpack_launch(sc_ctx* ctx, const int32_t* msgs, size_t stride, int qmax, int world, int Q_total, int Lmax, int cap_cells,
                                int32_t* len_all, int32_t* cost_all, int32_t* status_all, int64_t* offsets_all, int32_t* cells_all,
                                int64_t cells_capacity, int32_t* path_all, int32_t* truncated) {
    SC_HIP(ctx, hipMemsetAsync(truncated, 0, sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(gather_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const int32_t*)nullptr, (const int32_t*)nullptr, msgs,
                       (int)stride, qmax, world, Q_total, Lmax, offsets_all, len_all, cost_all, status_all);
    hipLaunchKernelGGL(gather_unpack_kernel, dim3((Q_total + 3) / 4), dim3(256), 0, ctx->stream, msgs, (int)stride, qmax, world, Q_total,
                       Lmax, cap_cells, (const int64_t*)offsets_all, (const int32_t*)len_all, (const int32_t*)status_all, cells_all,
                       (long long)cells_capacity, path_all, truncated);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
}

static bool gather_sizes_ok(int Q_total, int world, int Lmax, int cap_cells) {
    // message offsets are 32-bit words of an int-sized stride
    return Q_total > 0 && world >= 1 && Lmax > 0 && cap_cells > 0 && gather_stride(Q_total, world, cap_cells) <= (size_t)INT32_MAX;
}

// The two halves of sc_allgather_paths on their own (device pointers, enqueued, no host synchronisation): what a caller
// with a transport of its own -- or a test that plays several ranks on one GPU -- puts around the exchange.
extern "C" int sc_gather_pack(sc_ctx* ctx, const int32_t* path, const int32_t* len, const int32_t* cost, const int32_t* status, int Q_local,
                              int Q_total, int world, int rank, int Lmax, int cap_cells, int32_t* msg) {
    if (!ctx || !msg || Q_local < 0 || rank < 0 || rank >= world || !gather_sizes_ok(Q_total, world, Lmax, cap_cells) ||
        (Q_local > 0 && (!path || !len || !cost || !status)))
        return SC_ERR_INVALID;
    int q0, q1;
    sc_rank_range(Q_total, world, rank, &q0, &q1);
    if (q1 - q0 != Q_local) { snprintf(ctx->err, sizeof(ctx->err), "sc_gather_pack: rank %d of %d owns %d of %d queries, not %d", rank, world, q1 - q0, Q_total, Q_local); return SC_ERR_INVALID; }
    SC_HIP(ctx, hipSetDevice(ctx->device));
    int64_t* off_local; int32_t* own;
    int r = gather_scratch(ctx, Q_total, world, cap_cells, false, &off_local, &own, nullptr);
    if (r != SC_OK) return r;
    const int qmax = (Q_total + world - 1) / world;
    int tk = sc_time_begin(ctx, SC_K_GATHER);
    r = gather_pack_launch(ctx, path, len, cost, status, Q_local, qmax, Lmax, cap_cells, off_local, msg);
    sc_time_end(ctx, tk);
    return r;
}

extern "C" int sc_gather_unpack(sc_ctx* ctx, const int32_t* msgs, int world, int Q_total, int Lmax, int cap_cells, int32_t* len_all,
                                int32_t* cost_all, int32_t* status_all, int64_t* offsets_all, int32_t* cells_all, int64_t cells_capacity,
                                int32_t* path_all, int32_t* truncated) {
    if (!ctx || !msgs || !len_all || !cost_all || !status_all || !offsets_all || !truncated || !gather_sizes_ok(Q_total, world, Lmax, cap_cells) ||
        (cells_all && cells_capacity <= 0))
        return SC_ERR_INVALID;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const int qmax = (Q_total + world - 1) / world;
    int tk = sc_time_begin(ctx, SC_K_GATHER);
    int r = gather_unpack_launch(ctx, msgs, gather_stride(Q_total, world, cap_cells), qmax, world, Q_total, Lmax, cap_cells, len_all, cost_all,
                                 status_all, offsets_all, cells_all, cells_capacity, path_all, truncated);
    sc_time_end(ctx, tk);
    return r;
}

extern "C" int sc_allgather_paths(sc_ctx* ctx, const int32_t* path, const int32_t* len, const int32_t* cost, const int32_t* status,
                                  int Q_local, int Q_total, int Lmax, int cap_cells, int32_t* len_all, int32_t* cost_all, int32_t* status_all,
                                  int64_t* offsets_all, int32_t* cells_all, int64_t cells_capacity, int32_t* path_all, int32_t* truncated) {
    if (!ctx || !len_all || !cost_all || !status_all || !offsets_all || !truncated || Q_local < 0 ||
        (Q_local > 0 && (!path || !len || !cost || !status)) || (cells_all && cells_capacity <= 0))
        return SC_ERR_INVALID;
    if (!ctx->comm) { snprintf(ctx->err, sizeof(ctx->err), "sc_allgather_paths: no communicator (sc_comm_init / sc_comm_adopt)"); return SC_ERR_INVALID; }
    const int world = ctx->comm_ranks, rank = ctx->comm_rank;
    if (!gather_sizes_ok(Q_total, world, Lmax, cap_cells)) return SC_ERR_INVALID;
    int q0, q1;
    sc_rank_range(Q_total, world, rank, &q0, &q1);
    if (q1 - q0 != Q_local) { snprintf(ctx->err, sizeof(ctx->err), "sc_allgather_paths: rank %d of %d owns %d of %d queries, not %d", rank, world, q1 - q0, Q_total, Q_local); return SC_ERR_INVALID; }
    rccl_api* a = rccl();
    if (!a) return SC_ERR_NO_DEVICE;
    SC_HIP(ctx, hipSetDevice(ctx->device));
    const int qmax = (Q_total + world - 1) / world;
    const size_t stride = gather_stride(Q_total, world, cap_cells);
    int64_t* off_local; int32_t *msg, *msgs;
    int r = gather_scratch(ctx, Q_total, world, cap_cells, true, &off_local, &msg, &msgs);
    if (r != SC_OK) return r;
    int tk = sc_time_begin(ctx, SC_K_GATHER);
    r = gather_pack_launch(ctx, path, len, cost, status, Q_local, qmax, Lmax, cap_cells, off_local, msg);
    if (r != SC_OK) { sc_time_end(ctx, tk); return r; }
    const ncclResult_t nr = a->all_gather(msg, msgs, stride, ncclInt32, (ncclComm_t)ctx->comm, ctx->stream);
    if (nr != ncclSuccess) { sc_time_end(ctx, tk); return rccl_fail(ctx, "ncclAllGather", nr); }
    r = gather_unpack_launch(ctx, msgs, stride, qmax, world, Q_total, Lmax, cap_cells, len_all, cost_all, status_all, offsets_all, cells_all,
                             cells_capacity, path_all, truncated);
    sc_time_end(ctx, tk);
    if (r != SC_OK) return r;
    ctx->gather_bytes = (int64_t)stride * sizeof(int32_t) * world;
    return SC_OK;
}

extern "C" int sc_allgather_last_bytes(sc_ctx* ctx, int64_t* bytes) {
    if (!ctx || !bytes) return SC_ERR_INVALID;
    *bytes = ctx->gather_bytes;
    return SC_OK;
}
