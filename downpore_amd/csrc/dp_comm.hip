// libdownpore_hip.so — multi-GPU entry points of the C ABI: the scan is sharded by read, ranks own
// ascending contiguous read ranges, and after every round's scan the survivors (read id, hit count, segments) are
// all-gathered so that every GPU builds the identical index.  The exchange runs device to device on the context's own
// stream: counts first (variable sizes), then the payloads, concatenated in rank order = file order.
//   dp_comm_init        one process per GPU: an RCCL communicator over xGMI (librccl is loaded at run time, so a build of this
//                       library does not link against it - only its header is needed - and a process that never calls
//                       dp_comm_init never loads it);
//   dp_comm_init_local  one process driving several contexts (a Go host with one goroutine per GPU, or tests with several
//                       contexts on one GPU): every rank publishes device pointers, peers copy with hipMemcpyPeerAsync.
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <rccl/rccl.h>

#include "dp_common.h"

namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // an RCCL the process has loaded already (PyTorch brings its own next to its own HIP runtime) is the one that matches
        // the HIP runtime in use; otherwise the ROCm installation's
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (api.lib) break;
        }
        for (const char* n : names) {
            if (api.lib) break;
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!api.lib) {
            api.err = std::string("librccl not found: ") + dlerror();
            return;
        }
        auto sym = [&](const char* s) {
            void* p = dlsym(api.lib, s);
            if (!p) api.err = std::string("librccl lacks ") + s;
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    return &api;
}

// shared state of the ranks of a local (one-process) communicator
struct LocalGroup {
    std::mutex mu;
    std::condition_variable cv;
    int n = 0, refs = 0;
    uint64_t gen = 0;       // completed exchanges
    int arrived = 0, left = 0;
    bool failed = false;    // a rank left an exchange with an error: every wait of this group returns, now and later (sticky)
    // dp_allgather_blobs: host pointers published by the ranks
    struct BlobPub {
        const uint8_t* p = nullptr;
        uint64_t n = 0;
    };
    std::vector<BlobPub> blob;
    struct Pub {
        int device = 0;
        const void* counts = nullptr;   // device: uint64[2] = {survivors, segment ints of the survivors}
        const void* read = nullptr;     // device: uint32[survivors] read ids
        const void* nseeds = nullptr;   // device: uint32[survivors]
        const void* segs = nullptr;     // device: int32[segment ints]
        uint64_t n_surv = 0, n_ints = 0;
        hipEvent_t ready = nullptr;     // recorded on the publisher's stream once the buffers above are final
    };
    std::vector<Pub> pub;
};
}  // namespace

struct dp_comm {
    int n_ranks = 1, rank = 0;
    ncclComm_t nccl = nullptr;  // RCCL flavour
    LocalGroup* local = nullptr;  // one-process flavour
    bool dead = false;            // an exchange failed on this rank: the communicator was aborted and refuses further calls
    hipEvent_t ev = nullptr;
    // per-rank scratch on the owning context's device
    DevBuf d_cnt, d_pay, d_allpay, d_allsegs, d_blob;
    PinBuf h_cnt, h_out, h_blob, h_bsz;
    std::string err;
};

extern "C" int dp_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return DP_ERR_ARG;
    RcclApi* R = rccl_api();
    if (!R->err.empty()) return DP_ERR_STATE;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return DP_ERR_HIP;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id_out, &id, 128);
    return DP_OK;
}

extern "C" int dp_comm_init(dp_ctx* ctx, int n_ranks, int rank, const uint8_t* unique_id, dp_comm** out) {
    if (!ctx || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks || !unique_id) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_comm_init: bad arguments") : DP_ERR_ARG;
    *out = nullptr;
    RcclApi* R = rccl_api();
    if (!R->err.empty()) return dp_fail(ctx, DP_ERR_STATE, R->err.c_str());
    hipSetDevice(ctx->device);
    dp_comm* c = new dp_comm();
    c->n_ranks = n_ranks;
    c->rank = rank;
    ncclUniqueId id;
    memcpy(&id, unique_id, 128);
    const ncclResult_t r = R->CommInitRank(&c->nccl, n_ranks, id, rank);
    if (r != ncclSuccess) {
        std::string m = std::string("ncclCommInitRank: ") + R->GetErrorString(r);
        delete c;
        return dp_fail(ctx, DP_ERR_HIP, m.c_str());
    }
    hipEventCreateWithFlags(&c->ev, hipEventDisableTiming);
    *out = c;
    return DP_OK;
}

extern "C" int dp_comm_init_local(dp_ctx* const* ctxs, int n, dp_comm** out) {
    if (!ctxs || !out || n < 1) return DP_ERR_ARG;
    LocalGroup* g = new LocalGroup();
    g->n = n;
    g->refs = n;
    g->pub.resize((size_t)n);
    g->blob.resize((size_t)n);
    for (int r = 0; r < n; r++) {
        if (!ctxs[r]) {
            delete g;
            return DP_ERR_ARG;
        }
        dp_comm* c = new dp_comm();
        c->n_ranks = n;
        c->rank = r;
        c->local = g;
        hipSetDevice(ctxs[r]->device);
        hipEventCreateWithFlags(&c->ev, hipEventDisableTiming);
        hipEventCreateWithFlags(&g->pub[(size_t)r].ready, hipEventDisableTiming);
        g->pub[(size_t)r].device = ctxs[r]->device;
        out[r] = c;
    }
    // peers on different devices copy directly (xGMI) when the runtime allows it; otherwise hipMemcpyPeerAsync stages
    for (int a = 0; a < n; a++)
        for (int b = 0; b < n; b++)
            if (ctxs[a]->device != ctxs[b]->device) {
                int can = 0;
                hipDeviceCanAccessPeer(&can, ctxs[a]->device, ctxs[b]->device);
                if (can) {
                    hipSetDevice(ctxs[a]->device);
                    hipDeviceEnablePeerAccess(ctxs[b]->device, 0);  // (already enabled: harmless error)
                    (void)hipGetLastError();
                }
            }
    return DP_OK;
}

extern "C" void dp_comm_destroy(dp_comm* c) {
    if (!c) return;
    if (c->nccl) {  // (a communicator that failed mid-collective cannot be destroyed in the orderly way: that would wait for its peers)
        if (c->dead && rccl_api()->CommAbort) rccl_api()->CommAbort(c->nccl);
        else rccl_api()->CommDestroy(c->nccl);
    }
    if (c->local) {
        bool last;
        {
            std::lock_guard<std::mutex> lk(c->local->mu);
            last = --c->local->refs == 0;
        }
        if (last) {
            for (auto& p : c->local->pub)
                if (p.ready) hipEventDestroy(p.ready);
            delete c->local;
        }
    }
    if (c->ev) hipEventDestroy(c->ev);
    for (DevBuf* b : {&c->d_cnt, &c->d_pay, &c->d_allpay, &c->d_allsegs, &c->d_blob})
        if (b->p) dp_dev_free(b->p);
    for (PinBuf* b : {&c->h_cnt, &c->h_out, &c->h_blob, &c->h_bsz})
        if (b->p) hipHostFree(b->p);
    delete c;
}

// gathers read ids / hit counts of the survivors of the local scan (the compacted lists of dp_scan_reads) into one payload
__global__ void comm_pack_meta(const uint32_t* __restrict__ s_item, const uint32_t* __restrict__ s_count, uint32_t n, uint32_t lo,
                               uint32_t* __restrict__ read, uint32_t* __restrict__ nseeds) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    read[i] = s_item[i] + lo;
    nseeds[i] = s_count[i];
}

// the gathered survivors as the {read, hit count, segment offset} arrays dp_index_build_chunked works from (one workgroup: a
// round has a few thousand survivors, a few hundred thousand in the dense-seed regime)
__global__ __launch_bounds__(1024) void comm_install_survivors(const uint32_t* __restrict__ read, const uint32_t* __restrict__ nseeds, uint32_t n,
                                                              uint32_t* __restrict__ s_item, uint32_t* __restrict__ s_count,
                                                              uint64_t* __restrict__ s_off) {
    __shared__ unsigned long long sh[1024];
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    unsigned long long sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += 2ull * nseeds[i] + 1;
    sh[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        unsigned long long a = 0;
        if ((int)threadIdx.x >= d) a = sh[threadIdx.x - d];
        __syncthreads();
        sh[threadIdx.x] += a;
        __syncthreads();
    }
    unsigned long long pos = sh[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; i++) {
        s_item[i] = read[i];
        s_count[i] = nseeds[i];
        s_off[i] = pos;
        pos += 2ull * nseeds[i] + 1;
    }
}

static int comm_reserve(dp_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) ctx->retired_dev.push_back(b.p);
    b.p = nullptr;
    b.cap = 0;
    const size_t ncap = (bytes + bytes / 2 + 255) & ~(size_t)255;
    DP_HIP(dp_dev_malloc(&b.p, ncap));
    b.cap = ncap;
    return 0;
}
static int comm_pin(dp_ctx* ctx, PinBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) ctx->retired_pin.push_back(b.p);
    b.p = nullptr;
    b.cap = 0;
    const size_t ncap = (bytes + bytes / 2 + 4095) & ~(size_t)4095;
    DP_HIP(hipHostMalloc(&b.p, ncap, hipHostMallocDefault));
    b.cap = ncap;
    return 0;
}

// All-gather of the survivors of the last dp_scan_reads on `ctx` (which scanned this rank's read range; the `extra` items -
// the query windows - are scanned by every rank and are not exchanged).  Afterwards the context's device-resident scan
// output is [survivors of rank 0 | rank 1 | ... | this rank's extra items], i.e. what a single GPU scanning every read would
// hold, and `all` describes it the way dp_scan_reads would have (host copies included).
static int allgather_survivors_impl(dp_comm* c, dp_ctx* ctx, const dp_survivor_batch* local, dp_survivor_batch* all);

// The caller's own work of a round failed before it reached the exchange (or it gives up for any other reason): its peers
// must not wait for it.  Marks the communicator dead; peers inside or entering an exchange return DP_ERR_STATE (in-process
// group) or the error RCCL reports for an aborted communicator.
extern "C" void dp_comm_abort(dp_comm* c) {
    if (!c || c->dead) return;
    c->dead = true;
    if (c->local) {
        std::lock_guard<std::mutex> lk(c->local->mu);
        c->local->failed = true;
        c->local->cv.notify_all();
    }
    if (c->nccl && rccl_api()->CommAbort) {
        rccl_api()->CommAbort(c->nccl);
        c->nccl = nullptr;
    }
}

// A rank that leaves the exchange with an error must not leave its peers waiting for it: the in-process group is marked failed
// (both barriers give up, now and in every later call), an RCCL communicator is aborted (ncclCommAbort: the peers' pending
// collectives return an error instead of hanging) - a per-rank error stays a per-rank error code on every rank.
extern "C" int dp_allgather_survivors(dp_comm* c, dp_ctx* ctx, const dp_survivor_batch* local, dp_survivor_batch* all) {
    if (!c || !ctx || !local || !all) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_allgather_survivors: bad arguments") : DP_ERR_ARG;
    if (c->dead) return dp_fail(ctx, DP_ERR_STATE, "dp_allgather_survivors: the communicator failed in an earlier exchange");
    const int rc = allgather_survivors_impl(c, ctx, local, all);
    if (rc != DP_OK) dp_comm_abort(c);  // (marks the group failed AND aborts an RCCL communicator: the peers' collectives return)
    return rc;
}

static int allgather_survivors_impl(dp_comm* c, dp_ctx* ctx, const dp_survivor_batch* local, dp_survivor_batch* all) {
    hipSetDevice(ctx->device);
    if (const long fr = dp_tune("comm_fail_rank", -1); fr >= 0)  // test hook: this rank fails before it meets its peers
        if (fr == c->rank) return dp_fail(ctx, DP_ERR_STATE, "dp_allgather_survivors: injected failure (DP_COMM_FAIL_RANK)");
    const int N = c->n_ranks, me = c->rank;
    const uint32_t ns = local->n_survivors, ne = local->n_extra;
    // local layout of d_segs: survivors' segments first, then the extra items'
    uint64_t surv_ints = 0;
    if (ns) surv_ints = local->seg_off[ns - 1] + 2ull * local->n_seeds[ns - 1] + 1;
    const uint64_t extra_ints = local->n_segs - surv_ints;
    // ---- sizes
    if (comm_pin(ctx, c->h_cnt, (size_t)N * 16 + 64)) return DP_ERR_HIP;
    uint64_t* cnt = (uint64_t*)c->h_cnt.p;  // [N][2] = survivors, ints
    if (c->nccl) {
        RcclApi* R = rccl_api();
        if (comm_reserve(ctx, c->d_cnt, (size_t)(N + 1) * 16)) return DP_ERR_HIP;
        uint64_t mine[2] = {ns, surv_ints};
        uint64_t* d_mine = (uint64_t*)c->d_cnt.p + 2 * (size_t)N;
        DP_HIP(hipMemcpyAsync(d_mine, mine, 16, hipMemcpyHostToDevice, ctx->stream));
        ncclResult_t r = R->AllGather(d_mine, c->d_cnt.p, 2, ncclUint64, c->nccl, ctx->stream);
        if (r != ncclSuccess) return dp_fail(ctx, DP_ERR_HIP, R->GetErrorString(r));
        DP_HIP(hipMemcpyAsync(cnt, c->d_cnt.p, (size_t)N * 16, hipMemcpyDeviceToHost, ctx->stream));
        DP_HIP(dp_stream_sync(ctx));
    }
    // ---- this rank's payload: [read ids | hit counts] packed on the device; the segments are taken where they lie
    const uint32_t* s_item = (const uint32_t*)ctx->d_surv.p;
    const uint32_t* s_count = s_item + ctx->scan_items;
    if (comm_reserve(ctx, c->d_pay, (size_t)ns * 8 + 64)) return DP_ERR_HIP;
    uint32_t* d_read = (uint32_t*)c->d_pay.p;
    uint32_t* d_nseeds = d_read + ns;
    if (ns) {
        hipLaunchKernelGGL(comm_pack_meta, dim3((ns + 255) / 256), dim3(256), 0, ctx->stream, s_item, s_count, ns, ctx->cached_lo, d_read,
                           d_nseeds);
        DP_HIP(hipGetLastError());
    }
    if (c->local) {  // publish, wait for everybody, learn the sizes
        LocalGroup* g = c->local;
        DP_HIP(hipEventRecord(g->pub[(size_t)me].ready, ctx->stream));
        std::unique_lock<std::mutex> lk(g->mu);
        LocalGroup::Pub& p = g->pub[(size_t)me];
        p.read = d_read;
        p.nseeds = d_nseeds;
        p.segs = ctx->d_segs.p;
        p.n_surv = ns;
        p.n_ints = surv_ints;
        const uint64_t my_gen = g->gen;
        if (g->failed) return dp_fail(ctx, DP_ERR_STATE, "dp_allgather_survivors: a peer rank failed");
        if (++g->arrived == N) {
            g->arrived = 0;
            g->gen++;
            g->cv.notify_all();
        } else {
            g->cv.wait(lk, [&] { return g->gen != my_gen || g->failed; });
            if (g->gen == my_gen) return dp_fail(ctx, DP_ERR_STATE, "dp_allgather_survivors: a peer rank failed");
        }
        for (int r = 0; r < N; r++) {
            cnt[2 * r] = g->pub[(size_t)r].n_surv;
            cnt[2 * r + 1] = g->pub[(size_t)r].n_ints;
        }
    }
    uint64_t tot_surv = 0, tot_ints = 0, my_surv_off = 0, my_int_off = 0;
    for (int r = 0; r < N; r++) {
        if (r == me) {
            my_surv_off = tot_surv;
            my_int_off = tot_ints;
        }
        tot_surv += cnt[2 * r];
        tot_ints += cnt[2 * r + 1];
    }
    if (tot_surv > 0xfffffff0ull) return dp_fail(ctx, DP_ERR_CAPACITY, "dp_allgather_survivors: more than 2^32 survivors");
    // ---- gathered device buffers: metadata [read ids of all | hit counts of all]; segments [all survivors | my extra items] in a
    // buffer of their own, which BECOMES the context's scan output at the end (the two buffers change places: no copy)
    const uint64_t new_ints = tot_ints + extra_ints;
    if (comm_reserve(ctx, c->d_allpay, (size_t)tot_surv * 8 + 256)) return DP_ERR_HIP;
    if (comm_reserve(ctx, c->d_allsegs, (size_t)new_ints * 4 + 256)) return DP_ERR_HIP;
    uint32_t* a_read = (uint32_t*)c->d_allpay.p;
    uint32_t* a_nseeds = a_read + tot_surv;
    int32_t* a_segs = (int32_t*)c->d_allsegs.p;
    if (c->nccl) {
        // variable sizes: one broadcast per rank and array inside a group (RCCL fuses them); rank order = file order
        RcclApi* R = rccl_api();
        R->GroupStart();
        uint64_t so = 0, io = 0;
        ncclResult_t r = ncclSuccess;
        for (int q = 0; q < N && r == ncclSuccess; q++) {
            const uint64_t qs = cnt[2 * q], qi = cnt[2 * q + 1];
            if (qs) {
                r = R->Broadcast(q == me ? (const void*)d_read : (const void*)(a_read + so), a_read + so, qs, ncclUint32, q, c->nccl, ctx->stream);
                if (r == ncclSuccess)
                    r = R->Broadcast(q == me ? (const void*)d_nseeds : (const void*)(a_nseeds + so), a_nseeds + so, qs, ncclUint32, q, c->nccl, ctx->stream);
            }
            if (qi && r == ncclSuccess)
                r = R->Broadcast(q == me ? (const void*)ctx->d_segs.p : (const void*)(a_segs + io), a_segs + io, qi, ncclInt32, q, c->nccl, ctx->stream);
            so += qs;
            io += qi;
        }
        const ncclResult_t rg = R->GroupEnd();  // (the group is always closed; the first error of either is what is reported)
        if (r == ncclSuccess) r = rg;
        if (r != ncclSuccess) return dp_fail(ctx, DP_ERR_HIP, R->GetErrorString(r));
    } else {
        LocalGroup* g = c->local;
        uint64_t so = 0, io = 0;
        for (int q = 0; q < N; q++) {
            const LocalGroup::Pub& p = g->pub[(size_t)q];
            DP_HIP(hipStreamWaitEvent(ctx->stream, p.ready, 0));
            if (p.n_surv) {
                DP_HIP(hipMemcpyPeerAsync(a_read + so, ctx->device, p.read, p.device, p.n_surv * 4, ctx->stream));
                DP_HIP(hipMemcpyPeerAsync(a_nseeds + so, ctx->device, p.nseeds, p.device, p.n_surv * 4, ctx->stream));
            }
            if (p.n_ints) DP_HIP(hipMemcpyPeerAsync(a_segs + io, ctx->device, p.segs, p.device, p.n_ints * 4, ctx->stream));
            so += p.n_surv;
            io += p.n_ints;
        }
    }
    // my extra items follow the survivors
    if (extra_ints)
        DP_HIP(hipMemcpyAsync(a_segs + tot_ints, (const int32_t*)ctx->d_segs.p + surv_ints, extra_ints * 4, hipMemcpyDeviceToDevice, ctx->stream));
    // The gathered set described on the device the way dp_scan_reads describes its own survivors ({read, hit count, segment
    // offset} arrays): dp_index_build_chunked then chunks and indexes it where it lies (overlap.chunkWorker on the device)
    const uint32_t stride = (uint32_t)std::max<uint64_t>(1, tot_surv);
    if (dev_reserve(ctx, ctx->d_surv, (size_t)stride * 32 + 128)) return DP_ERR_HIP;
    {
        uint32_t* s_item = (uint32_t*)ctx->d_surv.p;
        uint32_t* s_count = s_item + stride;
        uint64_t* s_off = (uint64_t*)(s_count + stride + (stride & 1));
        hipLaunchKernelGGL(comm_install_survivors, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)a_read, (const uint32_t*)a_nseeds,
                           (uint32_t)tot_surv, s_item, s_count, s_off);
        DP_HIP(hipGetLastError());
    }
    // ---- host copies for the caller: the survivors' read ids and hit counts always; their segments only for a caller that
    // chunks on the host (dp_scan_fetch_mode(0)) - with fetch mode 1 they never leave the device (1-2 MB per round at k = 13,
    // ~100 MB in the dense regime); the extra items' (query windows') segments are the local scan's, already on the host
    const bool resident = ctx->scan_fetch_extras_only != 0;
    const size_t b_meta = (size_t)tot_surv * 8, b_off = ((size_t)tot_surv + ne + 2) * 8, b_ex = (size_t)ne * 4, b_segs = (size_t)new_ints * 4;
    if (comm_pin(ctx, c->h_out, b_meta + b_off + b_ex + b_segs + 256)) return DP_ERR_HIP;
    uint8_t* h = (uint8_t*)c->h_out.p;
    uint32_t* h_read = (uint32_t*)h;
    uint32_t* h_nseeds = h_read + tot_surv;
    uint64_t* h_off = (uint64_t*)(h + ((b_meta + 7) & ~(size_t)7));
    uint32_t* h_exn = (uint32_t*)((uint8_t*)h_off + b_off);
    int32_t* h_segs = (int32_t*)(((uintptr_t)((uint8_t*)h_exn + b_ex) + 15) & ~(uintptr_t)15);
    if (tot_surv) DP_HIP(hipMemcpyAsync(h_read, a_read, b_meta, hipMemcpyDeviceToHost, ctx->stream));
    if (new_ints && !resident) DP_HIP(hipMemcpyAsync(h_segs, a_segs, b_segs, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    if (c->local) {  // nobody may reuse its published buffers before every peer has copied from them
        LocalGroup* g = c->local;
        std::unique_lock<std::mutex> lk(g->mu);
        const uint64_t my_gen = g->gen;
        if (g->failed) return dp_fail(ctx, DP_ERR_STATE, "dp_allgather_survivors: a peer rank failed");
        if (++g->left == N) {
            g->left = 0;
            g->gen++;
            g->cv.notify_all();
        } else {
            g->cv.wait(lk, [&] { return g->gen != my_gen || g->failed; });
            if (g->gen == my_gen) return dp_fail(ctx, DP_ERR_STATE, "dp_allgather_survivors: a peer rank failed");
        }
    }
    // the merged array becomes the context's scan output (dp_index_build* / dp_find_overlaps read it): the buffers change places
    std::swap(ctx->d_segs, c->d_allsegs);
    ctx->n_segs = new_ints;
    ctx->scan_items = stride;
    ctx->last_surv_all = stride;
    ctx->chunk_lo = 0;  // (the installed survivor list holds read ids)
    uint64_t pos = 0;
    for (uint64_t i = 0; i < tot_surv; i++) {
        h_off[i] = pos;
        pos += 2ull * h_nseeds[i] + 1;
    }
    uint64_t* h_exoff = h_off + tot_surv + 1;
    for (uint32_t i = 0; i < ne; i++) {
        h_exn[i] = local->extra_n_seeds[i];
        h_exoff[i] = tot_ints + (local->extra_seg_off[i] - surv_ints);
        if (resident)  // (the query windows' segments: from the local scan's host copy to their place in the gathered layout)
            memcpy(h_segs + h_exoff[i], local->segs + local->extra_seg_off[i], (2 * (size_t)local->extra_n_seeds[i] + 1) * 4);
    }
    // (only now: `local` points into h_surv, the scan's own pinned output, which the gathered list replaces)
    if (pin_reserve(ctx, ctx->h_surv, (size_t)stride * 16 + 64)) return DP_ERR_HIP;
    memcpy(ctx->h_surv.p, h_read, (size_t)tot_surv * 4);
    memcpy((uint32_t*)ctx->h_surv.p + stride, h_nseeds, (size_t)tot_surv * 4);
    dp_survivor_batch o = *local;
    o.n_survivors = (uint32_t)tot_surv;
    o.read = h_read;
    o.n_seeds = h_nseeds;
    o.seg_off = h_off;
    o.n_extra = ne;
    o.extra_n_seeds = h_exn;
    o.extra_seg_off = h_exoff;
    o.segs = h_segs;
    o.n_segs = new_ints;
    *all = o;
    (void)my_surv_off;
    (void)my_int_off;
    return DP_OK;
}

// ---- all-gather of one variable-size byte string per rank (the round-parallel layout's result exchange) ---------------------
// Rank r contributes blob[0 .. n); every rank gets the ranks' strings back to back in rank order (*all_out, library-owned pinned
// memory, valid until the next call on this communicator) and their lengths (*sizes_out[n_ranks]).  RCCL flavour: sizes by
// ncclAllGather, payloads as one grouped broadcast per rank straight into the concatenation (exact sizes, no padding), one copy
// back; in-process flavour: host copies between the ranks' buffers.  Collective: every rank calls it.
static int allgather_blobs_impl(dp_comm* c, dp_ctx* ctx, const uint8_t* blob, uint64_t n, const uint8_t** all_out, const uint64_t** sizes_out) {
    hipSetDevice(ctx->device);
    const int N = c->n_ranks, me = c->rank;
    if (comm_pin(ctx, c->h_bsz, (size_t)N * 8 + 64)) return DP_ERR_HIP;
    uint64_t* sizes = (uint64_t*)c->h_bsz.p;
    if (c->local) {
        LocalGroup* g = c->local;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->blob[(size_t)me].p = blob;
            g->blob[(size_t)me].n = n;
            const uint64_t my_gen = g->gen;
            if (g->failed) return dp_fail(ctx, DP_ERR_STATE, "dp_allgather_blobs: a peer rank failed");
            if (++g->arrived == N) {
                g->arrived = 0;
                g->gen++;
                g->cv.notify_all();
            } else {
                g->cv.wait(lk, [&] { return g->gen != my_gen || g->failed; });
                if (g->gen == my_gen) return dp_fail(ctx, DP_ERR_STATE, "dp_allgather_blobs: a peer rank failed");
            }
        }
        uint64_t tot = 0;
        for (int r = 0; r < N; r++) tot += (sizes[r] = g->blob[(size_t)r].n);
        if (comm_pin(ctx, c->h_blob, tot + 64)) return DP_ERR_HIP;
        uint8_t* out = (uint8_t*)c->h_blob.p;
        uint64_t at = 0;
        for (int r = 0; r < N; r++) {
            if (sizes[r]) memcpy(out + at, g->blob[(size_t)r].p, sizes[r]);
            at += sizes[r];
        }
        std::unique_lock<std::mutex> lk(g->mu);  // (nobody may touch its blob again before every peer has copied it)
        const uint64_t my_gen = g->gen;
        if (g->failed) return dp_fail(ctx, DP_ERR_STATE, "dp_allgather_blobs: a peer rank failed");
        if (++g->left == N) {
            g->left = 0;
            g->gen++;
            g->cv.notify_all();
        } else {
            g->cv.wait(lk, [&] { return g->gen != my_gen || g->failed; });
            if (g->gen == my_gen) return dp_fail(ctx, DP_ERR_STATE, "dp_allgather_blobs: a peer rank failed");
        }
        *all_out = out;
        *sizes_out = sizes;
        return DP_OK;
    }
    if (!c->nccl) return dp_fail(ctx, DP_ERR_STATE, "dp_allgather_blobs: communicator without a transport");
    RcclApi* R = rccl_api();
    if (comm_reserve(ctx, c->d_cnt, (size_t)(N + 1) * 16)) return DP_ERR_HIP;
    uint64_t* d_mine = (uint64_t*)c->d_cnt.p + (size_t)N;
    const uint64_t mine = n;
    DP_HIP(hipMemcpyAsync(d_mine, &mine, 8, hipMemcpyHostToDevice, ctx->stream));
    ncclResult_t r = R->AllGather(d_mine, c->d_cnt.p, 1, ncclUint64, c->nccl, ctx->stream);
    if (r != ncclSuccess) return dp_fail(ctx, DP_ERR_HIP, R->GetErrorString(r));
    DP_HIP(hipMemcpyAsync(sizes, c->d_cnt.p, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    uint64_t tot = 0, my_off = 0;
    for (int q = 0; q < N; q++) {
        if (q == me) my_off = tot;
        tot += sizes[q];
    }
    if (comm_reserve(ctx, c->d_blob, (size_t)tot + 256)) return DP_ERR_HIP;
    if (comm_pin(ctx, c->h_blob, (size_t)tot + 64)) return DP_ERR_HIP;
    uint8_t* d_all = (uint8_t*)c->d_blob.p;
    if (n) DP_HIP(hipMemcpyAsync(d_all + my_off, blob, n, hipMemcpyHostToDevice, ctx->stream));
    R->GroupStart();
    uint64_t at = 0;
    for (int q = 0; q < N && r == ncclSuccess; q++) {
        if (sizes[q]) r = R->Broadcast(d_all + at, d_all + at, sizes[q], ncclUint8, q, c->nccl, ctx->stream);
        at += sizes[q];
    }
    const ncclResult_t rg = R->GroupEnd();
    if (r == ncclSuccess) r = rg;
    if (r != ncclSuccess) return dp_fail(ctx, DP_ERR_HIP, R->GetErrorString(r));
    if (tot) DP_HIP(hipMemcpyAsync(c->h_blob.p, d_all, tot, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    *all_out = (const uint8_t*)c->h_blob.p;
    *sizes_out = sizes;
    return DP_OK;
}

extern "C" int dp_allgather_blobs(dp_comm* c, dp_ctx* ctx, const uint8_t* blob, uint64_t n, const uint8_t** all_out, const uint64_t** sizes_out) {
    if (!c || !ctx || !all_out || !sizes_out || (n && !blob)) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_allgather_blobs: bad arguments") : DP_ERR_ARG;
    if (c->dead) return dp_fail(ctx, DP_ERR_STATE, "dp_allgather_blobs: the communicator failed in an earlier exchange");
    const int rc = allgather_blobs_impl(c, ctx, blob, n, all_out, sizes_out);
    if (rc != DP_OK) dp_comm_abort(c);  // (marks the group failed AND aborts an RCCL communicator: the peers' collectives return)
    return rc;
}

// ---- gather of one variable-size byte string per rank to ONE rank (the round-parallel layout's PAF text: only the rank that prints
// needs it) -------------------------------------------------------------------------------------------------------------------
// sizes[n_ranks] are known to every rank already (they travel inside the control blobs of dp_allgather_blobs), so there is no size
// exchange: RCCL flavour = one group of ncclSend (every rank but the root) / ncclRecv (the root, straight into the concatenation) and
// one copy back on the root; in-process flavour = the root copies from the peers' buffers.  Collective.
static int gather_blobs_impl(dp_comm* c, dp_ctx* ctx, const uint8_t* blob, uint64_t n, const uint64_t* sizes, int root, const uint8_t** all_out) {
    hipSetDevice(ctx->device);
    const int N = c->n_ranks, me = c->rank;
    if (sizes[me] != n) return dp_fail(ctx, DP_ERR_ARG, "dp_gather_blobs: sizes[rank] is not this rank's size");
    uint64_t tot = 0, my_off = 0;
    for (int q = 0; q < N; q++) {
        if (q == me) my_off = tot;
        tot += sizes[q];
    }
    *all_out = nullptr;
    if (c->local) {
        LocalGroup* g = c->local;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->blob[(size_t)me].p = blob;
            g->blob[(size_t)me].n = n;
            const uint64_t my_gen = g->gen;
            if (g->failed) return dp_fail(ctx, DP_ERR_STATE, "dp_gather_blobs: a peer rank failed");
            if (++g->arrived == N) {
                g->arrived = 0;
                g->gen++;
                g->cv.notify_all();
            } else {
                g->cv.wait(lk, [&] { return g->gen != my_gen || g->failed; });
                if (g->gen == my_gen) return dp_fail(ctx, DP_ERR_STATE, "dp_gather_blobs: a peer rank failed");
            }
        }
        if (me == root) {
            if (comm_pin(ctx, c->h_blob, tot + 64)) return DP_ERR_HIP;
            uint8_t* out = (uint8_t*)c->h_blob.p;
            uint64_t at = 0;
            for (int r = 0; r < N; r++) {
                if (g->blob[(size_t)r].n != sizes[r]) return dp_fail(ctx, DP_ERR_ARG, "dp_gather_blobs: a peer's size differs from sizes[]");
                if (sizes[r]) memcpy(out + at, g->blob[(size_t)r].p, sizes[r]);
                at += sizes[r];
            }
            *all_out = out;
        }
        std::unique_lock<std::mutex> lk(g->mu);  // (nobody may touch its blob again before the root has copied it)
        const uint64_t my_gen = g->gen;
        if (g->failed) return dp_fail(ctx, DP_ERR_STATE, "dp_gather_blobs: a peer rank failed");
        if (++g->left == N) {
            g->left = 0;
            g->gen++;
            g->cv.notify_all();
        } else {
            g->cv.wait(lk, [&] { return g->gen != my_gen || g->failed; });
            if (g->gen == my_gen) return dp_fail(ctx, DP_ERR_STATE, "dp_gather_blobs: a peer rank failed");
        }
        return DP_OK;
    }
    if (!c->nccl) return dp_fail(ctx, DP_ERR_STATE, "dp_gather_blobs: communicator without a transport");
    RcclApi* R = rccl_api();
    if (!R->Send || !R->Recv) return dp_fail(ctx, DP_ERR_STATE, "librccl lacks ncclSend / ncclRecv");
    ncclResult_t r = ncclSuccess;
    if (me == root) {
        if (comm_reserve(ctx, c->d_blob, (size_t)tot + 256)) return DP_ERR_HIP;
        if (comm_pin(ctx, c->h_blob, (size_t)tot + 64)) return DP_ERR_HIP;
        uint8_t* d_all = (uint8_t*)c->d_blob.p;
        R->GroupStart();
        uint64_t at = 0;
        for (int q = 0; q < N && r == ncclSuccess; q++) {
            if (q != root && sizes[q]) r = R->Recv(d_all + at, sizes[q], ncclUint8, q, c->nccl, ctx->stream);
            at += sizes[q];
        }
        const ncclResult_t rg = R->GroupEnd();
        if (r == ncclSuccess) r = rg;
        if (r != ncclSuccess) return dp_fail(ctx, DP_ERR_HIP, R->GetErrorString(r));
        // the peers' parts come back from the device; the root's own never left the host
        uint8_t* out = (uint8_t*)c->h_blob.p;
        if (my_off) DP_HIP(hipMemcpyAsync(out, d_all, my_off, hipMemcpyDeviceToHost, ctx->stream));
        if (tot > my_off + n) DP_HIP(hipMemcpyAsync(out + my_off + n, d_all + my_off + n, tot - my_off - n, hipMemcpyDeviceToHost, ctx->stream));
        if (n) memcpy(out + my_off, blob, n);
        DP_HIP(dp_stream_sync(ctx));
        *all_out = out;
    } else if (n) {
        if (comm_reserve(ctx, c->d_blob, (size_t)n + 256)) return DP_ERR_HIP;
        DP_HIP(hipMemcpyAsync(c->d_blob.p, blob, n, hipMemcpyHostToDevice, ctx->stream));
        R->GroupStart();
        r = R->Send(c->d_blob.p, n, ncclUint8, root, c->nccl, ctx->stream);
        const ncclResult_t rg = R->GroupEnd();
        if (r == ncclSuccess) r = rg;
        if (r != ncclSuccess) return dp_fail(ctx, DP_ERR_HIP, R->GetErrorString(r));
        DP_HIP(dp_stream_sync(ctx));  // (blob is the caller's again, d_blob this communicator's next exchange's)
    }
    return DP_OK;
}

// ---- all-gather of device-resident ranges, in place (round 5: the k-mer position index built in shares, dp_kindex.hip) --------------
// Every rank holds an array of `total` elements of `elem` bytes at `dst`; rank q owns the elements [first[q], first[q + 1]) - its
// share sits at `src` (or already in place when src is null).  Afterwards every rank's array is complete.  The shares are known to all
// ranks (no size exchange).  RCCL flavour: one grouped broadcast per rank straight between the device buffers (xGMI, no host copy);
// in-process flavour: every rank copies the peers' shares device to device once all have published theirs.  Internal (not part of
// the C ABI): called on the context's stream, returns after the stream has been waited for.
int dp_comm_allgather_ranges(dp_comm* c, dp_ctx* ctx, void* dst, size_t elem, const uint64_t* first, const void* src) {
    if (!c || !ctx || !dst || !first) return DP_ERR_ARG;
    if (c->dead) return dp_fail(ctx, DP_ERR_STATE, "index all-gather: the communicator failed in an earlier exchange");
    hipSetDevice(ctx->device);
    const int N = c->n_ranks, me = c->rank;
    uint8_t* d = (uint8_t*)dst;
    const void* mine = src ? src : (const void*)(d + first[me] * elem);
    int rc = DP_OK;
    if (c->local) {
        LocalGroup* g = c->local;
        auto barrier = [&](const char* what) -> int {
            std::unique_lock<std::mutex> lk(g->mu);
            const uint64_t my_gen = g->gen;
            if (g->failed) return dp_fail(ctx, DP_ERR_STATE, what);
            if (++g->arrived == N) {
                g->arrived = 0;
                g->gen++;
                g->cv.notify_all();
            } else {
                g->cv.wait(lk, [&] { return g->gen != my_gen || g->failed; });
                if (g->gen == my_gen) return dp_fail(ctx, DP_ERR_STATE, what);
            }
            return DP_OK;
        };
        // (the share must be final before a peer reads it: this rank's stream is waited for before it publishes)
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = dp_fail(ctx, DP_ERR_HIP, "index all-gather: stream");
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->blob[(size_t)me].p = (const uint8_t*)mine;
            g->blob[(size_t)me].n = (uint64_t)ctx->device;
        }
        if (rc == DP_OK) rc = barrier("index all-gather: a peer rank failed");
        if (rc == DP_OK) {
            for (int q = 0; q < N && rc == DP_OK; q++) {
                const uint64_t bytes = (first[q + 1] - first[q]) * elem;
                if (!bytes) continue;
                if (q == me) {
                    if (src && hipMemcpyAsync(d + first[q] * elem, src, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                        rc = dp_fail(ctx, DP_ERR_HIP, "index all-gather: copy");
                } else if (hipMemcpyPeerAsync(d + first[q] * elem, ctx->device, g->blob[(size_t)q].p, (int)g->blob[(size_t)q].n, bytes, ctx->stream) != hipSuccess) {
                    rc = dp_fail(ctx, DP_ERR_HIP, "index all-gather: peer copy");
                }
            }
            if (rc == DP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = dp_fail(ctx, DP_ERR_HIP, "index all-gather: stream");
        }
        // (nobody may free or overwrite its share before every peer has copied it)
        if (rc == DP_OK) rc = barrier("index all-gather: a peer rank failed");
    } else if (c->nccl) {
        RcclApi* R = rccl_api();
        ncclResult_t r = ncclSuccess;
        R->GroupStart();
        for (int q = 0; q < N && r == ncclSuccess; q++) {
            const uint64_t bytes = (first[q + 1] - first[q]) * elem;
            if (!bytes) continue;
            r = R->Broadcast(q == me ? mine : (const void*)(d + first[q] * elem), d + first[q] * elem, bytes, ncclUint8, q, c->nccl, ctx->stream);
        }
        const ncclResult_t rg = R->GroupEnd();
        if (r == ncclSuccess) r = rg;
        if (r != ncclSuccess) rc = dp_fail(ctx, DP_ERR_HIP, R->GetErrorString(r));
        else if (dp_stream_sync(ctx) != hipSuccess) rc = dp_fail(ctx, DP_ERR_HIP, "index all-gather: stream");
    } else {
        rc = dp_fail(ctx, DP_ERR_STATE, "index all-gather: communicator without a transport");
    }
    if (rc != DP_OK) dp_comm_abort(c);
    return rc;
}

extern "C" int dp_gather_blobs(dp_comm* c, dp_ctx* ctx, const uint8_t* blob, uint64_t n, const uint64_t* sizes, int root, const uint8_t** all_out) {
    if (!c || !ctx || !all_out || !sizes || (n && !blob) || root < 0 || root >= c->n_ranks)
        return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_gather_blobs: bad arguments") : DP_ERR_ARG;
    if (c->dead) return dp_fail(ctx, DP_ERR_STATE, "dp_gather_blobs: the communicator failed in an earlier exchange");
    const int rc = gather_blobs_impl(c, ctx, blob, n, sizes, root, all_out);
    if (rc != DP_OK) dp_comm_abort(c);
    return rc;
}

bool dp_comm_is_rccl(const dp_comm* c) { return c && c->nccl != nullptr; }  // (the one-process flavour exchanges through peer copies)
extern "C" int dp_comm_rank(const dp_comm* c) { return c ? c->rank : -1; }
extern "C" int dp_comm_size(const dp_comm* c) { return c ? c->n_ranks : 0; }
