// libdownpore_hip.so - gangs: several contexts whose per-round launches are issued as one (dp_gang.h has the device side and the
// launch template; this file has the rendezvous of the member threads and the C-ABI entry points).
//
// Every member is driven by its own host thread through the ordinary per-round calls.  Inside a round a member is always in one
// of three places: running host code (RUN), at a launch (LAUNCH: its argument block is deposited), or at a wait (SYNC).  The
// member whose arrival leaves nobody in RUN acts for the gang:
//   1. members at a launch: their deposits are grouped by kernel and issued - one launch per kernel, blockIdx.y = member - and
//      they go on;
//   2. otherwise members at a wait go on (each then waits for the gang's stream: everything deposited so far is on it);
//   3. otherwise members that want to start a round (START) go on - together, and only when no member is still getting ready
//      (PREP: it has a round but not yet its inputs), so that rounds of a gang begin in step and stay in step: same kernels in the
//      same order.  A member whose round takes a different path (a buffer overflow repeated, a fallback) simply forms a launch
//      group of its own until the next round.
// Waiting members spin (a rendezvous is a few microseconds, 26 of them per round).
#include <sched.h>
#include <time.h>

#include <cstdio>
#include <cstdlib>

#include <atomic>
#include <mutex>

#include "dp_common.h"
#include "dp_gang.h"

struct dp_gang {
    int n = 0, device = 0;
    dp_ctx* member[DP_GANG_MAX] = {};
    hipStream_t stream = nullptr;
    std::mutex mu;
    int state[DP_GANG_MAX] = {};
    const DpDeposit* dep[DP_GANG_MAX] = {};
    std::atomic<uint32_t> released[DP_GANG_MAX];
    long long off_since_ns[DP_GANG_MAX] = {};  // when the member last left a round (a member between two rounds is about to be back)
    // counters (dp_gang_counters)
    uint64_t deposits = 0, launches = 0, syncs = 0, starts = 0, started_members = 0;
    // sticky: a member reported an error inside a round (dp_fail) or a rendezvous waited longer than DP_GANG_TIMEOUT_S (30 s).  Every
    // waiting member is released, launches and waits of the members are plain ones on the shared stream from then on (in order, so
    // still correct), and dp_gang_round_begin returns DP_ERR_STATE: a caller that forgets dp_gang_round_end after an error - the
    // exported C and Go APIs rely on the caller for that - no longer leaves the other members spinning.
    std::atomic<bool> failed{false};
};

static long long gang_now_ns() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (long long)t.tv_sec * 1000000000ll + t.tv_nsec;
}
// A member that left its round less than this ago is expected back with its next round: members that are ready to start wait
// for it (DP_GANG_GRACE_US), so that the rounds of a gang keep beginning - and therefore launching - together.
static long long gang_grace_ns() {
    static const long long g = [] {
        const char* e = getenv("DP_GANG_GRACE_US");
        return (e ? atoll(e) : 150ll) * 1000ll;
    }();
    return g;
}

hipStream_t dp_ctx_stream(const dp_ctx* ctx) { return ctx->stream; }
bool dp_gang_active(const dp_ctx* ctx) { return ctx->gang && ctx->gang_in_round && !ctx->gang->failed.load(std::memory_order_acquire); }
void dp_gang_mark_failed(dp_ctx* ctx) {
    if (ctx && ctx->gang && ctx->gang_in_round) ctx->gang->failed.store(true, std::memory_order_release);
}

// with the gang's lock held: acts for the gang if nobody is running host code
static void gang_resolve(dp_gang* g) {
    int n_launch = 0, n_sync = 0, n_prep = 0, n_start = 0;
    for (int i = 0; i < g->n; i++) {
        switch (g->state[i]) {
            case DPG_RUN: return;
            case DPG_LAUNCH: n_launch++; break;
            case DPG_SYNC: n_sync++; break;
            case DPG_PREP: n_prep++; break;
            case DPG_START: n_start++; break;
            default: break;
        }
    }
    if (n_launch) {
        bool done[DP_GANG_MAX] = {};
        for (int i = 0; i < g->n; i++) {
            if (g->state[i] != DPG_LAUNCH || done[i]) continue;
            const DpDeposit* who[DP_GANG_MAX];
            int m = 0;
            for (int j = i; j < g->n; j++)
                if (g->state[j] == DPG_LAUNCH && !done[j] && g->dep[j]->issue == g->dep[i]->issue && g->dep[j]->block_x == g->dep[i]->block_x) {
                    who[m++] = g->dep[j];
                    done[j] = true;
                }
            g->dep[i]->issue(g->stream, who, m);
            g->launches++;
            g->deposits += (uint64_t)m;
        }
        for (int i = 0; i < g->n; i++)
            if (done[i]) {
                g->state[i] = DPG_RUN;
                g->released[i].fetch_add(1, std::memory_order_release);
            }
        return;
    }
    if (n_sync) {
        g->syncs++;
        for (int i = 0; i < g->n; i++)
            if (g->state[i] == DPG_SYNC) {
                g->state[i] = DPG_RUN;
                g->released[i].fetch_add(1, std::memory_order_release);
            }
        return;
    }
    if (n_prep || !n_start) return;
    if (n_start < g->n) {  // somebody is between two rounds: give it the time to come back (the waiting members ask again)
        const long long now = gang_now_ns();
        for (int i = 0; i < g->n; i++)
            if (g->state[i] == DPG_OFF && g->off_since_ns[i] && now - g->off_since_ns[i] < gang_grace_ns()) return;
    }
    g->starts++;
    g->started_members += (uint64_t)n_start;
    for (int i = 0; i < g->n; i++)
        if (g->state[i] == DPG_START) {
            g->member[i]->gang_round_members = n_start;
            g->state[i] = DPG_RUN;
            g->released[i].fetch_add(1, std::memory_order_release);
        }
}

static void gang_arrive(dp_ctx* ctx, int st, const DpDeposit* d) {
    dp_gang* g = ctx->gang;
    const int me = ctx->gang_slot;
    const uint32_t tk = g->released[me].load(std::memory_order_relaxed);
    {
        std::lock_guard<std::mutex> lk(g->mu);
        if (st == DPG_OFF) g->off_since_ns[me] = g->state[me] == DPG_RUN ? gang_now_ns() : 0;  // (0: it had no round after all)
        g->state[me] = st;
        g->dep[me] = d;
        gang_resolve(g);
    }
    if (st == DPG_OFF || st == DPG_PREP) return;
    unsigned spins = 0;
    static const bool debug = getenv("DP_GANG_DEBUG") != nullptr;
    static const long long timeout_ns = (getenv("DP_GANG_TIMEOUT_S") ? atoll(getenv("DP_GANG_TIMEOUT_S")) : 30ll) * 1000000000ll;
    long long wait_from = 0;
    timespec t0{0, 0};
    bool dumped = false;
    while (g->released[me].load(std::memory_order_acquire) == tk) {
        if (g->failed.load(std::memory_order_acquire)) return;  // (the caller goes on with plain launches / waits)
        if (++spins < 4096) {
            __builtin_ia32_pause();
            continue;
        }
        sched_yield();
        if ((spins & 255) == 0) {  // a rendezvous nobody completes: the gang gives up instead of spinning for ever
            const long long now = gang_now_ns();
            if (!wait_from) wait_from = now;
            else if (now - wait_from > timeout_ns) {
                g->failed.store(true, std::memory_order_release);
                ctx->err = "gang rendezvous timed out (a member left a round without dp_gang_round_end?)";
                return;
            }
        }
        if (st == DPG_START && (spins & 15) == 0) {  // (held back for a member between two rounds: its grace may be over)
            std::lock_guard<std::mutex> lk(g->mu);
            if (g->state[me] == DPG_START) gang_resolve(g);
        }
        if (debug && !dumped && (spins & 1023) == 0) {  // a member stuck for two seconds: who is where
            timespec t;
            clock_gettime(CLOCK_MONOTONIC, &t);
            if (!t0.tv_sec) t0 = t;
            if (t.tv_sec - t0.tv_sec >= 2) {
                dumped = true;
                std::lock_guard<std::mutex> lk(g->mu);
                static const char* names[] = {"OFF", "PREP", "START", "RUN", "LAUNCH", "SYNC"};
                fprintf(stderr, "[gang %p] member %d waits in %s for 2 s:", (void*)g, me, names[st]);
                for (int i = 0; i < g->n; i++)
                    fprintf(stderr, " %d=%s%s", i, names[g->state[i]], g->state[i] == DPG_LAUNCH ? (g->dep[i] ? "" : "(no deposit)") : "");
                for (int i = 0; i < g->n; i++)
                    if (g->state[i] == DPG_LAUNCH && g->dep[i]) fprintf(stderr, " [%d: issue %p grid %u block %u]", i, (void*)g->dep[i]->issue, g->dep[i]->grid_x, g->dep[i]->block_x);
                fprintf(stderr, " | stream %s\n", hipStreamQuery(g->stream) == hipSuccess ? "idle" : "busy");
            }
        }
    }
}

// A member that has to block on something another member may hold (a process-wide lock, a one-off build under a mutex) steps
// out of its round for that long: the others are not kept at their launches, its own launches and waits are plain ones.
bool dp_gang_pause(dp_ctx* ctx) {
    if (!dp_gang_active(ctx)) return false;
    ctx->gang_in_round = false;
    gang_arrive(ctx, DPG_OFF, nullptr);
    return true;
}
void dp_gang_resume(dp_ctx* ctx) {
    dp_gang* g = ctx->gang;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->state[ctx->gang_slot] = DPG_RUN;
    }
    ctx->gang_in_round = true;
}

void dp_gang_deposit(dp_ctx* ctx, const DpDeposit& d) { gang_arrive(ctx, DPG_LAUNCH, &d); }
void dp_gang_sync_point(dp_ctx* ctx) { gang_arrive(ctx, DPG_SYNC, nullptr); }

extern "C" int dp_gang_create(dp_ctx* const* ctxs, int n, dp_gang** out) {
    if (!ctxs || !out || n < 1 || n > DP_GANG_MAX) return DP_ERR_ARG;
    *out = nullptr;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return DP_ERR_ARG;
        if (ctxs[i]->gang) return dp_fail(ctxs[i], DP_ERR_STATE, "dp_gang_create: the context is a member of a gang already");
        if (ctxs[i]->device != ctxs[0]->device) return dp_fail(ctxs[i], DP_ERR_ARG, "dp_gang_create: contexts on different devices");
    }
    dp_ctx* ctx = ctxs[0];
    hipSetDevice(ctx->device);
    dp_gang* g = new dp_gang();
    g->n = n;
    g->device = ctx->device;
    hipError_t e = hipStreamCreate(&g->stream);
    if (e != hipSuccess) {
        delete g;
        return dp_fail(ctx, DP_ERR_HIP, "dp_gang_create: hipStreamCreate", e);
    }
    for (int i = 0; i < n; i++) {
        g->released[i].store(0);
        g->state[i] = DPG_OFF;
        g->member[i] = ctxs[i];
        hipStreamSynchronize(ctxs[i]->stream);  // (nothing of the member's own stream may still be queued when it changes streams)
        ctxs[i]->own_stream = ctxs[i]->stream;
        ctxs[i]->stream = g->stream;
        ctxs[i]->gang = g;
        ctxs[i]->gang_slot = i;
        ctxs[i]->gang_in_round = false;
    }
    *out = g;
    return DP_OK;
}

extern "C" void dp_gang_destroy(dp_gang* g) {
    if (!g) return;
    hipSetDevice(g->device);
    hipStreamSynchronize(g->stream);
    for (int i = 0; i < g->n; i++) {
        dp_ctx* c = g->member[i];
        if (!c) continue;
        c->stream = c->own_stream;
        c->own_stream = nullptr;
        c->gang = nullptr;
        c->gang_slot = -1;
        c->gang_in_round = false;
    }
    hipStreamDestroy(g->stream);
    delete g;
}

// A member that is destroyed before its gang (finalisers of a garbage-collected host run in any order): the gang forgets it -
// the slot counts as OFF for good, so the others never wait for it, and dp_gang_destroy no longer touches the freed context.
void dp_gang_forget(dp_ctx* ctx) {
    dp_gang* g = ctx->gang;
    if (!g) return;
    hipStreamSynchronize(g->stream);  // (launches of this member's last round may still be queued on the shared stream)
    {
        std::lock_guard<std::mutex> lk(g->mu);
        const int me = ctx->gang_slot;
        if (me >= 0 && me < g->n && g->member[me] == ctx) {
            g->member[me] = nullptr;
            g->state[me] = DPG_OFF;
            g->dep[me] = nullptr;
            g->off_since_ns[me] = 0;
            gang_resolve(g);
        }
    }
    ctx->stream = ctx->own_stream;
    ctx->own_stream = nullptr;
    ctx->gang = nullptr;
    ctx->gang_slot = -1;
    ctx->gang_in_round = false;
}

extern "C" int dp_gang_round_prepare(dp_ctx* ctx) {
    if (!ctx) return DP_ERR_ARG;
    if (!ctx->gang || ctx->gang_in_round) return DP_OK;
    gang_arrive(ctx, DPG_PREP, nullptr);
    return DP_OK;
}
extern "C" int dp_gang_round_begin(dp_ctx* ctx) {
    if (!ctx) return DP_ERR_ARG;
    if (!ctx->gang || ctx->gang_in_round) return DP_OK;
    ctx->gang_round_members = 1;
    if (ctx->gang->failed.load(std::memory_order_acquire)) return dp_fail(ctx, DP_ERR_STATE, "dp_gang_round_begin: the gang has failed (a member's error or a rendezvous timeout); destroy it");
    gang_arrive(ctx, DPG_START, nullptr);
    if (ctx->gang->failed.load(std::memory_order_acquire)) return dp_fail(ctx, DP_ERR_STATE, "dp_gang_round_begin: the gang has failed (a member's error or a rendezvous timeout); destroy it");
    ctx->gang_in_round = true;
    return DP_OK;
}
extern "C" int dp_gang_round_end(dp_ctx* ctx) {
    if (!ctx) return DP_ERR_ARG;
    if (!ctx->gang) return DP_OK;
    ctx->gang_in_round = false;
    gang_arrive(ctx, DPG_OFF, nullptr);
    return DP_OK;
}
extern "C" int dp_gang_round_members(const dp_ctx* ctx) { return ctx && ctx->gang ? ctx->gang_round_members : 1; }
extern "C" void dp_gang_counters(dp_gang* g, uint64_t* out /* [5] */) {
    if (!g || !out) return;
    std::lock_guard<std::mutex> lk(g->mu);
    out[0] = g->deposits;
    out[1] = g->launches;
    out[2] = g->syncs;
    out[3] = g->starts;
    out[4] = g->started_members;
}
