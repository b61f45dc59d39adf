// libdownpore_hip.so — context, read packing (A1), k-mer histogram (A22), per-round seed tables and the batched
// packed k-mer scan (A2 + A10).  CDNA4 / gfx950 only.
//
// Data layout in HBM
//   d_packed : all reads, 2 bit/base, first base in the two top bits of each byte (the reference's packedSequence
//              encoding, sequence/sequence.go:43-93).  Every read starts on a 16-byte boundary; padding and a 64-byte
//              tail are zero.  A k-mer start position is addressed by its ABSOLUTE base index a = 4*boff[read] + p.
//   d_bits   : 4^k-bit membership table (8 MiB at k=13; the reference uses a 4^k-BYTE []bool, seeds/seeds.go:13).
//   d_kmap   : dense int32 k-mer -> seed id (reference kmerMap, seeds/seeds.go:17); touched only on true hits.
//
// Scan kernel (one 1024-thread workgroup per CU, persistent; one WAVE owns one item at a time)
//   * the workgroup first builds a 1 Mbit (128 KiB) prefix filter of the round's seeds in LDS: bit = the first
//     min(k,10) bases of a seed.  Random sequence => ~n_seeds/2^20 pass rate, so only ~1 % of the k-mers go on to
//     the exact 4^k-bit table in L2/Infinity Cache.
//   * a lane handles one aligned group of 32 k-mer start positions: 12 bytes of packed read (three big-endian
//     dwords), 32 funnel-shift extractions, 32 one-byte LDS probes, then the (rare) exact probes.
//   * count pass -> per-item hit count; offsets pass -> exclusive scan of 2*count+1 over surviving items; write
//     pass -> wave prefix sums place each hit, a wave prefix-max supplies the previous hit for the gap.
// Algorithmic bytes per scan: packed bytes of the items + 4^k/8 + 8 B per written hit.
#include <sys/prctl.h>
#include <time.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "dp_common.h"
#include "dp_launch.h"

static std::string g_create_err;
// The scan kernels are persistent and fill every CU: two of them running at once (several contexts on one device) only
// slow each other down, so the device part of dp_scan is serialised per process.
// (a counting gate with the lock()/unlock() of a mutex: one scan in flight)
struct ScanGate {
    std::mutex mu;
    std::condition_variable cv;
    int avail = 1;
    void lock() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return avail > 0; });
        avail--;
    }
    void unlock() {
        {
            std::lock_guard<std::mutex> lk(mu);
            avail++;
        }
        cv.notify_one();
    }
};
static ScanGate g_scan_mu;

// Waiting for the context's stream.  Two ways:
//   spin  hipStreamSynchronize: the runtime busy-waits on the completion signal - the wake-up is immediate, and the waiting
//         thread occupies a core.  A round has five to seven waits of 50-500 us; with eight executor slots this is 12-18 % faster
//         per job than polling (0.44 against 0.54 ms per round) and costs about four more cores.
//   poll  record an event, poll it, sleep 20 microseconds (DP_TUNE=sync_poll_us=n) between polls (timer slack lowered to 1 us): every
//         wake-up is 50-70 us late, but a waiting thread costs nothing - what a process needs when it has fewer cores than
//         waiting threads (round 1: five slots, 17 core-ms of host work per round on a 16-core quota).
// dp_set_stream_wait() chooses (the host pipeline does, from its CPU budget and its number of slots); DP_SPIN_SYNC=0/1 in the
// environment overrides it.  sync_poll_us=0: blocking hipEventSynchronize instead of the poll loop.
static std::atomic<int> g_wait_spin{0};
extern "C" void dp_set_stream_wait(int spin) { g_wait_spin.store(spin ? 1 : 0); }
static std::atomic<long> g_timing_every{[] {
    const char* e = getenv("DP_KERNEL_TIMING");
    return e ? atol(e) : 8L;
}()};
extern "C" void dp_set_kernel_timing(int every) { g_timing_every.store(every < 0 ? 0 : every); }

#ifdef DP_COPY_LOG
#undef hipMemcpyAsync
namespace {
struct CopyLog {
    std::mutex mu;
    std::map<std::pair<std::string, int>, std::pair<uint64_t, uint64_t>> sites[5];
    ~CopyLog() {
        static const char* kinds[] = {"H2H", "H2D", "D2H", "D2D", "default"};
        for (int k = 0; k < 5; k++)
            for (auto& e : sites[k])
                fprintf(stderr, "[copylog] %s %s:%d calls %llu bytes %llu\n", kinds[k], e.first.first.c_str(), e.first.second,
                        (unsigned long long)e.second.first, (unsigned long long)e.second.second);
    }
} g_copy_log;
}  // namespace
hipError_t dp_copy_logged(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s, const char* file, int line) {
    {
        std::lock_guard<std::mutex> lk(g_copy_log.mu);
        auto& e = g_copy_log.sites[(int)kind < 5 ? (int)kind : 4][{file, line}];
        e.first++;
        e.second += n;
    }
    return hipMemcpyAsync(dst, src, n, kind, s);
}
#define hipMemcpyAsync(d_, s_, n_, k_, st_) dp_copy_logged((void*)(d_), (const void*)(s_), (n_), (k_), (st_), __FILE__, __LINE__)
#endif

const void* dp_stage(dp_ctx* ctx, const void* src, size_t bytes) {
    const size_t at = (ctx->stage_used + 63) & ~(size_t)63;
    if (at + bytes > ctx->stage_buf.size()) {
        if (ctx->stage_used) hipStreamSynchronize(ctx->stream);  // (copies out of the old block must be done before it moves)
        ctx->stage_used = 0;
        if (bytes > ctx->stage_buf.size()) ctx->stage_buf.resize(std::max<size_t>(bytes * 2, (size_t)4 << 20));
        return dp_stage(ctx, src, bytes);
    }
    uint8_t* dst = ctx->stage_buf.data() + at;
    memcpy(dst, src, bytes);
    ctx->stage_used = at + bytes;
    return dst;
}

hipStream_t dp_ctx_stream(const dp_ctx* ctx) { return ctx->stream; }

static std::map<std::string, std::string> env_tokens(const char* name) {
    std::map<std::string, std::string> m;
    const char* e = getenv(name);
    if (!e) return m;
    std::string s(e);
    size_t at = 0;
    while (at <= s.size()) {
        size_t end = s.find(',', at);
        if (end == std::string::npos) end = s.size();
        const std::string tok = s.substr(at, end - at);
        const size_t eq = tok.find('=');
        if (!tok.empty()) m[eq == std::string::npos ? tok : tok.substr(0, eq)] = eq == std::string::npos ? "1" : tok.substr(eq + 1);
        at = end + 1;
    }
    return m;
}
// (parsed again whenever the variable's text has changed: tests set it between jobs of one process)
struct EnvTokens {
    const char* name;
    std::mutex mu;
    std::string text;
    bool parsed = false;
    std::map<std::string, std::string> m;
    const std::map<std::string, std::string>& get() {  // (call with mu held)
        const char* e = getenv(name);
        if (!e) e = "";
        if (!parsed || text != e) {
            text = e;
            m = env_tokens(name);
            parsed = true;
        }
        return m;
    }
};
bool dp_debug(const char* what) {
    static EnvTokens t{"DP_DEBUG"};
    std::lock_guard<std::mutex> lk(t.mu);
    return t.get().count(what) != 0;
}
long dp_tune(const char* key, long dflt) {
    static EnvTokens t{"DP_TUNE"};
    std::lock_guard<std::mutex> lk(t.mu);
    const auto& m = t.get();
    const auto it = m.find(key);
    return it == m.end() ? dflt : atol(it->second.c_str());
}

hipError_t dp_stream_sync(dp_ctx* ctx) {
    ctx->stage_used = 0;  // (everything queued so far, copies out of the staging block included, is done when this returns)
    static const int env_spin = [] {
        const char* e = getenv("DP_SPIN_SYNC");
        return e ? (e[0] == '1' ? 1 : 0) : -1;
    }();
    static const long poll_ns = dp_tune("sync_poll_us", 20) * 1000L;
    const bool spin = env_spin >= 0 ? env_spin == 1 : g_wait_spin.load(std::memory_order_relaxed) != 0;
    if (spin || !ctx->ev_sync) return hipStreamSynchronize(ctx->stream);
    hipError_t e = hipEventRecord(ctx->ev_sync, ctx->stream);
    if (e != hipSuccess) return e;
    if (poll_ns <= 0) return hipEventSynchronize(ctx->ev_sync);
    static thread_local bool slack_set = false;
    if (!slack_set) {
        prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
        slack_set = true;
    }
    for (;;) {
        e = hipEventQuery(ctx->ev_sync);
        if (e != hipErrorNotReady) return e;
        timespec ts{0, poll_ns};
        nanosleep(&ts, nullptr);
    }
}

int dp_fail(dp_ctx* ctx, int code, const char* what, hipError_t e) {
    std::string s = what;
    if (e != hipSuccess) {
        s += ": ";
        s += hipGetErrorString(e);
    }
    if (ctx) ctx->err = s;
    else g_create_err = s;
    return code;
}

// Growing a buffer: twice the request (rounds vary in size and every growth is a hipMalloc / hipHostMalloc of megabytes),
// and the outgrown buffer is NOT freed here - hipFree / hipHostFree wait for the whole device, i.e. for every other
// executor slot's kernels, which showed up as 20-30 ms stalls a few times per hundred rounds.  Old buffers go to the
// context's retired list and are released with the context (their total is below the live size: geometric growth).
// DP_ALLOC_TRACE=1: one stderr line per growth (what, bytes, how long the allocation call took)
static bool alloc_trace() {
    static const bool on = dp_debug("alloc");
    return on;
}
static double alloc_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int dev_reserve(dp_ctx* ctx, DevBuf& b, size_t bytes, bool keep) {
    if (bytes <= b.cap) return 0;
    const double t0 = alloc_trace() ? alloc_now() : 0;
    // (per-round buffers get head room; the big resident ones - packed reads, k-mer index - are sized exactly)
    size_t ncap = bytes > ((size_t)256 << 20) ? bytes : std::max(bytes + bytes / 2, b.cap * 2);
    ncap = (ncap + 255) & ~(size_t)255;
    void* np = nullptr;
    DP_HIP(dp_dev_malloc(&np, ncap));
    if (b.p) {
        if (keep) DP_HIP(hipMemcpyAsync(np, b.p, b.cap, hipMemcpyDeviceToDevice, ctx->stream));
        ctx->retired_dev.push_back(b.p);  // (work already queued on the stream may still read it)
    }
    if (alloc_trace()) fprintf(stderr, "[alloc] device %zu -> %zu bytes, %.3f ms\n", b.cap, ncap, 1e3 * (alloc_now() - t0));
    b.p = np;
    b.cap = ncap;
    return 0;
}
int pin_reserve(dp_ctx* ctx, PinBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    size_t ncap = bytes > ((size_t)256 << 20) ? bytes : std::max(bytes + bytes / 2, b.cap * 2);
    ncap = (ncap + 4095) & ~(size_t)4095;
    void* np = nullptr;
    const double t0 = alloc_trace() ? alloc_now() : 0;
    DP_HIP(dp_pin_malloc(&np, ncap));
    if (alloc_trace()) fprintf(stderr, "[alloc] pinned %zu -> %zu bytes, %.3f ms\n", b.cap, ncap, 1e3 * (alloc_now() - t0));
    if (b.p) ctx->retired_pin.push_back(b.p);
    b.p = np;
    b.cap = ncap;
    return 0;
}

// ---- device and pinned blocks that outlive their context
// Keyed by device: one process may drive several GPUs (dp_comm_init_local, a Go host with one goroutine per GPU); a block is
// only ever handed back to the device it was allocated on, and the wait / hipFree that release it run with that device current.
// Round 5: blocks from 4 KiB on (were: from 32 MiB) and the pinned host blocks too (with 256 KiB as the smallest parked block a
// context's teardown was 2 ms, the ~30 smaller ones it still freed one by one; with 4 KiB 0.4 ms).  A context holds some fifty device and
// twenty-seven pinned buffers; giving them back to the driver one hipFree / hipHostFree at a time cost a `map` run 20 - 27 ms of
// its 95 (four contexts), and getting them again a part of its set-up.  What stays parked beyond the context that owned the
// reads is capped (DP_DEV_CACHE_MB, default 16384 - a config-3 `map` run parks 7.5 GB: six threads' contexts with a chain pool of 1 GB
// each; with 4096 half of it went back to the driver at the end of every run and was fetched again by the next; DP_PIN_CACHE_MB, default 2048); dp_release_device_caches() empties both.
namespace {
struct BigBlock {
    size_t cap;
    int device;
};
struct Parked {
    void* p;
    uint64_t seq;  // when it was parked: trimming lets the blocks go that have been lying here longest
};
struct BigCache {
    std::mutex mu;
    typedef std::multimap<std::pair<int, size_t>, Parked> Free;
    Free free_;                                 // (device, capacity) -> block
    std::map<uint64_t, Free::iterator> by_age;  // parking order -> the same blocks: trimming takes from the front
    std::unordered_map<void*, BigBlock> live;   // blocks handed out (>= the cache's smallest size)
    void park(int device, size_t cap, void* p) {  // (under mu)
        const uint64_t s = seq++;
        by_age.emplace(s, free_.emplace(std::make_pair(device, cap), Parked{p, s}));
        cached += cap;
    }
    void unpark(Free::iterator it) {  // (under mu)
        cached -= it->first.second;
        by_age.erase(it->second.seq);
        free_.erase(it);
    }
    size_t cached = 0;
    uint64_t seq = 0;
};
BigCache& big_cache() {
    static BigCache* c = new BigCache();  // (never destroyed: contexts may be torn down from static destructors)
    return *c;
}
BigCache& pin_cache() {
    static BigCache* c = new BigCache();
    return *c;
}
constexpr size_t kBig = (size_t)4 << 10;
constexpr size_t kPinMin = (size_t)4 << 10;
size_t cache_cap(const char* env, size_t dflt_mb) {
    const char* e = getenv(env);
    return (size_t)(e ? std::max(0, atoi(e)) : (int)dflt_mb) << 20;
}
struct DeviceGuard {  // makes `device` current for the scope, then restores the caller's
    int prev = -1;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) (void)hipSetDevice(device);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
// Blocks below 256 MB come in size classes (four per power of two): the per-round buffers grow by factors that depend on what a
// thread happened to be given, and a cache of exact sizes filled up with blocks nobody asked for again - 0.8 GB per config-3 `map` run.
size_t size_class(size_t bytes) {
    if (bytes >= ((size_t)256 << 20) || bytes < 4096) return bytes;
    size_t p = 4096;
    while (p * 2 <= bytes) p *= 2;
    const size_t step = p / 4;
    return (bytes + step - 1) / step * step;
}
// takes a parked block of about `bytes` (at most a quarter more) out of the cache
void* cache_take(BigCache& c, int dev, size_t bytes) {
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.free_.lower_bound({dev, bytes});
    if (it == c.free_.end() || it->first.first != dev || it->first.second > bytes + bytes / 4) return nullptr;
    void* p = it->second.p;
    c.live[p] = BigBlock{it->first.second, dev};
    c.unpark(it);
    return p;
}
// parked blocks go back to the driver until at most `keep` bytes stay - the ones parked longest ago first: what every run takes out
// and parks again (the packed reads, the staging copy: hundreds of MB, a hipMalloc of 0.4 - 0.6 s each) is young, what piles up is the
// buffers whose sizes differ from run to run and that nobody asks for again.  (Largest first - the first version - let exactly the
// expensive blocks go once the pile reached the cap: every fifth config-3 `map` run of a process took 0.51 s instead of 0.06.)
template <class F>
void cache_trim(BigCache& c, size_t keep, F&& release) {
    std::vector<std::pair<int, void*>> drop;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        while (c.cached > keep && !c.by_age.empty()) {
            auto old = c.by_age.begin()->second;
            drop.push_back({old->first.first, old->second.p});
            c.unpark(old);
        }
    }
    for (auto& q : drop) {
        DeviceGuard g(q.first);
        release(q.second);
    }
}
}  // namespace

size_t dp_dev_cached_bytes() {
    BigCache& c = big_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    return c.cached;
}

void dp_dev_trim() {
    cache_trim(big_cache(), 0, [](void* p) { (void)hipFree(p); });
}
// (a context that owned reads goes: what it and its borrowers parked stays for the next one, up to the cap)
// The cap is DP_DEV_CACHE_MB (default 16384) and never more than a quarter of what the device would have free with the cache empty:
// on a GPU somebody else has filled - torch in the same process, another process - little stays parked.  Embedders that live long
// call dp_release_device_caches() when they are done with a read set.
static void dp_dev_trim_to_cap() {
    static const size_t cap = cache_cap("DP_DEV_CACHE_MB", 16384);
    size_t keep = cap, free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) keep = std::min(keep, (free_b + dp_dev_cached_bytes()) / 4);
    else (void)hipGetLastError();
    cache_trim(big_cache(), keep, [](void* p) { (void)hipFree(p); });
}

namespace {
void kits_release();  // (parked streams and events, below)
}
extern "C" int64_t dp_release_device_caches() {
    const size_t before = dp_dev_cached_bytes() + [] {
        BigCache& c = pin_cache();
        std::lock_guard<std::mutex> lk(c.mu);
        return c.cached;
    }();
    dp_dev_trim();
    cache_trim(pin_cache(), 0, [](void* p) { (void)hipHostFree(p); });
    kits_release();
    return (int64_t)before;
}

hipError_t dp_dev_malloc(void** p, size_t bytes) {
    *p = nullptr;
    if (bytes < kBig) return hipMalloc(p, bytes);
    bytes = size_class(bytes);
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    BigCache& c = big_cache();
    if ((*p = cache_take(c, dev, bytes)) != nullptr) return hipSuccess;
    const double t0 = alloc_trace() ? alloc_now() : 0;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {  // out of memory with blocks parked in the cache: give them back and try again
        (void)hipGetLastError();
        dp_dev_trim();
        e = hipMalloc(p, bytes);
    }
    if (alloc_trace() && bytes >= ((size_t)1 << 20)) fprintf(stderr, "[alloc] device block %zu bytes on device %d from the driver, %.3f ms (parked now: %zu bytes)\n", bytes, dev, 1e3 * (alloc_now() - t0), dp_dev_cached_bytes());
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(c.mu);
        c.live[*p] = BigBlock{bytes, dev};
    }
    return e;
}

// quiet: the caller has already waited for everything that could use the block (dp_ctx_destroy: one wait for all its blocks)
static hipError_t dev_free_impl(void* p, bool quiet) {
    if (!p) return hipSuccess;
    BigCache& c = big_cache();
    BigBlock b{0, 0};
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.live.find(p);
        if (it != c.live.end()) {
            b = it->second;
            c.live.erase(it);
        }
    }
    if (!b.cap) return hipFree(p);
    // what hipFree promises its caller: nothing on the block's device uses it any more
    hipError_t e = hipSuccess;
    if (!quiet) {
        DeviceGuard g(b.device);
        e = hipDeviceSynchronize();
    }
    std::lock_guard<std::mutex> lk(c.mu);
    c.park(b.device, b.cap, p);
    return e;
}
hipError_t dp_dev_free(void* p) { return dev_free_impl(p, false); }

// pinned host blocks: the same cache (their users are a context's own stream, which the context waits for before it lets go)
hipError_t dp_pin_malloc(void** p, size_t bytes) {
    *p = nullptr;
    if (bytes >= kPinMin) bytes = size_class(bytes);
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    BigCache& c = pin_cache();
    if (bytes >= kPinMin && (*p = cache_take(c, dev, bytes)) != nullptr) return hipSuccess;
    hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        cache_trim(c, 0, [](void* q) { (void)hipHostFree(q); });
        e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    }
    if (e == hipSuccess && bytes >= kPinMin) {
        std::lock_guard<std::mutex> lk(c.mu);
        c.live[*p] = BigBlock{bytes, dev};
    }
    return e;
}
void dp_pin_free(void* p) {
    if (!p) return;
    static const size_t cap = cache_cap("DP_PIN_CACHE_MB", 2048);
    BigCache& c = pin_cache();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.live.find(p);
        if (it != c.live.end()) {
            const BigBlock b = it->second;
            c.live.erase(it);
            if (b.cap <= cap) {
                c.park(b.device, b.cap, p);
                p = nullptr;
            }
        }
    }
    if (p) (void)hipHostFree(p);
    else cache_trim(c, cap, [](void* q) { (void)hipHostFree(q); });  // (room is made by the blocks parked longest ago)
}

struct ZeroArgs {
    unsigned long long* p[4];
    unsigned long long n8[4];
    unsigned long long* dst[2];
    const unsigned long long* src[2];
    unsigned long long m8[2];
};
struct zero_regions_kernel {
    enum { THREADS = 256 };
    static __device__ void run(const ZeroArgs a) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (unsigned long long)gridDim.x * blockDim.x;
#pragma unroll
    for (int r = 0; r < 2; r++)  // (the fetches first: their loads cross the link while the stores below go out)
        for (unsigned long long j = i; j < a.m8[r]; j += stride) a.dst[r][j] = __builtin_nontemporal_load(&a.src[r][j]);
#pragma unroll
    for (int r = 0; r < 4; r++)
        for (unsigned long long j = i; j < a.n8[r]; j += stride) a.p[r][j] = 0ull;
}
};
// (a round has a dozen buffers to clear, and every hipMemsetAsync is a dispatch of its own: the command processor, not the
// memory, is what they cost - tools/micro/launch_rate.hip)
int dp_zero_fetch_regions(dp_ctx* ctx, const dp_zero_region* z, int nz, const dp_fetch_region* f, int nf) {
    ZeroArgs a;
    unsigned long long most = 0;
    for (int i = 0; i < 4; i++) {
        a.p[i] = i < nz ? (unsigned long long*)z[i].p : nullptr;
        a.n8[i] = i < nz && z[i].p ? (z[i].bytes + 7) / 8 : 0;
        most = std::max(most, a.n8[i]);
    }
    for (int i = 0; i < 2; i++) {
        a.dst[i] = i < nf ? (unsigned long long*)f[i].dst : nullptr;
        a.src[i] = i < nf ? (const unsigned long long*)f[i].src : nullptr;
        a.m8[i] = i < nf && f[i].dst && f[i].src ? (f[i].bytes + 7) / 8 : 0;
        most = std::max(most, a.m8[i]);
    }
    if (!most) return DP_OK;
    const uint32_t blocks = (uint32_t)std::min<unsigned long long>(4096, (most + 1023) / 1024);
    dp_launch<zero_regions_kernel>(ctx, dim3(std::max(1u, blocks)), dim3(256), a);
    DP_HIP(hipGetLastError());
    return DP_OK;
}
int dp_zero_regions(dp_ctx* ctx, const dp_zero_region* r, int n) { return dp_zero_fetch_regions(ctx, r, n, nullptr, 0); }

extern "C" const char* dp_version(void) { return "downpore_hip 0.1 (gfx950)"; }

extern "C" const char* dp_last_error(const dp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

// A context's stream and events outlive it: creating a stream takes the runtime 3 - 4 ms, one after another whatever the thread (a
// config-3 `map` run creates seven contexts: its sixth mapper thread started 20 ms after the first), so an idle stream is parked with
// its events when its context goes and handed to the next context on that device.  dp_release_device_caches() destroys the parked
// ones; streams whose priority was changed (dp_ctx_set_priority) are not kept.
namespace {
struct StreamKit {
    int device;
    hipStream_t stream;
    hipEvent_t ev[sizeof(dp_ctx::ev) / sizeof(dp_ctx::ev[0])];
    hipEvent_t ev_sync;
};
std::mutex g_kit_mu;
std::vector<StreamKit> g_kits;
constexpr size_t kMaxKits = 32;
bool kit_take(int device, dp_ctx* ctx) {
    std::lock_guard<std::mutex> lk(g_kit_mu);
    for (size_t i = g_kits.size(); i-- > 0;)
        if (g_kits[i].device == device) {
            ctx->stream = g_kits[i].stream;
            for (size_t j = 0; j < sizeof(ctx->ev) / sizeof(ctx->ev[0]); j++) ctx->ev[j] = g_kits[i].ev[j];
            ctx->ev_sync = g_kits[i].ev_sync;
            g_kits.erase(g_kits.begin() + (long)i);
            return true;
        }
    return false;
}
void kit_destroy(StreamKit& k) {
    for (auto& ev : k.ev)
        if (ev) hipEventDestroy(ev);
    if (k.ev_sync) hipEventDestroy(k.ev_sync);
    if (k.stream) hipStreamDestroy(k.stream);
}
// (the context's stream is idle: dp_ctx_destroy has waited for it)
void kit_put(dp_ctx* ctx) {
    StreamKit k;
    k.device = ctx->device;
    k.stream = ctx->stream;
    for (size_t j = 0; j < sizeof(ctx->ev) / sizeof(ctx->ev[0]); j++) k.ev[j] = ctx->ev[j];
    k.ev_sync = ctx->ev_sync;
    bool all = k.stream != nullptr && k.ev_sync != nullptr && !ctx->stream_priority_set;
    for (auto& ev : k.ev) all = all && ev != nullptr;
    {
        std::lock_guard<std::mutex> lk(g_kit_mu);
        if (all && g_kits.size() < kMaxKits) {
            g_kits.push_back(k);
            return;
        }
    }
    kit_destroy(k);
}
void kits_release() {
    std::vector<StreamKit> drop;
    {
        std::lock_guard<std::mutex> lk(g_kit_mu);
        drop.swap(g_kits);
    }
    for (StreamKit& k : drop) {
        DeviceGuard g(k.device);
        kit_destroy(k);
    }
}
}  // namespace

static std::mutex g_borrow_mu;  // borrower counts and pending owner destroys (dp_ctx_create_shared / dp_ctx_destroy)

extern "C" int dp_ctx_create(int device, dp_ctx** out) {
    if (!out) return DP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return dp_fail(nullptr, DP_ERR_NODEVICE, "no HIP device (this library has no CPU fallback)", e);
    if (device < 0 || device >= n) return dp_fail(nullptr, DP_ERR_ARG, "device index out of range");
    dp_ctx* ctx = new dp_ctx();
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess) {
        dp_fail(nullptr, DP_ERR_HIP, "hipSetDevice", e);
        delete ctx;
        return DP_ERR_HIP;
    }
    if (!kit_take(device, ctx)) {
        if ((e = hipStreamCreate(&ctx->stream)) != hipSuccess) {
            dp_fail(nullptr, DP_ERR_HIP, "hipStreamCreate", e);
            delete ctx;
            return DP_ERR_HIP;
        }
        for (auto& ev : ctx->ev) hipEventCreate(&ev);
        hipEventCreateWithFlags(&ctx->ev_sync, hipEventBlockingSync | hipEventDisableTiming);
    }
    *out = ctx;
    return DP_OK;
}

extern "C" int dp_ctx_create_shared(dp_ctx* src, dp_ctx** out) {
    if (!src || !out) return DP_ERR_ARG;
    int rc = dp_ctx_create(src->device, out);
    if (rc != 0) return rc;
    dp_ctx* c = *out;
    c->borrowed_reads = true;
    c->owner = src->owner ? src->owner : src;
    {
        std::lock_guard<std::mutex> lk(g_borrow_mu);
        c->owner->n_borrowers++;
    }
    c->n_reads = src->n_reads;
    c->total_bases = src->total_bases;
    c->packed_bytes = src->packed_bytes;
    c->d_packed = src->d_packed;
    c->d_boff = src->d_boff;
    c->d_len = src->d_len;
    c->h_boff = src->h_boff;
    c->h_len = src->h_len;
    c->d_values = src->d_values;
    c->n_values = src->n_values;
    c->d_qual = src->d_qual;
    c->d_qualoff = src->d_qualoff;
    c->d_hasq = src->d_hasq;
    return DP_OK;
}

extern "C" int dp_ctx_set_priority(dp_ctx* ctx, int high) {
    if (!ctx) return DP_ERR_ARG;
    hipSetDevice(ctx->device);
    int least = 0, greatest = 0;
    DP_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));  // numerically lower = higher priority
    DP_HIP(hipStreamSynchronize(ctx->stream));
    hipStream_t s = nullptr;
    DP_HIP(hipStreamCreateWithPriority(&s, hipStreamDefault, high ? greatest : least));
    hipStreamDestroy(ctx->stream);
    ctx->stream = s;
    ctx->stream_priority_set = true;
    return DP_OK;
}

static int upload_join(dp_ctx* ctx);  // (dp_reads_upload_rc_begin's thread, below)
extern "C" void dp_ctx_destroy(dp_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    (void)upload_join(ctx);  // (a read set still travelling: its thread uses this context's buffers)
    dp_stream_sync(ctx);
    {
        // contexts that borrow these reads (value table, k-mer index) are still alive - a garbage-collected host may finalise
        // the owner first: the resident data stays until the last borrower has gone, which then carries this call out.  "Are
        // there borrowers" + "remember the destroy" on this side and "was I the last" + "is a destroy pending" on the borrower's
        // side are one decision each, under one lock: exactly one of the two threads frees the owner.
        std::lock_guard<std::mutex> lk(g_borrow_mu);
        if (!ctx->owner && ctx->n_borrowers.load() > 0) {
            ctx->destroy_pending = true;
            return;
        }
    }
    if (ctx->borrowed_reads) ctx->d_packed.p = ctx->d_boff.p = ctx->d_len.p = ctx->d_values.p = ctx->d_qual.p = ctx->d_qualoff.p = ctx->d_hasq.p = nullptr;
    dp_ctx* const lender = ctx->owner;
    bool last_borrower = false;
    if (lender) {
        std::lock_guard<std::mutex> lk(g_borrow_mu);
        last_borrower = --lender->n_borrowers == 0 && lender->destroy_pending;
    }
    dp_kindex_free(ctx);
    dp_find_state_free(ctx);
    DevBuf* dbs[] = {&ctx->d_packed, &ctx->d_boff, &ctx->d_len, &ctx->d_bits, &ctx->d_kmap, &ctx->d_seeds, &ctx->d_items,
                     &ctx->d_counts, &ctx->d_segoff, &ctx->d_segs, &ctx->d_total, &ctx->d_seqrefs, &ctx->d_posting,
                     &ctx->d_seedsets, &ctx->d_pmeta, &ctx->d_qsegs, &ctx->d_qoff, &ctx->d_qsets, &ctx->d_qmeta,
                     &ctx->d_cand, &ctx->d_pool, &ctx->d_mrec, &ctx->d_ma, &ctx->d_mb, &ctx->d_cursor, &ctx->d_sched, &ctx->d_ignore, &ctx->d_surv, &ctx->d_values, &ctx->d_selwin, &ctx->d_seltop, &ctx->d_cin, &ctx->d_cout,
                     &ctx->d_kx_sz, &ctx->d_kx_lo, &ctx->d_kx_tmp, &ctx->d_kx_keys, &ctx->d_kx_vals, &ctx->d_manchor, &ctx->d_seeds_applied,
                     &ctx->d_pbase, &ctx->d_qbig, &ctx->d_pspec, &ctx->d_clist, &ctx->d_sa, &ctx->d_sb, &ctx->d_qual, &ctx->d_qualoff, &ctx->d_hasq, &ctx->d_cretry,
                     &ctx->d_chunk_meta, &ctx->d_nseqs};
    // one wait for everything this context's blocks could still be used by (its own stream is idle since dp_stream_sync above; a
    // gang's launches and a borrower's copies run on other streams of the device), then its blocks are parked without further waits
    (void)hipDeviceSynchronize();
    for (auto* b : dbs)
        if (b->p) dev_free_impl(b->p, true);
    if (ctx->d_kcounts) dev_free_impl(ctx->d_kcounts, true);
    for (void* q : ctx->retired_dev) dev_free_impl(q, true);
    for (void* q : ctx->retired_pin) dp_pin_free(q);
    PinBuf* pbs[] = {&ctx->h_counts, &ctx->h_segoff, &ctx->h_segs, &ctx->h_total, &ctx->h_mrec, &ctx->h_ma, &ctx->h_mb,
                     &ctx->h_seeds, &ctx->h_spack, &ctx->h_extra, &ctx->h_cursor, &ctx->h_cand, &ctx->h_cand_off, &ctx->h_cand_list, &ctx->h_mq, &ctx->h_mt, &ctx->h_moff, &ctx->h_surv, &ctx->h_ta, &ctx->h_tb, &ctx->h_qm, &ctx->h_qup, &ctx->h_seltop, &ctx->h_cin, &ctx->h_cout, &ctx->h_manchor, &ctx->h_manout, &ctx->h_ignore};
    for (auto* b : pbs)
        if (b->p) dp_pin_free(b->p);
    kit_put(ctx);
    const bool owner = !ctx->borrowed_reads;
    delete ctx;
    if (owner) dp_dev_trim_to_cap();  // (a context that owned reads goes: what is parked beyond it is bounded)
    if (last_borrower) dp_ctx_destroy(lender);
}

extern "C" uint32_t dp_reads_count(const dp_ctx* ctx) { return ctx ? ctx->n_reads : 0; }
extern "C" uint64_t dp_reads_total_bases(const dp_ctx* ctx) { return ctx ? ctx->total_bases : 0; }

// ---------------------------------------------------------------------------------------------------------------
// A1: 2-bit packing on the device.  One thread produces one packed dword (16 bases).

__global__ void pack_kernel(const uint8_t* __restrict__ ascii, const int64_t* __restrict__ aoff,
                            const uint64_t* __restrict__ boff, uint32_t n_reads, uint32_t* __restrict__ packed,
                            uint64_t n_dwords, const uint32_t* __restrict__ srcmap, uint64_t d_begin = 0) {
    uint64_t d = d_begin + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (dwords [d_begin, n_dwords) of the packed layout)
    if (d >= n_dwords) return;
    uint64_t byte = d * 4;
    // binary search: last read with boff[r] <= byte
    uint32_t lo = 0, hi = n_reads;  // boff has n_reads+1 entries; boff[n_reads] = total
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (boff[mid] <= byte) lo = mid;
        else hi = mid;
    }
    // srcmap (dp_reads_upload_rc): device read lo is host read srcmap>>1, reverse-complemented when bit 0 is set
    const uint32_t src_read = srcmap ? (srcmap[lo] >> 1) : lo;
    const bool rc = srcmap ? (srcmap[lo] & 1u) != 0 : false;
    int64_t len = aoff[src_read + 1] - aoff[src_read];
    int64_t base0 = (int64_t)(byte - boff[lo]) * 4;
    const uint8_t* src = ascii + aoff[src_read];
    uint32_t out = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int64_t p = base0 + b * 4 + j;
            uint32_t c = 0;
            if (p < len) {
                uint32_t ch = src[rc ? (len - 1 - p) : p];
                c = ((ch >> 1) ^ ((ch & 4) >> 2)) & 3;  // sequence/sequence.go:59,80
                if (rc) c = 3u - c;                      // complement (sequence.go:185-189)
            }
            v = (v << 2) | c;
        }
        out |= v << (8 * b);  // little-endian dword: byte b of the packed stream
    }
    packed[d] = out;
}

// Shared body of dp_reads_upload / dp_reads_upload_rc.  With first_paired < n_reads every host read r >= first_paired
// becomes TWO device reads: first_paired + 2*(r - first_paired) (forward) and the next id (its reverse complement,
// produced by the pack kernel; nothing but the forward ASCII crosses PCIe).
// The ASCII bases of a read set on their way to the device.  From pageable memory a gigabyte travels at 13-18 GB/s (the runtime
// stages it piece by piece on one thread: 55-75 ms for config 2's reads); here a few helper threads copy 4 MiB pieces into a ring of
// pinned blocks and every piece is sent on as soon as it is complete (pinned to device: ~40 GB/s), so the host copies, not the
// link, set the pace.  Small inputs go the plain way.  DP_UPLOAD_THREADS (default 4; 0 = plain copy; measured at config 2: 54 ms plain, 33-35 ms with 4, 6 or 8 helpers).
namespace {
std::mutex g_ring_mu;  // held by whoever uses the ring, for as long as it does
uint8_t* g_ring = nullptr;
size_t g_ring_bytes = 0;
uint8_t* ring_get(size_t bytes) {  // (caller holds g_ring_mu) null: no pinned memory to be had
    if (g_ring_bytes < bytes) {
        if (g_ring) hipHostFree(g_ring);
        g_ring = nullptr;
        g_ring_bytes = 0;
        if (hipHostMalloc((void**)&g_ring, bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            g_ring = nullptr;
            return nullptr;
        }
        g_ring_bytes = bytes;
    }
    return g_ring;
}
int upload_threads() {
    static const int n_thr = (int)std::max(0L, std::min(16L, dp_tune("upload_threads", 4)));
    return n_thr;
}
}  // namespace
static int upload_ascii(dp_ctx* ctx, void* d_dst, const uint8_t* src, uint64_t n) {
    const int n_thr = upload_threads();
    constexpr uint64_t kPiece = (uint64_t)4 << 20;
    if (n_thr == 0 || n < 16 * kPiece) {
        DP_HIP(hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, ctx->stream));
        return DP_OK;
    }
    const int n_slots = 2 * n_thr;
    // the ring is kept for the process (48 MiB pinned; pinning costs ~0.1 ms per MiB) and used by one upload at a time
    std::lock_guard<std::mutex> ring_lock(g_ring_mu);
    uint8_t* ring = ring_get((size_t)n_slots * kPiece);
    if (!ring) {
        DP_HIP(hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, ctx->stream));
        return DP_OK;
    }
    const uint64_t n_pieces = (n + kPiece - 1) / kPiece;
    std::vector<hipEvent_t> sent((size_t)n_slots, nullptr);
    for (auto& e : sent) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    std::mutex mu;
    std::condition_variable cv;
    std::vector<uint8_t> filled((size_t)n_pieces, 0);
    uint64_t next_fill = 0, next_send = 0;  // pieces handed to a helper / sent on, in order
    bool failed = false;
    auto helper = [&] {
        for (;;) {
            uint64_t i;
            {
                std::unique_lock<std::mutex> lk(mu);
                // a slot is free again once the piece that used it last has been sent and its copy has completed (checked by the
                // sender before it lets next_send pass it)
                cv.wait(lk, [&] { return failed || next_fill >= n_pieces || next_fill < next_send + (uint64_t)n_slots; });
                if (failed || next_fill >= n_pieces) return;
                i = next_fill++;
            }
            const uint64_t b = i * kPiece, len = std::min(kPiece, n - b);
            memcpy(ring + (i % (uint64_t)n_slots) * kPiece, src + b, len);
            {
                std::lock_guard<std::mutex> lk(mu);
                filled[(size_t)i] = 1;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_thr; t++) th.emplace_back(helper);
    hipError_t err = hipSuccess;
    for (uint64_t i = 0; i < n_pieces && err == hipSuccess; i++) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return filled[(size_t)i] != 0; });
        }
        const uint64_t b = i * kPiece, len = std::min(kPiece, n - b);
        const size_t slot = (size_t)(i % (uint64_t)n_slots);
        err = hipMemcpyAsync((uint8_t*)d_dst + b, ring + slot * kPiece, len, hipMemcpyHostToDevice, ctx->stream);
        if (err == hipSuccess) err = hipEventRecord(sent[slot], ctx->stream);
        // the slot piece i + 1 - n_slots .. may be refilled only when its copy is through: wait for the oldest outstanding one
        if (err == hipSuccess && i + 1 >= (uint64_t)n_slots / 2) {
            const uint64_t done_upto = i + 1 - (uint64_t)n_slots / 2;  // pieces < done_upto + 1 must have left their slots
            err = hipEventSynchronize(sent[(size_t)(done_upto % (uint64_t)n_slots)]);
            std::lock_guard<std::mutex> lk(mu);
            next_send = done_upto + 1;
        }
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (err != hipSuccess) failed = true;
        next_send = n_pieces + (uint64_t)n_slots;
    }
    cv.notify_all();
    for (auto& t : th) t.join();
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    for (auto& e : sent) hipEventDestroy(e);
    if (err != hipSuccess) return dp_fail(ctx, DP_ERR_HIP, "dp_reads_upload: staged copy", err);
    return DP_OK;
}

// What both forms of the upload do first: the new read set's tables (packed offsets, lengths) resident, room for the packed reads,
// the staging copies' device blocks and the two small tables the pack kernel reads (ASCII offsets, source map) on their way.
struct UploadPrep {
    void *d_ascii = nullptr, *d_aoff = nullptr, *d_map = nullptr;
    std::vector<int64_t> rel;      // ASCII offset of every host read (relative to the first)
    std::vector<uint32_t> srcmap;  // device read -> host read << 1 | reverse complement
    uint64_t nascii = 0, pos = 0;
    uint32_t n_reads = 0;
    bool nothing = false;          // no bases at all: the tables are resident, nothing travels
    bool packed_input = false;     // dp_reads_upload_packed_rc: no ASCII staging block
};
static int upload_join(dp_ctx* ctx);
static int reads_upload_prepare(dp_ctx* ctx, const uint8_t* bases, const int64_t* off, uint32_t n_host, uint32_t& first_paired, UploadPrep& P) {
    if (!ctx || !bases || !off) return DP_ERR_ARG;
    if (ctx->borrowed_reads) return dp_fail(ctx, DP_ERR_STATE, "dp_reads_upload on a context that borrows its reads");
    // contexts made with dp_ctx_create_shared hold plain copies of the resident buffers' addresses
    if (ctx->n_borrowers.load() > 0) return dp_fail(ctx, DP_ERR_STATE, "dp_reads_upload while contexts borrowing these reads exist");
    if (int rc = upload_join(ctx)) return rc;  // (a read set that was still travelling)
    if (first_paired > n_host) first_paired = n_host;
    hipSetDevice(ctx->device);
    dp_kindex_free(ctx);  // a position index of the previous read set is void
    for (DevBuf* qb : {&ctx->d_qual, &ctx->d_qualoff, &ctx->d_hasq})  // ... and so are its quality bytes
        if (qb->p) {
            dp_dev_free(qb->p);
            qb->p = nullptr;
            qb->cap = 0;
        }
    if (ctx->d_kcounts) {   // ... and so is its k-mer histogram
        dp_dev_free(ctx->d_kcounts);
        ctx->d_kcounts = nullptr;
        ctx->kcounts_k = 0;
    }
    const uint64_t nd64 = (uint64_t)first_paired + 2ull * (n_host - first_paired);
    if (nd64 > 0x7fffffffull) return dp_fail(ctx, DP_ERR_ARG, "too many reads");
    const uint32_t n_reads = (uint32_t)nd64;
    const bool paired = first_paired < n_host;
    std::vector<uint64_t> h_boff((size_t)n_reads + 1, 0);
    std::vector<uint32_t> h_len(n_reads, 0);
    if (paired) P.srcmap.resize(n_reads);
    uint64_t pos = 0, total = 0;
    for (uint32_t d = 0; d < n_reads; d++) {
        const uint32_t r = d < first_paired ? d : first_paired + (d - first_paired) / 2;
        const uint32_t isrc = d < first_paired ? 0u : ((d - first_paired) & 1u);
        int64_t len = off[r + 1] - off[r];
        if (len < 0 || len > 0x7fffffff) return dp_fail(ctx, DP_ERR_ARG, "read length out of range");
        h_boff[d] = pos;
        h_len[d] = (uint32_t)len;
        if (paired) P.srcmap[d] = (r << 1) | isrc;
        pos += ((uint64_t)(len + 3) / 4 + 15) & ~(uint64_t)15;
        total += (uint64_t)len;
    }
    h_boff[n_reads] = pos;
    // A resident buffer the new read set outgrows is released here and now (this call waits for the stream anyway) instead
    // of joining the retired list: a previous read set's gigabytes must not stay in HBM for the context's lifetime, and
    // they count against the k-mer index's free-memory test.
    DP_HIP(dp_stream_sync(ctx));
    {
        DevBuf* res[] = {&ctx->d_packed, &ctx->d_boff, &ctx->d_len};
        const size_t need[] = {(size_t)pos + 64, ((size_t)n_reads + 1) * 8, (size_t)n_reads * 4 + 4};
        for (int i = 0; i < 3; i++)
            if (res[i]->p && need[i] > res[i]->cap) {
                dp_dev_free(res[i]->p);
                res[i]->p = nullptr;
                res[i]->cap = 0;
            }
    }
    ctx->n_reads = n_reads;
    ctx->ignore_epoch = ~0ull;  // (cached per-read-set state of dp_scan_reads)
    ctx->ignore_shadow_valid = false;
    ctx->items_ptr = nullptr;
    ctx->h_boff.swap(h_boff);
    ctx->h_len.swap(h_len);
    ctx->packed_bytes = pos;
    ctx->total_bases = total;
    if (dev_reserve(ctx, ctx->d_packed, pos + 64)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_boff, ((size_t)n_reads + 1) * 8)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_len, (size_t)n_reads * 4 + 4)) return DP_ERR_HIP;
    DP_HIP(hipMemcpyAsync(ctx->d_boff.p, ctx->h_boff.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    DP_HIP(hipMemcpyAsync(ctx->d_len.p, ctx->h_len.data(), (size_t)n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
    DP_HIP(hipMemsetAsync((uint8_t*)ctx->d_packed.p + pos, 0, 64, ctx->stream));
    P.n_reads = n_reads;
    P.pos = pos;
    if (n_reads == 0 || pos == 0) {
        DP_HIP(dp_stream_sync(ctx));
        P.nothing = true;
        return DP_OK;
    }
    if (!P.packed_input) {
        P.nascii = (uint64_t)(off[n_host] - off[0]);
        DP_HIP(dp_dev_malloc(&P.d_ascii, P.nascii + 16));
        DP_HIP(dp_dev_malloc(&P.d_aoff, ((size_t)n_host + 1) * 8));
        P.rel.resize((size_t)n_host + 1);
        for (uint32_t r = 0; r <= n_host; r++) P.rel[r] = off[r] - off[0];
        DP_HIP(hipMemcpyAsync(P.d_aoff, P.rel.data(), ((size_t)n_host + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    if (paired) {
        DP_HIP(dp_dev_malloc(&P.d_map, (size_t)n_reads * 4));
        DP_HIP(hipMemcpyAsync(P.d_map, P.srcmap.data(), (size_t)n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    return DP_OK;
}

static int reads_upload_impl(dp_ctx* ctx, const uint8_t* bases, const int64_t* off, uint32_t n_host, uint32_t first_paired) {
    UploadPrep P;
    struct Temps {  // staging copies: released on every way out, error returns included (after the stream has drained)
        dp_ctx* c;
        UploadPrep& p;
        ~Temps() {
            if (!c) return;
            hipStreamSynchronize(c->stream);
            for (void* q : {p.d_ascii, p.d_aoff, p.d_map})
                if (q) dp_dev_free(q);
        }
    } temps{ctx, P};
    if (int rc = reads_upload_prepare(ctx, bases, off, n_host, first_paired, P)) return rc;
    if (P.nothing) return DP_OK;
    if (int rc = upload_ascii(ctx, P.d_ascii, bases + off[0], P.nascii)) return rc;
    uint64_t n_dwords = P.pos / 4;
    uint32_t blocks = (uint32_t)((n_dwords + 255) / 256);
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const uint8_t*)P.d_ascii, (const int64_t*)P.d_aoff,
                       (const uint64_t*)ctx->d_boff.p, P.n_reads, (uint32_t*)ctx->d_packed.p, n_dwords, (const uint32_t*)P.d_map, (uint64_t)0);
    DP_HIP(hipGetLastError());
    DP_HIP(dp_stream_sync(ctx));
    return DP_OK;
}

extern "C" int dp_reads_upload(dp_ctx* ctx, const uint8_t* bases, const int64_t* off, uint32_t n_reads) {
    return reads_upload_impl(ctx, bases, off, n_reads, n_reads);
}

extern "C" int dp_reads_upload_rc(dp_ctx* ctx, const uint8_t* bases, const int64_t* off, uint32_t n_reads, uint32_t first_paired) {
    return reads_upload_impl(ctx, bases, off, n_reads, first_paired);
}

// ---- reads that arrive packed (round 6).  The reference keeps a read as 2 bits per base from the moment it is read from its file
// (sequence.packedSequence, sequence/sequence.go:22-31: data []byte of ceil(len / 4) bytes, first base in a byte's top bits); a host that
// holds them that way hands them over that way - a quarter of the bytes over PCIe - and the device lays them out and makes the reverse
// strands from the packed forward ones.
//
// place_packed_kernel: one thread per dword of the device layout.  Forward reads are copied; 16 bases of a reverse strand are the
// 16 bases of the forward strand that end at len - 1 - base0, their order reversed (byte swap, nibble swap, pair swap) and
// complemented (3 - code = every bit inverted), bases beyond the read's end zero as everywhere in the layout.
__global__ void place_packed_kernel(const uint32_t* __restrict__ src, const uint64_t* __restrict__ soff, const uint32_t* __restrict__ slen,
                                    const uint64_t* __restrict__ boff, uint32_t n_reads, uint32_t* __restrict__ packed, uint64_t n_dwords,
                                    const uint32_t* __restrict__ srcmap) {
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_dwords) return;
    const uint64_t byte = d * 4;
    uint32_t lo = 0, hi = n_reads;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (boff[mid] <= byte) lo = mid;
        else hi = mid;
    }
    const uint32_t src_read = srcmap ? (srcmap[lo] >> 1) : lo;
    const bool rc = srcmap ? (srcmap[lo] & 1u) != 0 : false;
    const int64_t len = (int64_t)slen[src_read];
    const uint64_t rel = byte - boff[lo];              // byte of the read this dword starts at
    const uint64_t sbytes = ((uint64_t)len + 3) / 4;   // packed bytes of the read
    const uint32_t* sp = src + soff[src_read] / 4;     // (16-byte aligned; its padding up to the next read is zero)
    uint32_t out = 0;
    if (!rc) {
        if (rel < sbytes) {
            out = sp[rel / 4];
            const int64_t nb = len - (int64_t)rel * 4;  // bases of the read from this dword on
            if (nb < 16) out = __builtin_bswap32(__builtin_bswap32(out) & (0xffffffffu << (2 * (uint32_t)(16 - nb))));  // (whatever the host left in the last byte's unused bits)
        }
    } else {
        const int64_t base0 = (int64_t)rel * 4;
        const int64_t t = len - 16 - base0;  // forward base the window starts at (negative: the window hangs over the read's start)
        if (t > -16) {
            const int64_t tt = t < 0 ? 0 : t;
            const uint64_t b0 = (uint64_t)tt >> 2;                  // byte the window starts in
            const uint32_t d0 = sp[b0 / 4], d1 = sp[b0 / 4 + 1];    // (the slack behind the last read keeps d1 inside the block)
            const uint64_t x = ((uint64_t)__builtin_bswap32(d0) << 32) | (uint64_t)__builtin_bswap32(d1);  // the stream, first base on top
            const uint32_t sh = 8u * (uint32_t)(b0 & 3u) + 2u * (uint32_t)(tt & 3);
            uint32_t w = (uint32_t)((x << sh) >> 32);               // bases tt .. tt + 15, base tt on top
            if (t < 0) w >>= 2 * (uint32_t)(-t);                    // ... bases 0 .. 15 + t at the bottom, nothing above them
            w = __builtin_bswap32(w);
            w = ((w & 0x0f0f0f0fu) << 4) | ((w >> 4) & 0x0f0f0f0fu);
            w = ((w & 0x33333333u) << 2) | ((w >> 2) & 0x33333333u);  // base order reversed: forward base len - 1 - base0 on top
            w = ~w;
            if (t < 0) w &= 0xffffffffu << (2 * (uint32_t)(-t));    // bases beyond the read's end
            out = __builtin_bswap32(w);                               // byte b of the stream = byte b of the little-endian dword
        }
    }
    packed[d] = out;
}

// host memory handed out by dp_host_alloc is pinned: a copy from it needs no staging
static bool host_block_is_pinned(const void* p) {
    BigCache& c = pin_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    for (auto& kv : c.live) {
        const char* b = (const char*)kv.first;
        if ((const char*)p >= b && (const char*)p < b + kv.second.cap) return true;
    }
    return false;
}

extern "C" void* dp_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (dp_pin_malloc(&p, bytes < kPinMin ? kPinMin : bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
extern "C" void dp_host_free(void* p) { dp_pin_free(p); }

extern "C" int dp_reads_upload_packed_rc(dp_ctx* ctx, const uint8_t* packed, const uint32_t* lens, uint32_t n_host, uint32_t first_paired) {
    if (!ctx || !lens || (!packed && n_host)) return DP_ERR_ARG;
    // the host layout: read r at the sum over the reads before it of its packed bytes rounded up to 16
    std::vector<int64_t> off((size_t)n_host + 1, 0);
    std::vector<uint64_t> soff((size_t)n_host + 1, 0);
    for (uint32_t r = 0; r < n_host; r++) {
        if (lens[r] > 0x7fffffffu) return dp_fail(ctx, DP_ERR_ARG, "read length out of range");
        off[(size_t)r + 1] = off[r] + (int64_t)lens[r];
        soff[(size_t)r + 1] = soff[r] + ((((uint64_t)lens[r] + 3) / 4 + 15) & ~(uint64_t)15);
    }
    UploadPrep P;
    P.packed_input = true;
    struct Temps {
        dp_ctx* c;
        UploadPrep& p;
        void *d_src = nullptr, *d_soff = nullptr, *d_slen = nullptr;
        ~Temps() {
            hipStreamSynchronize(c->stream);
            for (void* q : {p.d_ascii, p.d_aoff, p.d_map, d_src, d_soff, d_slen})
                if (q) dp_dev_free(q);
        }
    } T{ctx, P};
    static const uint8_t nothing = 0;
    if (int rc = reads_upload_prepare(ctx, packed ? packed : &nothing, off.data(), n_host, first_paired, P)) return rc;
    if (P.nothing) return DP_OK;
    const uint64_t nsrc = soff[n_host];
    DP_HIP(dp_dev_malloc(&T.d_src, nsrc + 64));
    DP_HIP(dp_dev_malloc(&T.d_soff, ((size_t)n_host + 1) * 8));
    DP_HIP(dp_dev_malloc(&T.d_slen, (size_t)n_host * 4 + 4));
    DP_HIP(hipMemcpyAsync(T.d_soff, soff.data(), ((size_t)n_host + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    DP_HIP(hipMemcpyAsync(T.d_slen, lens, (size_t)n_host * 4, hipMemcpyHostToDevice, ctx->stream));
    DP_HIP(hipMemsetAsync((uint8_t*)T.d_src + nsrc, 0, 64, ctx->stream));
    if (host_block_is_pinned(packed)) DP_HIP(hipMemcpyAsync(T.d_src, packed, nsrc, hipMemcpyHostToDevice, ctx->stream));
    else if (int rc = upload_ascii(ctx, T.d_src, packed, nsrc)) return rc;
    const uint64_t n_dwords = P.pos / 4;
    hipLaunchKernelGGL(place_packed_kernel, dim3((uint32_t)((n_dwords + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)T.d_src,
                       (const uint64_t*)T.d_soff, (const uint32_t*)T.d_slen, (const uint64_t*)ctx->d_boff.p, P.n_reads, (uint32_t*)ctx->d_packed.p,
                       n_dwords, (const uint32_t*)P.d_map);
    DP_HIP(hipGetLastError());
    DP_HIP(dp_stream_sync(ctx));  // (soff and the caller's lens are read by copies that are through now)
    return DP_OK;
}

// ---- a read set that travels while its first reads are already worked on (round 5: `map` maps read 0 while read 40 000 is on the link)
struct dp_ctx::ReadsUpload {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t ready = 0;   // host reads [0, ready) are packed on the device (both strands)
    uint32_t n_host = 0;
    bool done = false;
    int rc = DP_OK;
    std::string error;
};

static int upload_join(dp_ctx* ctx) {
    if (!ctx) return DP_OK;
    std::shared_ptr<dp_ctx::ReadsUpload> U;
    {
        std::lock_guard<std::mutex> lk(ctx->upload_mu);
        U.swap(ctx->upload);
    }
    if (!U) return DP_OK;
    if (U->th.joinable()) U->th.join();
    std::lock_guard<std::mutex> lk(U->mu);  // (a borrower still inside its wait reads rc under this lock and keeps U alive)
    return U->rc == DP_OK ? DP_OK : dp_fail(ctx, U->rc, U->error.c_str());
}

// The thread that carries an upload: pieces of ASCII through the pinned ring to the device on a stream of its own, behind every piece the
// pack kernel for the reads that are complete with it, and the host told how far that is whenever a piece's event has fired.
static void upload_thread(dp_ctx* ctx, dp_ctx::ReadsUpload* U, const uint8_t* src, UploadPrep P, uint32_t first_paired) {
    auto publish = [&](uint32_t ready, bool done, hipError_t e, const char* what) {
        {
            std::lock_guard<std::mutex> lk(U->mu);
            if (ready > U->ready) U->ready = ready;
            if (e != hipSuccess && U->rc == DP_OK) {
                U->rc = DP_ERR_HIP;
                U->error = std::string(what) + ": " + hipGetErrorString(e);
            }
            if (done) U->done = true;
        }
        U->cv.notify_all();
    };
    hipError_t err = hipSetDevice(ctx->device);
    hipStream_t st = nullptr;
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const uint32_t n_host = U->n_host;
    const uint64_t n = P.nascii, n_dwords_all = P.pos / 4;
    uint32_t hr_sent = 0;    // host reads whose ASCII has been queued completely
    uint64_t d_done = 0;     // packed dwords whose pack launch has been queued
    auto pack_upto = [&](uint64_t ascii_end) {  // queues the pack of every read that is complete with ASCII [0, ascii_end)
        while (hr_sent < n_host && (uint64_t)P.rel[(size_t)hr_sent + 1] <= ascii_end) hr_sent++;
        const uint32_t D = hr_sent <= first_paired ? hr_sent : first_paired + 2 * (hr_sent - first_paired);
        const uint64_t d_end = hr_sent == n_host ? n_dwords_all : ctx->h_boff[D] / 4;
        if (d_end > d_done) {
            const uint32_t blocks = (uint32_t)((d_end - d_done + 255) / 256);
            hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, st, (const uint8_t*)P.d_ascii, (const int64_t*)P.d_aoff,
                               (const uint64_t*)ctx->d_boff.p, P.n_reads, (uint32_t*)ctx->d_packed.p, d_end, (const uint32_t*)P.d_map, d_done);
            d_done = d_end;
            return hipGetLastError();
        }
        return hipSuccess;
    };
    constexpr uint64_t kPiece = (uint64_t)4 << 20;
    const int n_thr = std::max(1, upload_threads());
    const int n_slots = 2 * n_thr;
    {
        std::unique_lock<std::mutex> ring_lock(g_ring_mu);
        uint8_t* ring = err == hipSuccess ? ring_get((size_t)n_slots * kPiece) : nullptr;
        if (err == hipSuccess && !ring) {  // no ring: one copy, one pack
            ring_lock.unlock();
            err = hipMemcpyAsync(P.d_ascii, src, n, hipMemcpyHostToDevice, st);
            if (err == hipSuccess) err = pack_upto(n);
        } else if (err == hipSuccess) {
            const uint64_t n_pieces = (n + kPiece - 1) / kPiece;
            std::vector<hipEvent_t> sent((size_t)n_slots, nullptr);
            for (auto& e : sent) hipEventCreateWithFlags(&e, hipEventDisableTiming);
            std::vector<uint32_t> piece_hr((size_t)n_pieces, 0);  // host reads packed once piece i's event has fired
            std::mutex mu;
            std::condition_variable cv;
            std::vector<uint8_t> filled((size_t)n_pieces, 0);
            uint64_t next_fill = 0, next_send = 0;
            bool failed = false;
            auto helper = [&] {
                for (;;) {
                    uint64_t i;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return failed || next_fill >= n_pieces || next_fill < next_send + (uint64_t)n_slots; });
                        if (failed || next_fill >= n_pieces) return;
                        i = next_fill++;
                    }
                    const uint64_t b = i * kPiece, len = std::min(kPiece, n - b);
                    memcpy(ring + (i % (uint64_t)n_slots) * kPiece, src + b, len);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        filled[(size_t)i] = 1;
                    }
                    cv.notify_all();
                }
            };
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; t++) th.emplace_back(helper);
            for (uint64_t i = 0; i < n_pieces && err == hipSuccess; i++) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return filled[(size_t)i] != 0; });
                }
                const uint64_t b = i * kPiece, len = std::min(kPiece, n - b);
                const size_t slot = (size_t)(i % (uint64_t)n_slots);
                err = hipMemcpyAsync((uint8_t*)P.d_ascii + b, ring + slot * kPiece, len, hipMemcpyHostToDevice, st);
                if (err == hipSuccess) err = pack_upto(b + len);
                piece_hr[(size_t)i] = hr_sent;
                if (err == hipSuccess) err = hipEventRecord(sent[slot], st);
                if (err == hipSuccess && i + 1 >= (uint64_t)n_slots / 2) {
                    const uint64_t done_upto = i + 1 - (uint64_t)n_slots / 2;
                    err = hipEventSynchronize(sent[(size_t)(done_upto % (uint64_t)n_slots)]);
                    if (err == hipSuccess) publish(piece_hr[(size_t)done_upto], false, hipSuccess, "");
                    std::lock_guard<std::mutex> lk(mu);
                    next_send = done_upto + 1;
                }
                cv.notify_all();
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (err != hipSuccess) failed = true;
                next_send = n_pieces + (uint64_t)n_slots;
            }
            cv.notify_all();
            for (auto& t : th) t.join();
            if (st) (void)hipStreamSynchronize(st);  // (the ring goes back only when nothing reads it any more)
            for (auto& e : sent) hipEventDestroy(e);
        }
    }
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    else if (st) (void)hipStreamSynchronize(st);
    for (void* q : {P.d_ascii, P.d_aoff, P.d_map})
        if (q) dev_free_impl(q, true);  // (their only user was this stream, which is idle)
    if (st) (void)hipStreamDestroy(st);
    publish(err == hipSuccess ? n_host : 0u, true, err, "dp_reads_upload_rc_begin: staged copy");
}

extern "C" int dp_reads_upload_rc_begin(dp_ctx* ctx, const uint8_t* bases, const int64_t* off, uint32_t n_reads, uint32_t first_paired, uint32_t ready_first) {
    if (!ctx) return DP_ERR_ARG;
    UploadPrep P;
    if (int rc = reads_upload_prepare(ctx, bases, off, n_reads, first_paired, P)) {
        hipStreamSynchronize(ctx->stream);
        for (void* q : {P.d_ascii, P.d_aoff, P.d_map})
            if (q) dp_dev_free(q);
        return rc;
    }
    if (P.nothing) return DP_OK;
    hipError_t e = dp_stream_sync(ctx);  // (the tables the pack kernel reads are resident; P's host vectors move into the thread)
    if (e != hipSuccess) {
        for (void* q : {P.d_ascii, P.d_aoff, P.d_map})
            if (q) dp_dev_free(q);
        return dp_fail(ctx, DP_ERR_HIP, "dp_reads_upload_rc_begin", e);
    }
    std::shared_ptr<dp_ctx::ReadsUpload> keep = std::make_shared<dp_ctx::ReadsUpload>();
    dp_ctx::ReadsUpload* U = keep.get();  // (the thread is joined before the owner's reference goes)
    U->n_host = n_reads;
    {
        std::lock_guard<std::mutex> lk(ctx->upload_mu);
        ctx->upload = keep;
    }
    const uint8_t* src = bases + off[0];
    U->th = std::thread([ctx, U, src, first_paired](UploadPrep Pm) { upload_thread(ctx, U, src, std::move(Pm), first_paired); }, std::move(P));
    return dp_reads_upload_wait(ctx, ready_first);
}

extern "C" int dp_reads_upload_wait(dp_ctx* ctx, uint32_t host_read_hi) {
    if (!ctx) return DP_ERR_ARG;
    dp_ctx* own = ctx->owner ? ctx->owner : ctx;
    if (host_read_hi == 0xffffffffu) return own == ctx ? upload_join(ctx) : DP_ERR_ARG;  // (everything, and the thread with it: the owner's call)
    std::shared_ptr<dp_ctx::ReadsUpload> U;
    {
        std::lock_guard<std::mutex> lk(own->upload_mu);
        U = own->upload;
    }
    if (!U) return DP_OK;
    std::unique_lock<std::mutex> lk(U->mu);
    const uint32_t want = std::min(host_read_hi, U->n_host);
    U->cv.wait(lk, [&] { return U->done || U->ready >= want; });
    return U->rc;  // (the message is handed over by the owner's final wait)
}

extern "C" int dp_reads_packed(dp_ctx* ctx, uint32_t read, uint8_t* out, uint64_t cap, uint64_t* n_bytes) {
    if (!ctx || read >= ctx->n_reads) return DP_ERR_ARG;
    hipSetDevice(ctx->device);
    uint64_t nb = ((uint64_t)ctx->h_len[read] + 3) / 4;
    if (n_bytes) *n_bytes = nb;
    if (nb > cap) return dp_fail(ctx, DP_ERR_ARG, "dp_reads_packed: buffer too small");
    DP_HIP(hipMemcpy(out, (uint8_t*)ctx->d_packed.p + ctx->h_boff[read], nb, hipMemcpyDeviceToHost));
    return DP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// shared device code: window of 32 k-mer start positions

struct Win {
    uint32_t w0, w1, w2;
};
__device__ __forceinline__ Win load_win(const uint8_t* __restrict__ packed, uint64_t g) {
    const uint32_t* p = (const uint32_t*)(packed + g * 8);
    Win w;
    w.w0 = __builtin_bswap32(p[0]);
    w.w1 = __builtin_bswap32(p[1]);
    w.w2 = __builtin_bswap32(p[2]);
    return w;
}
// 32-bit window whose top bits are the k-mer starting at position J (0..31) of the group
template <int J>
__device__ __forceinline__ uint32_t win_at(const Win& w) {
    constexpr int q = J >> 4, r = J & 15;
    uint32_t hi = q ? w.w1 : w.w0, lo = q ? w.w2 : w.w1;
    if (r == 0) return hi;
    return __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * r);
}
__device__ __forceinline__ uint32_t win_at_rt(const Win& w, int j) {
    int q = j >> 4, r = j & 15;
    uint32_t hi = q ? w.w1 : w.w0, lo = q ? w.w2 : w.w1;
    return r ? __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * r) : hi;
}
// bit mask of the positions of group g that lie inside [a0, a1)
__device__ __forceinline__ uint32_t valid_mask(uint64_t g, uint64_t a0, uint64_t a1) {
    int64_t lo = (int64_t)a0 - (int64_t)(g * 32), hi = (int64_t)a1 - (int64_t)(g * 32);
    if (lo < 0) lo = 0;
    if (hi > 32) hi = 32;
    if (hi <= lo) return 0u;
    uint32_t mh = hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u);
    uint32_t ml = (1u << lo) - 1u;  // lo < 32 here
    return mh & ~ml;
}

// ---------------------------------------------------------------------------------------------------------------
// A9 selection: one wave per window, one lane per block of k evaluated k-mers (seeds/seeds.go:84-128)

#define SEL_MAXTOP 64
__global__ __launch_bounds__(64) void select_kernel(const uint8_t* __restrict__ packed, const uint64_t* __restrict__ boff,
                                                    const dp_scan_item* __restrict__ win, uint32_t n, int k, int numSeeds,
                                                    const double* __restrict__ values, uint32_t* __restrict__ top,
                                                    uint32_t* __restrict__ evald, uint32_t stride, const uint8_t* __restrict__ qual,
                                                    const int64_t* __restrict__ qoff, const uint8_t* __restrict__ hasq) {
    __shared__ double bestV[64];
    __shared__ uint32_t bestS[64];
    __shared__ double topV[SEL_MAXTOP];
    __shared__ uint32_t topN[SEL_MAXTOP];
    const int lane = dp_lane();
    const uint32_t w = blockIdx.x;
    if (w >= n) return;
    const dp_scan_item it = win[w];
    const int64_t L = (int64_t)it.n_kmers;  // window length in bases
    const uint64_t A0 = boff[it.read] * 4 + it.start;
    const int sh = 32 - 2 * k;
    // FASTQ: the value of a k-mer is weighted by the quality byte of its middle base, q[nextIndex - k/2] (seeds.go:99-101)
    const uint8_t* q = (qual && hasq[it.read]) ? qual + qoff[it.read] + it.start : nullptr;
    if (lane < numSeeds) {
        topV[lane] = 0.0;
        topN[lane] = 0u;
    }
    // block b starts evaluating at nextIndex = k + 3k*b and exists while nextIndex < L - k
    const int64_t period = 3 * (int64_t)k;
    const int64_t nBlocks = (L - 2 * (int64_t)k) > 0 ? ((L - 2 * (int64_t)k) + period - 1) / period : 0;
    // evald (optional): every k-mer the loop below evaluates, slot = block * k + position in the block; unused slots ~0
    if (evald)
        for (uint32_t i = lane; i < stride; i += 64) evald[(uint64_t)w * stride + i] = 0xffffffffu;
    for (int64_t b0 = 0; b0 < nBlocks; b0 += 64) {
        const int64_t b = b0 + lane;
        double bv = 0.0;
        uint32_t bs = 0u;
        if (b < nBlocks) {
            const int64_t next0 = (int64_t)k + period * b;
            for (int i = 0; i < k; i++) {
                const int64_t ni = next0 + i;  // index of the base that completes the k-mer
                if (ni >= L) break;
                const uint64_t a = A0 + (uint64_t)(ni - k + 1);
                const Win wv = load_win(packed, a >> 5);
                const uint32_t kmer = win_at_rt(wv, (int)(a & 31)) >> sh;
                if (evald && (uint64_t)(b * k + i) < stride) evald[(uint64_t)w * stride + (uint64_t)(b * k + i)] = kmer;
                double v = values[kmer];
                if (q) v *= (double)q[ni + 1 - k / 2];
                if (v > bv) {
                    bv = v;
                    bs = kmer;
                }
            }
        }
        bestV[lane] = bv;
        bestS[lane] = bs;
        __syncthreads();
        if (lane == 0) {  // ascending insertion list, blocks in order (:107-120)
            const int64_t cnt = nBlocks - b0 < 64 ? nBlocks - b0 : 64;
            for (int64_t j = 0; j < cnt; j++) {
                const double v = bestV[j];
                int m = 0;
                for (; m < numSeeds && topV[m] < v; m++) {
                    if (m > 0) {
                        topV[m - 1] = topV[m];
                        topN[m - 1] = topN[m];
                    }
                }
                if (m > 0) {
                    topV[m - 1] = v;
                    topN[m - 1] = bestS[j];
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (lane < numSeeds) top[(uint64_t)w * (uint32_t)numSeeds + lane] = topN[lane];
}

// FASTQ quality bytes of the resident reads for the selection kernels: qual[off[r] + p] = (phred byte - 33) of base p of
// read r (same offsets as the ASCII bases of dp_reads_upload), has_qual[r] = 0 for reads without a usable quality line.
extern "C" int dp_quality_upload(dp_ctx* ctx, const uint8_t* qual, const int64_t* off, const uint8_t* has_qual, uint32_t n_reads) {
    if (!ctx || !qual || !off || !has_qual) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_quality_upload: bad arguments") : DP_ERR_ARG;
    if (ctx->borrowed_reads) return dp_fail(ctx, DP_ERR_STATE, "dp_quality_upload on a borrowing context");
    if (n_reads != ctx->n_reads) return dp_fail(ctx, DP_ERR_ARG, "dp_quality_upload: read count differs from dp_reads_upload");
    if (ctx->n_borrowers.load() > 0) return dp_fail(ctx, DP_ERR_STATE, "dp_quality_upload while contexts borrowing these reads exist");
    hipSetDevice(ctx->device);
    const size_t nb = (size_t)(off[n_reads] - off[0]);
    if (dev_reserve(ctx, ctx->d_qual, nb + 64)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_qualoff, ((size_t)n_reads + 1) * 8)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_hasq, (size_t)n_reads + 16)) return DP_ERR_HIP;
    std::vector<int64_t> rel((size_t)n_reads + 1);
    for (uint32_t r = 0; r <= n_reads; r++) rel[r] = off[r] - off[0];
    DP_HIP(hipMemcpyAsync(ctx->d_qual.p, qual + off[0], nb, hipMemcpyHostToDevice, ctx->stream));
    DP_HIP(hipMemcpyAsync(ctx->d_qualoff.p, rel.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    DP_HIP(hipMemcpyAsync(ctx->d_hasq.p, has_qual, n_reads, hipMemcpyHostToDevice, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    return DP_OK;
}

extern "C" int dp_values_upload(dp_ctx* ctx, const double* values, uint64_t n) {
    if (!ctx || !values || n == 0) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_values_upload: bad arguments") : DP_ERR_ARG;
    if (ctx->borrowed_reads) return dp_fail(ctx, DP_ERR_STATE, "dp_values_upload on a borrowing context");
    hipSetDevice(ctx->device);
    if (dev_reserve(ctx, ctx->d_values, n * sizeof(double))) return DP_ERR_HIP;
    DP_HIP(hipMemcpyAsync(ctx->d_values.p, values, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    ctx->n_values = n;
    ctx->values_total = 0;
    ctx->values_computed = false;
    return DP_OK;
}

static int select_impl(dp_ctx* ctx, const dp_scan_item* win, uint32_t n, int k, int num_seeds, uint32_t* top_out,
                       const uint32_t** evaluated_out, uint32_t stride) {
    if (k < 1 || k > 16 || num_seeds < 1 || num_seeds > SEL_MAXTOP) return dp_fail(ctx, DP_ERR_ARG, "dp_select_seeds: k in 1..16, num_seeds in 1..64");
    const dp_ctx* vsrc = ctx->owner ? ctx->owner : ctx;  // (a borrowing context outlives the owner's value tables: look them up there)
    if (!vsrc->d_values.p || vsrc->n_values != ((uint64_t)1 << (2 * k))) return dp_fail(ctx, DP_ERR_STATE, "dp_select_seeds: value table for this k not uploaded");
    if (n == 0) return DP_OK;
    hipSetDevice(ctx->device);
    for (uint32_t i = 0; i < n; i++) {
        if (win[i].read >= ctx->n_reads || (uint64_t)win[i].start + win[i].n_kmers > ctx->h_len[win[i].read])
            return dp_fail(ctx, DP_ERR_ARG, "dp_select_seeds: window outside its read");
    }
    const size_t wb = (size_t)n * sizeof(dp_scan_item), tb = (size_t)n * (size_t)num_seeds * 4, eb = (size_t)n * stride * 4;
    if (dev_reserve(ctx, ctx->d_selwin, wb)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_seltop, tb + eb)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_seltop, wb + tb + eb)) return DP_ERR_HIP;
    memcpy(ctx->h_seltop.p, win, wb);
    DP_HIP(hipMemcpyAsync(ctx->d_selwin.p, ctx->h_seltop.p, wb, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* d_ev = stride ? (uint32_t*)((uint8_t*)ctx->d_seltop.p + tb) : nullptr;
    hipLaunchKernelGGL(select_kernel, dim3(n), dim3(64), 0, ctx->stream, (const uint8_t*)ctx->d_packed.p,
                       (const uint64_t*)ctx->d_boff.p, (const dp_scan_item*)ctx->d_selwin.p, n, k, num_seeds,
                       (const double*)vsrc->d_values.p, (uint32_t*)ctx->d_seltop.p, d_ev, stride, (const uint8_t*)ctx->d_qual.p,
                       (const int64_t*)ctx->d_qualoff.p, (const uint8_t*)ctx->d_hasq.p);
    DP_HIP(hipGetLastError());
    DP_HIP(hipMemcpyAsync((uint8_t*)ctx->h_seltop.p + wb, ctx->d_seltop.p, tb + eb, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    memcpy(top_out, (uint8_t*)ctx->h_seltop.p + wb, tb);
    if (evaluated_out) *evaluated_out = (const uint32_t*)((uint8_t*)ctx->h_seltop.p + wb + tb);
    return DP_OK;
}

extern "C" int dp_select_seeds(dp_ctx* ctx, const dp_scan_item* win, uint32_t n, int k, int num_seeds, uint32_t* top_out) {
    if (!ctx || (n && (!win || !top_out))) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_select_seeds: bad arguments") : DP_ERR_ARG;
    return select_impl(ctx, win, n, k, num_seeds, top_out, nullptr, 0);
}

extern "C" int dp_select_windows(dp_ctx* ctx, const dp_scan_item* win, uint32_t n, int k, int num_seeds, uint32_t* top_out,
                                 const uint32_t** evaluated_out, uint32_t stride) {
    if (!ctx || (n && (!win || !top_out || !evaluated_out)) || stride == 0)
        return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_select_windows: bad arguments") : DP_ERR_ARG;
    return select_impl(ctx, win, n, k, num_seeds, top_out, evaluated_out, stride);
}

// ---------------------------------------------------------------------------------------------------------------
// A22: histogram of every k-mer of every read (util/sequtil/kmers.go:53-69): positions 0..len-k.

// Work items are stretches of 64 groups of 32 positions (2 048 bases) - not reads: `downpore map` counts the k-mers of a reference
// that is ONE sequence of millions of bases, and a wave per read meant one wave for all of it (8.2 ms for 4.6 Mb; now the chip).
// coff[r] = first item of read r (host-made prefix sums); the read of an item is found by bisection, once per 2 048 bases.
__global__ void hist_kernel(const uint8_t* __restrict__ packed, const uint64_t* __restrict__ boff,
                            const uint32_t* __restrict__ len, uint32_t n_reads, int k, const uint64_t* __restrict__ coff,
                            uint32_t* __restrict__ counts) {
    const int lane = dp_lane();
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t gw = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int sh = 32 - 2 * k;
    const uint64_t n_items = coff[n_reads];
    for (uint64_t it = gw; it < n_items; it += waves) {
        uint32_t lo = 0, hi = n_reads;  // largest r with coff[r] <= it
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (coff[mid] <= it) lo = mid;
            else hi = mid;
        }
        const uint32_t r = lo;
        const uint32_t L = len[r];
        if (L < (uint32_t)k) continue;  // (such a read has no item; kept for safety)
        const uint64_t a0 = boff[r] * 4, a1 = a0 + (L - k + 1);
        const uint64_t g0 = a0 >> 5, g1 = (a1 - 1) >> 5;
        const uint64_t g = g0 + (it - coff[r]) * 64 + (uint64_t)lane;
        if (g > g1) continue;
        Win w = load_win(packed, g);
        uint32_t vm = valid_mask(g, a0, a1);
        for (int j = 0; j < 32; j++)
            if ((vm >> j) & 1) atomicAdd(&counts[win_at_rt(w, j) >> sh], 1u);
    }
}

// adds the k-mer counts of the resident reads to d_counts (4^k zero-initialised uint32 on the device), on the context's stream
int dp_histogram_device(dp_ctx* ctx, int k, uint32_t* d_counts) {
    if (ctx->n_reads) {
        const dp_ctx* ow = ctx->owner ? ctx->owner : ctx;
        std::vector<uint64_t> coff((size_t)ctx->n_reads + 1);
        uint64_t run = 0;
        for (uint32_t r = 0; r < ctx->n_reads; r++) {
            coff[r] = run;
            const uint32_t L = ow->h_len[r];
            if (L < (uint32_t)k) continue;  // the reference would index out of range here; nothing to count
            const uint64_t a0 = ow->h_boff[r] * 4, a1 = a0 + (L - k + 1);
            run += (((a1 - 1) >> 5) - (a0 >> 5) + 1 + 63) / 64;
        }
        coff[ctx->n_reads] = run;
        void* d_coff = nullptr;
        DP_HIP(dp_dev_malloc(&d_coff, coff.size() * 8));
        DP_HIP(hipMemcpyAsync(d_coff, dp_stage(ctx, coff.data(), coff.size() * 8), coff.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        const uint32_t blocks = (uint32_t)std::min<uint64_t>(8192, std::max<uint64_t>(1, (run + 3) / 4));
        hipLaunchKernelGGL(hist_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const uint8_t*)ctx->d_packed.p,
                           (const uint64_t*)ctx->d_boff.p, (const uint32_t*)ctx->d_len.p, ctx->n_reads, k, (const uint64_t*)d_coff, d_counts);
        DP_HIP(hipGetLastError());
        // (freed once the kernel has read it: the block goes back to the cache after the stream has passed this point)
        DP_HIP(dp_stream_sync(ctx));
        dp_dev_free(d_coff);
    }
    return DP_OK;
}

extern "C" int dp_kmer_histogram(dp_ctx* ctx, int k, uint64_t* counts_out) {
    if (!ctx || !counts_out || k < 1 || k > 15) return DP_ERR_ARG;
    hipSetDevice(ctx->device);
    size_t n = (size_t)1 << (2 * k);
    void* d = nullptr;
    DP_HIP(dp_dev_malloc(&d, n * 4));
    DP_HIP(hipMemsetAsync(d, 0, n * 4, ctx->stream));
    if (dp_histogram_device(ctx, k, (uint32_t*)d) != 0) {
        dp_dev_free(d);
        return DP_ERR_HIP;
    }
    std::vector<uint32_t> tmp(n);
    DP_HIP(hipMemcpyAsync(tmp.data(), d, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    dp_dev_free(d);
    for (size_t i = 0; i < n; i++) counts_out[i] = tmp[i];
    return DP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// round state

__global__ void seeds_apply_kernel(const uint32_t* __restrict__ seeds, uint32_t n, uint32_t* __restrict__ bits,
                                   int32_t* __restrict__ kmap, int set) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t kmer = seeds[i];
    if (set) {
        atomicOr(&bits[kmer >> 5], 1u << (kmer & 31));
        kmap[kmer] = (int32_t)i;
    } else {
        bits[kmer >> 5] = 0;  // whole word: every bit in it belongs to a seed of the same (previous) round
        kmap[kmer] = 0;
    }
}

// d_bits / d_kmap := the current seed list (clear the list applied last, set this one); called before a scan kernel
static int seed_tables_ensure(dp_ctx* ctx) {
    if (!ctx->seeds_uploaded && ctx->n_seeds) {  // the scan kernels read the seed list many times: they get a device copy
        DP_HIP(hipMemcpyAsync(ctx->d_seeds.p, ctx->h_seeds.p, (size_t)ctx->n_seeds * 4, hipMemcpyHostToDevice, ctx->stream));
        ctx->seeds_uploaded = true;
    }
    if (!ctx->tables_dirty) return DP_OK;
    if (ctx->n_applied)
        hipLaunchKernelGGL(seeds_apply_kernel, dim3((ctx->n_applied + 255) / 256), dim3(256), 0, ctx->stream,
                           (const uint32_t*)ctx->d_seeds_applied.p, ctx->n_applied, (uint32_t*)ctx->d_bits.p, (int32_t*)ctx->d_kmap.p, 0);
    if (dev_reserve(ctx, ctx->d_seeds_applied, (size_t)ctx->n_seeds * 4 + 4)) return DP_ERR_HIP;
    if (ctx->n_seeds) {
        DP_HIP(hipMemcpyAsync(ctx->d_seeds_applied.p, ctx->d_seeds.p, (size_t)ctx->n_seeds * 4, hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(seeds_apply_kernel, dim3((ctx->n_seeds + 255) / 256), dim3(256), 0, ctx->stream,
                           (const uint32_t*)ctx->d_seeds.p, ctx->n_seeds, (uint32_t*)ctx->d_bits.p, (int32_t*)ctx->d_kmap.p, 1);
    }
    DP_HIP(hipGetLastError());
    ctx->n_applied = ctx->n_seeds;
    ctx->tables_dirty = false;
    return DP_OK;
}

extern "C" int dp_round_begin(dp_ctx* ctx, int k, const uint32_t* seed_kmers, uint32_t n_seeds) {
    if (!ctx || k < 4 || k > 15 || (n_seeds && !seed_kmers)) return DP_ERR_ARG;
    hipSetDevice(ctx->device);
    size_t nk = (size_t)1 << (2 * k);
    if (ctx->table_k != k) {
        if (dev_reserve(ctx, ctx->d_bits, nk / 8 + 64)) return DP_ERR_HIP;
        if (dev_reserve(ctx, ctx->d_kmap, nk * 4)) return DP_ERR_HIP;
        DP_HIP(hipMemsetAsync(ctx->d_bits.p, 0, nk / 8 + 64, ctx->stream));
        DP_HIP(hipMemsetAsync(ctx->d_kmap.p, 0, nk * 4, ctx->stream));
        ctx->table_k = k;
        ctx->n_seeds = 0;
        ctx->n_applied = 0;
    }
    for (uint32_t i = 0; i < n_seeds; i++)
        if (seed_kmers[i] >= nk) return dp_fail(ctx, DP_ERR_ARG, "seed k-mer out of range for k");
    if (dev_reserve(ctx, ctx->d_seeds, (size_t)n_seeds * 4 + 4)) return DP_ERR_HIP;
    // seed_kmers is borrowed only for the duration of the call: it is kept in a pinned block of the context.  Nothing is
    // copied to the device here - the k-mer index walk reads the block in place (dp_seeds_ptr), the scan kernels upload it
    // when they run (seed_tables_ensure).  (As a pageable hipMemcpyAsync this was the most expensive call of a round for the
    // calling thread: 80 us inside the runtime, every round, on every executor slot.)
    if (pin_reserve(ctx, ctx->h_seeds, (size_t)n_seeds * 4 + 4)) return DP_ERR_HIP;
    if (n_seeds) memcpy(ctx->h_seeds.p, seed_kmers, (size_t)n_seeds * 4);
    ctx->seeds_uploaded = false;
    // the membership bits and the k-mer -> seed-id map are only read by the scan kernels: they are brought up to date by
    // seed_tables_ensure() when a scan actually runs (a round served by the k-mer position index never needs them)
    ctx->tables_dirty = true;
    {
        // the round's kernels are bracketed by timing events in every N-th round of this context (dp_set_kernel_timing /
        // DP_KERNEL_TIMING; every event is a packet of its own for the command processor; 0 = never).  Untimed rounds report 0 ms.
        const long every = g_timing_every.load(std::memory_order_relaxed);
        ctx->timing_on = every > 0 && (ctx->round_serial++ % (uint64_t)every) == 0;
    }
    ctx->k = k;
    ctx->n_seeds = n_seeds;
    ctx->round_open = true;
    ctx->n_seqs = 0;
    return DP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// A2 + A10 scan

#define SCAN_THREADS 1024
#define BLOOM_LOG2 20
#define BLOOM_BYTES (1u << (BLOOM_LOG2 - 3))

template <int J>
struct ProbeLoop {
    static __device__ __forceinline__ uint32_t run(const Win& w, const uint8_t* bloom, int pshift) {
        uint32_t win = win_at<J>(w);
        uint32_t idx = win >> pshift;
        uint32_t byte = bloom[idx >> 3];
        uint32_t hit = (byte >> (idx & 7)) & 1u;
        return (hit << J) | ProbeLoop<J + 1>::run(w, bloom, pshift);
    }
};
template <>
struct ProbeLoop<32> {
    static __device__ __forceinline__ uint32_t run(const Win&, const uint8_t*, int) { return 0u; }
};

// ---- filter v2 (k >= 9): one LDS byte probe serves TWO adjacent k-mer positions.
// T1[c] for an 8-base core c: bit x     = some seed has 9-mer prefix  x.c  (x = the base before the core)
//                             bit 4 + y = some seed has 9-mer prefix  c.y  (y = the base after the core)
// The probe at the core that starts one base after position p answers position p (left extension by base p) and
// position p+1 (right extension by base p+9).  T2 is a second-level 2^19-bit filter on the k-mer's last 19 bits,
// consulted only for first-level hits, so the exact 4^k-bit table in L2 is reached by ~0.1 % of the k-mers.
#define T1_BYTES 65536u
#define T2_BYTES 16384u
#define T2_MASK 0x1FFFFu  // 2^17 bits
template <int T, bool NOLDS = false>
struct PairLoop {
    static __device__ __forceinline__ uint32_t run(const Win& w, const uint8_t* t1) {
        const uint32_t win = win_at<2 * T>(w);
        const uint32_t core = __builtin_amdgcn_ubfe(win, 14, 16);  // bases +1 .. +8
        const uint32_t byte = NOLDS ? (core & 0x11u) : t1[core];
        const uint32_t hitA = __builtin_amdgcn_ubfe(byte, win >> 30, 1);
        const uint32_t hitB = __builtin_amdgcn_ubfe(byte, __builtin_amdgcn_ubfe(win, 12, 2) | 4u, 1);
        return ((hitA | (hitB << 1)) << (2 * T)) | PairLoop<T + 1, NOLDS>::run(w, t1);
    }
};
template <bool NOLDS>
struct PairLoop<16, NOLDS> {
    static __device__ __forceinline__ uint32_t run(const Win&, const uint8_t*) { return 0u; }
};

template <int MODE, int FILTER>  // MODE 0 = count pass, 1 = write pass; FILTER 1 = 10-mer prefix filter, 2 = paired 9-mer filter
__global__ __launch_bounds__(SCAN_THREADS, FILTER >= 2 ? 8 : 4) void scan_kernel(const uint8_t* __restrict__ packed,
                                                            const uint64_t* __restrict__ boff,
                                                            const dp_scan_item* __restrict__ items, uint32_t n_items, int k,
                                                            const uint32_t* __restrict__ seeds, uint32_t n_seeds,
                                                            const uint32_t* __restrict__ bits, const int32_t* __restrict__ kmap,
                                                            uint32_t* __restrict__ counts, const uint64_t* __restrict__ segoff,
                                                            int32_t* __restrict__ segs, uint32_t dbg,
                                                            const uint32_t* __restrict__ sel, uint32_t n_sel) {
    constexpr uint32_t LDS_BYTES = FILTER == 1 ? BLOOM_BYTES : (T1_BYTES + T2_BYTES);  // FILTER 3 = timing experiment (no LDS probes)  // v2: 80 KiB -> two workgroups per CU
    __shared__ __attribute__((aligned(16))) uint8_t bloom[LDS_BYTES];  // FILTER 2: T1 | T2
    const int pb = k < 10 ? k : 10;       // prefix bases used by the v1 LDS filter
    const int pshift = 32 - 2 * pb;       // window -> filter index
    const int ksh = 32 - 2 * k;           // window -> k-mer
    {
        uint4* z = (uint4*)bloom;
        for (uint32_t i = threadIdx.x; i < LDS_BYTES / 16; i += SCAN_THREADS) z[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        uint32_t* bw = (uint32_t*)bloom;
        for (uint32_t s = threadIdx.x; s < n_seeds; s += SCAN_THREADS) {
            const uint32_t kmer = seeds[s];
            if (FILTER == 1) {
                uint32_t idx = kmer >> (2 * (k - pb));
                atomicOr(&bw[idx >> 5], 1u << (idx & 31));
            } else {
                const uint32_t P = kmer >> (2 * (k - 9));  // 9-mer prefix (18 bits)
                const uint32_t cl = P & 0xFFFFu, x = P >> 16;
                atomicOr(&bw[cl >> 2], (1u << x) << ((cl & 3) * 8));
                const uint32_t cr = P >> 2, y = P & 3u;
                atomicOr(&bw[cr >> 2], (16u << y) << ((cr & 3) * 8));
                const uint32_t h = kmer & T2_MASK;
                atomicOr(&bw[(T1_BYTES >> 2) + (h >> 5)], 1u << (h & 31));
            }
        }
        __syncthreads();
    }
    const int lane = dp_lane();
    const uint32_t waves = gridDim.x * (SCAN_THREADS / 64);
    const uint32_t gw = blockIdx.x * (SCAN_THREADS / 64) + (threadIdx.x >> 6);
    // sel (write pass of dp_scan_reads): the compacted list of surviving items, one per wave, instead of a strided walk
    // over all items that skips the non-survivors
    const uint32_t n_loop = sel ? n_sel : n_items;
    for (uint32_t li = gw; li < n_loop; li += waves) {
        const uint32_t it = sel ? sel[li] : li;
        const dp_scan_item item = items[it];
        if (MODE == 1 && !sel && counts[it] < item.min_seeds) continue;
        const uint64_t a0 = boff[item.read] * 4 + item.start;
        const uint64_t a1 = a0 + item.n_kmers;
        int cnt = 0;
        int carry = -k;  // position of the last hit so far ("-k": the first gap is the hit's own index)
        int written = 0;
        uint64_t outbase = 0;
        if (MODE == 1) outbase = segoff[it];
        if (item.n_kmers > 0) {
            const uint64_t g0 = a0 >> 5, g1 = (a1 - 1) >> 5;
            Win wn = {0, 0, 0};
            if (g0 + lane <= g1) wn = load_win(packed, g0 + lane);
            for (uint64_t gb = g0; gb <= g1; gb += 64) {
                const uint64_t g = gb + lane;
                uint32_t exact = 0;
                const Win w = wn;
                if (g + 64 <= g1) wn = load_win(packed, g + 64);  // software prefetch of the wave's next block
                if (g <= g1) {
                    uint32_t m = (FILTER == 1 ? ProbeLoop<0>::run(w, bloom, pshift) : FILTER == 3 ? PairLoop<0, true>::run(w, bloom) : PairLoop<0>::run(w, bloom));
                    if (g == g0 || g == g1) m &= valid_mask(g, a0, a1);  // only the item's first / last group is partial
                    if (dbg & 1) {  // timing experiments only (DP_SCAN_DEBUG): drop the candidate loop
                        cnt += __builtin_popcount(m);
                        m = 0;
                    }
                    // candidates of the low / high 16 positions separately: each half reads a fixed pair of window words
                    const uint64_t s01 = ((uint64_t)w.w0 << 32) | w.w1, s12 = ((uint64_t)w.w1 << 32) | w.w2;
                    uint32_t mlo = m & 0xffffu, mhi = m >> 16;
                    while (mlo) {
                        const int j = __builtin_ctz(mlo);
                        mlo &= mlo - 1;
                        const uint32_t kmer = (uint32_t)((s01 << (2 * j)) >> 32) >> ksh;
                        if (FILTER >= 2) {
                            const uint32_t h = kmer & T2_MASK;
                            if (!((bloom[T1_BYTES + (h >> 3)] >> (h & 7)) & 1u)) continue;
                        }
                        if ((bits[kmer >> 5] >> (kmer & 31)) & 1u) exact |= 1u << j;
                    }
                    while (mhi) {
                        const int j = __builtin_ctz(mhi);
                        mhi &= mhi - 1;
                        const uint32_t kmer = (uint32_t)((s12 << (2 * j)) >> 32) >> ksh;
                        if (FILTER >= 2) {
                            const uint32_t h = kmer & T2_MASK;
                            if (!((bloom[T1_BYTES + (h >> 3)] >> (h & 7)) & 1u)) continue;
                        }
                        if ((bits[kmer >> 5] >> (kmer & 31)) & 1u) exact |= 1u << (16 + j);
                    }
                }
                if (MODE == 0) {
                    cnt += __builtin_popcount(exact);
                } else {
                    int c = __builtin_popcount(exact);
                    int incl = wave_incl_sum(c);
                    int excl = incl - c;
                    int total = __shfl(incl, 63, 64);
                    int lastpos = exact ? (int)((int64_t)(g * 32 + 31 - __builtin_clz(exact)) - (int64_t)a0) : -0x40000000;
                    int pm = wave_incl_max(lastpos);
                    int up = __shfl_up(pm, 1, 64);
                    int prev = lane == 0 ? carry : max(carry, up);
                    int t = 0;
                    uint32_t e = exact;
                    while (e) {
                        int j = __builtin_ctz(e);
                        e &= e - 1;
                        uint32_t kmer = win_at_rt(w, j) >> ksh;
                        int p = (int)((int64_t)(g * 32 + j) - (int64_t)a0);
                        uint64_t o = outbase + 2 * (uint64_t)(written + excl + t);
                        segs[o] = p - (prev + k);
                        segs[o + 1] = kmap[kmer];
                        prev = p;
                        t++;
                    }
                    carry = max(carry, __shfl(pm, 63, 64));
                    written += total;
                }
            }
        }
        if (MODE == 0) {
            cnt = wave_sum(cnt);
            if (lane == 0) counts[it] = (uint32_t)cnt;
        } else if (lane == 0) {
            // final gap (sequence/asm_amd64.s:387-392): examined k-mers - last hit - 1; no hit: n_kmers + k - 1
            segs[outbase + 2 * (uint64_t)written] = (int)item.n_kmers - carry - 1;
        }
    }
}

// exclusive scan of (count >= min_seeds ? 2*count+1 : 0) over the items: tile sums -> scan of the tile sums -> offsets
#define OFF_TILE 1024
__device__ __forceinline__ uint32_t seg_len(const dp_scan_item* __restrict__ items, const uint32_t* __restrict__ counts, uint32_t i) {
    const uint32_t c = counts[i];
    return c >= items[i].min_seeds ? 2u * c + 1u : 0u;
}
__device__ __forceinline__ uint32_t block_incl_scan_1024(uint32_t v, uint32_t* sh /*16*/) {
    const int lane = dp_lane(), wave = threadIdx.x >> 6;
    uint32_t x = (uint32_t)wave_incl_sum((int)v);
    if (lane == 63) sh[wave] = x;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < 16 ? sh[lane] : 0u;
        w = (uint32_t)wave_incl_sum((int)w);
        if (lane < 16) sh[lane] = w;
    }
    __syncthreads();
    if (wave > 0) x += sh[wave - 1];
    return x;
}
__global__ __launch_bounds__(OFF_TILE) void offsets_tile_sums(const dp_scan_item* __restrict__ items,
                                                             const uint32_t* __restrict__ counts, uint32_t n,
                                                             uint64_t* __restrict__ tile_sum) {
    __shared__ uint32_t sh[16];
    const uint32_t i = blockIdx.x * OFF_TILE + threadIdx.x;
    uint32_t v = i < n ? seg_len(items, counts, i) : 0u;
    uint32_t x = block_incl_scan_1024(v, sh);
    if (threadIdx.x == OFF_TILE - 1) tile_sum[blockIdx.x] = x;
}
__global__ __launch_bounds__(1024) void offsets_scan_tiles(uint64_t* __restrict__ tile_sum, uint32_t n_tiles,
                                                           uint64_t* __restrict__ total, uint64_t* __restrict__ segoff,
                                                           uint32_t n) {
    // single workgroup; tiles are few (n/1024): serial carry over chunks of 1024 tiles
    __shared__ uint64_t part[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t t = base + threadIdx.x;
        part[threadIdx.x] = t < n_tiles ? tile_sum[t] : 0ull;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t run = carry;
            const uint32_t m = min(1024u, n_tiles - base);
            for (uint32_t j = 0; j < m; j++) {
                const uint64_t v = part[j];
                part[j] = run;
                run += v;
            }
            carry = run;
        }
        __syncthreads();
        if (t < n_tiles) tile_sum[t] = part[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total = carry;
        segoff[n] = carry;
    }
}
__global__ __launch_bounds__(OFF_TILE) void offsets_write(const dp_scan_item* __restrict__ items,
                                                         const uint32_t* __restrict__ counts, uint32_t n,
                                                         const uint64_t* __restrict__ tile_base, uint64_t* __restrict__ segoff) {
    __shared__ uint32_t sh[16];
    const uint32_t i = blockIdx.x * OFF_TILE + threadIdx.x;
    uint32_t v = i < n ? seg_len(items, counts, i) : 0u;
    uint32_t x = block_incl_scan_1024(v, sh);
    if (i < n) segoff[i] = tile_base[blockIdx.x] + (uint64_t)(x - v);
}

extern "C" int dp_scan(dp_ctx* ctx, const dp_scan_item* items, uint32_t n_items, dp_seedseq_batch* out) {
    if (!ctx || !out || (n_items && !items)) return DP_ERR_ARG;
    if (!ctx->round_open) return dp_fail(ctx, DP_ERR_STATE, "dp_scan before dp_round_begin");
    hipSetDevice(ctx->device);
    const int k = ctx->k;
    uint64_t bases = 0;
    for (uint32_t i = 0; i < n_items; i++) {
        const dp_scan_item& it = items[i];
        if (it.read >= ctx->n_reads) return dp_fail(ctx, DP_ERR_ARG, "scan item: read index out of range");
        // the examined k-mers may run past the read end only in the reference's over-scan corner cases (<~8+k bases),
        // which the caller must not batch; enforce that every examined base lies inside the read
        if ((uint64_t)it.start + it.n_kmers + (it.n_kmers ? k - 1 : 0) > ctx->h_len[it.read])
            return dp_fail(ctx, DP_ERR_ARG, "scan item: k-mer range exceeds the read");
        bases += it.n_kmers ? (uint64_t)it.n_kmers + k - 1 : 0;
    }
    if (dev_reserve(ctx, ctx->d_items, (size_t)n_items * sizeof(dp_scan_item) + 16)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_counts, (size_t)n_items * 4 + 16)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_segoff, ((size_t)n_items + 1) * 8)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_total, 16 + ((size_t)n_items / OFF_TILE + 2) * 8)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_counts, (size_t)n_items * 4 + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_segoff, ((size_t)n_items + 1) * 8)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_total, 16)) return DP_ERR_HIP;
    out->n_items = n_items;
    out->kernel_ms = 0;
    out->count_kernel_ms = 0;
    out->write_kernel_ms = 0;
    out->bases_scanned = bases;
    ctx->scan_items = n_items;
    if (n_items == 0) {
        ((uint64_t*)ctx->h_segoff.p)[0] = 0;
        out->n_seeds = (const uint32_t*)ctx->h_counts.p;
        out->seg_off = (const uint64_t*)ctx->h_segoff.p;
        out->segs = nullptr;
        out->n_segs = 0;
        ctx->n_segs = 0;
        return DP_OK;
    }
    ctx->items_ptr = nullptr;  // (dp_scan_reads' resident read items are overwritten)
    DP_HIP(hipMemcpyAsync(ctx->d_items.p, items, (size_t)n_items * sizeof(dp_scan_item), hipMemcpyHostToDevice, ctx->stream));
    int dev_cus = 256;
    hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const bool v2 = k >= 9;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)dev_cus * (v2 ? 2 : 1), ((uint64_t)n_items + 15) / 16);
    const uint32_t dbg = (uint32_t)dp_tune("scan_debug", 0);
    if (int rc = seed_tables_ensure(ctx)) return rc;
    std::unique_lock<ScanGate> scan_lock(g_scan_mu);
    DP_HIP(dp_mark(ctx, 0));
    hipLaunchKernelGGL(((dbg & 2) ? scan_kernel<0, 3> : v2 ? scan_kernel<0, 2> : scan_kernel<0, 1>), dim3(grid), dim3(SCAN_THREADS), 0, ctx->stream, (const uint8_t*)ctx->d_packed.p,
                       (const uint64_t*)ctx->d_boff.p, (const dp_scan_item*)ctx->d_items.p, n_items, k,
                       (const uint32_t*)ctx->d_seeds.p, ctx->n_seeds, (const uint32_t*)ctx->d_bits.p,
                       (const int32_t*)ctx->d_kmap.p, (uint32_t*)ctx->d_counts.p, (const uint64_t*)nullptr, (int32_t*)nullptr, dbg, (const uint32_t*)nullptr, 0u);
    DP_HIP(hipGetLastError());
    DP_HIP(dp_mark(ctx, 1));
    {
        const uint32_t n_tiles = (n_items + OFF_TILE - 1) / OFF_TILE;
        uint64_t* tiles = (uint64_t*)ctx->d_total.p + 2;
        hipLaunchKernelGGL(offsets_tile_sums, dim3(n_tiles), dim3(OFF_TILE), 0, ctx->stream, (const dp_scan_item*)ctx->d_items.p,
                           (const uint32_t*)ctx->d_counts.p, n_items, tiles);
        hipLaunchKernelGGL(offsets_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, tiles, n_tiles, (uint64_t*)ctx->d_total.p,
                           (uint64_t*)ctx->d_segoff.p, n_items);
        hipLaunchKernelGGL(offsets_write, dim3(n_tiles), dim3(OFF_TILE), 0, ctx->stream, (const dp_scan_item*)ctx->d_items.p,
                           (const uint32_t*)ctx->d_counts.p, n_items, (const uint64_t*)tiles, (uint64_t*)ctx->d_segoff.p);
        DP_HIP(hipGetLastError());
    }
    DP_HIP(dp_mark(ctx, 4));
    DP_HIP(hipMemcpyAsync(ctx->h_total.p, ctx->d_total.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(hipMemcpyAsync(ctx->h_counts.p, ctx->d_counts.p, (size_t)n_items * 4, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(hipMemcpyAsync(ctx->h_segoff.p, ctx->d_segoff.p, ((size_t)n_items + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    const uint64_t n_segs = *(uint64_t*)ctx->h_total.p;
    if (dev_reserve(ctx, ctx->d_segs, n_segs * 4 + 64)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_segs, n_segs * 4 + 64)) return DP_ERR_HIP;
    float ms0 = 0, ms1 = 0, msoff = 0;
    ms0 = dp_elapsed(ctx, 0, 1);
    msoff = dp_elapsed(ctx, 1, 4);
    if (n_segs) {
        DP_HIP(dp_mark(ctx, 2));
        hipLaunchKernelGGL((v2 ? scan_kernel<1, 2> : scan_kernel<1, 1>), dim3(grid), dim3(SCAN_THREADS), 0, ctx->stream,
                           (const uint8_t*)ctx->d_packed.p, (const uint64_t*)ctx->d_boff.p, (const dp_scan_item*)ctx->d_items.p,
                           n_items, k, (const uint32_t*)ctx->d_seeds.p, ctx->n_seeds, (const uint32_t*)ctx->d_bits.p,
                           (const int32_t*)ctx->d_kmap.p, (uint32_t*)ctx->d_counts.p, (const uint64_t*)ctx->d_segoff.p,
                           (int32_t*)ctx->d_segs.p, 0u, (const uint32_t*)nullptr, 0u);
        DP_HIP(hipGetLastError());
        DP_HIP(dp_mark(ctx, 3));
        DP_HIP(hipMemcpyAsync(ctx->h_segs.p, ctx->d_segs.p, n_segs * 4, hipMemcpyDeviceToHost, ctx->stream));
        DP_HIP(dp_stream_sync(ctx));
        ms1 = dp_elapsed(ctx, 2, 3);
    }
    if (scan_lock.owns_lock()) scan_lock.unlock();
    ctx->n_segs = n_segs;
    out->n_seeds = (const uint32_t*)ctx->h_counts.p;
    out->seg_off = (const uint64_t*)ctx->h_segoff.p;
    out->segs = (const int32_t*)ctx->h_segs.p;
    out->n_segs = n_segs;
    out->kernel_ms = (double)ms0 + (double)msoff + (double)ms1;
    out->count_kernel_ms = ms0;
    out->write_kernel_ms = ms1;
    return DP_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// dp_scan_reads: items generated and survivors compacted on the device

struct make_read_items_kernel {
    enum { THREADS = 256 };
    // flag_src (or null): the flags' pinned host copy, newer than the device's - this launch brings the block over as well (eight bytes
    // per thread, flag_words of them; its own read's flag every thread takes from the source itself)
    static __device__ void run(const uint32_t* __restrict__ len, const uint8_t* __restrict__ ignore, uint32_t lo,
                                       uint32_t hi, int k, int top_level, uint32_t min_seeds, dp_scan_item* __restrict__ items,
                                       const unsigned long long* __restrict__ flag_src, unsigned long long* __restrict__ flag_dst, uint32_t flag_words) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (flag_src) {
        if (i < flag_words) flag_dst[i] = flag_src[i];
        ignore = (const uint8_t*)flag_src;
    }
    if (lo + i >= hi) return;
    const uint32_t r = lo + i;
    const uint32_t L = len[r];
    int64_t n = (int64_t)L - k + 1;
    if (top_level && (L & 3u) == 0) n -= 4;  // top-level sequence with finalLen == 0 (asm:88-96)
    if (n < 0) n = 0;
    dp_scan_item it;
    it.read = r;
    it.start = 0;
    if (ignore[r]) {
        it.n_kmers = 0;
        it.min_seeds = 0xffffffffu;  // never a survivor
    } else {
        it.n_kmers = (uint32_t)n;
        it.min_seeds = min_seeds;
    }
    items[i] = it;
}
};

// tile sums / compaction of the survivor FLAG (count >= min_seeds), same tiling as the offsets scan
__global__ __launch_bounds__(OFF_TILE) void flag_tile_sums(const dp_scan_item* __restrict__ items, const uint32_t* __restrict__ counts,
                                                          uint32_t n, uint64_t* __restrict__ tile_sum) {
    __shared__ uint32_t sh[16];
    const uint32_t i = blockIdx.x * OFF_TILE + threadIdx.x;
    uint32_t v = (i < n && counts[i] >= items[i].min_seeds) ? 1u : 0u;
    uint32_t x = block_incl_scan_1024(v, sh);
    if (threadIdx.x == OFF_TILE - 1) tile_sum[blockIdx.x] = x;
}
__global__ __launch_bounds__(OFF_TILE) void compact_write(const dp_scan_item* __restrict__ items, const uint32_t* __restrict__ counts,
                                                         uint32_t n, const uint64_t* __restrict__ tile_base,
                                                         const uint64_t* __restrict__ segoff, uint32_t* __restrict__ s_item,
                                                         uint32_t* __restrict__ s_count, uint64_t* __restrict__ s_off,
                                                         uint4* __restrict__ s_pack) {
    __shared__ uint32_t sh[16];
    const uint32_t i = blockIdx.x * OFF_TILE + threadIdx.x;
    uint32_t v = (i < n && counts[i] >= items[i].min_seeds) ? 1u : 0u;
    uint32_t x = block_incl_scan_1024(v, sh);
    if (v) {
        const uint64_t slot = tile_base[blockIdx.x] + (uint64_t)(x - 1);
        const uint64_t so = segoff[i];
        s_item[slot] = i;
        s_count[slot] = counts[i];
        s_off[slot] = so;
        s_pack[slot] = make_uint4(i, counts[i], (uint32_t)so, (uint32_t)(so >> 32));  // the same triple, as the host fetches it
    }
}

// does dp_scan_reads answer from the resident k-mer position index (when that can be built) rather than by scanning?
static bool scan_wants_index(const dp_ctx* ctx) {
    bool use_index = ctx->total_bases >= 1000000000ull || (ctx->owner && ctx->owner->total_bases >= 1000000000ull);
    if (const char* e = getenv("DP_SCAN_INDEX")) use_index = e[0] == '1';
    return use_index;
}

extern "C" int dp_scan_prepare(dp_ctx* ctx, int k) {
    if (!ctx || k < 4 || k > 15) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_scan_prepare: k in 4..15") : DP_ERR_ARG;
    hipSetDevice(ctx->device);
    if (scan_wants_index(ctx)) {
        const int rc = dp_kindex_ensure(ctx, k);  // > 0: cannot be used for this k / read set, the rounds will scan
        if (rc < 0) return rc;
    }
    return DP_OK;
}

extern "C" int dp_scan_release(dp_ctx* ctx) {
    if (!ctx) return DP_ERR_ARG;
    if (ctx->borrowed_reads) return dp_fail(ctx, DP_ERR_STATE, "dp_scan_release on a context that borrows its reads");
    hipSetDevice(ctx->device);
    DP_HIP(dp_stream_sync(ctx));
    dp_kindex_free(ctx);
    if (ctx->d_kcounts) {
        dp_dev_free(ctx->d_kcounts);
        ctx->d_kcounts = nullptr;
        ctx->kcounts_k = 0;
    }
    return DP_OK;
}

#define DP_SCAN_AGAIN 1000  // internal: the index could not answer this round, run it again with the scan kernels

static int scan_reads_body(dp_ctx* ctx, const uint8_t* ignore, uint64_t ignore_epoch, uint32_t lo, uint32_t hi, int top_level,
                           uint32_t min_seeds, const dp_scan_item* extra, uint32_t n_extra, dp_survivor_batch* out, bool allow_index);

// Waits for the sequence number the sort pass of a one-go index step stores into the pinned flag when the scan's output is
// complete - not for the stream, which may already carry the chunk stage (dp_index_prechain).  Spins or polls as dp_stream_sync
// does; a flag that does not arrive within two seconds (a kernel that faulted) is left to the stream's own wait and its error.
static hipError_t kx_wait_done(dp_ctx* ctx, const uint32_t* flag, uint32_t seq) {
    static const int env_spin = [] {
        const char* e = getenv("DP_SPIN_SYNC");
        return e ? (e[0] == '1' ? 1 : 0) : -1;
    }();
    static const long poll_ns = dp_tune("sync_poll_us", 20) * 1000L;
    const bool spin = env_spin >= 0 ? env_spin == 1 : g_wait_spin.load(std::memory_order_relaxed) != 0;
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned it = 1;; it++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return hipSuccess;
        if (spin) {
            __builtin_ia32_pause();
            if (it & 0xfff) continue;
        } else {
            timespec ts{0, poll_ns > 0 ? poll_ns : 20000L};
            nanosleep(&ts, nullptr);
            if (it & 0x3f) continue;
        }
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 2000000000L) return dp_stream_sync(ctx);
    }
}

extern "C" int dp_scan_reads(dp_ctx* ctx, const uint8_t* ignore, uint64_t ignore_epoch, uint32_t lo, uint32_t hi, int top_level,
                             uint32_t min_seeds, const dp_scan_item* extra, uint32_t n_extra, dp_survivor_batch* out) {
    int rc = scan_reads_body(ctx, ignore, ignore_epoch, lo, hi, top_level, min_seeds, extra, n_extra, out, true);
    if (rc == DP_SCAN_AGAIN) rc = scan_reads_body(ctx, ignore, ignore_epoch, lo, hi, top_level, min_seeds, extra, n_extra, out, false);
    return rc;
}

static int scan_reads_body(dp_ctx* ctx, const uint8_t* ignore, uint64_t ignore_epoch, uint32_t lo, uint32_t hi, int top_level,
                           uint32_t min_seeds, const dp_scan_item* extra, uint32_t n_extra, dp_survivor_batch* out, bool allow_index) {
    if (!ctx || !out || !ignore || lo > hi || hi > ctx->n_reads || (n_extra && !extra)) return DP_ERR_ARG;
    if (!ctx->round_open) return dp_fail(ctx, DP_ERR_STATE, "dp_scan_reads before dp_round_begin");
    hipSetDevice(ctx->device);
    memset(out, 0, sizeof(*out));
    const int k = ctx->k;
    const uint32_t n_read_items = hi - lo, n_items = n_read_items + n_extra;
    for (uint32_t i = 0; i < n_extra; i++) {
        const dp_scan_item& it = extra[i];
        if (it.read >= ctx->n_reads) return dp_fail(ctx, DP_ERR_ARG, "scan item: read index out of range");
        if ((uint64_t)it.start + it.n_kmers + (it.n_kmers ? k - 1 : 0) > ctx->h_len[it.read])
            return dp_fail(ctx, DP_ERR_ARG, "scan item: k-mer range exceeds the read");
    }
    // bases examined over the non-ignored reads, and the flags on the device: brought up to date when the flags (epoch) or the range change -
    // which is nearly every round (a round's commit flags its query reads).  Round 6: the context keeps a pinned copy of the flags as the
    // device holds them; a call compares the caller's array with it eight bytes at a time, adjusts the two sums by the reads whose flag
    // changed and lets a kernel of the stream fetch the block - where it walked all reads (100 k branches a round) and handed 100 KB of
    // pageable memory to the runtime's copy path (a slot thread spent ~0.15 of its round's 0.73 ms there; five slots entering that path
    // together at a job's start could wait 8 ms for one another: profiles/r06/job_start_stall.txt).
    bool flags_dirty = false;
    struct FlagsGuard {  // (a way out of this call before the launch that brings the flags over: the next call starts from scratch)
        dp_ctx* c;
        bool& dirty;
        ~FlagsGuard() {
            if (dirty) {
                c->ignore_shadow_valid = false;
                c->ignore_epoch = ~0ull;
            }
        }
    } flagsGuard{ctx, flags_dirty};
    const size_t nr8 = ((size_t)ctx->n_reads + 7) & ~(size_t)7;
    if (ctx->ignore_epoch != ignore_epoch || ctx->cached_lo != lo || ctx->cached_hi != hi || ctx->cached_top != top_level ||
        ctx->cached_k != k) {
        const bool grown = !ctx->h_ignore.p || ctx->h_ignore.cap < nr8 + 64;
        if (pin_reserve(ctx, ctx->h_ignore, nr8 + 64)) return DP_ERR_HIP;
        if (dev_reserve(ctx, ctx->d_ignore, nr8 + 16)) return DP_ERR_HIP;
        uint8_t* shadow = (uint8_t*)ctx->h_ignore.p;
        auto kmers_of = [&](uint32_t r) -> uint64_t {
            int64_t n = (int64_t)ctx->h_len[r] - k + 1;
            if (top_level && (ctx->h_len[r] & 3u) == 0) n -= 4;
            return n > 0 ? (uint64_t)n + k - 1 : 0ull;
        };
        const bool same_view = ctx->ignore_shadow_valid && !grown && ctx->cached_lo == lo && ctx->cached_hi == hi && ctx->cached_top == top_level &&
                               ctx->cached_k == k;
        if (!same_view) {  // a new view (or the first call): everything once
            memcpy(shadow, ignore, ctx->n_reads);
            memset(shadow + ctx->n_reads, 0, nr8 + 64 - ctx->n_reads);
            uint64_t b = 0;
            uint32_t nr = 0;
            for (uint32_t r = lo; r < hi; r++) {
                if (shadow[r]) continue;
                b += kmers_of(r);
                nr++;
            }
            ctx->cached_bases = b;
            ctx->cached_reads = nr;
            ctx->ignore_shadow_valid = true;
        } else {  // the reads whose flag changed since the last call (the caller's array may change under this loop: what is READ here is
                  // what the device gets - the commit's speculation check covers flags set after a round's snapshot)
            const size_t words = (size_t)ctx->n_reads / 8;
            for (size_t w = 0; w < words; w++) {
                uint64_t now_w, old_w;
                memcpy(&now_w, ignore + 8 * w, 8);
                memcpy(&old_w, shadow + 8 * w, 8);
                if (now_w == old_w) continue;
                for (uint32_t r = (uint32_t)(8 * w); r < (uint32_t)(8 * w + 8); r++) {
                    const uint8_t v = (uint8_t)(now_w >> (8 * (r & 7u)));
                    if (v == shadow[r]) continue;
                    if (r >= lo && r < hi && (v != 0) != (shadow[r] != 0)) {
                        if (v) ctx->cached_bases -= kmers_of(r), ctx->cached_reads--;
                        else ctx->cached_bases += kmers_of(r), ctx->cached_reads++;
                    }
                    shadow[r] = v;
                }
            }
            for (uint32_t r = (uint32_t)(8 * words); r < ctx->n_reads; r++) {
                const uint8_t v = ignore[r];
                if (v == shadow[r]) continue;
                if (r >= lo && r < hi && (v != 0) != (shadow[r] != 0)) {
                    if (v) ctx->cached_bases -= kmers_of(r), ctx->cached_reads--;
                    else ctx->cached_bases += kmers_of(r), ctx->cached_reads++;
                }
                shadow[r] = v;
            }
        }
        flags_dirty = true;  // (brought over by the launch that makes the read items - below - or by one of its own)
        ctx->ignore_epoch = ignore_epoch;
        ctx->cached_lo = lo;
        ctx->cached_hi = hi;
        ctx->cached_top = top_level;
        ctx->cached_k = k;
    }
    uint64_t bases = ctx->cached_bases;
    for (uint32_t i = 0; i < n_extra; i++) bases += extra[i].n_kmers ? (uint64_t)extra[i].n_kmers + k - 1 : 0;
    out->bases_scanned = bases;
    out->reads_scanned = ctx->cached_reads;
    out->n_extra = n_extra;
    out->index_mode = 0;
    out->index_hits = 0;
    ctx->scan_items = n_items;
    ctx->chunk_lo = lo;
    if (pin_reserve(ctx, ctx->h_total, 128)) return DP_ERR_HIP;
    ctx->last_n_extra = n_extra;
    dp_index_prechain_cancel(ctx);
    if (n_items == 0) {
        ctx->n_segs = 0;
        return DP_OK;
    }
    const uint32_t n_tiles = (n_items + OFF_TILE - 1) / OFF_TILE;
    if (dev_reserve(ctx, ctx->d_items, (size_t)n_items * sizeof(dp_scan_item) + 16)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_counts, (size_t)n_items * 4 + 16)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_segoff, ((size_t)n_items + 1) * 8)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_total, 64 + ((size_t)n_tiles + 2) * 16)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_surv, (size_t)n_items * 32 + 128)) return DP_ERR_HIP;
    dp_scan_item* d_items = (dp_scan_item*)ctx->d_items.p;
    // the read items are a function of the read lengths, the flags and these arguments: they stay on the device from round to
    // round (the kernels only read them; the extra items behind them are this round's)
    if (n_read_items && (ctx->items_ptr != d_items || ctx->items_epoch != ignore_epoch || ctx->items_lo != lo || ctx->items_hi != hi ||
                         ctx->items_min != min_seeds || ctx->items_top != top_level || ctx->items_k != k)) {
        const uint32_t fw = flags_dirty ? (uint32_t)(nr8 / 8) : 0u;
        dp_launch<make_read_items_kernel>(ctx, dim3((std::max(n_read_items, fw) + 255) / 256), dim3(256),
                           (const uint32_t*)ctx->d_len.p, (const uint8_t*)ctx->d_ignore.p, lo, hi, k, top_level, min_seeds, d_items,
                           flags_dirty ? (const unsigned long long*)ctx->h_ignore.p : (const unsigned long long*)nullptr,
                           (unsigned long long*)ctx->d_ignore.p, fw);
        flags_dirty = false;
        ctx->items_ptr = d_items;
        ctx->items_epoch = ignore_epoch;
        ctx->items_lo = lo;
        ctx->items_hi = hi;
        ctx->items_min = min_seeds;
        ctx->items_top = top_level;
        ctx->items_k = k;
    }
    if (flags_dirty) {  // (no read items were made in this call)
        const dp_fetch_region f = {ctx->d_ignore.p, ctx->h_ignore.p, nr8};
        if (int rc = dp_zero_fetch_regions(ctx, nullptr, 0, &f, 1)) return rc;
        flags_dirty = false;
    }
    if (n_extra) {  // (borrowed from the caller: staged in a pinned block, fetched by a kernel of this stream)
        if (pin_reserve(ctx, ctx->h_extra, (size_t)n_extra * sizeof(dp_scan_item) + 64)) return DP_ERR_HIP;
        memcpy(ctx->h_extra.p, extra, (size_t)n_extra * sizeof(dp_scan_item));
    }
    ctx->extras_staged = n_extra > 0;
    auto fetch_extras = [&]() -> int {  // (the index step's first kernel does this itself: dp_kindex_count)
        if (!ctx->extras_staged) return DP_OK;
        ctx->extras_staged = false;
        const dp_fetch_region f = {d_items + n_read_items, ctx->h_extra.p, (size_t)n_extra * sizeof(dp_scan_item)};
        return dp_zero_fetch_regions(ctx, nullptr, 0, &f, 1);
    };
    int dev_cus = 256;
    hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const bool v2 = k >= 9;
    // persistent workgroups per CU: 2 fill every wave slot and all LDS of the CU (fastest scan in isolation) - and keep every
    // other slot's kernels out until the scan is done; DP_SCAN_WG_PER_CU=1 leaves half of each CU to them
    static const int wg_per_cu = (int)std::max(1L, std::min(2L, dp_tune("scan_wg_per_cu", 2)));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)dev_cus * (v2 ? wg_per_cu : 1), ((uint64_t)n_items + 15) / 16);
    uint64_t* totals = (uint64_t*)ctx->d_total.p;  // [0] n_segs, [1] n_survivors
    uint64_t* tilesA = totals + 4;
    uint64_t* tilesB = tilesA + n_tiles + 1;
    uint32_t* s_item = (uint32_t*)ctx->d_surv.p;
    uint32_t* s_count = s_item + n_items;
    uint64_t* s_off = (uint64_t*)(s_count + n_items + (n_items & 1));
    // {item, count, offset} per survivor: what the host wants of the compaction.  The kernel that makes the list writes it into a
    // pinned host block directly (nobody reads it on the device) - a 100 KB copy per round that is never submitted
    if (pin_reserve(ctx, ctx->h_spack, (size_t)n_items * 16 + 64)) return DP_ERR_HIP;
    uint4* s_pack = (uint4*)ctx->h_spack.p;
    // Resident k-mer position index instead of scanning (dp_kindex.hip): DP_SCAN_INDEX=1 forces it, =0 forbids it; by
    // default it is used from 1 Gbase up.  That is the break-even of a whole job: the build costs ~0.13 s per Gbase, a
    // round saves (scan 0.5 ms per Gbase) - (index step 0.25 ms), and a job has ~600 rounds per Gbase of 10 kb reads.
    bool use_index = allow_index && scan_wants_index(ctx);
    std::unique_lock<ScanGate> scan_lock(g_scan_mu, std::defer_lock);
    if (use_index) {
        int rc = dp_kindex_ensure(ctx, k);
        if (rc < 0) return rc;
        if (rc > 0) use_index = false;  // k > 14 or not enough free HBM for 8 B per base: scan
    }
    out->index_hits = 0;
    // Round 4: the index step in one go - count (+ hit records), offsets, fill from the records and sort/write are all launched
    // before the one wait, into buffers sized from this context's previous index-mode round (twice its segments, 1.5 x its hits, the
    // sort tier of 1.25 x its largest survivor); the rare round that outgrows a guess repeats fill + sort the old way after the
    // wait.  DP_KX_ONESHOT=0: count, wait, size, fill, sort, wait - as before.
    const char* ose = getenv("DP_KX_ONESHOT");  // (read per call: tests switch it between jobs of one process)
    const bool oneshot_env = !(ose && ose[0] == '0');
    bool oneshot = false;
    dp_kindex_oneshot one;
    memset(&one, 0, sizeof one);
    const bool mirror_wanted = ctx->scan_fetch_extras_only && n_extra > 0;
    if (use_index) {
        const std::vector<uint32_t>& lens = ctx->owner ? ctx->owner->h_len : ctx->h_len;
        if (ctx->kx_maxlen_reads != lens.size()) {  // (records hold 24 bits of read id and of position)
            uint32_t mx = 0;
            for (uint32_t L : lens) mx = std::max(mx, L);
            ctx->kx_maxlen = mx;
            ctx->kx_maxlen_reads = lens.size();
        }
        oneshot = oneshot_env && ctx->n_seeds > 0 && ctx->n_reads < (1u << 24) && ctx->kx_maxlen < (1u << 24);
        if (oneshot) {
            one.hits_guess = ctx->kx_prev_hits ? ctx->kx_prev_hits + ctx->kx_prev_hits / 2 : 0;
            one.surv_guess = ctx->kx_prev_surv ? ctx->kx_prev_surv + ctx->kx_prev_surv / 4 : 8192;
            const uint32_t mxg = ctx->kx_prev_max + ctx->kx_prev_max / 4;
            one.sort_cap = mxg <= 128 ? 256u : (mxg <= 512 ? 1024u : 4096u);
            one.min_seeds = min_seeds;
            const uint64_t want = std::max<uint64_t>(2 * ctx->kx_prev_segs + 65536, 1u << 20);
            if (dev_reserve(ctx, ctx->d_segs, want * 4 + 64)) return DP_ERR_HIP;
            one.seg_cap = (ctx->d_segs.cap - 64) / 4;
            if (mirror_wanted) {
                if (pin_reserve(ctx, ctx->h_segs, one.seg_cap * 4 + 64)) return DP_ERR_HIP;
                one.host_segs = (int32_t*)ctx->h_segs.p;
            }
            one.d_segs = (int32_t*)ctx->d_segs.p;
            // (where a chunk stage launched behind the step says that the step's own kernels are done)
            one.done_flag = (uint32_t*)ctx->h_total.p + 30;
            one.done_seq = ++ctx->kx_seq ? ctx->kx_seq : ++ctx->kx_seq;
            __atomic_store_n(one.done_flag, 0u, __ATOMIC_RELEASE);  // (whatever an earlier round or a fresh block left there)
        }
    }
    if (use_index) {
        // counts, segment offsets, survivor list and totals in three launches, no sort and no host round trip (dp_kindex.hip)
        // (views served whole - top_level == 0 - have a k-mer at every indexed position: the walk tests the ignore byte instead of
        // reading the read's item, and looks for extra items only on the round's query reads; DP_KX_FAST=0: off)
        dp_kindex_fast fastArgs = {nullptr, 0u, 0u};
        static const bool fast_off = false;
        if (!top_level && !fast_off && ctx->d_ignore.p) {
            uint32_t qmin = 0xffffffffu, qmax = 0;
            for (uint32_t i = 0; i < n_extra; i++) {
                qmin = std::min(qmin, extra[i].read);
                qmax = std::max(qmax, extra[i].read);
            }
            fastArgs.ign = (const uint8_t*)ctx->d_ignore.p;
            fastArgs.qlo = n_extra ? qmin : 0u;
            fastArgs.qspan = n_extra ? qmax - qmin + 1 : 0u;
        }
        int rc = dp_kindex_count(ctx, k, d_items, lo, hi, n_read_items, n_extra, (uint32_t*)ctx->d_counts.p, (uint64_t*)ctx->d_segoff.p,
                                 s_item, s_count, s_off, s_pack, totals, (unsigned long long*)ctx->h_total.p, oneshot ? &one : nullptr, &fastArgs);
        if (rc < 0) return rc;
        if (rc == 2) oneshot = false;   // (records not possible this round: the two-step form follows)
        else if (rc > 0) use_index = false;  // more items than its scan handles: this round is scanned
    }
    if (!use_index) oneshot = false;
    bool wait_flag = false;
    if (oneshot) {
        // dp_index_prechain: the chunk stage of the caller's coming dp_index_build_chunked goes behind the scan right now
        if (int rc = dp_index_prechain_launch(ctx, totals, n_extra, one.seg_cap, one.done_flag, one.done_seq)) return rc;
        // (its first kernel tells the host that the scan's own kernels are done: the host waits for that word, not for the stream)
        wait_flag = ctx->pc_launched && !ctx->timing_on;
    }
    out->index_mode = use_index ? 1u : 0u;
    if (!use_index) {
        if (int rc = fetch_extras()) return rc;
        if (int rc = seed_tables_ensure(ctx)) return rc;
        scan_lock.lock();  // (see dp_scan)
        DP_HIP(dp_mark(ctx, 0));
        hipLaunchKernelGGL((v2 ? scan_kernel<0, 2> : scan_kernel<0, 1>), dim3(grid), dim3(SCAN_THREADS), 0, ctx->stream,
                           (const uint8_t*)ctx->d_packed.p, (const uint64_t*)ctx->d_boff.p, (const dp_scan_item*)d_items, n_items, k,
                           (const uint32_t*)ctx->d_seeds.p, ctx->n_seeds, (const uint32_t*)ctx->d_bits.p,
                           (const int32_t*)ctx->d_kmap.p, (uint32_t*)ctx->d_counts.p, (const uint64_t*)nullptr, (int32_t*)nullptr, 0u,
                           (const uint32_t*)nullptr, 0u);
        DP_HIP(hipGetLastError());
        DP_HIP(dp_mark(ctx, 1));
        hipLaunchKernelGGL(offsets_tile_sums, dim3(n_tiles), dim3(OFF_TILE), 0, ctx->stream, (const dp_scan_item*)d_items,
                           (const uint32_t*)ctx->d_counts.p, n_items, tilesA);
        hipLaunchKernelGGL(offsets_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, tilesA, n_tiles, totals, (uint64_t*)ctx->d_segoff.p, n_items);
        hipLaunchKernelGGL(offsets_write, dim3(n_tiles), dim3(OFF_TILE), 0, ctx->stream, (const dp_scan_item*)d_items,
                           (const uint32_t*)ctx->d_counts.p, n_items, (const uint64_t*)tilesA, (uint64_t*)ctx->d_segoff.p);
        hipLaunchKernelGGL(flag_tile_sums, dim3(n_tiles), dim3(OFF_TILE), 0, ctx->stream, (const dp_scan_item*)d_items,
                           (const uint32_t*)ctx->d_counts.p, n_items, tilesB);
        hipLaunchKernelGGL(offsets_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, tilesB, n_tiles, totals + 1, tilesB + n_tiles, 0u);
        hipLaunchKernelGGL(compact_write, dim3(n_tiles), dim3(OFF_TILE), 0, ctx->stream, (const dp_scan_item*)d_items,
                           (const uint32_t*)ctx->d_counts.p, n_items, (const uint64_t*)tilesB, (const uint64_t*)ctx->d_segoff.p, s_item,
                           s_count, s_off, s_pack);
    }
    DP_HIP(hipGetLastError());
    DP_HIP(dp_mark(ctx, 4));
    if (!use_index) {  // (stored into the pinned block by a launch of this stream, not copied by the runtime; the index step's
                       // last kernel has stored them itself)
        const dp_fetch_region f = {ctx->h_total.p, totals, 48};
        if (int rc = dp_zero_fetch_regions(ctx, nullptr, 0, &f, 1)) return rc;
    }
    if (wait_flag) DP_HIP(kx_wait_done(ctx, one.done_flag, one.done_seq));
    else DP_HIP(dp_stream_sync(ctx));
    {
        // DP_SCAN_RELEASE_EARLY=1 opens the gate here, after the count pass, so that the short write pass overlaps the
        // next slot's count pass.  Measured: it does not - the count pass is a persistent grid that owns every CU's LDS, so
        // the write pass queues behind it (0.08 -> 0.2 ms) and the slot only gets slower.  Off by default.
        static const bool early = false;
        if (early && scan_lock.owns_lock()) scan_lock.unlock();
    }
    const uint64_t n_segs = ((uint64_t*)ctx->h_total.p)[0];
    const uint64_t n_surv_all = ((uint64_t*)ctx->h_total.p)[1];  // surviving reads + all extra items
    if (use_index) {
        ctx->kx_hits = ((uint64_t*)ctx->h_total.p)[2];
        out->index_hits = ctx->kx_hits;
    }
    const uint32_t kx_max_count = (uint32_t)((uint64_t*)ctx->h_total.p)[3];
    bool oneshot_done = false, queued_more = false;
    if (oneshot) {
        // did the guesses hold?  (the kernels behind the count step gave up on their own where a buffer was too small)
        const bool rec_full = (uint32_t)((uint64_t*)ctx->h_total.p)[6] != 0;
        oneshot_done = n_segs <= one.seg_cap && !rec_full && kx_max_count <= one.sort_cap;
        ctx->kx_oneshot_rounds++;
        if (!oneshot_done) ctx->kx_oneshot_redone++;
        ctx->kx_prev_hits = ((uint64_t*)ctx->h_total.p)[2];
        ctx->kx_prev_segs = n_segs;
        ctx->kx_prev_max = kx_max_count;
        ctx->kx_prev_surv = (uint32_t)n_surv_all;
        static const bool dbg1 = dp_debug("kx_oneshot");
        if (dbg1 && !oneshot_done)
            fprintf(stderr, "[kx] one-go step repeated: segs %llu / cap %llu, records %s, largest survivor %u / sort %u (%llu of %llu rounds)\n",
                    (unsigned long long)n_segs, (unsigned long long)one.seg_cap, rec_full ? "full" : "ok", kx_max_count, one.sort_cap,
                    (unsigned long long)ctx->kx_oneshot_redone, (unsigned long long)ctx->kx_oneshot_rounds);
        if (!oneshot_done) dp_index_prechain_cancel(ctx);  // (the chunk stage behind it found nothing to chunk and said so)
        if (!oneshot_done && kx_max_count <= 4096)
            if (int rc = dp_kindex_refill(ctx, (const dp_scan_item*)d_items, n_read_items, n_extra)) return rc;
    }
    if (dev_reserve(ctx, ctx->d_segs, n_segs * 4 + 64)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_segs, n_segs * 4 + 64)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_surv, n_surv_all * 32 + 128)) return DP_ERR_HIP;
    ctx->last_surv_all = (uint32_t)n_surv_all;
    float ms0 = 0, ms1 = 0, msoff = 0;
    ms0 = dp_elapsed(ctx, 0, 1);
    msoff = oneshot ? 0.f : dp_elapsed(ctx, 1, 4);  // (one-go step: fill + sort lie between these two marks, and are ms1 below)
    uint32_t* h_item = (uint32_t*)ctx->h_surv.p;
    uint32_t* h_count = h_item + n_surv_all;
    uint64_t* h_off = (uint64_t*)(h_count + n_surv_all + (n_surv_all & 1));
    const uint32_t* h_pack = (const uint32_t*)ctx->h_spack.p;  // (complete since the wait after the count pass)
    // dp_scan_fetch_mode(1) with the index step: the only segments the host wants are the extra items' - the sort kernel stores
    // them into the pinned block as it writes them (no store launch afterwards)
    const bool mirror_extras = use_index && ctx->scan_fetch_extras_only && n_extra > 0;
    if (n_segs) {
        if (!oneshot_done) DP_HIP(dp_mark(ctx, 2));
        if (oneshot_done) {
            // (fill + sort ran behind the counting step already)
        } else if (use_index) {
            int rc = dp_kindex_write(ctx, k, (const dp_scan_item*)d_items, lo, hi, n_read_items, n_extra, (const uint32_t*)s_item,
                                     (uint32_t)n_surv_all, kx_max_count, (const uint32_t*)ctx->d_counts.p, (const uint64_t*)ctx->d_segoff.p,
                                     (const uint64_t*)totals, (int32_t*)ctx->d_segs.p, mirror_extras ? (int32_t*)ctx->h_segs.p : (int32_t*)nullptr);
            if (rc < 0) return rc;
            if (rc > 0) {  // a survivor with more hits than the in-LDS sort holds: the scan kernels answer this round
                DP_HIP(dp_stream_sync(ctx));
                return DP_SCAN_AGAIN;
            }
        } else {
            // one wave per SURVIVOR (the compacted list), not a strided walk over every item
            const uint32_t wgrid = (uint32_t)std::min<uint64_t>(grid, (n_surv_all + 15) / 16);
            hipLaunchKernelGGL((v2 ? scan_kernel<1, 2> : scan_kernel<1, 1>), dim3(std::max(1u, wgrid)), dim3(SCAN_THREADS), 0, ctx->stream,
                               (const uint8_t*)ctx->d_packed.p, (const uint64_t*)ctx->d_boff.p, (const dp_scan_item*)d_items, n_items, k,
                               (const uint32_t*)ctx->d_seeds.p, ctx->n_seeds, (const uint32_t*)ctx->d_bits.p,
                               (const int32_t*)ctx->d_kmap.p, (uint32_t*)ctx->d_counts.p, (const uint64_t*)ctx->d_segoff.p,
                               (int32_t*)ctx->d_segs.p, 0u, (const uint32_t*)s_item, (uint32_t)n_surv_all);
        }
        DP_HIP(hipGetLastError());
        if (!oneshot_done) DP_HIP(dp_mark(ctx, 3));
        if (n_segs * 4 > ((uint64_t)8 << 20) && !oneshot_done) {
            // dense-seed regime: tens of MB go back to the host; let the next slot's scan start while they travel
            if (!ctx->timing_on) DP_HIP(hipEventRecord(ctx->ev[3], ctx->stream));
            DP_HIP(hipEventSynchronize(ctx->ev[3]));
            if (scan_lock.owns_lock()) scan_lock.unlock();
        }
        // dp_scan_fetch_mode(1): the survivors' segments stay on the device (the caller chunks and indexes them there); only the
        // extra items' - the query windows', which follow the survivors in the scan output - come to the host, at their offsets
        uint64_t from = 0;
        if (ctx->scan_fetch_extras_only) {
            if (!n_extra || mirror_extras) from = n_segs;
            else if (use_index) from = std::min<uint64_t>(n_segs, ((uint64_t*)ctx->h_total.p)[5]);
        }
        if (from < n_segs) {
            if (n_segs - from <= ((uint64_t)1 << 20)) {  // a few hundred KB: stored by a launch of this stream (8-byte words)
                const uint64_t f0 = from & ~(uint64_t)1;
                const dp_fetch_region f = {(int32_t*)ctx->h_segs.p + f0, (const int32_t*)ctx->d_segs.p + f0, (n_segs - f0) * 4};
                if (int rc = dp_zero_fetch_regions(ctx, nullptr, 0, &f, 1)) return rc;
            } else {
                DP_HIP(hipMemcpyAsync((int32_t*)ctx->h_segs.p + from, (const int32_t*)ctx->d_segs.p + from, (n_segs - from) * 4,
                                      hipMemcpyDeviceToHost, ctx->stream));
            }
            queued_more = true;
        }
    }
    // (a one-go step that held: everything the host reads arrived with the first wait, and what may be queued behind it - the
    // chunk stage - is not this call's to wait for)
    if (!oneshot_done || queued_more) DP_HIP(dp_stream_sync(ctx));
    if (n_segs) ms1 = dp_elapsed(ctx, 2, 3);
    if (scan_lock.owns_lock()) scan_lock.unlock();
    ctx->n_segs = n_segs;
    for (uint64_t i = 0; i < n_surv_all; i++) {
        h_item[i] = h_pack[4 * i];
        h_count[i] = h_pack[4 * i + 1];
        h_off[i] = (uint64_t)h_pack[4 * i + 2] | ((uint64_t)h_pack[4 * i + 3] << 32);
    }
    // survivors of the read range come first (item index < n_read_items), the extra items after them (all present)
    uint64_t ns = 0;
    while (ns < n_surv_all && h_item[ns] < n_read_items) ns++;
    if (n_surv_all - ns != n_extra) return dp_fail(ctx, DP_ERR_STATE, "dp_scan_reads: internal compaction mismatch");
    for (uint64_t i = 0; i < ns; i++) h_item[i] += lo;  // item index -> read id
    out->n_survivors = (uint32_t)ns;
    out->read = h_item;
    out->n_seeds = h_count;
    out->seg_off = h_off;
    out->extra_n_seeds = h_count + ns;
    out->extra_seg_off = h_off + ns;
    out->segs = (const int32_t*)ctx->h_segs.p;
    out->n_segs = n_segs;
    out->kernel_ms = (double)ms0 + (double)msoff + (double)ms1;
    out->count_kernel_ms = ms0;
    out->write_kernel_ms = ms1;
    return DP_OK;
}

extern "C" int dp_scan_fetch_mode(dp_ctx* ctx, int extras_only) {
    if (!ctx) return DP_ERR_ARG;
    ctx->scan_fetch_extras_only = extras_only ? 1 : 0;
    return DP_OK;
}

// the whole segment array of the last dp_scan_reads, for a caller that left the survivors' part on the device (fetch mode 1)
// and needs it after all
extern "C" int dp_scan_fetch_segments(dp_ctx* ctx, const int32_t** segs_out, uint64_t* n_segs) {
    if (!ctx || !segs_out) return DP_ERR_ARG;
    hipSetDevice(ctx->device);
    if (ctx->n_segs) {
        if (pin_reserve(ctx, ctx->h_segs, ctx->n_segs * 4 + 64)) return DP_ERR_HIP;
        DP_HIP(hipMemcpyAsync(ctx->h_segs.p, ctx->d_segs.p, ctx->n_segs * 4, hipMemcpyDeviceToHost, ctx->stream));
        DP_HIP(dp_stream_sync(ctx));
    }
    *segs_out = (const int32_t*)ctx->h_segs.p;
    if (n_segs) *n_segs = ctx->n_segs;
    return DP_OK;
}

extern "C" int dp_scan_device_buffers(dp_ctx* ctx, void** segs_dev, uint64_t* n_segs) {
    if (!ctx) return DP_ERR_ARG;
    if (segs_dev) *segs_dev = ctx->d_segs.p;
    if (n_segs) *n_segs = ctx->n_segs;
    return DP_OK;
}

extern "C" int dp_scan_import_segments(dp_ctx* ctx, const int32_t* segs, uint64_t n_segs) {
    if (!ctx || (n_segs && !segs)) return DP_ERR_ARG;
    hipSetDevice(ctx->device);
    if (dev_reserve(ctx, ctx->d_segs, n_segs * 4 + 64)) return DP_ERR_HIP;
    if (n_segs) DP_HIP(hipMemcpyAsync(ctx->d_segs.p, segs, n_segs * 4, hipMemcpyHostToDevice, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    ctx->n_segs = n_segs;
    return DP_OK;
}
