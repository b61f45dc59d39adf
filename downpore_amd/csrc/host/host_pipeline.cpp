// The overlap command's round loop (commands/overlap.go:96-195) as a pipeline: Planner (PrepareQueries chain ahead of
// execution), executor slots (speculative rounds), in-order commit with the speculation check; plus the multi-rank
// entry points (round-parallel supersteps, scan-shard halves).
#include <sys/mman.h>
#include <unistd.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

#include "host_util.hpp"

namespace dph {

// ---------------------------------------------------------------------------------------------------------------
// WindowCache

struct WindowCache::Impl {
    dp_ctx* ctx;
    const ReadSet* reads = nullptr;
    ValueView hostValues = ValueView((const double*)nullptr);
    int k;
    // windows per chunk: 8192 (one selection kernel + one copy back per chunk; the producer is busy half of a config-2 job
    // with that, with 2048 it could not keep up) - except the first four (1024, 1024, 2048, 4096), which the first plans of a
    // job wait for: every slot stands still until the chunk its first plan needs is there
    static constexpr uint32_t CW = 8192;
    static uint32_t chunkOf(uint32_t w) { return w < 1024 ? 0u : w < 2048 ? 1u : w < 4096 ? 2u : w < CW ? 3u : 3u + w / CW; }
    static size_t chunkBegin(uint32_t c) { return c == 0 ? 0 : c <= 3 ? (size_t)512 << c : (size_t)(c - 3) * CW; }
    static constexpr uint32_t SOFT_CAP = 11;  // chunks kept ahead of the release point unless somebody waits for more
    struct Chunk {
        std::vector<uint32_t> spec, kmers;
    };
    // (a chunk's buffers are 11.5 MB at config 2, a dozen of them are alive at a time: a handle's next job takes over the last
    // job's instead of mapping - and zero-filling - 130 MB again and giving 130 MB back)
    static std::mutex& sparesMu() {
        static std::mutex m;
        return m;
    }
    static std::vector<std::unique_ptr<Chunk>>& spares() {
        static std::vector<std::unique_ptr<Chunk>> v;
        return v;
    }
    std::mutex mu;
    std::condition_variable cvProduced, cvSpace;
    std::map<uint32_t, std::unique_ptr<Chunk>> live;  // produced chunks by number
    std::vector<std::unique_ptr<Chunk>> pool;         // recycled buffers
    uint32_t produced = 0, released = 0, nChunks = 0, wanted = 0;
    bool stop = false, failed = false;
    std::string error;
    std::thread th;
};

WindowCache::WindowCache(dp_ctx* ctx, const ReadSet& reads, i64 overlap, int k, int numSeeds_, ValueView hostValues) : d(new Impl()) {
    d->ctx = ctx;
    d->reads = &reads;
    d->hostValues = hostValues;
    d->k = k;
    numSeeds = numSeeds_;
    first.resize(reads.size() + 1);
    wins.reserve(reads.size() * 2);
    i64 maxLen = 0;
    for (size_t r = 0; r < reads.size(); r++) {
        first[r] = (uint32_t)wins.size();
        const i64 L = reads.length(r);
        if (L < overlap * 2) {
            wins.push_back({(uint32_t)r, 0u, (uint32_t)L});
            maxLen = std::max(maxLen, L);
        } else {
            wins.push_back({(uint32_t)r, 0u, (uint32_t)overlap});
            wins.push_back({(uint32_t)r, (uint32_t)(L - overlap), (uint32_t)overlap});
            maxLen = std::max(maxLen, overlap);
        }
    }
    first[reads.size()] = (uint32_t)wins.size();
    seedCount.assign(wins.size(), 0);
    const i64 blocks = maxLen - 2 * k > 0 ? (maxLen - 2 * k + 3 * k - 1) / (3 * k) : 0;
    stride = (uint32_t)std::max<i64>(1, blocks * k);
    d->nChunks = wins.empty() ? 0u : Impl::chunkOf((uint32_t)wins.size() - 1) + 1;
    d->th = std::thread([this] { producer(); });
}

WindowCache::~WindowCache() {
    {
        std::lock_guard<std::mutex> lk(d->mu);
        d->stop = true;
    }
    d->cvSpace.notify_all();
    d->cvProduced.notify_all();
    if (d->th.joinable()) d->th.join();
    std::lock_guard<std::mutex> lk(Impl::sparesMu());
    auto& sp = Impl::spares();
    for (auto& kv : d->live)
        if (sp.size() < 16 && kv.second) sp.push_back(std::move(kv.second));
    for (auto& ch : d->pool)
        if (sp.size() < 16 && ch) sp.push_back(std::move(ch));
}

size_t WindowCache::releaseSpares() {
    std::lock_guard<std::mutex> lk(Impl::sparesMu());
    size_t freed = 0;
    for (auto& ch : Impl::spares())
        if (ch) freed += (ch->spec.capacity() + ch->kmers.capacity()) * 4;
    std::vector<std::unique_ptr<Impl::Chunk>>().swap(Impl::spares());
    return freed;
}

void WindowCache::producer() {
    struct Reg {
        Reg() { sampleProfRegister("cache"); }
        ~Reg() { sampleProfUnregister(); }
    } reg;
    std::vector<dp_scan_item> items;
    for (uint32_t c = 0; c < d->nChunks; c++) {
        std::unique_ptr<Impl::Chunk> ch;
        {
            std::unique_lock<std::mutex> lk(d->mu);
            d->cvSpace.wait(lk, [&] { return d->stop || c < d->released + Impl::SOFT_CAP || d->wanted >= c; });
            if (d->stop) return;
            if (c < d->released) {  // everything in it is committed already (a job restarted far ahead): nothing to produce
                d->produced = c + 1;
                producedWins.store((uint32_t)std::min(wins.size(), Impl::chunkBegin(c + 1)), std::memory_order_release);
                d->cvProduced.notify_all();
                continue;
            }
            if (!d->pool.empty()) {
                ch = std::move(d->pool.back());
                d->pool.pop_back();
            }
        }
        if (!ch) {
            std::lock_guard<std::mutex> lk(Impl::sparesMu());
            auto& sp = Impl::spares();
            if (!sp.empty()) {
                ch = std::move(sp.back());
                sp.pop_back();
            }
        }
        if (!ch) ch.reset(new Impl::Chunk());
        const size_t w0 = Impl::chunkBegin(c), w1 = std::min(wins.size(), Impl::chunkBegin(c + 1)), n = w1 - w0;
        items.resize(n);
        for (size_t i = 0; i < n; i++) {
            items[i].read = wins[w0 + i].read;
            items[i].start = wins[w0 + i].start;
            items[i].n_kmers = wins[w0 + i].len;  // window length in bases
            items[i].min_seeds = 0;
        }
        ch->spec.resize(n * (size_t)numSeeds);
        ch->kmers.resize(n * (size_t)stride);
        const uint32_t* ev = nullptr;
        int rc = 0;
        if (d->ctx) {
            rc = dp_select_windows(d->ctx, items.data(), (uint32_t)n, d->k, numSeeds, ch->spec.data(), &ev, stride);
        } else {  // host stand-in for the selection kernel: the same walk, block winners and evaluated k-mers (AddSeeds :62-156)
            SeedIndex none(d->k, 21);
            const uint32_t mask = (uint32_t)(((uint64_t)1 << (2 * d->k)) - 1);
            for (size_t i = 0; i < n; i++) {
                const Win& wn = wins[w0 + i];
                const char* sq = d->reads->seq(wn.read) + wn.start;
                const i64 L = wn.len;
                none.selectSeeds(sq, L, numSeeds, d->hostValues, ch->spec.data() + i * (size_t)numSeeds, false);
                uint32_t* km = ch->kmers.data() + i * (size_t)stride;
                std::fill(km, km + stride, 0xffffffffu);
                uint32_t at = 0, kmer = 0;
                auto kmerAt = [&](i64 pos) {
                    uint32_t v = 0;
                    for (int j = 0; j < d->k; j++) v = (v << 2) | baseCode((unsigned char)sq[pos + j]);
                    return v;
                };
                if (L >= d->k) kmer = kmerAt(0);
                i64 nextIndex = d->k;
                while (nextIndex < L - d->k) {
                    for (int j = 0; nextIndex < L && j < d->k; j++) {
                        kmer = ((kmer << 2) | baseCode((unsigned char)sq[nextIndex])) & mask;
                        nextIndex++;
                        if (at < stride) km[at++] = kmer;
                    }
                    nextIndex += d->k;
                    if (nextIndex < L - d->k) kmer = kmerAt(nextIndex);
                    nextIndex += d->k;
                }
            }
        }
        if (rc == 0) {
            // the evaluated k-mers leave the library's pinned block (11 MB per chunk of 8192 windows at config 2) and every window
            // gets its seed count, in a few slices on the worker pool: alone, this thread took as long over a chunk as the plan
            // chain takes to walk it, and the planner waited for the cache a quarter of a job
            const size_t parts = n >= 1024 ? 6 : 1;
            parallelFor(parts, [&](size_t part) {
                const size_t i0 = n * part / parts, i1 = n * (part + 1) / parts;
                if (d->ctx) memcpy(ch->kmers.data() + i0 * (size_t)stride, ev + i0 * (size_t)stride, (i1 - i0) * (size_t)stride * 4);
                std::vector<uint32_t> both((size_t)numSeeds * 2);
                for (size_t i = i0; i < i1; i++) {  // commitSeeds enters every selected k-mer and its reverse complement
                    const uint32_t* sp = ch->spec.data() + i * (size_t)numSeeds;
                    for (int j = 0; j < numSeeds; j++) {
                        both[(size_t)j * 2] = sp[j];
                        both[(size_t)j * 2 + 1] = reverseComplementKmer(sp[j], d->k);
                    }
                    std::sort(both.begin(), both.end());
                    seedCount[w0 + i] = (uint16_t)(std::unique(both.begin(), both.end()) - both.begin());
                }
            });
        }
        std::lock_guard<std::mutex> lk(d->mu);
        if (rc != 0) {
            d->failed = true;
            d->error = dp_last_error(d->ctx);
            d->cvProduced.notify_all();
            return;
        }
        d->live[c] = std::move(ch);
        d->produced = c + 1;
        producedWins.store((uint32_t)w1, std::memory_order_release);
        d->cvProduced.notify_all();
    }
}

bool WindowCache::get(uint32_t w, const uint32_t** spec, const uint32_t** kmers, std::string* err) {
    const uint32_t c = Impl::chunkOf(w);
    std::unique_lock<std::mutex> lk(d->mu);
    if (c >= d->produced) {
        if (c > d->wanted) {
            d->wanted = c;
            d->cvSpace.notify_all();
        }
        const double t0 = now();
        d->cvProduced.wait(lk, [&] { return d->failed || d->stop || c < d->produced; });
        g_prof.cacheWaitUs += (long long)((now() - t0) * 1e6);
    }
    auto it = d->live.find(c);
    if (d->failed || it == d->live.end()) {
        if (err) *err = d->failed ? d->error : "window cache: chunk released before use";
        return false;
    }
    const size_t i = w - Impl::chunkBegin(c);
    *spec = it->second->spec.data() + i * (size_t)numSeeds;
    *kmers = it->second->kmers.data() + i * (size_t)stride;
    return true;
}

bool WindowCache::getRun(uint32_t w, const uint32_t** spec, const uint32_t** kmers, uint32_t* wEnd, std::string* err) {
    if (!get(w, spec, kmers, err)) return false;
    *wEnd = (uint32_t)std::min(wins.size(), Impl::chunkBegin(Impl::chunkOf(w) + 1));
    return true;
}

void WindowCache::release(size_t belowRead) {
    const uint32_t w = belowRead < first.size() ? first[belowRead] : (uint32_t)wins.size();
    const uint32_t c = Impl::chunkOf(w);
    std::lock_guard<std::mutex> lk(d->mu);
    if (c <= d->released) return;
    d->released = c;
    for (auto it = d->live.begin(); it != d->live.end() && it->first < c;) {
        d->pool.push_back(std::move(it->second));
        it = d->live.erase(it);
    }
    d->cvSpace.notify_all();
}

// ---------------------------------------------------------------------------------------------------------------
// Planner: the PrepareQueries chain (overlap.go:157-214 seed selection + commands/overlap.go:128-143 bookkeeping)

struct Planner::Impl {
    ReadSet& reads;
    OverlapParams p;
    ValueView values;
    bool threaded;
    dp_ctx* selCtx;
    WindowCache* winCache = nullptr;
    SeedIndex index;  // selection-side seed set of the plan being computed (inline mode)
    std::mutex mu;
    std::mutex computeMu;     // inline mode: one caller at a time extends the chain (the selection index is shared)
    std::condition_variable cv;
    std::map<i64, std::shared_ptr<RoundPlan>> cache;  // the confirmed chain: contiguous from `base`
    // plans computed ahead of their predecessor, by (round, firstSequence they assumed): they join the chain when the
    // predecessor's firstOut turns out to be that firstSequence, and are dropped otherwise
    std::map<std::pair<i64, i64>, std::shared_ptr<RoundPlan>> pending;
    i64 wantUpTo = -1;        // prefetch target (highest requested round + depth)
    i64 base = 0;             // rounds below are committed and gone
    i64 startFirstIn = 0;     // firstSequence of round `base` (the committed state): the chain can always restart here
    uint64_t epoch = 0;       // bumped whenever an ignore flag is set
    bool stop = false;
    bool fromCache = false;   // plans come from the window cache (QueryEdges with the cache's numSeeds)
    long testDelayUs = 0;     // DPH_TEST_PLAN_DELAY_US: sleep before every compute (tests/test_planner_epoch.py)
    struct Lane {
        std::thread th;
        SeedIndex index;
        bool busy = false;
        i64 round = -1, firstIn = -1;
        i64 maxFlagged = -1;  // largest read id flagged since this lane's compute started (-1: none)
        Lane(int k) : index(k, 21) {}
    };
    std::vector<std::unique_ptr<Lane>> lanes;
    // round-parallel ownership (setOwnership): rounds m with m % ownWorld == ownRank are this planner's to compute
    int ownRank = 0, ownWorld = 1;
    mutable std::unordered_map<i64, i64> predMemo;  // firstIn -> predictFirstOut(firstIn) under the flags of predEpoch
    mutable uint64_t predEpoch = ~0ull;
    mutable bool predStarved = false;  // a guess was wanted for windows the window cache has not produced yet: whoever waits, waits briefly
    Impl(ReadSet& r, const OverlapParams& pp, ValueView v, bool t, dp_ctx* sc)
        : reads(r), p(pp), values(v), threaded(t), selCtx(sc), index(pp.k, 21) {}
};

// Lanes: the plan of round r+1 starts where the plan of round r ends, and where that is is known with certainty only when
// r's plan is finished - but it can be guessed in a microsecond: the budget test (overlap.go:57-60) looks at the number of
// seeds once per read, and an untouched window adds exactly the seeds of its cached selection (WindowCache::seedsOf; a
// re-selected window adds as many, bar a reverse complement that is a seed already).  A free lane therefore starts the
// next plan from the guessed firstSequence while the one in front of it is still being computed; the guess is checked when
// the predecessor joins the chain (promote()), and a plan that started from a wrong one is computed again.
Planner::Planner(ReadSet& reads, const OverlapParams& p, ValueView values, bool threaded, dp_ctx* selCtx, WindowCache* cache)
    : d(new Impl(reads, p, values, threaded, selCtx)) {
    d->winCache = cache;
    // flag epochs are unique per planner (= per job): device contexts that outlive a job (OverlapRun::shutdown(keepContexts))
    // remember the epoch of the flags they hold, and a new job's epoch n must not look like the old job's epoch n
    static std::atomic<uint64_t> serial{0};
    d->epoch = (serial.fetch_add(1) + 1) << 32;
    d->testDelayUs = dph_tune("plan_delay_us", 0);  // (test hook)
    d->fromCache = cache && (p.queryType & 1) && !(p.queryType & 8) && p.numSeeds == cache->numSeeds;
    if (threaded) setLanes(1);
}

// Lanes for a run of `world` ranks (in the round-parallel mode every rank's planner walks the whole chain while its GPU
// executes one round in `world`, so the chain has to be `world` times faster than a GPU) next to `slots` executor threads.
int Planner::lanesFor(int world, int slots) {
    if (const char* e = getenv("DPH_PLAN_LANES")) return std::max(1, std::min(8, atoi(e)));
    const int spare = (int)hostThreads() - std::max(1, slots) - 4;  // slots, window cache, formatter, commit
    // (ownership mode - setOwnership - leaves a rank's planner 1 / world of the chain: three lanes as on one GPU; without it the
    // whole chain has to be walked `world` times faster than a GPU executes)
    (void)world;
    return std::max(1, std::min(3, spare));
}
// ... and how many it may grow to while the slots turn out to wait for their plans (OverlapRun::step: a lane more whenever the slots
// spent more than 10 % of the last 64 rounds waiting - the fast hosts' 2-6 % stay below that, and a fourth lane there makes every plan
// 10 % slower for nothing (`ab_adapt.txt`)).  The pool's hosts differ by a third in single-thread speed: on the slower ones
// a plan takes 0.45 ms instead of 0.33, three lanes are 84 % busy and the slots waited 18 ms each per job; on the fast ones three are
// enough and more threads are only more contention.
int Planner::lanesMax(int world, int slots) {
    if (getenv("DPH_PLAN_LANES")) return lanesFor(world, slots);
    const int spare = (int)hostThreads() - std::max(1, slots) - 4;
    return std::max(lanesFor(world, slots), std::min(6, spare));
}
int Planner::lanes() const {
    std::lock_guard<std::mutex> lk(d->mu);
    return (int)d->lanes.size();
}

void Planner::setLanes(int n) {
    std::lock_guard<std::mutex> lk(d->mu);
    if (!d->threaded) return;
    if (!d->fromCache) n = 1;  // guessing needs the window cache's counts
    while ((int)d->lanes.size() < n) {
        const size_t i = d->lanes.size();
        d->lanes.emplace_back(new Impl::Lane(d->p.k));
        d->lanes[i]->th = std::thread([this, i] { laneMain(i); });
    }
    g_prof.planLanes = (long long)d->lanes.size();
}

Planner::~Planner() {
    {
        std::lock_guard<std::mutex> lk(d->mu);
        d->stop = true;
    }
    d->cv.notify_all();
    for (auto& l : d->lanes)
        if (l->th.joinable()) l->th.join();
}

std::shared_ptr<RoundPlan> Planner::compute(i64 round, i64 firstIn, SeedIndex& index) {
    const double tc0 = now();
    struct Tick {
        double t0, c0;
        ~Tick() {
            g_prof.planComputes++;
            g_prof.planUs += (long long)((now() - t0) * 1e6);
            g_prof.plannerCpuUs += (long long)((threadCpuNow() - c0) * 1e6);
        }
    } tick{tc0, threadCpuNow()};
    auto plan = std::make_shared<RoundPlan>();
    plan->round = round;
    plan->firstIn = firstIn;
    index.reset();
    // the device path needs the window in the resident (cached-view) form and at most 64 list slots
    dp_ctx* sel = (d->selCtx && d->p.numSeeds <= 64) ? d->selCtx : nullptr;
    Overlapper lap(sel, d->reads, index, d->p.chunkSize, d->p.numWorkers, d->p.overlapSize, d->p.numSeeds, d->p.minHits);
    lap.setWindowCache(d->winCache);
    const int nw = lap.PrepareQueries(d->p.numSeeds, d->p.seedBatchSize, d->values, firstIn, d->p.queryBatchSize, d->p.queryType);
    if (nw < 0) {
        plan->error = lap.err;
        plan->failed = true;
    }
    plan->empty = nw <= 0;
    plan->windows = lap.windows();
    plan->seedMap = index.seedMap;
    index.buildRcTable();  // (the executor slot takes both as they are: SeedIndex::adopt)
    plan->rcOf = index.rcOf;
    // firstSequence = max query SequenceID + 1 (commands/overlap.go:135-142); windows are in ascending read order
    plan->firstOut = nw ? (i64)plan->windows.back().read + 1 : firstIn;
    return plan;
}

// First round the confirmed chain lacks and the firstSequence it starts from; false once the chain has ended.
bool Planner::chainHead(i64* round, i64* firstIn) const {
    i64 m = d->base, f = d->startFirstIn;
    for (;;) {
        auto it = d->cache.find(m);
        if (it == d->cache.end()) break;
        if (it->second->empty) return false;
        f = it->second->firstOut;
        m++;
    }
    *round = m;
    *firstIn = f;
    return true;
}

// Plans computed ahead join the chain as far as their assumptions hold; what can no longer join is dropped.
void Planner::promote() {
    if (d->ownWorld > 1) return;  // (ownership mode: plans stay where get() looks them up - by round and assumed start)
    i64 m = 0, f = 0;
    bool alive;
    while ((alive = chainHead(&m, &f))) {
        auto it = d->pending.find({m, f});
        if (it == d->pending.end()) break;
        d->cache[m] = it->second;
        d->pending.erase(it);
    }
    for (auto it = d->pending.begin(); it != d->pending.end();) {
        if (!alive || it->first.first < m || (it->first.first == m && it->first.second != f)) {
            g_prof.planDiscarded++;
            it = d->pending.erase(it);
        } else {
            ++it;
        }
    }
}

// Where the plan that starts at firstIn will end if none of its windows is re-selected (prepareFromCache's loop with the
// windows' seed counts); firstIn itself for a plan without windows, -1 while the window cache has not got that far.
i64 Planner::predictFirstOut(i64 firstIn) const {
    const WindowCache& wc = *d->winCache;
    const i64 n = (i64)d->reads.size();
    if (firstIn != 0 && firstIn >= n) return firstIn;
    i64 sent = 0, seeds = 0, last = -1;
    for (i64 r = firstIn; r < n && sent < d->p.queryBatchSize; r++) {
        if (flagLoad(&d->reads.ignore[(size_t)r])) continue;
        sent++;
        if (seeds >= d->p.seedBatchSize) break;
        for (uint32_t w = wc.first[(size_t)r]; w < wc.first[(size_t)r + 1]; w++) {
            const int c = wc.seedsOf(w);
            if (c < 0) return -1;
            seeds += c;
        }
        last = r;
    }
    return last < 0 ? firstIn : last + 1;
}

void Planner::setOwnership(int rank, int world) {
    std::lock_guard<std::mutex> lk(d->mu);
    static const bool off = false;
    if (off || world < 2 || !d->threaded || !d->fromCache) {
        d->ownRank = 0;
        d->ownWorld = 1;
    } else {
        d->ownRank = rank;
        d->ownWorld = world;
    }
    d->cv.notify_all();
}
bool Planner::sparse() const { return d->ownWorld > 1; }

i64 Planner::predictMemo(i64 firstIn) const {
    if (d->predEpoch != d->epoch) {
        d->predMemo.clear();
        d->predEpoch = d->epoch;
    }
    auto it = d->predMemo.find(firstIn);
    if (it != d->predMemo.end()) return it->second;
    const i64 fo = predictFirstOut(firstIn);
    if (fo >= 0) d->predMemo[firstIn] = fo;
    return fo;
}

// Ownership mode: where the chain - committed truth at `base`, this rank's own finished plans where there are some, guesses
// everywhere else - says round `round` starts.  0: *firstIn is set.  1: the chain ends before `round` (*ended = the owned empty
// plan that says so, or null when a guess does).  -1: cannot tell yet (the window cache has not got that far).
int Planner::sparseEstimate(i64 round, i64* firstIn, std::shared_ptr<RoundPlan>* ended) const {
    i64 m = d->base, f = d->startFirstIn;
    if (ended) ended->reset();
    while (m < round) {
        if (m % d->ownWorld == d->ownRank) {
            auto it = d->pending.find({m, f});
            if (it != d->pending.end()) {
                if (it->second->empty) {
                    if (ended) *ended = it->second;
                    return 1;
                }
                f = it->second->firstOut;
                m++;
                continue;
            }
        }
        const i64 fo = predictMemo(f);
        if (fo < 0) {
            d->predStarved = true;
            return -1;
        }
        if (fo == f) return 1;  // (the guess: no window left - whether that is so is the owner's plan to say)
        f = fo;
        m++;
    }
    *firstIn = f;
    return 0;
}

// Ownership mode: the first owned round up to wantUpTo that nobody has planned or is planning from where the chain says it starts.
bool Planner::sparseNext(i64* round, i64* firstIn) const {
    i64 m = d->base;
    while (m % d->ownWorld != d->ownRank) m++;
    for (; m <= d->wantUpTo; m += d->ownWorld) {
        i64 f = 0;
        const int st = sparseEstimate(m, &f, nullptr);
        if (st != 0) return false;  // the chain ends before m / cannot tell yet
        if (d->pending.count({m, f})) continue;
        bool inFlight = false;
        for (auto& l : d->lanes)
            if (l->busy && l->round == m && l->firstIn == f) inFlight = true;
        if (inFlight) continue;
        *round = m;
        *firstIn = f;
        return true;
    }
    return false;
}

// The first plan nobody has computed or is computing, following the chain through the plans computed ahead and, past a
// plan still being computed, through the guess of where it ends.
bool Planner::nextWork(i64* round, i64* firstIn) const {
    if (d->ownWorld > 1) return sparseNext(round, firstIn);
    i64 m = 0, f = 0;
    if (!chainHead(&m, &f)) return false;
    while (m <= d->wantUpTo) {
        auto it = d->pending.find({m, f});
        if (it != d->pending.end()) {
            if (it->second->empty) return false;
            f = it->second->firstOut;
            m++;
            continue;
        }
        bool inFlight = false;
        for (auto& l : d->lanes)
            if (l->busy && l->round == m && l->firstIn == f) inFlight = true;
        if (!inFlight) {
            *round = m;
            *firstIn = f;
            return true;
        }
        if (d->lanes.size() < 2) return false;
        const i64 fo = predictFirstOut(f);
        if (fo < 0 || fo == f) return false;  // cannot tell yet / the chain is about to end: wait for the real thing
        f = fo;
        m++;
    }
    return false;
}

void Planner::laneMain(size_t li) {
    struct Reg {
        Reg() { sampleProfRegister("lane"); }
        ~Reg() { sampleProfUnregister(); }
    } reg;
    std::unique_lock<std::mutex> lk(d->mu);  // (setLanes may still be growing the vector)
    Impl::Lane& me = *d->lanes[li];
    for (;;) {
        if (d->stop) return;
        i64 m = 0, firstIn = 0;
        d->predStarved = false;
        if (!nextWork(&m, &firstIn)) {
            // (the window cache's producer does not announce its progress: a guess that waits for it is asked for again shortly)
            if (d->predStarved) d->cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(200));  // (system clock: the form ThreadSanitizer follows)
            else d->cv.wait(lk);
            continue;
        }
        me.busy = true;
        me.round = m;
        me.firstIn = firstIn;
        me.maxFlagged = -1;
        const uint64_t e0 = d->epoch;
        lk.unlock();
        static const bool dbg = dph_debug("planner");
        if (dbg) fprintf(stderr, "[planner %zu] computing plan %lld (firstIn %lld, wantUpTo %lld, base %lld)\n", li, (long long)m, (long long)firstIn, (long long)d->wantUpTo, (long long)d->base);
        std::shared_ptr<RoundPlan> plan = compute(m, firstIn, me.index);
        if (d->testDelayUs > 0) usleep((useconds_t)d->testDelayUs);  // test hook: flags arrive after this plan has read them
        if (dbg) fprintf(stderr, "[planner %zu] plan %lld done: %zu windows, %zu seeds, firstOut %lld, empty %d failed %d\n", li, (long long)m, plan->windows.size(), plan->seedMap.size(), (long long)plan->firstOut, (int)plan->empty, (int)plan->failed);
        lk.lock();
        me.busy = false;
        if (d->stop) return;
        // discard if flags that could matter changed meanwhile.  PrepareQueries looks at ignore[r] for every r >= firstIn,
        // so the plan is stale as soon as ANY read flagged during the compute has id >= firstIn - i.e. when the largest
        // one does (a commit usually flags ids on both sides of firstIn).  Whether it continues the chain is promote()'s test.
        if ((d->epoch != e0 && me.maxFlagged >= firstIn) || m < d->base) g_prof.planDiscarded++;
        else d->pending[{m, firstIn}] = plan;
        if (d->ownWorld > 1) {  // (ownership mode: nothing joins a confirmed chain - plans are looked up by (round, where the chain says it starts))
            for (auto it = d->pending.begin(); it != d->pending.end();)
                it = it->first.first < d->base ? d->pending.erase(it) : std::next(it);
        } else {
            promote();
        }
        d->cv.notify_all();
    }
}

std::shared_ptr<const RoundPlan> Planner::get(i64 round) {
    std::unique_lock<std::mutex> lk(d->mu);
    if (!d->threaded) {  // inline chain: compute the first missing plan until `round` is there (or the chain has ended)
        lk.unlock();
        std::lock_guard<std::mutex> cl(d->computeMu);  // executor slots call concurrently
        lk.lock();
        for (;;) {
            auto it = d->cache.find(round);
            if (it != d->cache.end()) return it->second;
            i64 m = d->base, firstIn = d->startFirstIn;
            for (;;) {
                auto e = d->cache.find(m);
                if (e == d->cache.end()) break;
                if (e->second->empty) return e->second;
                firstIn = e->second->firstOut;
                m++;
            }
            if (m > round) return nullptr;
            lk.unlock();
            auto plan = compute(m, firstIn, d->index);
            lk.lock();
            d->cache[m] = plan;
        }
    }
    // plans computed beyond the highest round anybody asked for (DPH_PLAN_DEPTH): the slots ask in bursts - a commit that was
    // holding the issue window releases several rounds at once - and a lane needs 0.3 ms per plan
    static const i64 depth = std::max(1L, dph_tune("plan_depth", 6));
    if (d->ownWorld > 1 && round % d->ownWorld != d->ownRank) {
        // somebody asks for a round this rank does not own (not the round-parallel workers): the whole chain from here on
        d->ownWorld = 1;
        d->ownRank = 0;
        promote();
    }
    if (d->ownWorld > 1) {
        // ownership mode: the plan of an owned round from where the chain - truth, own plans, guesses - says it starts
        const i64 want = round + depth * d->ownWorld;
        if (want > d->wantUpTo) d->wantUpTo = want;
        d->cv.notify_all();
        for (;;) {
            i64 f = 0;
            std::shared_ptr<RoundPlan> ended;
            const int st = sparseEstimate(round, &f, &ended);
            if (st == 1) {
                if (ended) return ended;
                auto e = std::make_shared<RoundPlan>();  // "the input ends before this round" as a guess says it: an empty plan
                e->round = -3;                           // of no round - a result made from it is never valid at the commit, the
                e->empty = true;                         // round is simply planned again once the committed chain has reached it
                return e;
            }
            if (st == 0) {
                auto it = d->pending.find({round, f});
                if (it != d->pending.end()) return it->second;
            }
            if (st < 0) d->cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(200));  // (system clock: the form ThreadSanitizer follows)
            else d->cv.wait(lk);
        }
    }
    if (round + depth > d->wantUpTo) d->wantUpTo = round + depth;
    d->cv.notify_all();
    for (;;) {
        auto it = d->cache.find(round);
        if (it != d->cache.end()) return it->second;
        // the chain may have ended before `round`
        for (i64 m = d->base; m < round; m++) {
            auto e = d->cache.find(m);
            if (e == d->cache.end()) break;
            if (e->second->empty) return e->second;
        }
        d->cv.wait(lk);
    }
}

i64 Planner::applyIgnores(const std::vector<int>& ids, i64 committedRound) {
    std::lock_guard<std::mutex> lk(d->mu);
    i64 maxNew = -1;
    for (int id : ids) {
        if (!d->reads.ignore[(size_t)id]) {
            flagStore(&d->reads.ignore[(size_t)id], 1);
            if (id > maxNew) maxNew = id;
        }
    }
    if (maxNew < 0) return -1;
    d->epoch++;
    for (auto& l : d->lanes)
        if (maxNew > l->maxFlagged) l->maxFlagged = maxNew;
    for (auto it = d->pending.begin(); it != d->pending.end();) {  // plans computed ahead looked at the flags too
        if (it->first.second <= maxNew) {
            g_prof.planErased++;
            it = d->pending.erase(it);
        } else {
            ++it;
        }
    }
    // every cached plan of a later round that starts at or before a newly flagged read may change
    i64 firstBad = -1;
    for (auto it = d->cache.begin(); it != d->cache.end(); ++it) {
        if (it->first > committedRound && it->second->firstIn <= maxNew) {
            firstBad = it->first;
            break;
        }
    }
    if (firstBad >= 0) {
        auto lb = d->cache.lower_bound(firstBad);
        g_prof.planErased += (long long)std::distance(lb, d->cache.end());
        d->cache.erase(lb, d->cache.end());
    }
    d->cv.notify_all();
    return firstBad;
}

uint64_t Planner::ignoreEpoch() {
    std::lock_guard<std::mutex> lk(d->mu);
    return d->epoch;
}

void Planner::dropBefore(i64 round, i64 firstInOfRound) {
    std::lock_guard<std::mutex> lk(d->mu);
    // `round` is the first uncommitted round and starts at firstInOfRound whatever this planner has cached (in a
    // multi-rank run other ranks executed rounds this planner never looked at)
    d->cache.erase(d->cache.begin(), d->cache.lower_bound(round));
    d->base = round;
    d->startFirstIn = firstInOfRound;
    if (d->ownWorld > 1)
        for (auto it = d->pending.begin(); it != d->pending.end();) it = it->first.first < round ? d->pending.erase(it) : std::next(it);
    if (d->winCache) {  // (a lane may still be reading the windows of a plan that started from an older guess)
        i64 keep = firstInOfRound;
        for (auto& l : d->lanes)
            if (l->busy && l->firstIn < keep) keep = l->firstIn;
        d->winCache->release((size_t)keep);
    }
    auto it = d->cache.find(round);
    if (it != d->cache.end() && it->second->firstIn != firstInOfRound) d->cache.erase(it, d->cache.end());  // stale chain
    promote();
    d->cv.notify_all();
}

// ---------------------------------------------------------------------------------------------------------------
// commands/overlap.go Run :96-195

OverlapRun::~OverlapRun() { shutdown(); }

void OverlapRun::HugeTable::assign(const double* src, size_t count) {
    memcpy(reserve(count), src, count * sizeof(double));
}

double* OverlapRun::HugeTable::reserve(size_t count) {
    const size_t huge = (size_t)2 << 20;
    if (base_ && bytes == (count * sizeof(double) + huge - 1) / huge * huge) {
        // the table of the previous job on this handle had the same size: its pages are mapped and touched already (unmapping
        // 134 MB and faulting them in again under the next copy cost 10 ms + half of the copy's time per job)
        n = count;
        return p;
    }
    clear();
    bytes = (count * sizeof(double) + huge - 1) / huge * huge;
    void* m = mmap(nullptr, bytes + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) throw std::bad_alloc();
    base_ = m;
    mapped_ = bytes + huge;
    p = (double*)(((uintptr_t)m + huge - 1) / huge * huge);
    madvise(p, bytes, MADV_HUGEPAGE);  // advisory: falls back to small pages silently
    n = count;
    return p;
}

const double* OverlapRun::fullValues() {
    if (values.size() == 0 && !valueLut.empty()) {
        const size_t nk = (size_t)1 << (2 * p.k);
        double* dst = values.reserve(nk);
        if (codes8_) {
            const uint8_t* codes = (const uint8_t*)valueCodes.data();
            for (size_t i = 0; i < nk; i++) dst[i] = valueLut[codes[i]];
        } else {
            const uint16_t* codes = (const uint16_t*)valueCodes.data();
            for (size_t i = 0; i < nk; i++) dst[i] = valueLut[codes[i]];
        }
    }
    return values.data();
}

void OverlapRun::HugeTable::clear() {
    if (base_) munmap(base_, mapped_);
    base_ = nullptr;
    p = nullptr;
    n = bytes = mapped_ = 0;
}

// keepContexts: the job is over but the handle goes on to another one on the same resident reads (dph_overlap_reset): the
// executor slots' and the planner's device contexts - streams, per-round device and pinned buffers - stay; tearing them down
// and building them again cost 25 ms between two jobs of 0.2 s (some sixty buffers per context, every free a device-wide wait).
void OverlapRun::shutdown(bool keepContexts) {
    sampleProfStop();
    {
        std::lock_guard<std::mutex> lk(pmu_);
        stopWorkers_ = true;
    }
    cvWork_.notify_all();
    for (auto& t : workers_)
        if (t.joinable()) t.join();
    workers_.clear();
    ready_.clear();
    redo_.clear();
    textPool.reset();  // (finishes the jobs still queued; their results are dropped with their rounds)
    if (planner) g_prof.print();
    planner.reset();
    winCache.reset();
    for (auto& sl : slots) {
        sl->lap.reset();
        sl->index.reset();
    }
    if (keepContexts) return;
    if (plannerCtx) dp_ctx_destroy(plannerCtx);
    plannerCtx = nullptr;
    for (auto& sl : slots) {
        if (sl->ownsCtx && sl->ctx) dp_ctx_destroy(sl->ctx);
        sl->ctx = nullptr;
    }
    slots.clear();
}


int OverlapRun::init(dp_ctx* c, ReadSet* r, const OverlapParams& params, const double* valuesOrNull, int nSlots) {
    ctx = c;
    reads = r;
    p = params;
    reads->himem = p.himem;
    errText.clear();
    error.clear();
    paf.clear();
    pafLines = 0;
    badBack = emptyMatch = 0;
    numQuerySeqs = 0;
    last = RoundStats();
    total = RoundStats();
    char line[160];
    snprintf(line, sizeof line, "Counting all %d-mers in the input...\n", p.k);
    errText += line;
    bool valuesOnDevice = false;
    double tm = now();
    auto mark = [&](const char* what) {
        if (!g_prof.on) return;
        const double t = now();
        fprintf(stderr, "[setup] %-28s %.1f ms\n", what, 1e3 * (t - tm));
        tm = t;
    };
    {
        // The k-mer position index first, when this read set gets one: its radix-sort build leaves the k-mer histogram
        // behind, which is what the value table starts from (no separate counting pass then).
        int rc = dp_scan_prepare(ctx, p.k);
        if (rc != 0) {
            error = dp_last_error(ctx);
            return rc;
        }
        mark("k-mer position index");
    }
    if (valuesOrNull) {
        values.assign(valuesOrNull, (size_t)1 << (2 * p.k));
    } else if (dph_tune("host_values", 0)) {  // (tests) histogram on the GPU, table on the host
        std::vector<uint64_t> counts((size_t)1 << (2 * p.k));
        int rc = dp_kmer_histogram(ctx, p.k, counts.data());
        if (rc != 0) {
            error = dp_last_error(ctx);
            return rc;
        }
        mark("k-mer histogram");
        std::vector<double> v = kmerValuesFromCounts(counts, p.k);
        mark("value table (host)");
        values.assign(v.data(), v.size());
    } else {  // KmerOccurrences + value table + 1 % cut on the GPU; the table stays resident for dp_select_seeds
        int rc = dp_kmer_values(ctx, p.k, nullptr);
        if (rc != 0) {
            error = dp_last_error(ctx);
            return rc;
        }
        valuesOnDevice = true;
        mark("value table (device)");
    }
    if (reads->isFastq && !reads->qual.empty() && qualityCtx != ctx) {
        // FASTQ: the selection kernels weight values by the quality bytes.  Normally resident since the reads were uploaded
        // (dph_overlap_open); a caller that drives init() on its own context gets them here - before any context borrows the
        // reads (the download thread below does), which dp_quality_upload refuses
        int rc = dp_quality_upload(ctx, (const uint8_t*)reads->qual.data(), reads->off.data(), reads->hasQual.data(), (uint32_t)reads->size());
        if (rc != 0) {
            error = dp_last_error(ctx);
            return rc;
        }
        qualityCtx = ctx;
        mark("quality upload");
    }
    // The host copy of the table (the planner re-selects a window on the host when its speculation did not hold) travels
    // on a second context while the executor slots, the planner's context and the window cache are set up.
    std::thread dl;
    int dlRc = 0;
    std::string dlErr;
    struct Joiner {  // (error returns below must not leave the thread running)
        std::thread& t;
        ~Joiner() {
            if (t.joinable()) t.join();
        }
    } joiner{dl};
    valueLut.clear();
    if (!valuesOnDevice) valueCodes.clear();
    uint64_t dlTotal = 0;
    int dlOverflow = 0;
    if (valuesOnDevice) {
        // 2 bytes per k-mer instead of 8 (dp_values_download_codes): a quarter of the copy and of the host pages to touch
        const size_t nk = (size_t)1 << (2 * p.k);
        values.clear();
        uint16_t* dst = (uint16_t*)valueCodes.reserve((nk + 3) / 4);
        dl = std::thread([this, dst, nk, &dlRc, &dlErr, &dlTotal, &dlOverflow] {
            dp_ctx* c2 = nullptr;
            dlRc = dp_ctx_create_shared(ctx, &c2);
            if (dlRc != 0) {
                dlErr = dp_last_error(nullptr);
                return;
            }
            // one byte per k-mer where that is enough, else two, else the doubles
            codes8_ = false;
            dlOverflow = 1;
            {
                dlRc = dp_values_download_codes8(c2, (uint8_t*)dst, nk, &dlTotal, &dlOverflow);
                codes8_ = dlRc == 0 && !dlOverflow;
            }
            if (dlRc == 0 && dlOverflow) dlRc = dp_values_download_codes(c2, dst, nk, &dlTotal, &dlOverflow);
            if (dlRc == 0 && dlOverflow) {  // a valued k-mer seen more than 65535 times: the table as doubles
                double* full = values.reserve(nk);
                dlRc = dp_values_download(c2, full, nk);
            }
            if (dlRc != 0) dlErr = dp_last_error(c2);
            dp_ctx_destroy(c2);
        });
    }
    errText += "Counting complete. Starting indexing and querying...";
    {
        if (!dph_tune("host_select", 0) && p.numSeeds <= 64 && !valuesOnDevice) {  // value table resident for dp_select_seeds
            int rc = dp_values_upload(ctx, values.data(), values.size());
            if (rc != 0) {
                error = dp_last_error(ctx);
                return rc;
            }
        }
    }
    mark("values upload");
    // contexts kept by a reset() of this handle are taken over when they fit (same owner context, same number of slots)
    const bool reuseSlots = !slots.empty() && (int)slots.size() == std::max(1, nSlots) && slots[0]->ctx == ctx;
    if (!reuseSlots) {
            for (auto& sl : slots)
            if (sl->ownsCtx && sl->ctx) dp_ctx_destroy(sl->ctx);
        slots.clear();
    }
    {
        textPool.reset();
        // formatter threads (DPH_TEXT_THREADS): a round's text is ~0.26 ms of one thread, so two of them cap a run at 7.7 rounds per ms -
        // which is where the rounds arrived in round 5 (0.135 ms each): three where the host has the threads for it (commit's wait for
        // its round's text 46 -> 40 ms per job, 0.141 - 0.152 -> 0.133 - 0.141 ms per round; four and six: the same as three)
        int nText = hostThreads() >= 12 ? 3 : 2;
        if (const char* te = getenv("DPH_TEXT_THREADS")) nText = std::max(1, std::min(16, atoi(te)));
        textPool.reset(new TextPool(nText));
    }
    setHostThreadShare((unsigned)std::max(1, nSlots));
    // every executor slot's thread waits for its stream five to seven times per round: busy waits when this process has the
    // cores for it (the worker pool's size is the CPU budget: cgroup quota, or DP_HOST_THREADS for ranks sharing a host)
    dp_set_stream_wait(hostThreads() >= (unsigned)std::max(1, nSlots) + 3u ? 1 : 0);
    for (int i = 0; i < std::max(1, nSlots); i++) {
        if (reuseSlots) {
            slots[(size_t)i]->index.reset(new SeedIndex(p.k));
            slots[(size_t)i]->comm = (size_t)i < slotComms.size() ? slotComms[(size_t)i] : nullptr;
            continue;
        }
        std::unique_ptr<ExecSlot> sl(new ExecSlot());
        sl->slotNo = i;
        if (i == 0) {
            sl->ctx = ctx;
        } else {
            int rc = dp_ctx_create_shared(ctx, &sl->ctx);
            if (rc != 0) {
                error = dp_last_error(nullptr);
                return rc;
            }
            sl->ownsCtx = true;
        }
        sl->index.reset(new SeedIndex(p.k));
        if ((size_t)i < slotComms.size()) sl->comm = slotComms[(size_t)i];
        slots.push_back(std::move(sl));
    }
    mark("executor slots");
    // (tests: DP_TUNE=no_planner_thread=1 plans on the calling thread, host_select=1 keeps the speculative seed selection on host threads)
    const bool nothread = dph_tune("no_planner_thread", 0) != 0;
    const bool wantPlannerCtx = !dph_tune("host_select", 0) && p.numSeeds <= 64;
    if (plannerCtx && !wantPlannerCtx) {
        dp_ctx_destroy(plannerCtx);
        plannerCtx = nullptr;
    }
    if (wantPlannerCtx && !plannerCtx) {
        int rc = dp_ctx_create_shared(ctx, &plannerCtx);
        if (rc != 0) {
            error = dp_last_error(nullptr);
            return rc;
        }
        // the plan chain is sequential and waits for one small selection kernel per plan: it should not queue behind the
        // executor slots' kernels.  (Measured neutral on config 2 - there the planner's wait grows because the process is
        // at its CPU quota, not because of the GPU queue - kept because it is the right order of service.)
        dp_ctx_set_priority(plannerCtx, 1);
    }
    winCache.reset();
    {
        // QueryEdges without WeightEdges (the overlap command): the windows' speculative selection and evaluated k-mers come
        // from a producer thread that runs ahead of the plan chain on the planner's context.  DP_WINDOW_CACHE=0: per-plan
        // dp_select_seeds calls and base-by-base speculation checks (the round-1 path).
        const char* wc = getenv("DP_WINDOW_CACHE");
        if (plannerCtx && p.queryType == 1 && !(wc && wc[0] == '0'))
            winCache.reset(new WindowCache(plannerCtx, *reads, p.overlapSize, p.k, p.numSeeds));
    }
    if (dl.joinable()) {
        dl.join();
        if (dlRc != 0) {
            error = dlErr;
            return dlRc;
        }
        if (valuesOnDevice && !dlOverflow) {  // overlap.go:73-88, once per possible count
            valueLut.resize(65536);
            const double tf = (double)dlTotal, targetFreq = 0.000005;
            for (uint32_t c = 0; c < 65536; c++) {
                const double freq = (double)c / tf;
                valueLut[c] = c < 3 ? 0.0 : freq <= targetFreq ? 1.0 - (targetFreq - freq) : 1.0 - (freq - targetFreq);
            }
        } else {
            valueCodes.clear();
        }
        mark("values copy (overlapped)");
    }
    planner.reset(new Planner(*reads, p, valueView(), !nothread, plannerCtx, winCache.get()));
    planner->setLanes(std::max(Planner::lanesFor(world_, nSlots), adaptLanes_));
    mark("planner");
    firstSequence = 0;
    round = 0;
    done = false;
    stopWorkers_ = issueEnd_ = draining_ = false;
    adaptRounds_ = 0;  // (the adaptive lane growth measures inside one job: not across the idle gap since the last one)
    adaptT_ = 0;
    adaptWait_ = planWaitUs_.load(std::memory_order_relaxed);
    nextIssue_ = 0;
    inflight_ = workerRc_ = 0;
    flagRound_.assign(reads->size(), -1);
    shardLo = 0;
    shardHi = reads->size();
    initEnd_ = now();
    return 0;
}

// seeds.NewSeedIndex + the plan's seeds on host and device; queries are built after the scan
int OverlapRun::beginRound(ExecSlot& sl, const RoundPlan& plan) {
    const double tb0 = now();
    if (dp_comm* cm = sl.comm ? sl.comm : comm)
        if (const long fr = dph_tune("fail_begin_rank", -1); fr >= 0)  // test hook: this rank's round fails before it reaches any exchange
            if (fr == dp_comm_rank(cm)) {
                sl.error = "injected failure before the exchange (DPH_FAIL_BEGIN_RANK)";
                return -1;
            }
    // seeds.NewSeedIndex(k) per round (:125): the plan's seed list and reverse-complement table, as the planner left them
    if (plan.rcOf.size() == plan.seedMap.size()) {
        sl.index->adopt(plan.seedMap, plan.rcOf);
    } else {
        sl.index->reset();
        for (uint32_t km : plan.seedMap) sl.index->addSeedKmer(km);
        sl.index->buildRcTable();
    }
    const double tb1 = now();
    g_prof.add(0, tb1 - tb0);
    sl.lap.reset(new Overlapper(sl.ctx, *reads, *sl.index, p.chunkSize, p.numWorkers, p.overlapSize, p.numSeeds, p.minHits));
    if (sl.comm && shardQueries && exchangeOrdered_) {
        // this rank's share of the round's query windows (contiguous: the ranks' results joined in rank order are in query order)
        const size_t nw = plan.windows.size(), nr = (size_t)dp_comm_size(sl.comm), me = (size_t)dp_comm_rank(sl.comm);
        const size_t w0 = nw * me / nr, w1 = nw * (me + 1) / nr;
        sl.lap->setWindows(std::vector<Overlapper::Window>(plan.windows.begin() + (i64)w0, plan.windows.begin() + (i64)w1));
    } else {
        sl.lap->setWindows(plan.windows);
    }
    sl.lap->setTextPool(textPool.get());
    {
        // chunkWorker on the device (dp_index_build_chunked) wherever the consensus runs there too; DP_DEVICE_CHUNK=0: host chunks
        static const bool deviceChunk = [] {
            const char* e = getenv("DP_DEVICE_CHUNK");
            const char* c = getenv("DP_DEVICE_CONSENSUS");
            return !(e && e[0] == '0') && !(c && c[0] == '0');
        }();
        sl.lap->setDeviceChunking(deviceChunk);  // (scan-shard mode too: the gathered survivors are chunked where the exchange put them)
        // dp_index_prechain (the chunk stage launched behind the un-waited scan) shortens a round's chain of waits: 0.460 -> 0.440 ms
        // per round with one slot, 0.194 -> 0.186 with three - and nothing with five or six (0.154 / 0.158: the GPU is the limit there,
        // and the guess-sized bit matrices cost what the wait saved; profiles/r04/ab_prechain_s*.txt).  Default: up to three slots.
        const char* pce = getenv("DPH_PRECHAIN");  // (read per round: tests switch it between jobs of one process)
        const int prechainEnv = pce ? (pce[0] == '0' ? 0 : 1) : -1;
        const bool prechain = prechainEnv >= 0 ? prechainEnv == 1 : slots.size() <= 3;
        sl.lap->setPrechain(prechain && deviceChunk && sl.comm == nullptr && comm == nullptr);
    }
    sl.lap->setIgnoreView(reads->ignore.data(), planner->ignoreEpoch());
    const double tb2 = now();
    g_prof.add(1, tb2 - tb1);
    g_prof.add(18, now() - tb2);
    int rc = dp_round_begin(sl.ctx, p.k, sl.index->seedMap.data(), (uint32_t)sl.index->seedMap.size());
    if (rc != 0) {
        sl.error = dp_last_error(sl.ctx);
        return rc;
    }
    g_prof.add(2, now() - tb2);
    return 0;
}

int OverlapRun::finishRound(ExecSlot& sl, const Survivors& all, RoundResult& out) {
    RoundStats& st = out.st;
    double t0 = now();
    int rc = sl.lap->IndexSurvivors(all, st);
    if (rc != 0) {
        sl.error = sl.lap->err;
        return rc;
    }
    out.indexedReads = all.read;
    out.numQuerySeqs = 0;
    for (const SeedQuery& q : sl.lap->queries)
        if (q.ID >= out.numQuerySeqs) out.numQuerySeqs = q.ID + 1;
    double t1 = now();
    st.t_index = t1 - t0;
    // DP_DEVICE_CONSENSUS=0: matches come back to the host, BuildConsensus / finalCheckWorker run on the worker pool (the
    // round-1 path, still what a window the device flags falls back to)
    static const bool deviceConsensus = [] {
        const char* e = getenv("DP_DEVICE_CONSENSUS");
        return !(e && e[0] == '0');
    }();
    double t2;
    if (deviceConsensus && (sl.lap->queries.size() % 2) == 0) {
        rc = sl.lap->FindOverlapsAndFinalCheck(sl.matchPool, p.overlapSize, out.paf, out.fs, &out.ignores, st, &out.text);
        if (rc != 0) {
            sl.error = sl.lap->err;
            return rc;
        }
        t2 = now();
        st.t_query = t2 - t1;
    } else {
        std::vector<SeedMatch*>& matches = sl.matches;
        rc = sl.lap->FindOverlaps(sl.matchPool, matches, st);
        if (rc != 0) {
            sl.error = sl.lap->err;
            return rc;
        }
        t2 = now();
        st.t_query = t2 - t1;
        rc = finalCheck(sl.index->arena, *sl.index, *reads, matches, out.numQuerySeqs, p.overlapSize, out.paf, out.fs, &out.ignores,
                        sl.ctx, &sl.consJobs, &sl.error, &st);
        if (rc != 0) return rc;
    }
    st.n_paf = out.fs.lines;
    st.t_consensus = now() - t2;
    const int k = p.k;
    // algorithmic bytes of the scan: packed bytes of the scanned items + bit table + 8 B per hit
    st.scan_bytes = st.scan_bases / 4 + (((uint64_t)1 << (2 * k)) / 8) + 8 * st.n_hits;
    st.count_bytes = st.scan_bases / 4 + (((uint64_t)1 << (2 * k)) / 8);
    if (st.idx_rounds) {
        // served by the resident k-mer position index: no read is streamed.  Minimum traffic = two table entries per seed
        // k-mer + each occurrence once (8 B) + one count per item
        st.count_bytes = 16 * st.n_seeds + 8 * st.idx_hits + 4 * st.scan_items;
        st.scan_bytes = st.count_bytes + 8 * st.n_hits;
    }
    return 0;
}

// A slot's round in the sharded layouts ends in collectives its peers take part in: whatever made this rank's round fail - the
// plan, dp_round_begin, the scan, the exchange itself - the slot's communicator is aborted, so that the peers of this round's
// exchanges (and, behind their batons, of every later one) return an error instead of waiting for a rank that will not come.
int OverlapRun::executeRoundOn(ExecSlot& sl, i64 r, RoundResult& out) {
    const int rc = executeRoundOnImpl(sl, r, out);
    if (rc != 0 && sl.comm) dp_comm_abort(sl.comm);
    return rc;
}

int OverlapRun::executeRoundOnImpl(ExecSlot& sl, i64 r, RoundResult& out) {
    out = RoundResult();
    out.round = r;
    double t0 = now();
    const double tc0 = g_prof.on ? threadCpuNow() : 0;
    struct SlotCpu {
        double t0;
        ~SlotCpu() {
            if (g_prof.on) g_prof.slotCpuUs += (long long)((threadCpuNow() - t0) * 1e6);
        }
    } slotCpu{tc0};
    struct ExchangeTurn {  // however this round ends, the slots behind it in the batch's exchange order get their turn - in order:
        OverlapRun* run;   // a slot passes a baton on only once it holds it (the slots' collectives are issued in the same order on
        int index;         // every rank, also around a slot whose round turned out empty)
        bool on;
        void pass(int& turn) {
            std::unique_lock<std::mutex> lk(run->exchangeMu_);
            run->exchangeCv_.wait(lk, [&] { return turn >= index; });
            turn = std::max(turn, index + 1);
            run->exchangeCv_.notify_all();
        }
        ~ExchangeTurn() {
            if (!on) return;
            pass(run->exchangeTurn_);
            pass(run->resultTurn_);
        }
    } exchangeTurn{this, sl.slotNo, sl.comm != nullptr && exchangeOrdered_};
    static const bool dbgExec = dph_debug("planner");
    if (dbgExec) fprintf(stderr, "[exec] round %lld waiting for its plan\n", (long long)r);
    std::shared_ptr<const RoundPlan> plan = planner->get(r);
    if (dbgExec) fprintf(stderr, "[exec] round %lld got plan\n", (long long)r);
    static const bool startTrace = dph_debug("start");  // a job's first rounds, in ms since the end of its set-up
    const double tPlan = now();
    {
        const long long waitedUs = (long long)((now() - t0) * 1e6);
        g_prof.getWaitUs += waitedUs;
        planWaitUs_.fetch_add(waitedUs, std::memory_order_relaxed);  // (this handle's own: what its adaptive lane growth looks at)
    }
    if (plan && plan->failed) {
        sl.error = "seed selection failed: " + plan->error;
        return -1;
    }
    out.planRound = plan ? plan->round : -2;
    if (!plan || plan->empty || plan->round != r) {
        out.empty = true;
        if (plan) out.firstIn = out.firstOut = plan->firstOut;
        return 0;
    }
    out.empty = false;
    out.firstIn = plan->firstIn;
    out.firstOut = plan->firstOut;
    out.queryReads.reserve(plan->windows.size());
    for (const auto& w : plan->windows) out.queryReads.push_back(w.read);
    int rc = beginRound(sl, *plan);
    if (rc) return rc;
    out.st.n_seeds = plan->seedMap.size();
    out.st.gang_members = 1;
    double t1 = now();
    out.st.t_prepare = t1 - t0;
    const bool sharded = sl.comm != nullptr;  // scan-shard: this rank scans its reads, the survivors of all ranks are exchanged
    rc = sl.lap->ScanLocal(sharded ? shardLo : 0, sharded ? shardHi : reads->size(), sl.local, out.st);
    static const bool dbgX = dph_debug("exchange");
    if (dbgX) fprintf(stderr, "[x %p] slot %d round %lld scanned rc %d, waiting for turn (turn %d)\n", (void*)this, sl.slotNo, (long long)r, rc, exchangeTurn_);
    if (sharded && exchangeOrdered_) {  // this slot's turn among the batch's exchanges (taken also by a slot whose scan failed)
        std::unique_lock<std::mutex> lk(exchangeMu_);
        exchangeCv_.wait(lk, [&] { return exchangeTurn_ >= sl.slotNo; });
    }
    if (dbgX) fprintf(stderr, "[x %p] slot %d round %lld exchanging\n", (void*)this, sl.slotNo, (long long)r);
    if (rc == 0 && sharded) rc = sl.lap->ExchangeSurvivors(sl.comm, sl.gathered);
    if (dbgX) fprintf(stderr, "[x %p] slot %d round %lld exchanged rc %d\n", (void*)this, sl.slotNo, (long long)r, rc);
    if (sharded && exchangeOrdered_) {
        std::lock_guard<std::mutex> lk(exchangeMu_);
        exchangeTurn_ = std::max(exchangeTurn_, sl.slotNo + 1);
        exchangeCv_.notify_all();
    }
    if (rc != 0) {
        sl.error = sl.lap->err;
        if (sharded) dp_comm_abort(sl.comm);  // (the peers of this round's exchange must not wait for a rank that will not come)
        return rc;
    }
    out.st.t_scan = now() - t1;
    rc = finishRound(sl, sharded ? sl.gathered : sl.local, out);
    if (dbgX) fprintf(stderr, "[x %p] slot %d round %lld finished rc %d\n", (void*)this, sl.slotNo, (long long)r, rc);
    if (sharded && shardQueries && exchangeOrdered_) {
        // the ranks' shares of the round joined: [text bytes | ignores | counters] per rank, rank order = query order
        if (rc == 0) out.takeText();
        std::string blob;
        {
            const uint64_t hdr[8] = {(uint64_t)out.paf.size(), (uint64_t)out.ignores.size(), (uint64_t)out.fs.badBack, (uint64_t)out.fs.emptyMatch,
                                     out.fs.lines, out.fs.hits, out.fs.qHits, (uint64_t)(rc == 0 ? 1 : 0)};
            blob.append((const char*)hdr, sizeof hdr);
            blob.append(out.paf);
            blob.append((const char*)out.ignores.data(), out.ignores.size() * sizeof(int));
        }
        {
            std::unique_lock<std::mutex> lk(exchangeMu_);
            exchangeCv_.wait(lk, [&] { return resultTurn_ >= sl.slotNo; });
        }
        const uint8_t* all = nullptr;
        const uint64_t* sizes = nullptr;
        int rc2 = rc == 0 ? dp_allgather_blobs(sl.comm, sl.ctx, (const uint8_t*)blob.data(), blob.size(), &all, &sizes) : rc;
        {
            std::lock_guard<std::mutex> lk(exchangeMu_);
            resultTurn_ = std::max(resultTurn_, sl.slotNo + 1);
            exchangeCv_.notify_all();
        }
        if (rc == 0 && rc2 != 0) {
            sl.error = dp_last_error(sl.ctx);
            rc = rc2;
        }
        if (rc != 0) {
            dp_comm_abort(sl.comm);
        } else {
            out.paf.clear();
            out.ignores.clear();
            out.fs = FinalCheckStats();
            const int nr = dp_comm_size(sl.comm);
            const uint8_t* q = all;
            for (int i = 0; i < nr; i++) {
                uint64_t hdr[8];
                memcpy(hdr, q, sizeof hdr);
                const uint8_t* body = q + sizeof hdr;
                if (!hdr[7]) {
                    sl.error = "a peer rank failed in this round";
                    rc = -1;
                    break;
                }
                out.paf.append((const char*)body, (size_t)hdr[0]);
                const int* ig = (const int*)(body + hdr[0]);
                out.ignores.insert(out.ignores.end(), ig, ig + hdr[1]);
                out.fs.badBack += (i64)hdr[2];
                out.fs.emptyMatch += (i64)hdr[3];
                out.fs.lines += hdr[4];
                out.fs.hits += hdr[5];
                out.fs.qHits += hdr[6];
                q += sizes[i];
            }
            out.st.n_paf = out.fs.lines;
            out.numQuerySeqs = (i64)plan->windows.size();  // (one query id per window, commands/overlap.go:133-142)
        }
    }
    if (dbgExec) fprintf(stderr, "[exec] round %lld finished rc %d\n", (long long)r, rc);
    const double t2 = now();
    static const bool slowTrace = dph_debug("slow");  // rounds that took a slot more than 3 ms, with where the time went
    if (slowTrace && t2 - t0 > 3e-3)
        fprintf(stderr, "[slow] round %lld slot %d at %.1f ms of its job: %.2f ms = plan wait %.2f + prepare %.2f + scan / count %.2f + index %.2f + query / chain / consensus %.2f + rest %.2f\n",
                (long long)r, sl.slotNo, 1e3 * (t0 - initEnd_), 1e3 * (t2 - t0), 1e3 * (tPlan - t0), 1e3 * (out.st.t_prepare - (tPlan - t0)), 1e3 * out.st.t_scan,
                1e3 * out.st.t_index, 1e3 * (out.st.t_query + out.st.t_consensus),
                1e3 * ((t2 - t0) - (out.st.t_prepare + out.st.t_scan + out.st.t_index + out.st.t_query + out.st.t_consensus)));
    if (startTrace && r < 12)
        fprintf(stderr, "[start] round %lld slot %d: asked for its plan at %.2f ms, had it at %.2f, scanned / counted at %.2f, finished at %.2f\n", (long long)r, sl.slotNo,
                1e3 * (t0 - initEnd_), 1e3 * (tPlan - initEnd_), 1e3 * (t1 + out.st.t_scan - initEnd_), 1e3 * (t2 - initEnd_));
    g_prof.add(14, t2 - t0);
    g_prof.add(15, (t2 - t0) - (out.st.t_prepare + out.st.t_scan + out.st.t_index + out.st.t_query + out.st.t_consensus));
    return rc;
}

int OverlapRun::executeRound(i64 r, RoundResult& out) {
    int rc = executeRoundOn(*slots[0], r, out);
    if (rc) error = slots[0]->error;
    return rc;
}

int OverlapRun::executeRounds(const std::vector<i64>& rounds, std::vector<RoundResult>& outs) {
    const size_t n = std::min(rounds.size(), slots.size());
    outs.assign(n, RoundResult());
    std::vector<int> rcs(n, 0);
    if (n == 1) {
        rcs[0] = executeRoundOn(*slots[0], rounds[0], outs[0]);
    } else {
        std::vector<std::thread> th;
        for (size_t i = 0; i < n; i++) th.emplace_back([&, i] { rcs[i] = executeRoundOn(*slots[i], rounds[i], outs[i]); });
        for (auto& t : th) t.join();
    }
    for (size_t i = 0; i < n; i++)
        if (rcs[i]) {
            error = slots[i]->error;
            return rcs[i];
        }
    return 0;
}

// A commit has a heavy half that only the committing thread's own data sees (the round's text) and a light half that the
// executor slots look at (the commit point, the flags, the planner's chain): step() runs the first without the pipeline's
// lock - a slot that finishes a round meanwhile hands it in at once instead of queueing behind half a megabyte of text.
void OverlapRun::commitText(RoundResult& r) {
    char line[200];
    {
        const double tw = now();
        r.takeText();  // (the round's PAF text may still be with a formatter thread)
        g_prof.textWaitUs += (long long)((now() - tw) * 1e6);
    }
    if (round == 0)
        snprintf(line, sizeof line, "Using query sets of around %lld sequences against %lld sequences.\n", (long long)r.firstOut,
                 (long long)reads->size());
    else
        snprintf(line, sizeof line, "Using query set with %lld  sequences starting from %lld sequences against %lld sequences.\n",
                 (long long)r.numQuerySeqs, (long long)r.firstOut, (long long)reads->size());
    errText += line;
    snprintf(line, sizeof line, "Total %lld hits across %lld overlaps.\n", (long long)r.fs.hits, (long long)r.fs.qHits);
    errText += line;
    badBack += r.fs.badBack;
    emptyMatch += r.fs.emptyMatch;
    if (paf.empty()) paf = std::move(r.paf);  // (the formatter's buffer itself: host_capi.cpp moves it on)
    else paf += r.paf;
    pafLines += (i64)r.fs.lines;
    r.st.timed_rounds = (r.st.k_chain_ms > 0 || r.st.k_query_ms > 0 || r.st.k_cons_ms > 0 || r.st.k_count_ms > 0 || r.st.k_scan_ms > 0) ? 1 : 0;
    last = r.st;
    total.add(r.st);
}

void OverlapRun::commitState(RoundResult& r) {
    firstSequence = r.firstOut;
    numQuerySeqs = r.numQuerySeqs;
    g_prof.ignores += (long long)r.ignores.size();
    for (int id : r.ignores)
        if (!reads->ignore[(size_t)id] && flagRound_[(size_t)id] < 0) flagRound_[(size_t)id] = (int32_t)round;
    planner->applyIgnores(r.ignores, round);
    round++;
    planner->dropBefore(round, firstSequence);
}

void OverlapRun::commitOne(RoundResult& r) {
    commitText(r);
    commitState(r);
}

int OverlapRun::commitResults(std::vector<RoundResult>& results) {
    paf.clear();
    pafLines = 0;
    int committed = 0;
    std::vector<uint8_t> newly;  // flags set by rounds committed in THIS call
    std::vector<int> newIds;
    for (RoundResult& r : results) {
        if (r.round != round) break;
        if (r.empty) {
            if (emptyResultValid(r)) done = true;  // (a stale one is not committed: the caller executes `round` again)
            break;
        }
        // speculation check: the round ran against the flags at the start of this batch.  It is exact iff no read
        // flagged by the earlier rounds of the batch could have been one of its queries (id >= firstIn) or entered
        // its index.
        bool ok = true;
        if (!newIds.empty()) {
            for (int id : newIds)
                if (id >= r.firstIn) {
                    ok = false;
                    break;
                }
            if (ok) {
                if (newly.empty()) {
                    newly.assign(reads->size(), 0);
                    for (int id : newIds) newly[(size_t)id] = 1;
                }
                for (uint32_t rd : r.indexedReads)
                    if (newly[rd]) {
                        ok = false;
                        break;
                    }
            }
        }
        if (!ok) break;
        for (int id : r.ignores)
            if (!reads->ignore[(size_t)id]) {
                newIds.push_back(id);
                if (!newly.empty()) newly[(size_t)id] = 1;
            }
        commitOne(r);
        committed++;
    }
    return committed;
}

void OverlapRun::setRanks(int rank, int world) {
    rank_ = rank;
    world_ = std::max(1, world);
    if (planner) planner->setLanes(std::max(Planner::lanesFor(world_, (int)slots.size()), adaptLanes_));
}

void OverlapRun::startWorkers() {
    if (workers_.empty()) sampleProfStart();
    if (!workers_.empty()) return;
    nextIssue_ = round;
    while (nextIssue_ % world_ != rank_) nextIssue_++;  // first round this rank owns
    if (world_ > 1 && planner) planner->setOwnership(rank_, world_);  // (the rounds dealt to the ranks: each plans its own)
    for (size_t i = 0; i < slots.size(); i++) workers_.emplace_back([this, i] { workerMain(i); });
}

void OverlapRun::workerMain(size_t si) {
    struct Reg {
        Reg() { sampleProfRegister("slot"); }
        ~Reg() { sampleProfUnregister(); }
    } reg;
    ExecSlot& sl = *slots[si];
    // rounds issued ahead of the commit point (owned ones only): commits are in order, so a round that takes longer than its
    // neighbours holds the window; DPH_ISSUE_WINDOW = rounds beyond the slot count (config 2, six slots: +2 0.370, +6 0.344, +12 0.338, +24 0.343 ms per round)
    static const i64 extra = std::max(0L, dph_tune("issue_window", 10));
    const i64 window = ((i64)slots.size() + extra) * world_;
    std::unique_lock<std::mutex> lk(pmu_);
    for (;;) {
        cvWork_.wait(lk, [&] {
            // after the end of the input was seen, only the owned round of the current superstep is still issued (it
            // comes back empty and lets every rank finish the same superstep)
            const bool more = nextIssue_ < round + window && (!issueEnd_ || nextIssue_ < round + world_);
            return stopWorkers_ || (!draining_ && workerRc_ == 0 && (!redo_.empty() || more));
        });
        if (stopWorkers_) return;
        i64 r;
        if (!redo_.empty()) {
            r = redo_.front();
            redo_.pop_front();
        } else {
            r = nextIssue_;
            nextIssue_ += world_;
        }
        const i64 snap = round;
        inflight_++;
        lk.unlock();
        RoundResult res;
        const int rc = executeRoundOn(sl, r, res);
        res.snapshot = snap;
        res.round = r;
        lk.lock();
        inflight_--;
        if (rc != 0 && workerRc_ == 0) {
            workerRc_ = rc;
            workerErr_ = sl.error;
        }
        if (rc == 0) {
            // flags only accumulate: an exhausted input stays exhausted.  Only a real plan of THIS round says so - an empty result
            // from a guessed "the input ends here" plan (ownership mode, round -3) or from another round's empty plan is rejected at
            // the commit anyway, and believing it here drained this rank's pipeline until then
            if (res.empty && (res.planRound == r || res.planRound == -2)) issueEnd_ = true;
            if (!draining_) ready_[r] = std::move(res);
        }
        cvDone_.notify_all();
    }
}

// A round executed against a flag snapshot is exact iff no read flagged since then is one of its queries or entered
// its index, and its plan continues the committed chain.
// "No more queries" ends the command (commands/overlap.go:130), so an empty result is checked like any other: it must
// come from the plan of THIS round (planner->get() hands out an earlier empty plan of a speculative chain when there is
// one, and that chain may have been erased by applyIgnores since) and that plan must start at the committed
// firstSequence.  Flags only accumulate, so an empty plan computed from the right firstIn stays empty.
bool OverlapRun::emptyResultValid(const RoundResult& r) const {
    if (r.planRound == -2) return true;  // inline planner with nothing left to compute
    return r.planRound == round && r.firstIn == firstSequence;
}

bool OverlapRun::resultValid(const RoundResult& r) const {
    if (r.firstIn != firstSequence) return false;
    for (uint32_t id : r.queryReads)
        if (flagRound_[id] >= r.snapshot) return false;
    for (uint32_t id : r.indexedReads)
        if (flagRound_[id] >= r.snapshot) return false;
    return true;
}

int OverlapRun::step() {
    if (done) return 0;
    startWorkers();
    const double t0 = now();
    std::unique_lock<std::mutex> lk(pmu_);
    paf.clear();
    pafLines = 0;
    int committed = 0;
    for (;;) {
        if (workerRc_ != 0) {
            error = workerErr_;
            return workerRc_;
        }
        if (roundLimit >= 0 && round >= roundLimit) break;  // (a step commits every finished round it finds: callers that want exactly n rounds)
        auto it = ready_.find(round);
        if (it == ready_.end()) {
            if (committed) break;
            const double tw = now();
            cvDone_.wait(lk);
            g_prof.commitWaitUs += (long long)((now() - tw) * 1e6);
            continue;
        }
        RoundResult res = std::move(it->second);
        ready_.erase(it);
        if (res.empty) {
            if (emptyResultValid(res)) {
                done = true;
                break;
            }
            g_prof.rejected++;
            redo_.push_back(round);
            issueEnd_ = false;
            cvWork_.notify_all();
            continue;
        }
        g_prof.executed++;
        if (!resultValid(res)) {
            g_prof.rejected++;
            redo_.push_back(round);
            cvWork_.notify_all();
            continue;
        }
        lk.unlock();
        const double tc = now();
        commitText(res);
        const double tc1 = now();
        lk.lock();
        commitState(res);
        g_prof.commitTextUs += (long long)((tc1 - tc) * 1e6);
        g_prof.commitStateUs += (long long)((now() - tc1) * 1e6);
        committed++;
        g_prof.committed++;
        cvWork_.notify_all();
        if (planner && world_ == 1 && ++adaptRounds_ >= 64) {  // do the slots wait for their plans?  then the planner gets another lane
            const double tn = now();
            const long long w = planWaitUs_.load(std::memory_order_relaxed);  // (per handle: other handles' waits are not this planner's)
            if (adaptT_ > 0 && round >= 192) {  // (not the job's first rounds: the slots wait for the first plans whatever the host)
                const double waited = (double)(w - adaptWait_) * 1e-6, span = (tn - adaptT_) * (double)std::max<size_t>(1, slots.size());
                const int have = planner->lanes();
                if (waited > 0.10 * span && have < Planner::lanesMax(world_, (int)slots.size())) {
                    planner->setLanes(have + 1);
                    adaptLanes_ = have + 1;  // (kept for the handle's next job: its planner starts with as many)
                }
            }
            adaptT_ = tn;
            adaptWait_ = w;
            adaptRounds_ = 0;
        }
    }
    g_prof.execUs += (long long)((now() - t0) * 1e6);
    if (done) sampleProfStop();
    return committed;
}

// This rank's contribution to a superstep: its next owned round (waited for) and, up to maxRounds in all, the owned rounds
// after it that are finished already - an exchange that takes as long as several rounds then carries several rounds, and the
// ranks need not agree on how many: commitGathered commits whatever prefix of the gathered rounds is contiguous and valid,
// the rest stays with its owner for the next superstep.
int OverlapRun::waitOwned(std::vector<RoundResult>& outs, int maxRounds) {
    startWorkers();
    std::unique_lock<std::mutex> lk(pmu_);
    i64 mine = round;
    while (mine % world_ != rank_) mine++;
    cvWork_.notify_all();
    cvDone_.wait(lk, [&] { return workerRc_ != 0 || ready_.count(mine) != 0; });
    if (workerRc_ != 0) {
        error = workerErr_;
        return workerRc_;
    }
    outs.clear();
    for (int i = 0; i < std::max(1, maxRounds); i++, mine += world_) {
        auto it = ready_.find(mine);
        if (it == ready_.end()) break;
        it->second.takeText();       // (into the result itself: it may be contributed again if its predecessor is rejected)
        outs.push_back(it->second);  // a copy: the result stays here until it is committed or rejected
        if (it->second.empty) break;
    }
    return 0;
}

int OverlapRun::commitGathered(std::vector<RoundResult>& results) {
    std::unique_lock<std::mutex> lk(pmu_);
    paf.clear();
    pafLines = 0;
    std::sort(results.begin(), results.end(), [](const RoundResult& a, const RoundResult& b) { return a.round < b.round; });
    int committed = 0;
    for (RoundResult& r : results) {
        if (r.round != round) break;
        const bool owned = r.round % world_ == rank_;
        if (r.empty) {
            if (emptyResultValid(r)) {
                done = true;
                break;
            }
            g_prof.rejected++;  // stale "end of input": every rank takes the same decision, the owner executes it again
            issueEnd_ = false;
            if (owned) {
                ready_.erase(r.round);
                redo_.push_back(r.round);
            }
            break;
        }
        g_prof.executed++;
        if (!resultValid(r)) {  // every rank takes the same decision; the owner executes the round again
            g_prof.rejected++;
            if (owned) {
                ready_.erase(r.round);
                redo_.push_back(r.round);
            }
            break;
        }
        if (owned) ready_.erase(r.round);
        commitOne(r);
        committed++;
        g_prof.committed++;
    }
    cvWork_.notify_all();
    return committed;
}

void OverlapRun::drain() {
    std::unique_lock<std::mutex> lk(pmu_);
    if (workers_.empty()) return;
    draining_ = true;
    cvDone_.wait(lk, [&] { return inflight_ == 0; });
    ready_.clear();
    redo_.clear();
    nextIssue_ = round;
    while (nextIssue_ % world_ != rank_) nextIssue_++;
    issueEnd_ = false;
    draining_ = false;
    cvWork_.notify_all();
}

// ---- scan-shard mode: plan + local scan, then (after the survivor exchange) the rest of the round
int OverlapRun::roundPrepareAndScan() {
    if (done) return 0;
    cur = RoundResult();
    cur.round = round;
    double t0 = now();
    curPlan = planner->get(round);
    if (curPlan && curPlan->failed) {
        error = "seed selection failed: " + curPlan->error;
        return -1;
    }
    if (!curPlan || curPlan->empty) {
        done = true;
        return 0;
    }
    cur.empty = false;
    cur.firstIn = curPlan->firstIn;
    cur.firstOut = curPlan->firstOut;
    ExecSlot& sl = *slots[0];
    int rc = beginRound(sl, *curPlan);
    if (rc) {
        error = sl.error;
        return rc;
    }
    sl.lap->setDeviceChunking(false);  // (the survivors of this scan are exchanged: their segments are wanted on the host)
    cur.st.n_seeds = curPlan->seedMap.size();
    double t1 = now();
    cur.st.t_prepare = t1 - t0;
    rc = sl.lap->ScanLocal(shardLo, shardHi, sl.local, cur.st);
    if (rc != 0) {
        error = sl.lap->err;
        return rc;
    }
    cur.st.t_scan = now() - t1;
    return 1;
}

// A failed rank stops taking part in the survivor exchanges: its communicators are aborted so that the other ranks' exchanges
// return an error (dp_comm_abort) - a per-rank failure ends the job on every rank instead of hanging it.
void OverlapRun::abortComms() {
    if (comm) dp_comm_abort(comm);
    for (dp_comm* c : slotComms)
        if (c) dp_comm_abort(c);
}

int OverlapRun::roundSharded() {
    if (!comm) {
        error = "roundSharded without a communicator";
        return -1;
    }
    int rc = roundPrepareAndScan();
    if (rc < 0) abortComms();
    if (rc <= 0) return rc;
    ExecSlot& sl = *slots[0];
    rc = sl.lap->ExchangeSurvivors(comm, gathered_);
    if (rc != 0) {
        error = sl.lap->err;
        abortComms();
        return rc;
    }
    rc = roundFinish(gathered_);
    if (rc < 0) abortComms();
    return rc < 0 ? rc : 1;
}

int OverlapRun::roundsShardedBatch() {
    if (done) return 0;
    {
        shardQueries = !dph_tune("no_shard_queries", 0);
    }
    if (slotComms.size() < slots.size()) {
        error = "roundsShardedBatch: fewer communicators than executor slots";
        return -1;
    }
    for (int attempt = 0; attempt < 4; attempt++) {
        std::vector<i64> rounds;
        for (size_t i = 0; i < slots.size(); i++) rounds.push_back(round + (i64)i);
        std::vector<RoundResult> outs;
        {
            std::lock_guard<std::mutex> lk(exchangeMu_);
            exchangeTurn_ = 0;
            resultTurn_ = 0;
            exchangeOrdered_ = true;
        }
        int rc = executeRounds(rounds, outs);
        exchangeOrdered_ = false;
        if (rc != 0) {
            abortComms();  // this rank leaves the job: the peers' next exchanges fail instead of waiting for it
            return rc < 0 ? rc : -1;
        }
        // the first round of a batch ran against the committed flags: it commits, or ends the command (an empty result that
        // came from a plan of a chain erased meanwhile is executed again - on every rank alike)
        const int c = commitResults(outs);
        if (c > 0 || done) return c;
    }
    error = "roundsShardedBatch: the first round of a batch keeps being rejected";
    return -1;
}

int OverlapRun::roundFinish(const Survivors& all) {
    int rc = finishRound(*slots[0], all, cur);
    if (rc) {
        error = slots[0]->error;
        return rc;
    }
    paf.clear();
    pafLines = 0;
    commitOne(cur);
    return 0;
}

}  // namespace dph
