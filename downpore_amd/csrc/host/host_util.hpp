// Internal helpers shared by the host translation units (not part of the mirrored reference interface).
#pragma once
// DP_DEBUG=a,b,c / DP_TUNE=key=value,...: see dp_common.h (the same two variables; the host library reads them for its own names)
#include <sys/resource.h>
#include <x86intrin.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "dph.hpp"

namespace dph {

static inline double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline double threadCpuNow() {  // CPU time consumed by the calling thread
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// -DDPH_FINE: cycle counters around the phases of the consensus stage (tools/host_consensus_profile.py); compiled out otherwise
#ifdef DPH_FINE
struct FineProfile {
    std::atomic<long long> cyc[16];
    const char* name[16] = {"unRC", "basesCovered+trim", "sharedByTwo", "reduced", "alignCore", "trimToBestSeed", "contig",
                            "pafEmit", "", "", "", "", "", "", "", ""};
    FineProfile() {
        for (auto& c : cyc) c = 0;
    }
    ~FineProfile() {
        long long tot = 0;
        for (int i = 0; i < 8; i++) tot += cyc[i].load();
        if (!tot) return;
        fprintf(stderr, "[fine]");
        for (int i = 0; i < 8; i++) fprintf(stderr, " %s %.1f%%", name[i], 100.0 * cyc[i].load() / tot);
        fprintf(stderr, " | total %.3f Gcycles\n", tot / 1e9);
        if (cyc[8].load())
            fprintf(stderr, "[fine] consensus steps %lld: uniform %.1f%%, out of step %.1f%%, in step but disagreeing %.1f%%, <2 live %.1f%%\n",
                    cyc[8].load(), 100.0 * cyc[9].load() / cyc[8].load(), 100.0 * cyc[10].load() / cyc[8].load(),
                    100.0 * cyc[11].load() / cyc[8].load(), 100.0 * cyc[12].load() / cyc[8].load());
    }
};
extern FineProfile g_fine;
struct FineScope {
    int i;
    unsigned long long t0;
    explicit FineScope(int i_) : i(i_), t0(__rdtsc()) {}
    ~FineScope() { g_fine.cyc[i] += (long long)(__rdtsc() - t0); }
};
#define FINE(i) FineScope fineScope##i(i)
#else
#define FINE(i)
#endif

// DPH_PROFILE=1: pipeline counters printed to stderr when a run shuts down
struct PipeProfile {
    std::atomic<long long> executed{0}, committed{0}, rejected{0}, discarded{0}, ignores{0}, planComputes{0}, planErased{0},
        planDiscarded{0}, hostGroups{0}, reselected{0};  // hostGroups: query windows the device consensus left to the host path
    std::atomic<long long> planTouchCyc{0}, planReselCyc{0}, planCommitCyc{0}, planOtherCyc{0};  // prepareFromCache, TSC cycles
    std::atomic<long long> cacheWaitUs{0};  // planner waiting for the window cache's producer
    std::atomic<long long> commitWaitUs{0}, commitTextUs{0}, commitStateUs{0}, commitKeepUs{0}, formatUs{0}, textWaitUs{0}, planLanes{0};  // the committing thread: waiting for the next round in order, text, state, handing the text on
    std::atomic<long long> planUs{0}, getWaitUs{0}, execUs{0}, commitUs{0}, consensusCpuUs{0}, selectCpuUs{0}, slotCpuUs{0}, plannerCpuUs{0};
    std::atomic<long long> sub[19];
    std::atomic<long long> subCpu[19];  // CPU time of the calling thread since its previous add(): the sections are consecutive
    const char* subName[19] = {"prep.indexReset", "prep.newOverlapper", "prep.roundBegin", "scan.call", "scan.copy", "idx.copy",
                               "idx.chunk", "idx.build", "idx.queries", "qry.call", "qry.matches", "fc.collate", "fc.parallel",
                               "fc.merge", "round.total", "round.tail", "plan.speculate", "plan.commitLoop", "prep.gangStart"};
    PipeProfile() {
        for (auto& x : sub) x = 0;
        for (auto& x : subCpu) x = 0;
    }
    void add(int i, double sec) {
        if (!on) return;  // (19 counters on two cache lines, bumped twenty times per round by every slot thread: only when somebody reads them)
        sub[i] += (long long)(sec * 1e6);
        if (on) {
            static thread_local double last = 0;
            const double c = threadCpuNow();
            if (last > 0) subCpu[i] += (long long)((c - last) * 1e6);
            last = c;
        }
    }
    bool on = getenv("DPH_PROFILE") != nullptr;
    void print() {
        if (!on) return;
        fprintf(stderr,
                "[pipe] rounds executed %lld committed %lld rejected %lld discarded %lld | new ignores %lld | plans computed %lld "
                "(%.2f ms each) erased %lld thrown away %lld | plan wait %.1f ms, execute %.1f ms, commit %.1f ms\n",
                executed.load(), committed.load(), rejected.load(), discarded.load(), ignores.load(), planComputes.load(),
                planComputes.load() ? planUs.load() / 1e3 / planComputes.load() : 0.0, planErased.load(), planDiscarded.load(),
                getWaitUs.load() / 1e3, execUs.load() / 1e3, commitUs.load() / 1e3);
        fprintf(stderr, "[pipe] query windows done by the host consensus path: %lld; planner waited %.1f ms for the window cache\n",
                hostGroups.load(), cacheWaitUs.load() / 1e3);
        fprintf(stderr, "[pipe] windows re-selected on the host (speculation did not hold): %lld\n", reselected.load());
        {
            const double tot = (double)(planTouchCyc.load() + planReselCyc.load() + planCommitCyc.load() + planOtherCyc.load());
            if (tot > 0)
                fprintf(stderr, "[pipe] plan from the window cache: touch test %.0f%%, re-selection %.0f%%, seed commits %.0f%%, rest %.0f%% (%.2f Gcycles)\n",
                        100 * planTouchCyc.load() / tot, 100 * planReselCyc.load() / tot, 100 * planCommitCyc.load() / tot,
                        100 * planOtherCyc.load() / tot, tot / 1e9);
        }
        const double n = (double)std::max<long long>(1, executed.load());
        struct rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        fprintf(stderr, "[pipe] host threads %u, process CPU time user %.2f s sys %.2f s, page faults minor %ld major %ld, context switches voluntary %ld involuntary %ld\n",
                hostThreads(), ru.ru_utime.tv_sec + ru.ru_utime.tv_usec / 1e6, ru.ru_stime.tv_sec + ru.ru_stime.tv_usec / 1e6, ru.ru_minflt,
                ru.ru_majflt, ru.ru_nvcsw, ru.ru_nivcsw);
        {
            const double nn = (double)std::max<long long>(1, executed.load());
            fprintf(stderr, "[pipe] thread CPU per round (ms): consensus items %.2f, seed-selection items %.2f, slot threads %.2f, planner thread %.2f\n",
                    consensusCpuUs.load() / 1e3 / nn, selectCpuUs.load() / 1e3 / nn, slotCpuUs.load() / 1e3 / nn,
                    plannerCpuUs.load() / 1e3 / nn);
        }
        fprintf(stderr, "[pipe] per executed round (ms):");
        for (int i = 0; i < 19; i++) fprintf(stderr, " %s %.3f", subName[i], sub[i].load() / 1e3 / n);
        fprintf(stderr, "\n[pipe] thread CPU up to the end of each section (ms):");
        for (int i = 0; i < 19; i++) fprintf(stderr, " %s %.3f", subName[i], subCpu[i].load() / 1e3 / n);
        fprintf(stderr, "\n");
    }
};
extern PipeProfile g_prof;  // defined in host_pool.cpp

}  // namespace dph

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
namespace dph {
inline std::map<std::string, std::string> envTokens(const char* name) {
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
struct EnvTokens {  // (parsed again whenever the variable's text has changed: tests set it between jobs of one process)
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
            m = envTokens(name);
            parsed = true;
        }
        return m;
    }
};
inline bool dph_debug(const char* what) {
    static EnvTokens t{"DP_DEBUG"};
    std::lock_guard<std::mutex> lk(t.mu);
    return t.get().count(what) != 0;
}
inline long dph_tune(const char* key, long dflt) {
    static EnvTokens t{"DP_TUNE"};
    std::lock_guard<std::mutex> lk(t.mu);
    const auto& m = t.get();
    const auto it = m.find(key);
    return it == m.end() ? dflt : atol(it->second.c_str());
}
}  // namespace dph
