// Process-wide worker pool (sized from the cgroup CPU quota) and the DPH_PROFILE counters.
#include <dlfcn.h>
#include <execinfo.h>
#include <malloc.h>
#include <signal.h>
#include <pthread.h>
#include <sched.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <map>
#include <string>
#include <vector>
#include <cstring>
#include <mutex>
#include <thread>

#include "host_util.hpp"

namespace dph {

// Eight executor-slot threads allocate and release hundreds of kilobytes to megabytes per round (result text, segment copies,
// per-round objects).  With glibc's defaults every other free hands memory back to the kernel (heap trim, munmap of large
// chunks) and the next round faults it in again; those calls take the process-wide mmap lock for writing and stall every
// thread's page faults - measured as 12 % of the whole job (two processes with four slots each on one GPU ran 19 % faster
// than one process with eight).  Keep freed memory in the allocator instead.  DPH_MALLOC_DEFAULTS=1 leaves glibc alone.
static const bool g_malloc_tuned = [] {
    mallopt(M_MMAP_THRESHOLD, 32 << 20);      // (the largest value glibc accepts; fixes the threshold)
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    return true;
}();


// DPH_SEGV_TRACE=1: a crashing host thread prints its frames (module + offset: addr2line on the in-tree build) before dying
static void segvTrace(int sig) {
    void* frames[48];
    const int n = backtrace(frames, 48);
    const char msg[] = "[dph] fatal signal, frames of the crashing thread:\n";
    if (write(2, msg, sizeof msg - 1) < 0) {}
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
static const bool g_segv_trace = [] {
    if (dph::dph_debug("segv")) {
        signal(SIGSEGV, segvTrace);
        signal(SIGBUS, segvTrace);
        signal(SIGABRT, segvTrace);
        return true;
    }
    return false;
}();

// DPH_SAMPLE_PROF=1: wall-clock sampling of the pipeline's threads (executor slots, planner lanes, window cache, formatters):
// a sampler thread signals each registered thread every 0.2 ms, the thread records its frames.  A sample is attributed to
// the innermost frame inside our own libraries - the call site of the HIP API call (or host code) the thread is in, as
// module+offset (tools/sample_resolve.py turns them into file:line with addr2line on the in-tree build; the runtime
// libraries carry no symbols of their own) - and to the module of the innermost frame overall (runtime, libc, driver).
namespace {
struct ProfSample {
    int n, role;
    void* f[20];
};
constexpr int kProfMax = 1 << 20;
ProfSample* g_profSamples = nullptr;
std::atomic<int> g_profCount{0};
std::atomic<bool> g_profRun{false};
std::mutex g_profMu;
std::vector<std::pair<pthread_t, bool>> g_profThreads;  // (thread, alive)
std::vector<std::string> g_profRoles;
thread_local int t_profRole = -1;
std::thread g_profSampler;
void profTick(int) {
    const int i = g_profCount.fetch_add(1, std::memory_order_relaxed);
    if (i < kProfMax) {
        g_profSamples[i].role = t_profRole;
        g_profSamples[i].n = backtrace(g_profSamples[i].f, 20);
    }
}
std::string moduleOf(const Dl_info& di) {
    std::string m = di.dli_fname ? di.dli_fname : "?";
    const size_t sl = m.rfind('/');
    return sl == std::string::npos ? m : m.substr(sl + 1);
}
void profReport() {
    g_profRun = false;
    if (g_profSampler.joinable()) g_profSampler.join();
    const int n = std::min(g_profCount.load(), kProfMax);
    std::map<std::string, std::map<std::string, int>> sites;  // role -> "module+off | leaf module" -> samples
    std::map<std::string, int> perRole;
    for (int i = 0; i < n; i++) {
        const ProfSample& sm = g_profSamples[i];
        std::string role = sm.role >= 0 && (size_t)sm.role < g_profRoles.size() ? g_profRoles[(size_t)sm.role] : "?";
        std::string leafMod = "?", site;
        for (int j = 2; j < sm.n; j++) {  // 0 = this handler, 1 = the signal trampoline
            Dl_info di;
            if (!dladdr(sm.f[j], &di)) continue;
            const std::string mod = moduleOf(di);
            if (j == 2) leafMod = mod;
            if (mod.find("libdownpore_") != std::string::npos) {
                char b[64];
                snprintf(b, sizeof b, "+0x%zx", (size_t)((char*)sm.f[j] - (char*)di.dli_fbase));
                site = mod + b + (j == 2 ? " own" : "");
                break;
            }
        }
        if (site.empty()) site = "(no frame of ours)";
        sites[role][site + " | " + leafMod]++;
        perRole[role]++;
    }
    fprintf(stderr, "[sample] %d samples at 0.2 ms\n", n);
    for (auto& rs : sites) {
        std::vector<std::pair<int, std::string>> order;
        for (auto& kv : rs.second) order.push_back({kv.second, kv.first});
        std::sort(order.rbegin(), order.rend());
        const int tot = perRole[rs.first];
        for (size_t i = 0; i < order.size() && i < 45; i++)
            fprintf(stderr, "[sample] %-10s %6d %5.1f%%  %s\n", rs.first.c_str(), order[i].first, 100.0 * order[i].first / std::max(1, tot), order[i].second.c_str());
    }
}
const bool g_profOn = [] {
    if (!dph::dph_debug("sample_prof")) return false;
    g_profSamples = new ProfSample[kProfMax];
    void* warm[4];
    backtrace(warm, 4);  // (loads the unwinder now, not inside the first signal)
    struct sigaction sa = {};
    sa.sa_handler = profTick;
    sa.sa_flags = SA_RESTART;
    sigaction(SIGPROF, &sa, nullptr);
    atexit(profReport);
    return true;
}();
}  // namespace
// a pipeline thread announces itself (role: "slot", "lane", "cache", "text", "commit"); unregistered when it ends
void sampleProfRegister(const char* role) {
    if (!g_profOn) return;
    std::lock_guard<std::mutex> lk(g_profMu);
    size_t r = 0;
    for (; r < g_profRoles.size(); r++)
        if (g_profRoles[r] == role) break;
    if (r == g_profRoles.size()) g_profRoles.push_back(role);
    t_profRole = (int)r;
    g_profThreads.push_back({pthread_self(), true});
}
void sampleProfUnregister() {
    if (!g_profOn) return;
    std::lock_guard<std::mutex> lk(g_profMu);
    for (auto& t : g_profThreads)
        if (t.second && pthread_equal(t.first, pthread_self())) t.second = false;
}
// sampling runs between these two (the rounds of a job; set-up and everything else of the process stay out of the picture)
void sampleProfStart() {
    if (!g_profOn || g_profRun.exchange(true)) return;
    if (g_profSampler.joinable()) g_profSampler.join();
    g_profSampler = std::thread([] {
        while (g_profRun.load()) {
            {
                std::lock_guard<std::mutex> lk(g_profMu);
                for (auto& t : g_profThreads)
                    if (t.second) pthread_kill(t.first, SIGPROF);
            }
            usleep(200);
        }
    });
}
void sampleProfStop() {
    if (!g_profOn) return;
    g_profRun = false;
}

PipeProfile g_prof;
#ifdef DPH_FINE
FineProfile g_fine;
#endif

void profilePrint() { g_prof.print(); }
void setHostThreadShare(unsigned) {}  // kept for callers; the shared pool needs no per-slot split
// CPUs this process may actually use: the cgroup CPU quota (containers often expose every host CPU but cap the CPU
// time; exceeding the cap gets the whole process throttled for the rest of the scheduler period) or the CPU count.
static unsigned cpuBudget() {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    double quota = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char a[64] = {0};
        long long period = 0;
        if (fscanf(f, "%63s %lld", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) quota = atof(a) / (double)period;
        fclose(f);
    } else {
        long long q = -1, per = 0;
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(g, "%lld", &q) != 1) q = -1;
            fclose(g);
        }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(g, "%lld", &per) != 1) per = 0;
            fclose(g);
        }
        if (q > 0 && per > 0) quota = (double)q / (double)per;
    }
    if (quota >= 1.0 && quota < (double)hw) {
        // Under a quota the pool must not be able to use all of it: the executor-slot threads, the planner and the HIP
        // runtime's own threads run next to it, and a process that overdraws its quota is stopped whole for the rest of
        // the 100 ms scheduler period (seen as 15-30 ms stalls of every slot a few times per hundred rounds, GPU idle).
        hw = (unsigned)quota;
        const unsigned reserve = std::max(2u, hw / 5);
        hw = hw > reserve + 1 ? hw - reserve : 1;
    }
    return hw;
}

unsigned hostThreads() {
    static unsigned n = [] {
        const char* e = getenv("DP_HOST_THREADS");
        unsigned v = e ? (unsigned)atoi(e) : cpuBudget();
        if (v == 0) v = 1;
        return std::min(v, 96u);
    }();
    return n;
}

namespace {
struct PoolJob {
    size_t n = 0;
    const std::function<void(size_t)>* fn = nullptr;
    std::atomic<size_t> next{0}, done{0};
    std::mutex mu;
    std::condition_variable cv;
};
class WorkPool {
   public:
    static WorkPool& get() {
        static WorkPool* p = new WorkPool();  // intentionally leaked: workers may outlive static destruction order
        return *p;
    }
    void run(size_t n, const std::function<void(size_t)>& fn) {
        if (n == 0) return;
        if (n == 1 || threads_.empty()) {
            for (size_t i = 0; i < n; i++) fn(i);
            return;
        }
        auto job = std::make_shared<PoolJob>();
        job->n = n;
        job->fn = &fn;
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.push_back(job);
        }
        cv_.notify_all();
        work(*job);
        if (job->done.load(std::memory_order_acquire) < n) {  // items still running on pool threads
            std::unique_lock<std::mutex> jl(job->mu);
            job->cv.wait(jl, [&] { return job->done.load(std::memory_order_acquire) >= n; });
        }
        std::lock_guard<std::mutex> lk(mu_);
        for (auto it = jobs_.begin(); it != jobs_.end(); ++it)
            if (it->get() == job.get()) {
                jobs_.erase(it);
                break;
            }
    }

   private:
    WorkPool() {
        const unsigned n = hostThreads();
        // Workers sleep between jobs and are woken together by the submitting thread; the scheduler tends to leave such
        // short bursts stacked on the waker's CPU.  Each worker is therefore pinned to its own CPU of the allowed set
        // (spread evenly) when DP_PIN_WORKERS=1.
        std::vector<int> cpus;
        const long pinMode = dph::dph_tune("pin_workers", 0);
        const char pinBuf[2] = {(char)('0' + (pinMode & 7)), 0};
        const char* pin = pinBuf;
        if (pin[0] != '0') {  // 1: one worker per physical core, spread over all cores; 2: the same within NUMA node 0
            cpu_set_t set;
            CPU_ZERO(&set);
            if (sched_getaffinity(0, sizeof set, &set) == 0)
                for (int c = 0; c < CPU_SETSIZE; c++) {
                    if (!CPU_ISSET(c, &set)) continue;
                    char path[128];
                    int first = c, node0 = 1;
                    snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
                    if (FILE* f = fopen(path, "r")) {
                        if (fscanf(f, "%d", &first) != 1) first = c;
                        fclose(f);
                    }
                    if (first != c) continue;  // SMT sibling of a lower-numbered CPU
                    if (pin[0] == '2') {
                        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/node0", c);
                        node0 = access(path, F_OK) == 0;
                    }
                    if (node0) cpus.push_back(c);
                }
        }
        for (unsigned i = 1; i < n; i++) {
            threads_.emplace_back([this] { loop(); });
            if (cpus.size() >= n) {
                cpu_set_t one;
                CPU_ZERO(&one);
                CPU_SET(cpus[(size_t)i * cpus.size() / n], &one);
                pthread_setaffinity_np(threads_.back().native_handle(), sizeof one, &one);
            }
        }
        for (auto& t : threads_) t.detach();
    }
    static void work(PoolJob& j) {
        for (;;) {
            const size_t i = j.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j.n) return;
            (*j.fn)(i);
            if (j.done.fetch_add(1, std::memory_order_acq_rel) + 1 == j.n) {
                std::lock_guard<std::mutex> jl(j.mu);
                j.cv.notify_all();
            }
        }
    }
    void loop() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            std::shared_ptr<PoolJob> job;
            for (auto& j : jobs_)
                if (j->next.load(std::memory_order_relaxed) < j->n) {
                    job = j;
                    break;
                }
            if (!job) {
                cv_.wait(lk);
                continue;
            }
            lk.unlock();
            work(*job);
            lk.lock();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::shared_ptr<PoolJob>> jobs_;
    std::vector<std::thread> threads_;
};
}  // namespace

void parallelFor(size_t n, const std::function<void(size_t)>& fn) { WorkPool::get().run(n, fn); }

}  // namespace dph
