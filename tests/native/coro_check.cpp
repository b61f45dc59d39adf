// Stand-alone check of downpore_amd/csrc/host/host_coro.hpp, built by tests/test_host_coroutines.py in every flavour the header has:
// the x86-64 switch, the <ucontext.h> fallback (-DDPH_CORO_UCONTEXT), and the fallback with AddressSanitizer's / ThreadSanitizer's
// fiber annotations.  Coroutines on recycled stacks keep a deep frame alive across switches; exit code 0 = every frame came back.
#define DPH_CORO_IMPLEMENTATION
#include "host_coro.hpp"

#include <cstdio>
#include <memory>
#include <vector>

using namespace dph;

struct T {
    CoroPoint at, *main = nullptr;
    int id = 0, yields = 0;
    long sum = 0;
    bool done = false, bad = false;
};

static void body(void* arg) {
    T* t = (T*)arg;
    volatile char pad[16384];
    for (int y = 0; y < t->yields; y++) {
        for (size_t i = 0; i < sizeof(pad); i += 32) pad[i] = (char)(t->id * 3 + y);
        coroSwitch(t->at, *t->main);
        for (size_t i = 0; i < sizeof(pad); i += 32)
            if (pad[i] != (char)(t->id * 3 + y)) t->bad = true;
        t->sum += t->id;
    }
    t->done = true;
    coroSwitch(t->at, *t->main, true);
}

int main() {
    const size_t stackBytes = (size_t)128 << 10;
    const int nTasks = 200, yields = 11, wave = 9;
    CoroPoint mainPt;
    std::vector<std::unique_ptr<char[]>> store;
    std::vector<char*> freeStacks;
    long sum = 0, want = 0;
    for (int base = 0; base < nTasks; base += wave) {
        std::vector<std::unique_ptr<T>> live;
        std::vector<char*> mine;
        for (int i = base; i < nTasks && i < base + wave; i++) {
            if (freeStacks.empty()) {
                store.emplace_back(new char[stackBytes]);
                freeStacks.push_back(store.back().get());
            }
            live.emplace_back(new T());
            T& t = *live.back();
            t.main = &mainPt;
            t.id = i + 1;
            t.yields = yields;
            want += (long)t.id * yields;
            mine.push_back(freeStacks.back());
            freeStacks.pop_back();
            coroStart(t.at, mine.back(), stackBytes, &body, &t);
            coroSwitch(mainPt, t.at);
        }
        for (bool any = true; any;) {
            any = false;
            for (auto& t : live)
                if (!t->done) {
                    coroSwitch(mainPt, t->at);
                    any = true;
                }
        }
        for (size_t i = 0; i < live.size(); i++) {
            if (live[i]->bad) {
                printf("task %d: frame damaged\n", live[i]->id);
                return 2;
            }
            sum += live[i]->sum;
            coroRelease(live[i]->at);
            freeStacks.push_back(mine[i]);
        }
    }
    printf("%s sum %ld want %ld\n", DPH_CORO_ASM ? "asm" : "ucontext", sum, want);
    return sum == want ? 0 : 1;
}
