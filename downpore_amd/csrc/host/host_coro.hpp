// Stackful coroutines of the mapper's read tasks (host_map.cpp): a switch between two stacks of ONE thread.
//
// On x86-64 the switch is sixteen instructions of our own (glibc's swapcontext saves and restores the signal mask with a system
// call per switch, 300 k of them per config-3 `map` run).  Everywhere else - another architecture, a build under AddressSanitizer or
// ThreadSanitizer (which have to be told about every stack switch and know how to follow swapcontext), a CET shadow-stack build
// (a `ret` into a hand-laid frame faults there), or -DDPH_CORO_UCONTEXT - the same interface runs on <ucontext.h>.  The coroutines
// do integer work and never change MXCSR or the x87 control word, which the hand-written switch therefore does not carry.
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define DPH_CORO_ASAN 1
#endif
#if __has_feature(thread_sanitizer)
#define DPH_CORO_TSAN 1
#endif
#endif
#if defined(__SANITIZE_ADDRESS__) && !defined(DPH_CORO_ASAN)
#define DPH_CORO_ASAN 1
#endif
#if defined(__SANITIZE_THREAD__) && !defined(DPH_CORO_TSAN)
#define DPH_CORO_TSAN 1
#endif

#if defined(__x86_64__) && !defined(DPH_CORO_ASAN) && !defined(DPH_CORO_TSAN) && !defined(__CET__) && !defined(DPH_CORO_UCONTEXT)
#define DPH_CORO_ASM 1
#else
#define DPH_CORO_ASM 0
#include <ucontext.h>
#endif

#ifdef DPH_CORO_ASAN
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#endif
#ifdef DPH_CORO_TSAN
extern "C" void* __tsan_get_current_fiber(void);
extern "C" void* __tsan_create_fiber(unsigned flags);
extern "C" void __tsan_destroy_fiber(void* fiber);
extern "C" void __tsan_switch_to_fiber(void* fiber, unsigned flags);
#endif

#if DPH_CORO_ASM
// Switches stacks: the callee-saved registers of the System V x86-64 ABI go to the current stack, its top to *save_sp, and the
// same registers come back from the stack `load_sp` points at.  A fresh coroutine's stack is laid out by coroStart() so that the
// first switch to it "returns" into coroBoot.
extern "C" void dph_coro_switch(void** save_sp, void* load_sp);
#ifdef DPH_CORO_IMPLEMENTATION
__asm__(
    ".text\n"
    ".globl dph_coro_switch\n"
    ".hidden dph_coro_switch\n"
    ".type dph_coro_switch,@function\n"
    "dph_coro_switch:\n"
    "    pushq %rbp\n"
    "    pushq %rbx\n"
    "    pushq %r12\n"
    "    pushq %r13\n"
    "    pushq %r14\n"
    "    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n"
    "    popq %r14\n"
    "    popq %r13\n"
    "    popq %r12\n"
    "    popq %rbx\n"
    "    popq %rbp\n"
    "    ret\n"
    ".size dph_coro_switch,.-dph_coro_switch\n");
#endif
#endif

namespace dph {

// A place execution can be switched away from and back to: a coroutine, or the thread's own stack while a coroutine runs.
struct CoroPoint {
#if DPH_CORO_ASM
    void* sp = nullptr;  // the saved stack pointer while switched out
#else
    ucontext_t uc;
#endif
    void (*entry)(void*) = nullptr;  // (coroutines only) first frame; must leave through coroSwitch(.., final = true), never return
    void* arg = nullptr;
#ifdef DPH_CORO_ASAN
    const void* stackBottom = nullptr;  // what AddressSanitizer is told when execution moves here
    size_t stackSize = 0;
    void* fakeStack = nullptr;
#endif
#ifdef DPH_CORO_TSAN
    void* fiber = nullptr;
    bool ownsFiber = false;
#endif
};

namespace coro_detail {
inline thread_local CoroPoint* g_to = nullptr;    // the point the switch in progress goes to
inline thread_local CoroPoint* g_from = nullptr;  // ... and the one it left
inline void arrived() {  // first thing on the destination's stack
#ifdef DPH_CORO_ASAN
    const void* bottom = nullptr;
    size_t size = 0;
    __sanitizer_finish_switch_fiber(g_to->fakeStack, &bottom, &size);
    if (g_from && !g_from->stackBottom) {  // (the thread's own stack: learnt when it is left for the first time)
        g_from->stackBottom = bottom;
        g_from->stackSize = size;
    }
#endif
}
inline void boot() {
    CoroPoint* me = g_to;
    arrived();
    me->entry(me->arg);
    __builtin_trap();  // (an entry that returned: there is nothing to return to)
}
}  // namespace coro_detail

// Lays out `pt` so that the first switch to it starts entry(arg) on [stack, stack + bytes).
inline void coroStart(CoroPoint& pt, char* stack, size_t bytes, void (*entry)(void*), void* arg) {
    pt.entry = entry;
    pt.arg = arg;
#ifdef DPH_CORO_ASAN
    pt.stackBottom = stack;
    pt.stackSize = bytes;
    pt.fakeStack = nullptr;
#endif
#ifdef DPH_CORO_TSAN
    pt.fiber = __tsan_create_fiber(0);
    pt.ownsFiber = true;
#endif
#if DPH_CORO_ASM
    // as dph_coro_switch leaves a stack: six zeroed callee-saved registers, then the address `ret` jumps to; at that `ret` the
    // stack pointer is 16-byte aligned + 8, as at any function's first instruction
    uintptr_t top = ((uintptr_t)stack + bytes) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;  // (the slot a return address of boot's caller would take: keeps the alignment)
    *--sp = (void*)&coro_detail::boot;
    for (int i = 0; i < 6; i++) *--sp = nullptr;
    pt.sp = (void*)sp;
#else
    getcontext(&pt.uc);
    pt.uc.uc_stack.ss_sp = stack;
    pt.uc.uc_stack.ss_size = bytes;
    pt.uc.uc_link = nullptr;
    makecontext(&pt.uc, (void (*)())&coro_detail::boot, 0);
#endif
}

// Leaves the running stack at `save` and continues at `load`.  final: the running coroutine will never be resumed (its stack may be
// reused as soon as the switch has happened).
inline void coroSwitch(CoroPoint& save, CoroPoint& load, bool final = false) {
    coro_detail::g_from = &save;
    coro_detail::g_to = &load;
#ifdef DPH_CORO_ASAN
    __sanitizer_start_switch_fiber(final ? nullptr : &save.fakeStack, load.stackBottom, load.stackSize);
#endif
#ifdef DPH_CORO_TSAN
    if (!save.fiber) save.fiber = __tsan_get_current_fiber();
    __tsan_switch_to_fiber(load.fiber, 0);
#endif
#if DPH_CORO_ASM
    dph_coro_switch(&save.sp, load.sp);
#else
    swapcontext(&save.uc, &load.uc);
#endif
    coro_detail::arrived();  // (somebody switched back to `save`)
}

// A finished coroutine's point before its stack is reused or freed.
inline void coroRelease(CoroPoint& pt) {
#ifdef DPH_CORO_TSAN
    if (pt.ownsFiber && pt.fiber) __tsan_destroy_fiber(pt.fiber);
    pt.fiber = nullptr;
    pt.ownsFiber = false;
#endif
    (void)pt;
}

}  // namespace dph
