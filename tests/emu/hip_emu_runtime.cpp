// TEST INFRASTRUCTURE.  Fiber scheduler behind tests/emu/fakehip/hip/hip_runtime.h: one fiber (its own stack, a hand-written x86-64
// register switch: glibc's swapcontext makes a signal-mask system call per switch, which was most of the emulated run time) per GPU thread of the running
// workgroup; a fiber blocks at a workgroup barrier or at a wavefront rendezvous and the scheduler releases a group when all of its
// live members have arrived (convergent use of the collectives is assumed; anything else is reported as a deadlock).
#if !defined(__x86_64__)
#error "the fiber switch below is x86-64 System V only"
#endif
// void hipemu_switch(void** save_sp, void* load_sp): push the callee-saved registers, park the stack pointer, adopt the other one
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch,.-hipemu_switch
)");
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <map>
#include <string>
#include <chrono>
#include <thread>
#include <sched.h>

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
namespace hipemu {
struct ThreadCtx { dim3 tid, bid, bdim, gdim; };

namespace {
enum State { RUN, AT_BARRIER, AT_WAVE, DONE };
struct Fiber { void* sp; char* stack; State st; ThreadCtx tc; double xval; int xsrc; double xgot; double xval2; const void* xptr; int xbytes; };
// (all scheduler state is per OS thread: with HIPEMU_CONCURRENT=1 every workgroup of a launch runs on a thread of its own)
thread_local std::vector<char> gatherb_buf;   // [wave][64][64 bytes]: the operands of the wave's last byte gather
thread_local std::vector<double> gather_buf;   // [wave][64][2]: the operands of the wave's last gather, filled when the wave is released
const size_t STACK = 1 << 20;
thread_local std::vector<Fiber> fibers;
thread_local std::vector<char*> stacks;
thread_local void* sched_sp = nullptr;
thread_local int cur_fiber = -1;
thread_local const std::function<void()>* cur_body = nullptr;

void trampoline() {
    (*cur_body)();
    fibers[cur_fiber].st = DONE;
    hipemu_switch(&fibers[cur_fiber].sp, sched_sp);
    __builtin_trap();   // (a finished fiber is never resumed)
}
void yield_as(State s) {
    Fiber& f = fibers[cur_fiber];
    f.st = s;
    hipemu_switch(&f.sp, sched_sp);
}
}  // namespace

struct LdsArray { double* base; size_t bytes; };
std::vector<LdsArray>& lds_arrays() { static std::vector<LdsArray> v; return v; }
void register_dynamic_lds(double* base, size_t bytes) {
    for (auto& a : lds_arrays()) if (a.base == base) return;
    lds_arrays().push_back(LdsArray{base, bytes});
}
static const double CANARY = -7.25e77;

thread_local ThreadCtx* cur_ptr = nullptr;
static inline ThreadCtx& cur() { return *cur_ptr; }
void barrier() { yield_as(AT_BARRIER); }
void wave_sync() { fibers[cur_fiber].xsrc = -1; yield_as(AT_WAVE); }
double wave_exchange(double v, int src_lane) {
    Fiber& f = fibers[cur_fiber];
    f.xval = v; f.xsrc = src_lane;
    yield_as(AT_WAVE);
    return fibers[cur_fiber].xgot;
}

void wave_gather2(double a, double b, const double** all) {
    Fiber& f = fibers[cur_fiber];
    f.xval = a; f.xval2 = b; f.xsrc = -2;
    yield_as(AT_WAVE);
    *all = &gather_buf[(size_t)(cur_fiber / 64) * 128];
}

const char* wave_allgather(const void* mine, int nbytes) {
    Fiber& f = fibers[cur_fiber];
    f.xptr = mine; f.xbytes = nbytes; f.xsrc = -3;
    yield_as(AT_WAVE);
    return &gatherb_buf[(size_t)(cur_fiber / 64) * 4096];
}

static void segv_backtrace(int) {   // a kernel bug on the CPU: say where (symbols: build with -g, resolve with addr2line)
    void* frames[48];
    const int n = backtrace(frames, 48);
    const char msg[] = "hipemu: SIGSEGV inside an emulated kernel; backtrace:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    _exit(139);
}

const char* launch_name = "?";
namespace {
struct ProfRow { double s; long n; long fibers; };
std::map<std::string, ProfRow>& prof() { static std::map<std::string, ProfRow> m; return m; }
void prof_dump() { for (auto& kv : prof()) fprintf(stderr, "hipemu prof: %-40s %9.3f s  %8ld launches %12ld fibers\n", kv.first.c_str(), kv.second.s, kv.second.n, kv.second.fibers); }
}
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body) {
    static const bool profiling = getenv("HIPEMU_PROF") != nullptr;
    static bool registered = false;
    if (profiling && !registered) { atexit(prof_dump); registered = true; }
    const auto t_launch0 = std::chrono::steady_clock::now();
    const std::string name_now = launch_name;
    const int nt = (int)(block.x * block.y * block.z);
    static bool handler_installed = false;
    if (!handler_installed) {
        static char altstack[1 << 16];
        stack_t ss; ss.ss_sp = altstack; ss.ss_size = sizeof(altstack); ss.ss_flags = 0;
        sigaltstack(&ss, nullptr);
        struct sigaction sa; memset(&sa, 0, sizeof(sa)); sa.sa_handler = segv_backtrace; sa.sa_flags = SA_ONSTACK;
        sigaction(SIGSEGV, &sa, nullptr);
        handler_installed = true;
    }
    if (const char* e = getenv("HIPEMU_LDS_SHRINK")) { size_t cut = (size_t)atol(e); lds_bytes = lds_bytes > cut ? lds_bytes - cut : 0; }   // (self-test of the guard)
    // everything beyond the dynamic LDS this launch asked for is a canary: a kernel writing past its allocation is caught below
    for (auto& a : lds_arrays()) for (size_t i = (lds_bytes + 7) / 8; i < a.bytes / 8; ++i) a.base[i] = CANARY;
    // HIPEMU_CONCURRENT=1: the workgroups of a launch run side by side, one OS thread each (their LDS is thread_local) -- what kernels
    // whose workgroups wait for each other need (the cooperative chains); the default runs them one after another on this thread
    const bool concurrent = getenv("HIPEMU_CONCURRENT") != nullptr && atoi(getenv("HIPEMU_CONCURRENT")) != 0 && grid.x * grid.y * grid.z > 1;
    auto run_block = [&](unsigned bx, unsigned by, unsigned bz, bool check_lds) {
        gather_buf.assign((size_t)((nt + 63) / 64) * 128, 0.0);
        gatherb_buf.assign((size_t)((nt + 63) / 64) * 4096, 0);
        while ((int)stacks.size() < nt) stacks.push_back((char*)malloc(STACK));
        fibers.assign(nt, Fiber());
        cur_body = &body;
        for (int t = 0; t < nt; ++t) {
            Fiber& f = fibers[t];
            f.stack = stacks[t]; f.st = RUN;
            f.tc.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.tc.bid = dim3(bx, by, bz); f.tc.bdim = block; f.tc.gdim = grid;
            // a fresh stack as hipemu_switch expects to find it: [mxcsr / x87 control word][r15 r14 r13 r12 rbx rbp][return address =
            // trampoline][a null return address for trampoline itself: it never returns], entered with rsp = 8 mod 16 like any call
            uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
            void** sp = (void**)(top - 8);
            *sp = nullptr;
            *--sp = (void*)&trampoline;
            for (int k = 0; k < 6; ++k) *--sp = nullptr;
            --sp;
            { unsigned int mx; unsigned short cw; asm volatile("stmxcsr %0" : "=m"(mx)); asm volatile("fnstcw %0" : "=m"(cw)); ((unsigned int*)sp)[0] = mx; ((unsigned short*)sp)[2] = cw; }
            f.sp = (void*)sp;
        }
        while (true) {
            bool progressed = false;
            for (int t = 0; t < nt; ++t) if (fibers[t].st == RUN) {
                cur_fiber = t; cur_ptr = &fibers[t].tc;
                hipemu_switch(&sched_sp, fibers[t].sp);
                progressed = true;
            }
            // release wavefronts whose live lanes have all arrived
            const int nw = (nt + 63) / 64;
            for (int w = 0; w < nw; ++w) {
                int lo = w * 64, hi = std::min(nt, lo + 64), waiting = 0, live = 0;
                for (int t = lo; t < hi; ++t) { if (fibers[t].st != DONE) ++live; if (fibers[t].st == AT_WAVE) ++waiting; }
                if (live && waiting == live) {
                    for (int t = lo; t < hi; ++t) if (fibers[t].st == AT_WAVE) {
                        int s = fibers[t].xsrc;
                        fibers[t].xgot = (s >= 0 && lo + s < hi) ? fibers[lo + s].xval : 0.0;
                        if (s == -3) memcpy(&gatherb_buf[(size_t)w * 4096 + (size_t)(t - lo) * 64], fibers[t].xptr, (size_t)fibers[t].xbytes);
                        if (s == -2) { gather_buf[(size_t)w * 128 + 2 * (t - lo)] = fibers[t].xval; gather_buf[(size_t)w * 128 + 2 * (t - lo) + 1] = fibers[t].xval2; }
                    }
                    for (int t = lo; t < hi; ++t) if (fibers[t].st == AT_WAVE) fibers[t].st = RUN;
                    progressed = true;
                }
            }
            int live = 0, atb = 0;
            for (int t = 0; t < nt; ++t) { if (fibers[t].st != DONE) ++live; if (fibers[t].st == AT_BARRIER) ++atb; }
            if (live == 0) {
                if (check_lds)   // (the registry holds the launching thread's LDS arrays)
                    for (auto& a : lds_arrays()) for (size_t i = (lds_bytes + 7) / 8; i < a.bytes / 8; ++i)
                        if (a.base[i] != CANARY) { fprintf(stderr, "hipemu: write past the %zu bytes of dynamic LDS (double index %zu) in block %u\n", lds_bytes, i, bx); abort(); }
                break;
            }
            if (atb == live) { for (int t = 0; t < nt; ++t) if (fibers[t].st == AT_BARRIER) fibers[t].st = RUN; progressed = true; }
            if (!progressed) { fprintf(stderr, "hipemu: deadlock (divergent barrier / collective) in block %u\n", bx); abort(); }
        }
        cur_fiber = -1;
    };
    if (concurrent) {
        std::vector<std::thread> ths;
        for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx)
            ths.emplace_back([&run_block, bx, by, bz]() {
                run_block(bx, by, bz, false);
                for (char* st : stacks) free(st);
                stacks.clear();
            });
        for (auto& t : ths) t.join();
    } else {
        for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) run_block(bx, by, bz, true);
    }
    cur_fiber = -1;
    if (profiling) { ProfRow& r = prof()[name_now]; r.s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_launch0).count(); r.n += 1; r.fibers += (long)grid.x * grid.y * grid.z * nt; }
}
}  // namespace hipemu
void hipemu_yield() { sched_yield(); }
