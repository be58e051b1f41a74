// tests/hipemu/hipemu_rt.cpp -- the fiber scheduler behind tests/hipemu/include/hip/hip_runtime.h (TEST INFRASTRUCTURE).
//
// launch(): the workgroups of a grid are handed out in ascending blockIdx order to a few OS threads (so a workgroup that spins on a
// flag of a LOWER-numbered workgroup -- decoupled look-back -- always makes progress); a workgroup runs as blockDim fibers on its
// OS thread, switched by hand (x86-64: callee-saved registers + stack pointer) in round-robin order at every barrier / cross-lane
// operation, which gives the wave64 lock step the kernels rely on.
#include <hip/hip_runtime.h>

#include <execinfo.h>
#include <mutex>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>
#include <thread>
#include <vector>

#if !defined(__x86_64__)
#error "hipemu's fiber switch is written for x86-64"
#endif

extern "C" void hipemu_switch(void** save_sp, void* new_sp);
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
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

thread_local Block* t_block = nullptr;
thread_local Fiber* t_fiber = nullptr;

static size_t stack_bytes()
{
    static const size_t n = [] { const char* e = std::getenv("HIPEMU_STACK_KB"); return (size_t)(e ? std::atoi(e) : 256) * 1024; }();
    return n;
}

// stacks: one slab per OS thread in flight, recycled between launches
struct Slab { char* base; size_t bytes; };
static std::mutex g_slab_mutex;
static std::vector<Slab> g_slabs;
static Slab acquire_slab(size_t bytes)
{
    {
        std::lock_guard<std::mutex> lock(g_slab_mutex);
        for (size_t i = 0; i < g_slabs.size(); i++)
            if (g_slabs[i].bytes >= bytes) { Slab s = g_slabs[i]; g_slabs.erase(g_slabs.begin() + (long)i); return s; }
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { std::perror("hipemu: mmap of fiber stacks"); std::abort(); }
    return Slab{ (char*)p, bytes };
}
static void release_slab(Slab s)
{
    std::lock_guard<std::mutex> lock(g_slab_mutex);
    g_slabs.push_back(s);
}

[[noreturn]] static void deadlock(Block* b)
{
    std::fprintf(stderr, "hipemu: DEADLOCK in kernel %s, workgroup (%u,%u,%u): %u fibers alive, barrier has %u\n", b->kernel_name, b->bid.x, b->bid.y,
                 b->bid.z, b->alive, b->bar_arrived);
    unsigned shown = 0;
    for (unsigned i = 0; i < b->nthreads && shown < (std::getenv("HIPEMU_VERBOSE") ? 1024u : 24u); i++) {
        const Fiber& f = b->fibers[i];
        if (f.done) continue;
        if (!std::getenv("HIPEMU_VERBOSE") && i > 0 && !b->fibers[i - 1].done && b->fibers[i - 1].wait_what == f.wait_what && i + 1 < b->nthreads) continue;
        std::fprintf(stderr, "  thread %u (wave %u lane %u) waits at: %s\n", i, f.wave, f.lane, f.wait_what ? f.wait_what : "(running)");
        shown++;
    }
    std::fprintf(stderr, "  (lanes of one wave waiting at different cross-lane sites = control flow the hardware resolves with exec masks; not supported)\n");
    std::abort();
}

static inline void switch_to(Block* b, unsigned next)
{
    Fiber* me = t_fiber;
    Fiber* to = &b->fibers[next];
    b->cur = next;
    t_fiber = to;
    hipemu_switch(&me->sp, to->sp);
}

// HIPEMU_ORDER: which runnable fiber goes next when one blocks.  The hardware fixes no order between the waves of a workgroup, and
// within a wave it executes every lane of an instruction together, so between two synchronisation points no result may depend on
// the order the emulator happens to run lanes in.  "forward" (default) visits fibers in ascending order, "reverse" in descending
// order, "random:<seed>" picks pseudo-randomly: a parity test that passes under one order and fails under another has found either
// a missing barrier between waves or an unlisted lock-step point (build_emu.py: LOCKSTEP_POINTS) inside one.
static int order_mode(unsigned* seed)
{
    static int mode = -1; static unsigned s0 = 1;
    if (mode < 0) {
        const char* e = std::getenv("HIPEMU_ORDER");
        mode = 0;
        if (e && std::strcmp(e, "reverse") == 0) mode = 1;
        else if (e && std::strncmp(e, "random", 6) == 0) { mode = 2; s0 = e[6] == ':' ? (unsigned)std::atoi(e + 7) * 2654435761u + 1u : 12345u; }
    }
    *seed = s0;
    return mode;
}

void yield()
{
    Block* b = t_block;
    unsigned seed;
    const int mode = order_mode(&seed);
    // (a fiber that waits is re-run only to look at its gate again; the bound is on such looks without anybody arriving anywhere --
    // generous under the random order, where the one fiber that can make progress is drawn with probability 1 / #alive)
    if (++b->spin > (mode == 2 ? 256u : 8u) * b->nthreads + 64u) deadlock(b);
    unsigned n = b->cur;
    if (mode == 2) {
        // a few uniform draws for a runnable fiber, then the first runnable one in ascending order from the last draw
        for (int tries = 0; tries < 8; tries++) {
            b->rng = b->rng * 1664525u + 1013904223u + seed;
            n = (b->rng >> 8) % b->nthreads;
            if (!b->fibers[n].done && n != b->cur) break;
        }
        for (unsigned k = 0; k < b->nthreads; k++) {
            if (!b->fibers[n].done && n != b->cur) break;
            n = (n + 1 == b->nthreads) ? 0 : n + 1;
        }
        if (b->fibers[n].done) return;
    } else {
        for (unsigned k = 0; k < b->nthreads; k++) {
            if (mode == 1) n = (n == 0) ? b->nthreads - 1 : n - 1;
            else n = (n + 1 == b->nthreads) ? 0 : n + 1;
            if (!b->fibers[n].done) break;
        }
    }
    if (n == b->cur) return;
    switch_to(b, n);
}

static inline void release_block_barrier(Block* b)
{
    b->res_and[b->bar_gen & 1] = b->acc_and;
    b->res_or[b->bar_gen & 1] = b->acc_or;
    b->res_count[b->bar_gen & 1] = b->acc_count;
    b->acc_and = 1; b->acc_or = 0; b->acc_count = 0;
    b->bar_arrived = 0;
    b->bar_gen++;
    b->spin = 0;
    for (unsigned w = 0; w < b->nwaves; w++) b->waves[w].at_barrier = 0;       // (before any released lane runs on to a cross-lane operation)
}
static inline uint64_t group_mask(Scope scope, unsigned lane)
{
    return scope == SCOPE_QUAD ? (0xFull << (lane & ~3u)) : scope == SCOPE_ROW ? (0xFFFFull << (lane & ~15u)) : ~0ull;
}
static inline unsigned group_index(Scope scope, unsigned lane) { return scope == SCOPE_QUAD ? lane >> 2 : scope == SCOPE_ROW ? lane >> 4 : 0u; }

[[noreturn]] static void site_mismatch(Block* b, const char* a, const char* c)
{
    std::fprintf(stderr, "hipemu: kernel %s, workgroup (%u,%u,%u): lanes of one group met at DIFFERENT cross-lane operations:\n    %s\n    %s\n"
                 "  (divergent control flow around cross-lane operations: the hardware pairs them by exec mask, fibers cannot)\n",
                 b->kernel_name, b->bid.x, b->bid.y, b->bid.z, a, c);
    std::abort();
}
// all alive lanes of the group have deposited: check that they stand at the same site, open the gate
static inline void release_group(Wave& w, Scope scope, unsigned lane, Block* b)
{
    ScopeState& s = w.sc[scope];
    const uint64_t gm = group_mask(scope, lane);
    const unsigned parity = s.parity[group_index(scope, lane)];
    const uint64_t part = s.active[parity] & gm;
    if (part) {
        const char* first = s.site[__builtin_ctzll(part)];
        for (uint64_t m = part; m; m &= m - 1) if (s.site[__builtin_ctzll(m)] != first) site_mismatch(b, first, s.site[__builtin_ctzll(m)]);
    }
    const unsigned g = group_index(scope, lane);
    s.arrived[g] = 0;
    s.active[parity ^ 1] &= ~gm;          // the buffer the NEXT operation of this group deposits into (everybody has read it: they are all here)
    s.parity[g] = parity ^ 1;
    s.gen[g]++;
    b->spin = 0;
}

void block_barrier_count(int pred, int* r_count)
{
    Block* b = t_block;
    const unsigned gen = b->bar_gen;
    b->acc_count += (pred != 0);
    block_barrier(1, nullptr, nullptr);
    *r_count = b->res_count[gen & 1];
}

void block_barrier(int pred, int* r_and, int* r_or)
{
    Block* b = t_block;
    Fiber* f = t_fiber;
    const unsigned gen = b->bar_gen;
    b->acc_and &= (pred != 0);
    b->acc_or |= (pred != 0);
    b->spin = 0;
    if (++b->bar_arrived == b->alive) release_block_barrier(b);
    else {
        f->wait_what = "__syncthreads";
        Wave& w = b->waves[f->wave];
        w.at_barrier |= 1ull << f->lane;
        while (b->bar_gen == gen) yield();
        f->wait_what = nullptr;
    }
    if (r_and) *r_and = b->res_and[gen & 1];
    if (r_or) *r_or = b->res_or[gen & 1];
}

const uint64_t* wave_exchange(uint64_t mine, uint64_t* active, const char* what, Scope scope)
{
    Block* b = t_block;
    Fiber* f = t_fiber;
    Wave& w = b->waves[f->wave];
    ScopeState& s = w.sc[scope];
    const unsigned g = group_index(scope, f->lane);
    const unsigned p = s.parity[g];
    const uint64_t gm = group_mask(scope, f->lane);
    s.xchg[p][f->lane] = mine;
    s.site[f->lane] = what;
    s.active[p] |= 1ull << f->lane;
    const unsigned gen = s.gen[g];
    f->wait_what = what;
    b->spin = 0;
    if (++s.arrived[g] == (unsigned)__builtin_popcountll(w.alive_mask & gm)) release_group(w, scope, f->lane, b);
    else {
        while (s.gen[g] == gen) {
            // Exec-mask semantics: the rest of the group left this code path (e.g. `continue`d out of a loop that holds a ballot)
            // and waits at the workgroup barrier, which cannot open before the lanes standing here arrive there -- so they will
            // never take part, and the operation runs with the lanes present, as the hardware runs it with a partial exec mask.
            if (s.arrived[g] + (unsigned)__builtin_popcountll(w.at_barrier & gm) == (unsigned)__builtin_popcountll(w.alive_mask & gm)) {
                release_group(w, scope, f->lane, b);
                break;
            }
            yield();
        }
    }
    f->wait_what = nullptr;
    *active = s.active[p] & gm;
    return s.xchg[p];
}

static void fiber_main()
{
    Block* b = t_block;
    (*b->body)();
    // exit: leave the barriers consistent for the fibers that are still running
    Fiber* f = t_fiber;
    f->done = true;
    b->alive--;
    b->spin = 0;
    Wave& w = b->waves[f->wave];
    w.alive_mask &= ~(1ull << f->lane);
    for (int sc = 0; sc < 3; sc++) {
        const unsigned g = group_index((Scope)sc, f->lane);
        const unsigned left = (unsigned)__builtin_popcountll(w.alive_mask & group_mask((Scope)sc, f->lane));
        if (left > 0 && w.sc[sc].arrived[g] == left) release_group(w, (Scope)sc, f->lane, b);
    }
    if (b->alive > 0 && b->bar_arrived == b->alive) release_block_barrier(b);
    if (b->alive == 0) {
        void* dummy;
        hipemu_switch(&dummy, b->sched_sp);
    }
    unsigned n = b->cur;
    for (;;) { n = (n + 1 == b->nthreads) ? 0 : n + 1; if (!b->fibers[n].done) break; }
    b->cur = n;
    t_fiber = &b->fibers[n];
    void* dummy;
    hipemu_switch(&dummy, t_fiber->sp);
    std::abort();      // a finished fiber is never resumed
}

static void run_block(const char* name, dim3 grid, dim3 dim, unsigned linear, size_t shmem, const std::function<void()>& body)
{
    Block b{};
    b.grid = grid; b.dim = dim;
    b.bid = dim3(linear % grid.x, (linear / grid.x) % grid.y, linear / (grid.x * grid.y));
    b.nthreads = dim.x * dim.y * dim.z;
    b.nwaves = (b.nthreads + 63) / 64;
    b.alive = b.nthreads;
    b.acc_and = 1; b.acc_or = 0; b.acc_count = 0;
    b.body = &body;
    b.kernel_name = name;
    std::vector<Fiber> fibers(b.nthreads);
    std::vector<Wave> waves(b.nwaves);
    std::vector<uint64_t> dyn((shmem + 7) / 8 + 1);
    b.fibers = fibers.data(); b.waves = waves.data(); b.dyn_lds = dyn.data();
    const size_t sb = stack_bytes();
    Slab slab = acquire_slab(sb * b.nthreads);
    for (unsigned i = 0; i < b.nthreads; i++) {
        Fiber& f = fibers[i];
        f.flat = i;
        f.tid.x = i % dim.x; f.tid.y = (i / dim.x) % dim.y; f.tid.z = i / (dim.x * dim.y);
        f.wave = i >> 6; f.lane = i & 63; f.done = false; f.wait_what = nullptr;
        // initial frame: six zeroed callee-saved registers, then the "return address" fiber_main, placed so that the stack is
        // aligned as at a function entry (rsp % 16 == 8 after the ret)
        char* top = slab.base + sb * (i + 1);
        uint64_t* sp = reinterpret_cast<uint64_t*>(reinterpret_cast<uintptr_t>(top) & ~uintptr_t(15));
        *--sp = 0;                                   // one pad word: the return-address slot lands on a multiple of 16
        *--sp = reinterpret_cast<uint64_t>(&fiber_main);
        for (int r = 0; r < 6; r++) *--sp = 0;
        f.sp = sp;
    }
    for (unsigned wv = 0; wv < b.nwaves; wv++) {
        Wave& w = waves[wv];
        std::memset(&w, 0, sizeof w);
        const unsigned n = std::min(64u, b.nthreads - 64u * wv);
        w.alive_mask = n == 64 ? ~0ull : ((1ull << n) - 1ull);
    }
    Block* prev_block = t_block; Fiber* prev_fiber = t_fiber;
    t_block = &b;
    t_fiber = &fibers[0];
    b.cur = 0;
    hipemu_switch(&b.sched_sp, fibers[0].sp);
    t_block = prev_block; t_fiber = prev_fiber;
    release_slab(slab);
}

// a fault inside a kernel: say which kernel / workgroup / thread, whether it was the fiber's stack, and where
static void on_fault(int sig, siginfo_t* info, void*)
{
    Block* b = t_block;
    Fiber* f = t_fiber;
    char buf[512];
    int n = std::snprintf(buf, sizeof buf, "hipemu: signal %d at address %p in kernel %s, workgroup (%u,%u,%u), thread %u (sp slot %p)\n", sig, info->si_addr,
                          b ? b->kernel_name : "(none)", b ? b->bid.x : 0, b ? b->bid.y : 0, b ? b->bid.z : 0, f ? f->flat : 0, f ? f->sp : nullptr);
    (void)!write(2, buf, (size_t)n);
    void* bt[48];
    const int k = backtrace(bt, 48);
    backtrace_symbols_fd(bt, k, 2);
    _exit(139);
}
static void install_fault_handler()
{
    static std::once_flag once;
    std::call_once(once, [] {
        static char alt[1 << 16];
        stack_t ss{}; ss.ss_sp = alt; ss.ss_size = sizeof alt;
        sigaltstack(&ss, nullptr);
        struct sigaction sa{};
        sa.sa_sigaction = on_fault; sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigaction(SIGSEGV, &sa, nullptr); sigaction(SIGBUS, &sa, nullptr); sigaction(SIGFPE, &sa, nullptr);
    });
}

static unsigned worker_count()
{
    static const unsigned n = [] {
        const char* e = std::getenv("HIPEMU_THREADS");
        unsigned v = e ? (unsigned)std::atoi(e) : std::thread::hardware_concurrency();
        return v ? v : 1u;
    }();
    return n;
}

void launch(const char* name, dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body)
{
    if (std::getenv("HIPEMU_FAULT_HANDLER")) install_fault_handler();
    const unsigned nblocks = grid.x * grid.y * grid.z;
    if (nblocks == 0 || block.x * block.y * block.z == 0) return;
    if (std::getenv("HIPEMU_TRACE")) std::fprintf(stderr, "hipemu: %s <<<(%u,%u,%u), (%u,%u,%u), %zu>>>\n", name, grid.x, grid.y, grid.z, block.x, block.y, block.z, shmem);
    std::atomic<unsigned> next{ 0 };
    auto work = [&]() {
        for (;;) {
            const unsigned i = next.fetch_add(1);
            if (i >= nblocks) break;
            run_block(name, grid, block, i, shmem, body);
        }
    };
    const unsigned nw = std::min(worker_count(), nblocks);
    if (nw <= 1) { work(); return; }
    std::vector<std::thread> threads;
    for (unsigned t = 0; t < nw; t++) threads.emplace_back(work);
    for (auto& t : threads) t.join();
}

}   // namespace hipemu
