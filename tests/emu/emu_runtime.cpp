// tests/emu/emu_runtime.cpp -- TEST HARNESS ONLY (see tests/emu/hip/hip_runtime.h).
// Fiber scheduler that executes one HIP block at a time (plain launches) or a whole cooperative grid at once
// on the calling OS thread, plus host stand-ins for the handful of hip* runtime calls the engine makes.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" void emu_ctx_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
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
.size emu_ctx_switch, .-emu_ctx_switch
)");

namespace emu {

Fiber *cur = nullptr;
uint3_ g_blockDim, g_gridDim;

static const size_t kStack = 128 * 1024;      // plain launches: one block at a time
static const size_t kCoopStack = 64 * 1024;   // cooperative launches: every block of the grid is resident
static const unsigned kMaxThreads = 1024;
static char *g_stacks = nullptr;
static size_t g_stacks_bytes = 0;
static void *g_sched_sp = nullptr;
static const std::function<void()> *g_body = nullptr;

struct WaveSync {
    // exchange slots per log2 width: a lane that has left a narrow exchange may enter a wider one (a different call site)
    // before its partner has read the narrow one's value
    unsigned long long slot[7][2][64];
    // one (arrived, generation) pair per (log2 width, segment)
    unsigned arrived[7][64];
    unsigned gen[7][64];
};

// per-block state: barrier bookkeeping, wave exchange slots, dynamic LDS, the block's fibers
struct Block {
    uint3_ idx;
    std::vector<Fiber> fibers;
    std::vector<WaveSync> waves;
    std::vector<char> dyn;
    unsigned alive = 0, bar_arrived = 0, bar_gen = 0;
};
static std::vector<Block> g_blocks;

static inline Block &blk() { return *static_cast<Block *>(cur->blk); }
uint3_ block_idx() { return blk().idx; }
void *dyn_shared() { return blk().dyn.data(); }

static void yield_to_scheduler() { emu_ctx_switch(&cur->sp, g_sched_sp); }
void yield() { yield_to_scheduler(); }

static bool g_resident_next = false;
void next_launch_resident() { g_resident_next = true; }
bool take_resident_flag() {
    const bool f = g_resident_next;
    g_resident_next = false;
    return f;
}

static void fiber_entry() {
    (*g_body)();
    Block &b = blk();
    cur->done = true;
    --b.alive;
    // a finished thread no longer participates in barriers (hardware: terminated waves)
    if (b.alive && b.bar_arrived == b.alive) {
        b.bar_arrived = 0;
        ++b.bar_gen;
    }
    yield_to_scheduler();
    std::abort();  // never resumed
}

void syncthreads() {
    Block &b = blk();
    unsigned my = b.bar_gen;
    if (++b.bar_arrived == b.alive) {
        b.bar_arrived = 0;
        ++b.bar_gen;
        return;
    }
    while (b.bar_gen == my) yield_to_scheduler();
}

static int ilog2(int w) {
    int l = 0;
    while ((1 << l) < w) ++l;
    return l;
}

unsigned long long shfl_exchange(unsigned long long v, int src, int width) {
    if (width <= 1) return v;
    const unsigned lane = cur->linear & 63u;
    WaveSync &w = blk().waves[cur->linear >> 6];
    const int lw = ilog2(width), seg = (int)lane / width;
    const unsigned my = w.gen[lw][seg];
    w.slot[lw][my & 1][lane] = v;
    if (++w.arrived[lw][seg] == (unsigned)width) {
        w.arrived[lw][seg] = 0;
        ++w.gen[lw][seg];
    } else {
        while (w.gen[lw][seg] == my) yield_to_scheduler();
    }
    return w.slot[lw][my & 1][src];
}

unsigned long long ballot_exchange(bool pred) {
    const unsigned lane = cur->linear & 63u;
    WaveSync &w = blk().waves[cur->linear >> 6];
    const int lw = 6, seg = 0;
    const unsigned my = w.gen[lw][seg];
    w.slot[lw][my & 1][lane] = pred ? 1ull : 0ull;
    if (++w.arrived[lw][seg] == 64u) {
        w.arrived[lw][seg] = 0;
        ++w.gen[lw][seg];
    } else {
        while (w.gen[lw][seg] == my) yield_to_scheduler();
    }
    unsigned long long m = 0;
    for (unsigned l = 0; l < 64u; ++l) m |= (w.slot[lw][my & 1][l] & 1ull) << l;
    return m;
}

// the `static` stand-ins of every __shared__ variable of the library (hip_runtime.h puts them into the section "emu_shared")
extern "C" char __start_emu_shared[] __attribute__((weak));
extern "C" char __stop_emu_shared[] __attribute__((weak));

// a workgroup finds garbage in its LDS, not zeros and not its own leftovers in a known state: poison before every block of a plain
// launch (a cooperative grid's blocks share the statics and may not use them: launch_coop poisons once)
static void poison_static_shared() {
    if (__start_emu_shared && __stop_emu_shared > __start_emu_shared)
        std::memset(__start_emu_shared, 0xCD, (size_t)(__stop_emu_shared - __start_emu_shared));
}

static void init_block(Block &b, uint3_ idx, dim3 block, unsigned nthreads, size_t shmem, char *stacks, size_t stack_bytes) {
    b.idx = idx;
    b.fibers.resize(nthreads);
    b.waves.assign((nthreads + 63) / 64, WaveSync());
    std::memset(b.waves.data(), 0, b.waves.size() * sizeof(WaveSync));
    b.dyn.assign(shmem + 64, (char)0xCD);  // (dynamic LDS: poisoned, like hipMalloc's memory)
    b.alive = nthreads;
    b.bar_arrived = 0;
    b.bar_gen = 0;
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber &f = b.fibers[t];
        f.linear = t;
        f.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
        f.done = false;
        f.blk = &b;
        uintptr_t top = reinterpret_cast<uintptr_t>(stacks + stack_bytes * (t + 1));
        top &= ~uintptr_t(15);
        void **sp = reinterpret_cast<void **>(top);
        *--sp = nullptr;                                  // fake return address
        *--sp = reinterpret_cast<void *>(&fiber_entry);   // popped by ret
        for (int r = 0; r < 6; ++r) *--sp = nullptr;      // rbp rbx r12-r15
        f.sp = sp;
    }
}

static void ensure_stacks(size_t bytes) {
    if (bytes <= g_stacks_bytes) return;
    std::free(g_stacks);
    g_stacks = static_cast<char *>(std::malloc(bytes));
    g_stacks_bytes = bytes;
}

static unsigned check_block(dim3 block) {
    const unsigned nthreads = block.x * block.y * block.z;
    if (nthreads == 0 || nthreads > kMaxThreads || (nthreads & 63u)) {
        std::fprintf(stderr, "emu: unsupported block size %u\n", nthreads);
        std::abort();
    }
    return nthreads;
}

// SLK_EMU_ORDER: the order in which the scheduler visits the fibers between two synchronisation points.  Any order is a legal
// execution of a correctly synchronised kernel, so every test must pass under every setting -- a missing __syncthreads() /
// SLK_WAVE_SYNC whose reader happens to run after its writer in ascending thread order shows up under "reverse":
//   (unset) / forward   threads 0, 1, 2, ... of block 0, then block 1, ...
//   reverse             the last thread of the last resident block first
//   alternate           forward and reverse on alternating sweeps
static int fiber_order() {
    static int order = -1;
    if (order < 0) {
        const char *e = std::getenv("SLK_EMU_ORDER");
        order = !e ? 0 : !std::strcmp(e, "reverse") ? 1 : !std::strcmp(e, "alternate") ? 2 : 0;
    }
    return order;
}

// run every fiber of blocks [first, first + count) round-robin until all are done
static void run_blocks(size_t first, size_t count) {
    unsigned long long spins = 0;
    const int order = fiber_order();
    for (;;) {
        unsigned alive = 0;
        const bool rev = order == 1 || (order == 2 && (spins & 1));
        for (size_t kk = 0; kk < count; ++kk) {
            Block &b = g_blocks[first + (rev ? count - 1 - kk : kk)];
            const size_t nf = b.fibers.size();
            for (size_t t = 0; t < nf; ++t) {
                Fiber &f = b.fibers[rev ? nf - 1 - t : t];
                if (f.done) continue;
                cur = &f;
                emu_ctx_switch(&g_sched_sp, cur->sp);
            }
            alive += b.alive;
        }
        if (!alive) break;
        if (++spins > 2000000000ull) {
            std::fprintf(stderr, "emu: grid appears deadlocked\n");
            std::abort();
        }
    }
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body) {
    const unsigned nthreads = check_block(block);
    ensure_stacks(kStack * kMaxThreads);
    g_body = &body;
    g_blockDim = {block.x, block.y, block.z};
    g_gridDim = {grid.x, grid.y, grid.z};
    g_blocks.resize(1);
    // one block at a time: `static` stand-ins for __shared__ variables belong to the running block
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                init_block(g_blocks[0], uint3_{bx, by, bz}, block, nthreads, shmem, g_stacks, kStack);
                poison_static_shared();
                run_blocks(0, 1);
            }
    cur = nullptr;
}

// hipLaunchCooperativeKernel: every block of the grid is resident at once (blocks may wait for each other through
// global memory; __builtin_amdgcn_s_sleep yields).  Such kernels must keep their LDS in the dynamic region
// (HIP_DYNAMIC_SHARED): the `static` stand-in for __shared__ variables would be shared by all blocks.
void launch_coop(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body) {
    const unsigned nthreads = check_block(block);
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0 || nblocks > 64) {
        std::fprintf(stderr, "emu: cooperative grid of %zu blocks unsupported\n", nblocks);
        std::abort();
    }
    ensure_stacks(kCoopStack * nthreads * nblocks);
    g_body = &body;
    g_blockDim = {block.x, block.y, block.z};
    g_gridDim = {grid.x, grid.y, grid.z};
    g_blocks.resize(nblocks);
    size_t k = 0;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx, ++k)
                init_block(g_blocks[k], uint3_{bx, by, bz}, block, nthreads, shmem, g_stacks + kCoopStack * nthreads * k, kCoopStack);
    poison_static_shared();
    run_blocks(0, nblocks);
    cur = nullptr;
}

}  // namespace emu

struct emu_event_ {
    std::chrono::steady_clock::time_point t;
};

extern "C" {
hipError_t hipMalloc(void **p, size_t n) {
    void *q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 256)) return hipErrorOutOfMemory;
    std::memset(q, 0xCD, n);  // poison: uninitialised reads show up
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipPeekAtLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu error"; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    std::memset(p, 0, sizeof(*p));
    // keeps emulated grids tiny; SLK_EMU_CUS widens them (cooperative grids of several workgroups)
    const char *cus = std::getenv("SLK_EMU_CUS");
    p->multiProcessorCount = cus ? std::atoi(cus) : 2;
    std::snprintf(p->name, sizeof(p->name), "emu");
    std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "emu");
    p->totalGlobalMem = size_t(1) << 34;
    p->sharedMemPerBlock = size_t(160) * 1024;
    p->maxSharedMemoryPerMultiProcessor = size_t(160) * 1024;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event_(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
}
