// tests/emu/emu_runtime.cpp -- TEST HARNESS ONLY (see tests/emu/hip/hip_runtime.h).
// Fiber scheduler that executes one HIP block at a time on the calling OS thread, plus
// host stand-ins for the handful of hip* runtime calls the engine makes.
#include <hip/hip_runtime.h>

#include <chrono>
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
uint3_ g_blockIdx, g_blockDim, g_gridDim;

static const size_t kStack = 128 * 1024;
static const unsigned kMaxThreads = 1024;
static char *g_stacks = nullptr;
static std::vector<Fiber> g_fibers;
static void *g_sched_sp = nullptr;
static const std::function<void()> *g_body = nullptr;
static std::vector<char> g_dyn;

struct WaveSync {
    unsigned long long slot[2][64];
    // one (arrived, generation) pair per (log2 width, segment)
    unsigned arrived[7][64];
    unsigned gen[7][64];
};
static std::vector<WaveSync> g_waves;
static unsigned g_alive = 0, g_bar_arrived = 0, g_bar_gen = 0;

void *dyn_shared() { return g_dyn.data(); }

static void yield_to_scheduler() { emu_ctx_switch(&cur->sp, g_sched_sp); }

static void fiber_entry() {
    (*g_body)();
    cur->done = true;
    --g_alive;
    // a finished thread no longer participates in barriers (hardware: terminated waves)
    if (g_alive && g_bar_arrived == g_alive) {
        g_bar_arrived = 0;
        ++g_bar_gen;
    }
    yield_to_scheduler();
    std::abort();  // never resumed
}

void syncthreads() {
    unsigned my = g_bar_gen;
    if (++g_bar_arrived == g_alive) {
        g_bar_arrived = 0;
        ++g_bar_gen;
        return;
    }
    while (g_bar_gen == my) yield_to_scheduler();
}

static int ilog2(int w) {
    int l = 0;
    while ((1 << l) < w) ++l;
    return l;
}

unsigned long long shfl_exchange(unsigned long long v, int src, int width) {
    if (width <= 1) return v;
    const unsigned lane = cur->linear & 63u;
    WaveSync &w = g_waves[cur->linear >> 6];
    const int lw = ilog2(width), seg = (int)lane / width;
    const unsigned my = w.gen[lw][seg];
    w.slot[my & 1][lane] = v;
    if (++w.arrived[lw][seg] == (unsigned)width) {
        w.arrived[lw][seg] = 0;
        ++w.gen[lw][seg];
    } else {
        while (w.gen[lw][seg] == my) yield_to_scheduler();
    }
    return w.slot[my & 1][src];
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body) {
    const unsigned nthreads = block.x * block.y * block.z;
    if (nthreads == 0 || nthreads > kMaxThreads || (nthreads & 63u)) {
        std::fprintf(stderr, "emu: unsupported block size %u\n", nthreads);
        std::abort();
    }
    if (!g_stacks) g_stacks = static_cast<char *>(std::malloc(kStack * kMaxThreads));
    g_fibers.resize(nthreads);
    g_waves.resize((nthreads + 63) / 64);
    g_dyn.assign(shmem + 64, 0);
    g_body = &body;
    g_blockDim = {block.x, block.y, block.z};
    g_gridDim = {grid.x, grid.y, grid.z};
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = {bx, by, bz};
                std::memset(g_waves.data(), 0, g_waves.size() * sizeof(WaveSync));
                g_alive = nthreads;
                g_bar_arrived = 0;
                for (unsigned t = 0; t < nthreads; ++t) {
                    Fiber &f = g_fibers[t];
                    f.linear = t;
                    f.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
                    f.done = false;
                    uintptr_t top = reinterpret_cast<uintptr_t>(g_stacks + kStack * (t + 1));
                    top &= ~uintptr_t(15);
                    void **sp = reinterpret_cast<void **>(top);
                    *--sp = nullptr;                                  // fake return address
                    *--sp = reinterpret_cast<void *>(&fiber_entry);   // popped by ret
                    for (int r = 0; r < 6; ++r) *--sp = nullptr;      // rbp rbx r12-r15
                    f.sp = sp;
                }
                unsigned long long spins = 0;
                while (g_alive) {
                    for (unsigned t = 0; t < nthreads; ++t) {
                        if (g_fibers[t].done) continue;
                        cur = &g_fibers[t];
                        emu_ctx_switch(&g_sched_sp, cur->sp);
                    }
                    if (++spins > 2000000000ull) {
                        std::fprintf(stderr, "emu: block (%u,%u,%u) appears deadlocked\n", bx, by, bz);
                        std::abort();
                    }
                }
            }
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
    p->multiProcessorCount = 2;  // keeps emulated grids tiny
    std::snprintf(p->name, sizeof(p->name), "emu");
    std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "emu");
    p->totalGlobalMem = size_t(1) << 34;
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
