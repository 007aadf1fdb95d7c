// tests/emu/hip/hip_runtime.h -- TEST HARNESS ONLY.
//
// A tiny single-threaded stand-in for <hip/hip_runtime.h> so that the UNMODIFIED kernel
// sources in spotlight_amd/csrc/*.hip can be compiled by g++ and stepped through on a box
// without a GPU (this container has none).  Every HIP thread of a block is a fiber; fibers
// run round-robin and only switch at __syncthreads()/__shfl*() (the only places HIP threads
// may communicate), so indexing/segment/sort-key logic, the C ABI and the host control
// flow are exercised for real.  It says nothing about speed, memory ordering or occupancy
// and it is never loaded by spotlight_amd (tests/emu/build_emu.py builds
// tests/emu/_build/libspotlight_emu.so; only tests/ open it).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
// __shared__ variables: statics gathered in one ELF section, which the runtime fills with a poison pattern before every block starts
// (emu_runtime.cpp: poison_static_shared) -- LDS holds whatever the last workgroup left there, never zeros
#define __shared__ static __attribute__((section("emu_shared")))
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define HIP_DYNAMIC_SHARED(type, var) type *var = reinterpret_cast<type *>(::emu::dyn_shared());

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };

#define SLK_EMU_VEC2(T, N) struct N { T x, y; };
#define SLK_EMU_VEC4(T, N) struct N { T x, y, z, w; };
SLK_EMU_VEC2(float, float2) SLK_EMU_VEC4(float, float4)
SLK_EMU_VEC2(unsigned, uint2) SLK_EMU_VEC4(unsigned, uint4)
SLK_EMU_VEC2(int, int2) SLK_EMU_VEC4(int, int4)
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }

typedef int hipError_t;
typedef struct emu_stream_ *hipStream_t;
typedef struct emu_event_ *hipEvent_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipErrorOutOfMemory 2
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

struct hipDeviceProp_t {
    int multiProcessorCount;
    char name[64];
    char gcnArchName[64];
    size_t totalGlobalMem;
    size_t sharedMemPerBlock;
    size_t maxSharedMemoryPerMultiProcessor;
};

namespace emu {
struct Fiber {
    void *sp;
    uint3_ tid;
    unsigned linear;
    bool done;
    void *blk;  // the fiber's block (emu_runtime.cpp)
};
extern Fiber *cur;
extern uint3_ g_blockDim, g_gridDim;
uint3_ block_idx();
void *dyn_shared();
void yield();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
void launch_coop(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
void next_launch_resident();   // SLK_RESIDENT_GRID_LAUNCH(): the next plain launch runs all of its blocks concurrently
bool take_resident_flag();
void syncthreads();
unsigned long long shfl_exchange(unsigned long long v, int src_lane_in_wave, int width);
unsigned long long ballot_exchange(bool pred);
}  // namespace emu

#define threadIdx (::emu::cur->tid)
#define blockIdx (::emu::block_idx())
#define blockDim (::emu::g_blockDim)
#define gridDim (::emu::g_gridDim)
#define warpSize 64

template <class... KArgs, class... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem,
                                      hipStream_t, Args... args) {
    if (emu::take_resident_flag())
        emu::launch_coop(grid, block, shmem, [=]() { kernel(static_cast<KArgs>(args)...); });
    else
        emu::launch(grid, block, shmem, [=]() { kernel(static_cast<KArgs>(args)...); });
}

// cooperative launch: all blocks resident, arguments passed as an array of pointers (one kernel parameter here)
template <class A>
static inline hipError_t hipLaunchCooperativeKernel(void (*kernel)(A), dim3 grid, dim3 block, void **args, size_t shmem,
                                                    hipStream_t) {
    const A a = *static_cast<const A *>(args[0]);
    emu::launch_coop(grid, block, shmem, [=]() { kernel(a); });
    return hipSuccess;
}

static inline void __syncthreads() { emu::syncthreads(); }
// agent-scope relaxed atomics (gfx950: sc1 accesses that bypass the CU's L1): fibers never pre-empt and there are no
// caches here, so plain accesses; a sleeping wave lets the other fibers (= other workgroups) run
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
template <class T, class V>
static inline T emu_fetch_add(T *p, V v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class V>
static inline T emu_fetch_or(T *p, V v) { T o = *p; *p = (T)(o | (T)v); return o; }
#define __hip_atomic_fetch_add(p, v, order, scope) emu_fetch_add((p), (v))
#define __hip_atomic_fetch_or(p, v, order, scope) emu_fetch_or((p), (v))
static inline void __builtin_amdgcn_s_sleep(int) { emu::yield(); }
static inline int __lane_id() { return (int)(emu::cur->linear & 63u); }

template <class T>
static inline T emu_shfl_to(T v, int src, int width) {
    static_assert(sizeof(T) <= 8, "shfl of <= 8 byte types only");
    unsigned long long bits = 0;
    std::memcpy(&bits, &v, sizeof(T));
    bits = emu::shfl_exchange(bits, src, width);
    T out;
    std::memcpy(&out, &bits, sizeof(T));
    return out;
}
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    int self = __lane_id();
    int idx = self ^ mask;
    if (idx >= ((self + width) & ~(width - 1))) idx = self;
    return emu_shfl_to(v, idx, width);
}
template <class T>
static inline T __shfl(T v, int src, int width = 64) {
    int self = __lane_id();
    int idx = (src & (width - 1)) + (self & ~(width - 1));
    return emu_shfl_to(v, idx, width);
}
template <class T>
static inline T __shfl_up(T v, unsigned delta, int width = 64) {
    int self = __lane_id();
    int idx = self - (int)delta;
    if (idx < (self & ~(width - 1))) idx = self;
    return emu_shfl_to(v, idx, width);
}
template <class T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int self = __lane_id();
    int idx = self + (int)delta;
    if ((idx & ~(width - 1)) != (self & ~(width - 1))) idx = self;
    return emu_shfl_to(v, idx, width);
}

// atomics: fibers never pre-empt, so plain read-modify-write is atomic.
template <class T>
static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned *p, int v) { unsigned o = *p; *p = o + (unsigned)v; return o; }
template <class T>
static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T>
static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T>
static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <class T>
static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
static inline void __threadfence() {}

#define __expf expf
static inline int __ffs(int x) { return __builtin_ffs(x); }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
// wave-wide vote: every lane of the wave must call it (as for the shuffles)
static inline unsigned long long __ballot(int pred) { return emu::ballot_exchange(pred != 0); }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
template <class T>
static inline T __ldg(const T *p) { return *p; }

extern "C" {
hipError_t hipMalloc(void **p, size_t n);
#define hipHostMallocDefault 0u
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }  // host == device memory here
hipError_t hipFree(void *p);
static inline hipError_t hipHostFree(void *p) { return hipFree(p); }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipDeviceSynchronize(void);
hipError_t hipGetLastError(void);
hipError_t hipPeekAtLastError(void);
const char *hipGetErrorString(hipError_t e);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipEventCreate(hipEvent_t *e);
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(0x10); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
// stream flavours (CU masks, priorities) only change where and when kernels run: the emulator executes everything in order
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *) { *s = reinterpret_cast<hipStream_t>(0x20); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = reinterpret_cast<hipStream_t>(0x30); return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
}
template <class T>
static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc(reinterpret_cast<void **>(p), n); }
// every kernel "fits": the emulator runs a resident grid's blocks as concurrent fibers whatever their footprint
template <class F>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, F, int, size_t) { *n = 8; return hipSuccess; }
template <class T>
static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f) { return hipHostMalloc(reinterpret_cast<void **>(p), n, f); }
