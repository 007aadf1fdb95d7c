// Micro-benchmark (diagnostic, not part of the library): the latencies the persistent epoch kernel (csrc/slk_epoch.hip)
// is made of, measured with s_memtime inside one wave:
//   A  dependent chain of PLAIN 8-B loads, lines this CU has never touched (first pass) and again (second pass)
//   B  dependent chain of sc1 (agent-scope) 8-B loads: lines last written by an earlier kernel / re-read
//   C  16 independent sc1 8-B loads issued together, one wait
//   D  16 sc1 8-B stores to distinct lines + s_waitcnt vmcnt(0) (the "drain" before a barrier arrival)
//   E  returning device-scope atomic add (the barrier arrival)
//   F  two workgroups ping-pong one word with sc1 stores / sc1 polling loads: one-way hand-over latency
//   G  a block of dependent VALU work (expf + divide + sqrt, the per-pair loss and the Adagrad update): cycles per element
// build: hipcc --offload-arch=gfx950 -O3 -o coherent_latency scripts/micro/coherent_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define AG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// buf[i * STRIDE] holds the index of the next element: a random cycle over n lines
__global__ void k_chain(const unsigned long long *buf, int n, int stride, int sc1, unsigned long long *out) {
    if (threadIdx.x != 0) return;
    for (int pass = 0; pass < 2; ++pass) {
        unsigned long long i = 0;
        drain();
        const unsigned long long t0 = now();
        for (int k = 0; k < n; ++k)
            i = sc1 ? __hip_atomic_load(buf + i * stride, AG) : *(volatile const unsigned long long *)(buf + i * stride);
        drain();
        const unsigned long long t1 = now();
        out[pass] = (t1 - t0) / n;
        out[2] = i;
    }
}

__global__ void k_batch(unsigned long long *buf, int stride, unsigned long long *out) {
    if (threadIdx.x != 0) return;
    unsigned long long v[16];
    drain();
    unsigned long long t0 = now();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = __hip_atomic_load(buf + (size_t)(k * 37 + 5) * stride, AG);
    drain();
    unsigned long long t1 = now();
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
    out[0] = t1 - t0;
    // D: stores
    drain();
    t0 = now();
#pragma unroll
    for (int k = 0; k < 16; ++k) __hip_atomic_store(buf + (size_t)(k * 41 + 900) * stride, s + k, AG);
    drain();
    t1 = now();
    out[1] = t1 - t0;
    // E: returning atomics, one after the other
    t0 = now();
    unsigned acc = 0;
    for (int k = 0; k < 16; ++k) acc += __hip_atomic_fetch_add((unsigned *)(buf + (size_t)2000 * stride), 1u + (acc & 1u), AG);
    drain();
    t1 = now();
    out[2] = (t1 - t0) / 16;
    out[3] = acc;
}

// C2/C3/D2/D3: 16 independent accesses per lane, ALL lanes of `blockDim.x / 64` waves active (each lane its own line), by flavour:
//   0 plain 8-B, 1 sc1 8-B (agent-scope atomic), 2 sc1 16-B (raw buffer op, aux = sc1), 3 plain 16-B
typedef int v4i __attribute__((ext_vector_type(4)));
template <int FLAVOUR, bool STORE>
__global__ void k_flavour(unsigned long long *buf, size_t base_words, unsigned long long *out) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0x7fffffff, 0x00020000);
    const size_t lane_off = base_words + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 64;  // 512 B apart
    unsigned long long acc = 0;
    v4i acc4 = {0, 0, 0, 0};
    __syncthreads();
    drain();
    const unsigned long long t0 = now();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const size_t w = lane_off + (size_t)k * 16384 * 64 / 8;  // each k a different region
        if (!STORE) {
            if (FLAVOUR == 0) acc += *(volatile unsigned long long *)(buf + w);
            if (FLAVOUR == 1) acc += __hip_atomic_load(buf + w, AG);
            if (FLAVOUR == 2) acc4 += __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(w * 8), 0, 16);
            if (FLAVOUR == 3) acc4 += *(volatile v4i *)(buf + w);
        } else {
            const v4i x = {k, k, k, k};
            if (FLAVOUR == 0) *(volatile unsigned long long *)(buf + w) = (unsigned long long)k;
            if (FLAVOUR == 1) __hip_atomic_store(buf + w, (unsigned long long)k, AG);
            if (FLAVOUR == 2) __builtin_amdgcn_raw_buffer_store_b128(x, rs, (int)(w * 8), 0, 16);
            if (FLAVOUR == 3) *(volatile v4i *)(buf + w) = x;
        }
    }
    drain();
    const unsigned long long t1 = now();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc + acc4.x + acc4.y == 0x123456789ull) out[1] = acc;
}

// F: block 0 and block `peer` bounce a counter: even values written by block 0, odd by the peer
__global__ void k_pingpong(unsigned *word, int peer, int rounds, unsigned long long *out) {
    if (threadIdx.x != 0 || (blockIdx.x != 0 && blockIdx.x != (unsigned)peer)) return;
    const bool a = blockIdx.x == 0;
    drain();
    const unsigned long long t0 = now();
    for (int r = 0; r < rounds; ++r) {
        unsigned spins = 0;
        if (a) {
            __hip_atomic_store(word, 2u * r + 1u, AG);
            while (__hip_atomic_load(word, AG) != 2u * r + 2u && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
        } else {
            while (__hip_atomic_load(word, AG) != 2u * r + 1u && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(word, 2u * r + 2u, AG);
        }
    }
    drain();
    if (a) out[0] = (now() - t0) / (2ull * rounds);  // one-way hops
}

__global__ void k_valu(float *p, int iters, unsigned long long *out) {
    float x = p[threadIdx.x], s = p[threadIdx.x + 64], acc = 0.0f;
    drain();
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = now();
    for (int i = 0; i < iters; ++i) {
        const float sg = 1.0f / (1.0f + expf(-x));          // sigmoid (bpr loss)
        const float g = -(sg * (1.0f - sg)) * 0.001f;
        s += g * g;                                          // Adagrad
        x += -0.01f * (g / (sqrtf(s) + 1e-10f));
        acc += sg;
    }
    const unsigned long long t1 = now();
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = (t1 - t0) / iters;
        out[1] = t1 - t0;   // shader-cycle counter ticks ...
        out[2] = w1 - w0;   // ... over this many ticks of the constant 100 MHz wall clock
    }
    p[threadIdx.x] = x + acc;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n = 512, stride = 512;  // 4 KB apart: every element its own line and DRAM page
    std::vector<unsigned long long> h((size_t)4096 * stride, 0);
    // random cycle over lines 0..n-1
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    unsigned seed = 12345;
    for (int i = n - 1; i > 0; --i) { seed = seed * 1664525u + 1013904223u; int j = seed % (i + 1); std::swap(perm[i], perm[j]); }
    for (int i = 0; i < n; ++i) h[(size_t)perm[i] * stride] = perm[(i + 1) % n];
    unsigned long long *buf, *out;
    CK(hipMalloc(&buf, h.size() * 8));
    CK(hipMalloc(&out, 64 * 8));
    unsigned long long res[8];
    int clk = 0;
    CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    printf("shader clock (attr) %.0f MHz; numbers below in shader cycles per operation\n", clk / 1e3);
    for (int sc1 = 0; sc1 < 2; ++sc1) {
        CK(hipMemcpy(buf, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, buf, n, stride, sc1, out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(res, out, 24, hipMemcpyDeviceToHost));
        printf("%s dependent 8-B load chain over %d lines 4 KB apart: first pass %llu, second pass %llu cycles per load\n",
               sc1 ? "B sc1  " : "A plain", n, res[0], res[1]);
    }
    hipLaunchKernelGGL(k_batch, dim3(1), dim3(64), 0, 0, buf, stride, out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(res, out, 32, hipMemcpyDeviceToHost));
    printf("C 16 independent sc1 8-B loads + one wait: %llu cycles\nD 16 sc1 8-B stores + drain: %llu cycles\n"
           "E returning device-scope atomic add, dependent: %llu cycles each\n", res[0], res[1], res[2]);
    {
        unsigned long long *big;
        const size_t words = (size_t)1 << 27;  // 1 GiB
        CK(hipMalloc(&big, words * 8));
        CK(hipMemset(big, 0, words * 8));
        const char *names[4] = {"plain 8-B", "sc1 8-B (atomic)", "sc1 16-B (buffer op)", "plain 16-B"};
        for (int threads : {64, 256}) {
            for (int blocks : {1, 64}) {
                for (int fl = 0; fl < 4; ++fl) {
                    unsigned long long r[2][2];
                    for (int st = 0; st < 2; ++st) {
                        for (int rep = 0; rep < 2; ++rep) {  // second repetition: lines warm in L2
                            const size_t base = (size_t)(fl * 2 + st) * 1024;  // distinct 8-KB-aligned offsets within the lane stride? no: shift by words
                            dim3 g(blocks), b(threads);
                            if (st == 0) {
                                if (fl == 0) hipLaunchKernelGGL((k_flavour<0, false>), g, b, 0, 0, big, base, out);
                                if (fl == 1) hipLaunchKernelGGL((k_flavour<1, false>), g, b, 0, 0, big, base, out);
                                if (fl == 2) hipLaunchKernelGGL((k_flavour<2, false>), g, b, 0, 0, big, base, out);
                                if (fl == 3) hipLaunchKernelGGL((k_flavour<3, false>), g, b, 0, 0, big, base, out);
                            } else {
                                if (fl == 0) hipLaunchKernelGGL((k_flavour<0, true>), g, b, 0, 0, big, base, out);
                                if (fl == 1) hipLaunchKernelGGL((k_flavour<1, true>), g, b, 0, 0, big, base, out);
                                if (fl == 2) hipLaunchKernelGGL((k_flavour<2, true>), g, b, 0, 0, big, base, out);
                                if (fl == 3) hipLaunchKernelGGL((k_flavour<3, true>), g, b, 0, 0, big, base, out);
                            }
                            CK(hipDeviceSynchronize());
                            CK(hipMemcpy(res, out, 8, hipMemcpyDeviceToHost));
                            r[st][rep] = res[0];
                        }
                    }
                    printf("H %3d threads x %2d blocks, 16 independent %-22s per lane: loads %5llu / %5llu cycles (cold / warm), "
                           "stores + drain %5llu / %5llu\n", threads, blocks, names[fl], r[0][0], r[0][1], r[1][0], r[1][1]);
                }
            }
        }
        CK(hipFree(big));
    }
    unsigned *word;
    CK(hipMalloc(&word, 256));
    for (int peer : {1, 8, 9, 63}) {
        CK(hipMemset(word, 0, 256));
        hipLaunchKernelGGL(k_pingpong, dim3(64), dim3(64), 0, 0, word, peer, 200, out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(res, out, 8, hipMemcpyDeviceToHost));
        printf("F ping-pong block 0 <-> block %d (sc1 store -> sc1 poll): %llu cycles per one-way hand-over\n", peer, res[0]);
    }
    float *fp;
    CK(hipMalloc(&fp, 1024));
    CK(hipMemset(fp, 0, 1024));
    for (int iters : {1000, 100000}) {
        for (int blocks : {1, 64}) {
            hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(64), 0, 0, fp, iters, out);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(res, out, 24, hipMemcpyDeviceToHost));
            printf("G sigmoid + Adagrad update chain (one wave per SIMD, %d blocks, %d iterations): %llu counter ticks per element; "
                   "%llu counter ticks in %llu wall ticks of 10 ns => counter runs at %.0f MHz, %.3f us per element\n", blocks, iters,
                   res[0], res[1], res[2], (double)res[1] / ((double)res[2] * 0.01), (double)res[2] * 0.01 / iters);
        }
    }
    return 0;
}
