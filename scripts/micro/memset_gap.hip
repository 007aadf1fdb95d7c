// Micro-benchmark (diagnostic, not part of the library): where do the ~240 us GPU-idle gaps in
// front of rocPRIM's radix sort come from?  Times, with hipEvents on one stream:
//   A  busy kernel -> busy kernel                       (baseline launch gap)
//   B  busy kernel -> hipMemsetAsync(4 KB) -> kernel    (rocPRIM's histogram clear)
//   C  busy kernel -> fill kernel(4 KB)    -> kernel    (same clear done by a kernel)
//   D  busy kernel -> rocprim::radix_sort_pairs(1M u32/u32, 24 bits)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <rocprim/rocprim.hpp>

__global__ void busy(float *p, int iters) {
    float x = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) x = x * 1.000001f + 0.5f;
    p[threadIdx.x] = x;
}
__global__ void fill(unsigned *p, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    float *buf; unsigned *small; unsigned *k0, *k1, *v0, *v1; void *tmp = nullptr; size_t tmpsz = 0;
    const size_t N = 1 << 20;
    CK(hipMalloc(&buf, 4096)); CK(hipMalloc(&small, 4096));
    CK(hipMalloc(&k0, N * 4)); CK(hipMalloc(&k1, N * 4)); CK(hipMalloc(&v0, N * 4)); CK(hipMalloc(&v1, N * 4));
    CK(hipMemset(k0, 0x5a, N * 4)); CK(hipMemset(v0, 1, N * 4));
    CK(rocprim::radix_sort_pairs(nullptr, tmpsz, k0, k1, v0, v1, N, 0u, 24u, 0));
    CK(hipMalloc(&tmp, tmpsz));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int IT = 20000;  // ~ a few hundred us
    hipStream_t created = s;
    for (int pass = 0; pass < 2; ++pass) {
    s = pass ? (hipStream_t)0 : created;
    printf("---- %s stream\n", pass ? "NULL (legacy default)" : "created");
    for (int variant = 0; variant < 5; ++variant) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int k = 0; k < 10; ++k) {
                hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, s, buf, IT);
                if (variant == 1) CK(hipMemsetAsync(small, 0, 4096, s));
                if (variant == 2) hipLaunchKernelGGL(fill, dim3(4), dim3(256), 0, s, small, 1024);
                if (variant == 3) CK(rocprim::radix_sort_pairs(tmp, tmpsz, k0, k1, v0, v1, N, 0u, 24u, s));
                if (variant == 4) { size_t q = 0; CK(rocprim::radix_sort_pairs(nullptr, q, k0, k1, v0, v1, N, 0u, 24u, s)); }
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const char *names[] = {"A kernel only", "B + hipMemsetAsync", "C + fill kernel", "D + rocprim sort 1M", "E + rocprim size query"};
            if (rep == 2) printf("%-24s %8.1f us per iteration\n", names[variant], ms * 100.0f);
        }
    }
    }
    return 0;
}
