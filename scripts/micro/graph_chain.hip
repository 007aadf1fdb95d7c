// Micro-benchmark (diagnostic, not part of the library): what a chain of small DEPENDENT kernels costs per kernel on one
// stream -- the shape of the launch path at minibatches of a few thousand interactions (two row passes per minibatch) --
//   A  plain launches on a stream (what slk_bilinear_train does above the persistent kernel's range)
//   B  the same chain captured once into a hipGraph: instantiate time, then time per graph launch
//   C  the graph re-instantiated every time (what a chain whose kernel arguments change per epoch would need)
// for kernels that do ~nothing (launch-bound) and kernels of ~5 us of work.
// build: hipcc --offload-arch=gfx950 -O3 -o graph_chain scripts/micro/graph_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_work(float *p, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = p[i];
    for (int k = 0; k < iters; ++k) x = x * 1.0000001f + 1e-7f;
    p[i] = x;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const int n = 1 << 16, chain = 512;
    float *d = nullptr;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemset(d, 0, n * sizeof(float)));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int iters : {1, 2000}) {
        auto enqueue = [&]() {
            for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n, iters);
        };
        enqueue();
        CK(hipStreamSynchronize(s));
        double t0 = now_us();
        for (int r = 0; r < 5; ++r) enqueue();
        const double t_enq = now_us();
        CK(hipStreamSynchronize(s));
        double t1 = now_us();
        printf("iters %4d  A plain launches: %.2f us per kernel (host enqueue %.2f us per kernel)\n", iters, (t1 - t0) / (5 * chain),
               (t_enq - t0) / (5 * chain));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        enqueue();
        CK(hipStreamEndCapture(s, &g));
        t0 = now_us();
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const double t_inst = now_us() - t0;
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = now_us();
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t1 = now_us();
        printf("iters %4d  B graph: instantiate %.1f us (%.2f per node), launch %.2f us per kernel\n", iters, t_inst, t_inst / chain,
               (t1 - t0) / (5 * chain));
        t0 = now_us();
        for (int r = 0; r < 3; ++r) {
            hipGraphExec_t ge2;
            CK(hipGraphInstantiate(&ge2, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge2, s));
            CK(hipStreamSynchronize(s));
            CK(hipGraphExecDestroy(ge2));
        }
        t1 = now_us();
        printf("iters %4d  C instantiate + launch every time: %.2f us per kernel\n", iters, (t1 - t0) / (3 * chain));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
