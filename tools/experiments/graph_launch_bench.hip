// Launch cost of a three-kernel dependent chain (what an MD step of a small cell is: k_eval -> k_eval_collect_md -> k_frame_sum) as three
// hipLaunchKernelGGL calls against one hipGraphLaunch of the captured chain, each step waited for.  hipcc --offload-arch=gfx950 graph_launch_bench.hip && ./a.out
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k1(double *x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * 1.0000001 + 1e-9; }
__global__ void k2(double *x, double *y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] += x[i]; }
__global__ void k3(double *y, double *out, int n) { if (blockIdx.x == 0 && threadIdx.x == 0) { double s = 0; for (int i = 0; i < 64; i++) s += y[i]; out[0] = s; } }
int main() {
    const int n = 128 * 64, iters = 3000;
    double *x, *y, *out;
    hipMalloc(&x, 8 * n); hipMalloc(&y, 8 * n); hipMalloc(&out, 8);
    hipMemset(x, 0, 8 * n); hipMemset(y, 0, 8 * n);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    auto chain = [&]() {
        hipLaunchKernelGGL(k1, dim3(128), dim3(64), 0, st, x, n);
        hipLaunchKernelGGL(k2, dim3(128), dim3(64), 0, st, x, y, n);
        hipLaunchKernelGGL(k3, dim3(1), dim3(64), 0, st, y, out, n);
    };
    for (int i = 0; i < 200; i++) { chain(); hipStreamSynchronize(st); }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; i++) { chain(); hipStreamSynchronize(st); }
    double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    chain();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 200; i++) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; i++) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }
    double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
    printf("three launches + wait: %.2f us per step;  one graph launch + wait: %.2f us per step\n", plain, graph);
    return 0;
}
