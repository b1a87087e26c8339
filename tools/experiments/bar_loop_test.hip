// The MD-step pattern without a fetch kernel: the host stores a block straight into device memory (large BAR), fences, launches a
// kernel that reads it and answers into pinned host memory, polls the answer -- 20 000 times, counting stale reads.
//     hipcc --offload-arch=gfx950 -O2 -w bar_loop_test.hip -o bar_loop_test && for k in 0 1; do ./bar_loop_test $k; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <immintrin.h>
__global__ void rd(const int *p, volatile int *answer, int n, int seq) {
    __shared__ int part[256];
    int s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) { answer[0] = part[0]; __threadfence_system(); __hip_atomic_store((int *)answer + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
int main(int argc, char **argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    void *p = nullptr;
    hipError_t e = kind == 0 ? hipExtMallocWithFlags(&p, 1 << 16, hipDeviceMallocFinegrained) : hipMalloc(&p, 1 << 16);
    if (e != hipSuccess) { printf("alloc failed\n"); return 1; }
    int *ans; hipHostMalloc((void **)&ans, 64, hipHostMallocDefault);
    ans[0] = ans[1] = 0;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    volatile int *hp = (volatile int *)p;
    const int n = 896, reps = 20000;                       // 3.5 KB: a 128-atom block
    int stale = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int rep = 1; rep <= reps; rep++) {
        const int v = (rep * 2654435761u) >> 20;
        for (int i = 0; i < n; i++) hp[i] = v + (i & 3);
        _mm_sfence();
        hipLaunchKernelGGL(rd, dim3(1), dim3(256), 0, st, (const int *)p, ans, n, rep);
        while (__atomic_load_n(ans + 1, __ATOMIC_ACQUIRE) != rep) {}
        const int want = n * v + (n / 4) * 6;
        if (ans[0] != want) stale++;
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("kind %d (%s): %d launches, %d stale reads, %.2f us per round trip\n", kind, kind == 0 ? "fine-grained device" : "plain hipMalloc", reps, stale, us);
    return 0;
}
