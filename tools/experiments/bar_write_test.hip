// Can the host write straight into (fine-grained) device memory on this box -- a staging block that needs no fetch kernel?
//     hipcc --offload-arch=gfx950 -O2 -w bar_write_test.hip -o bar_write_test && for k in 0 1 2; do ./bar_write_test $k; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <unistd.h>
#include <sys/wait.h>
__global__ void rd(const int *p, int *out, int n) { int s = 0; for (int i = threadIdx.x; i < n; i += 64) s += p[i]; atomicAdd(out, s); }
int main(int argc, char **argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 0;          // run once per kind: a host store that faults ends the process
    int large = 0;
    hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, 0);
    void *p = nullptr;
    hipError_t e = kind == 0 ? hipExtMallocWithFlags(&p, 1 << 16, hipDeviceMallocFinegrained)
                 : kind == 1 ? hipMalloc(&p, 1 << 16) : hipExtMallocWithFlags(&p, 1 << 16, hipDeviceMallocUncached);
    printf("large BAR %d; kind %d (%s): alloc %s\n", large, kind, kind == 0 ? "fine-grained device" : kind == 1 ? "plain hipMalloc" : "uncached device", hipGetErrorString(e));
    fflush(stdout);
    if (e != hipSuccess) return 0;
    int *out; hipMalloc(&out, 4);
    volatile int *hp = (volatile int *)p;
    const int n = 1024;
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(out, 0, 4); hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) hp[i] = rep + 1;           // host stores into device memory
        auto t1 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(rd, dim3(1), dim3(64), 0, 0, (const int *)p, out, n);
        int h = 0; hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
        printf("  rep %d: host wrote 4 KB in %.2f us, kernel summed %d (expected %d)\n", rep,
               std::chrono::duration<double, std::micro>(t1 - t0).count(), h, n * (rep + 1));
        fflush(stdout);
    }
    return 0;
}
