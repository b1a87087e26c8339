// LDS read throughput on gfx950 for the access shapes of k_featurize3's stage 1: how many cycles of the CU's LDS does one
// wave-instruction take -- full-wave distinct addresses, broadcasts (3 or 9 distinct addresses per wave), and reads with part
// of the lanes masked off?  16 waves per CU (1024 threads), every wave issues independent reads back to back.
//     hipcc --offload-arch=gfx950 -O3 -w lds_read_bench.hip -o lds_read_bench && ./lds_read_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double __attribute__((ext_vector_type(2))) d2;
template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned long long *cyc, double *sink, int iters) {
    __shared__ __align__(16) double buf[8192];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 8192; i += 1024) buf[i] = 1e-300 * i;
    __syncthreads();
    const int j = (lane & 31) / 9, n = (lane & 31) % 9, half = lane >> 5;
    int a;                                           // element index
    bool on = true;
    if (MODE == 0) a = lane * 2;                     // b128, 64 distinct addresses, conflict free
    if (MODE == 1) a = j * 4 + half * 2;             // b128, 6 distinct addresses (the P pairs of stage 1)
    if (MODE == 2) a = lane;                         // b64, distinct
    if (MODE == 3) a = n;                            // b64, 9 distinct (the Q values of the centre role)
    if (MODE == 4) { a = j * 4 + half * 2; on = n < 4; }        // b128, 24 of 64 lanes active
    if (MODE == 5) { a = lane * 2; on = lane < 16; }            // b128, 16 contiguous lanes active
    if (MODE == 6) a = (n & 3) * 4 + half * 2;       // b128 of quads 32 B apart (neighbour role's clamped quads)
    if (MODE == 7) a = 0;                            // b32 broadcast
    a += wave * 256;
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (on) {
        for (int i = 0; i < iters; i++) {
            const int o = (i & 7) * 16;
            if (MODE == 2 || MODE == 3) {
                const __attribute__((address_space(3))) double *p = (const __attribute__((address_space(3))) double *)(buf + a + o);
                double x0 = p[0], x1 = p[64], x2 = p[128], x3 = p[192];
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
                acc0 += x0; acc1 += x1; acc2 += x2; acc3 += x3;
            } else if (MODE == 7) {
                const __attribute__((address_space(3))) int *p = (const __attribute__((address_space(3))) int *)(buf + a + o);
                int x0 = p[0], x1 = p[64], x2 = p[128], x3 = p[192];
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
                acc0 += x0; acc1 += x1; acc2 += x2; acc3 += x3;
            } else {
                const d2 *p = (const d2 *)(buf + a + o);
                d2 x0 = __builtin_nontemporal_load(p), x1 = __builtin_nontemporal_load(p + 32), x2 = __builtin_nontemporal_load(p + 64), x3 = __builtin_nontemporal_load(p + 96);
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
                acc0 += x0.x; acc1 += x1.y; acc2 += x2.x; acc3 += x3.y;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (t == 0) cyc[blockIdx.x] = t2 - t0;
    (void)t1;
    if (acc0 + acc1 + acc2 + acc3 == 12345.678) sink[0] = acc0;
}
template <int MODE> void run(const char *what) {
    unsigned long long *cyc; double *sink;
    hipMalloc(&cyc, 8 * 256); hipMalloc(&sink, 8);
    const int iters = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, cyc, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, cyc, 8 * 256, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; i++) avg += (double)h[i] / 256;
    printf("%-62s %.2f LDS cycles per wave-instruction (16 waves per CU)\n", what, avg / (16.0 * 4 * iters));
    hipFree(cyc); hipFree(sink);
}
// does a DS read beyond the workgroup's allocation return zero?
__global__ void oob(double *out) {
    extern __shared__ double dyn[];
    dyn[threadIdx.x] = 7.0;
    __syncthreads();
    const unsigned addr = 0xfffffff0u - 64u * threadIdx.x;                      // "negative" byte offsets
    double v;
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[threadIdx.x] = v;
}
int main() {
    run<0>("b128, 64 distinct addresses");
    run<1>("b128, 6 distinct addresses (P pairs: 3 rows x 2 halves)");
    run<6>("b128, 8 distinct addresses 32 B apart");
    run<4>("b128, 6 distinct addresses, 24 of 64 lanes active");
    run<5>("b128, lanes 0-15 active");
    run<2>("b64, 64 distinct addresses");
    run<3>("b64, 9 distinct addresses (Q values)");
    run<7>("b32, one address");
    double *o; hipMalloc(&o, 64 * 8);
    hipLaunchKernelGGL(oob, dim3(1), dim3(64), 1024, 0, o);
    double h[64]; hipMemcpy(h, o, 64 * 8, hipMemcpyDeviceToHost);
    int nz = 0; for (int i = 0; i < 64; i++) nz += h[i] != 0.0;
    printf("DS reads at wrapped (negative) addresses: %d of 64 lanes non-zero (0 = out-of-range reads return zero)\n", nz);
    return 0;
}
