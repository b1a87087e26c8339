// Does v_mfma_f64_16x16x4 share the issue / double-precision pipe with the vector ALU on gfx950?  (Both fp64 peaks are
// quoted as 78.6 TF.)  One workgroup per CU, three groups of four waves (one wave of each group per SIMD); every group
// runs one of: m = independent v_mfma_f64_16x16x4, f = v_fma_f64 (8 chains), i = 32-bit integer multiply-add, - = idle.
//     hipcc --offload-arch=gfx950 -O3 dp_pipe_bench.hip -o dp_pipe_bench && ./dp_pipe_bench
// Measured on MI355X (ticks of s_memtime = shader cycles, ~1.9 GHz under this load):
//   m alone 64.0 cycles per MFMA;  f alone (1 wave / SIMD) 7.5 per v_fma_f64;  i alone 12.5 per multiply-add pair
//   m + f: MFMA 64.0, v_fma_f64 39.5;   m + i: MFMA 64.0, integer 44.5   -> an fp64 MFMA holds the SIMD's vector issue
//   for its 64 cycles (about 1.6 other vector instructions get through per MFMA): matrix and vector work add up.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(768) k(int kinds, int iters, unsigned long long *cyc, double *sink) {
    const int wave = threadIdx.x >> 6, slot = wave >> 2;
    const int grp = (kinds >> (4 * slot)) & 15;          // 0 mfma, 1 fma64, 2 int, 15 idle
    if (grp == 15) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double out = 0;
    if (grp == 0) {
        double4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        double x = threadIdx.x * 1e-3, y = 1.0 + x;
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
        }
        out = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (grp == 3) {                               // q: the four-block 4x4x4 form (one double per lane in and out)
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        double x = threadIdx.x * 1e-3, y = 1.0 + x;
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, y, a3, 0, 0, 0);
        }
        out = a0 + a1 + a2 + a3;
    } else if (grp == 1) {
        double f[8];
        for (int q = 0; q < 8; q++) f[q] = threadIdx.x * 1e-3 + q;
        const double m = 1.0000001, c = 1e-9;
        for (int i = 0; i < iters; i++)
#pragma unroll
            for (int q = 0; q < 8; q++) f[q] = __builtin_fma(f[q], m, c);
        for (int q = 0; q < 8; q++) out += f[q];
    } else {
        unsigned u[8];
        for (int q = 0; q < 8; q++) u[q] = threadIdx.x + q;
        for (int i = 0; i < iters; i++)
#pragma unroll
            for (int q = 0; q < 8; q++) u[q] = u[q] * 1664525u + 1013904223u;
        for (int q = 0; q < 8; q++) out += u[q];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) atomicMax(&cyc[slot], t1 - t0);
    if (out == 12345.678) sink[0] = out;
}
int main() {
    unsigned long long *cyc; double *sink;
    hipMalloc(&cyc, 64); hipMalloc(&sink, 8);
    const int iters = 20000;
    const char *cfgs[] = {"q--", "qf-", "qi-", "qqq", "m--", "f--", "i--", "mf-", "mi-", "fi-", "mfi", "fff", "iii", "mmm", "mff", "ffi"};
    for (const char *cfg : cfgs) {
        int kinds = 0;
        for (int q = 0; q < 3; q++) kinds |= (cfg[q] == 'm' ? 0 : cfg[q] == 'f' ? 1 : cfg[q] == 'i' ? 2 : cfg[q] == 'q' ? 3 : 15) << (4 * q);
        hipMemset(cyc, 0, 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(768), 0, 0, kinds, iters, cyc, sink);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("%s: %.3f ms;", cfg, ms);
        for (int q = 0; q < 3; q++) {
            if (cfg[q] == 'm') printf("  wave %d mfma %.1f cycles/instr", q, (double)h[q] / (4.0 * iters));
            if (cfg[q] == 'q') printf("  wave %d mfma4x4x4 %.1f cycles/instr", q, (double)h[q] / (4.0 * iters));
            if (cfg[q] == 'f') printf("  wave %d fma64 %.2f", q, (double)h[q] / (8.0 * iters));
            if (cfg[q] == 'i') printf("  wave %d int-mad %.2f", q, (double)h[q] / (8.0 * iters));
        }
        printf("\n");
    }
    return 0;
}
