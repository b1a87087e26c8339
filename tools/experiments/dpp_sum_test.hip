// wave-wide sum of doubles through DPP (no LDS crossbar): correctness against a serial sum.  hipcc --offload-arch=gfx950 dpp_sum_test.hip && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(h2, l2);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_take<0xB1, 0xf>(v);        // quad_perm [1,0,3,2]
    v += dpp_take<0x4E, 0xf>(v);        // quad_perm [2,3,0,1]
    v += dpp_take<0x141, 0xf>(v);       // row_half_mirror
    v += dpp_take<0x140, 0xf>(v);       // row_mirror: every lane of a row holds the row's sum
    v += dpp_take<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    v += dpp_take<0x143, 0xc>(v);       // row_bcast:31 into rows 2 and 3: lanes 48..63 hold the wave's sum
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__global__ void k(const double *in, double *out) {
    out[blockIdx.x * 64 + threadIdx.x] = wave_sum_dpp(in[blockIdx.x * 64 + threadIdx.x]);
}
int main() {
    const int nb = 64;
    double *h = (double *)malloc(8 * 64 * nb), *d, *o, *ho = (double *)malloc(8 * 64 * nb);
    srand(1);
    for (int i = 0; i < 64 * nb; i++) h[i] = (rand() / (double)RAND_MAX - 0.5) * pow(10.0, rand() % 6);
    hipMalloc(&d, 8 * 64 * nb); hipMalloc(&o, 8 * 64 * nb);
    hipMemcpy(d, h, 8 * 64 * nb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, d, o);
    hipMemcpy(ho, o, 8 * 64 * nb, hipMemcpyDeviceToHost);
    double worst = 0; int uniform = 1;
    for (int b = 0; b < nb; b++) {
        long double s = 0, a = 0;
        for (int i = 0; i < 64; i++) { s += h[b * 64 + i]; a += fabsl(h[b * 64 + i]); }
        for (int i = 0; i < 64; i++) { worst = fmax(worst, fabs((double)(ho[b * 64 + i] - s)) / (double)a); uniform &= ho[b * 64 + i] == ho[b * 64]; }
    }
    printf("worst |err| / sum|x| = %.3e, all lanes equal: %d\n", worst, uniform);
    return !(worst < 1e-15 && uniform);
}
