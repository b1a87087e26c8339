// micro-benchmark: cost of LDS accumulate flavours on gfx950 (cycles per wave-instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void k(long long *out, int iters, int nwaves_active) {
    __shared__ double buf[8192];
    __shared__ unsigned long long ibuf[4096];
    int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 8192; i += blockDim.x) buf[i] = 0;
    for (int i = t; i < 4096; i += blockDim.x) ibuf[i] = 0;
    __syncthreads();
    if (wave >= nwaves_active) return;
    double v = 1.0 + lane * 1e-3;
    int base = wave * 512;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        int idx = base + ((lane + i * 7) & 63) + (i & 3) * 64;   // distinct address per lane
        if (MODE == 0) __hip_atomic_fetch_add(&buf[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 1) __hip_atomic_fetch_add(&ibuf[idx], (unsigned long long)(v * 1e6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) __hip_atomic_fetch_add((unsigned *)&ibuf[idx], (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) __hip_atomic_fetch_add((float *)&buf[idx], (float)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 4) { buf[idx] += v; }
        if (MODE == 5) { if (lane < 4) __hip_atomic_fetch_add(&buf[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (MODE == 6) { int j = base + (lane & 3) + (i & 3) * 64;  // 16-way same-address
                         __hip_atomic_fetch_add(&buf[j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (MODE == 7) { double2 x = *(double2 *)&buf[base + (lane & 3) * 2 + (i & 7) * 16]; v += x.x * 1e-30 + x.y * 1e-30; }
        if (MODE == 8) { double x = buf[base + (i & 63)]; v += x * 1e-30; }   // uniform broadcast b64 read
        if (MODE == 9) { v += __shfl(v, i & 63) * 1e-30; }                    // readlane-like broadcast
        if (MODE == 10) { v = v * 1.0000001 + 1e-9; }                          // dependent fp64 fma
    }
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (v == 12345.678) out[100] = (long long)buf[lane];
}
template <int MODE> void run(const char *name) {
    long long *d; hipMalloc(&d, 8 * 4096); long long h[16];
    for (int nw : {1, 4, 8}) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(512), 0, 0, d, 2000, nw);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-28s waves=%d  cycles/instr(wave0)=%.1f\n", name, nw, h[0] / 2000.0);
    }
    hipFree(d);
}
int main() {
    run<0>("ds_add_f64"); run<1>("ds_add_u64"); run<2>("ds_add_u32"); run<3>("ds_add_f32");
    run<4>("plain RMW b64"); run<5>("ds_add_f64 4 lanes"); run<6>("ds_add_f64 16-way conflict");
    run<7>("ds_read_b128 (4 addr)"); run<8>("ds_read_b64 uniform"); run<9>("shfl uniform"); run<10>("fp64 fma dep");
    return 0;
}
