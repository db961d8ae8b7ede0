// Microbenchmark (GPU box): does a loop whose BODY is tens of KB of straight-line code run at the rate of a small loop?
// One wave per SIMD; body = UNR x 114 independent v_mfma_f64_4x4x4_4b (8-byte instructions) or v_fma_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int UNR>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, int nit, double a0, double b0)
{
    { int anchor = 0; asm volatile("" : "+a"(anchor)); }
    double acc[3][38];
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 38; ++i) acc[j][i] = 0;
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < nit; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
            for (int s = 0; s < 38; ++s)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (MODE == 0) acc[j][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[j][s], 0, 0, 0);
                    else acc[j][s] = __builtin_fma(a, b, acc[j][s]);
                }
            asm volatile("" ::: "memory");
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double sum = 0;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 38; ++i) sum += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int UNR>
void run(const char *name, double *out, long long *cyc, int wgs)
{
    long long h;
    const int nit = 4096 / UNR;
    k<MODE, UNR><<<wgs, 256>>>(out, cyc, nit, 1.0, 2.0);
    k<MODE, UNR><<<wgs, 256>>>(out, cyc, nit, 1.0, 2.0);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-14s body %5.1f KB, %3d workgroups: %6.2f ticks per instruction\n", name, UNR * 114 * 8 / 1024.0, wgs, (double)h / (nit * UNR * 114.0));
}
int main()
{
    double *out; long long *cyc;
    (void)hipMalloc(&out, sizeof(double) * 256 * 256);
    (void)hipMalloc(&cyc, 64);
    for (int wgs : {1, 256}) {
        run<0, 1>("mfma f64 4x4x4", out, cyc, wgs);
        run<0, 8>("mfma f64 4x4x4", out, cyc, wgs);
        run<0, 16>("mfma f64 4x4x4", out, cyc, wgs);
        run<0, 32>("mfma f64 4x4x4", out, cyc, wgs);
        run<0, 64>("mfma f64 4x4x4", out, cyc, wgs);
        run<0, 128>("mfma f64 4x4x4", out, cyc, wgs);
        run<1, 1>("v_fma_f64", out, cyc, wgs);
        run<1, 16>("v_fma_f64", out, cyc, wgs);
        run<1, 64>("v_fma_f64", out, cyc, wgs);
        run<1, 128>("v_fma_f64", out, cyc, wgs);
    }
    return 0;
}
