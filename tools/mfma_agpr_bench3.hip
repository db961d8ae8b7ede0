// Microbenchmark (GPU box): v_mfma_f64_4x4x4_4b issue rate, one wave per SIMD, when the kernel occupies (nearly) the whole
// 512-entry register file: 228 accumulator AGPRs + ~230 live VGPRs, the A operands rotating through the high VGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 512
template <int NLIVE, int LDSKB>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, double a0, double b0)
{
    extern __shared__ double sh[];
    { int anchor = 0; asm volatile("" : "+a"(anchor)); }
    double acc[3][38];
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 38; ++i) acc[j][i] = 0;
    double live[NLIVE > 0 ? NLIVE : 1];
    for (int i = 0; i < NLIVE; ++i) live[i] = a0 + threadIdx.x * 1e-3 * (i + 1);
    double b[3];
    for (int i = 0; i < 3; ++i) b[i] = b0 - threadIdx.x * 1e-3 * (i + 1);
    if (LDSKB) sh[threadIdx.x] = a0;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int s = 0; s < 38; ++s) {
            const double av = NLIVE > 0 ? live[(s * 5) % NLIVE] : a0;
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b[j], acc[j][s], 0, 0, 0);
        }
        if (NLIVE > 0) {
#pragma unroll
            for (int i = 0; i < NLIVE; ++i) asm volatile("" : "+v"(live[i]));     // all of them stay live, in VGPRs
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double sum = 0;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 38; ++i) sum += acc[j][i];
    for (int i = 0; i < NLIVE; ++i) sum += live[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum + (LDSKB ? sh[threadIdx.x] : 0.0);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NLIVE, int LDSKB>
void run(const char *name, double *out, long long *cyc)
{
    long long h;
    (void)hipFuncSetAttribute((const void *)k<NLIVE, LDSKB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<NLIVE, LDSKB><<<256, 256, LDSKB * 1024>>>(out, cyc, 1.0, 2.0);
    k<NLIVE, LDSKB><<<256, 256, LDSKB * 1024>>>(out, cyc, 1.0, 2.0);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-60s %6.1f ticks per matrix instruction\n", name, (double)h / (N_IT * 114.0));
}
int main()
{
    double *out; long long *cyc;
    (void)hipMalloc(&out, sizeof(double) * 256 * 256);
    (void)hipMalloc(&cyc, 64);
    run<0, 0>("228 AGPR accumulators, few VGPRs", out, cyc);
    run<19, 0>("... + 19 live A operands", out, cyc);
    run<60, 0>("... + 60 live doubles (120 VGPRs)", out, cyc);
    run<110, 0>("... + 110 live doubles (220 VGPRs)", out, cyc);
    run<110, 146>("... + 110 live doubles, 146 KB of LDS", out, cyc);
    return 0;
}
