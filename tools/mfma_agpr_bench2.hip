// Microbenchmark (GPU box): issue rate of v_mfma_f64_4x4x4_4b from ONE wave per SIMD under the operand patterns of the
// three-pair T63 kernel: distinct A operands, three B operands per A, accumulators 76 registers apart.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 512
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, const double *src, double a0, double b0)
{
    { int anchor = 0; asm volatile("" : "+a"(anchor)); }
    double acc[3][38];
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 38; ++i) acc[j][i] = 0;
    double a[8], b[3];
    for (int i = 0; i < 8; ++i) a[i] = a0 + threadIdx.x * 1e-3 * (i + 1);
    for (int i = 0; i < 3; ++i) b[i] = b0 - threadIdx.x * 1e-3 * (i + 1);
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int s = 0; s < 38; ++s) {
            const double av = (MODE & 1) ? a[s & 7] : a[0];
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, (MODE & 2) ? b[j] : b[0], acc[j][s], 0, 0, 0);
            if ((MODE & 4) && (s & 7) == 7) {     // new B operands now and then (as per quad)
#pragma unroll
                for (int j = 0; j < 3; ++j) b[j] = b[j] * 1.0000001 + 1e-9;
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double sum = 0;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 38; ++i) sum += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char *name, double *out, long long *cyc)
{
    long long h;
    k<MODE><<<256, 256>>>(out, cyc, out, 1.0, 2.0);
    k<MODE><<<256, 256>>>(out, cyc, out, 1.0, 2.0);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-60s %6.1f ticks per matrix instruction\n", name, (double)h / (N_IT * 114.0));
}
int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * 256 * 256);
    hipMalloc(&cyc, 64);
    run<0>("114 accumulators [3][38], one A, one B", out, cyc);
    run<1>("... eight A operands in turn", out, cyc);
    run<2>("... three B operands per A", out, cyc);
    run<3>("... both", out, cyc);
    run<7>("... both, B recomputed every 8 slots", out, cyc);
    return 0;
}
