// Microbenchmark (GPU box): does v_mfma_f64_4x4x4_4b issue at the same rate with its accumulators in AGPRs (a[..] destination
// form) as with VGPR accumulators, from ONE wave per SIMD?  And next to interleaved FP64 VALU / LDS reads?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_agpr_bench.hip -o gpurun_out/mfma_agpr && gpurun_out/mfma_agpr
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 1024
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, double a0, double b0)
{
    __shared__ double sh[2048];
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3, c = a0 * 0.5;
    if (MODE & 1) { int anchor = 0; asm volatile("" : "+a"(anchor)); }
    sh[threadIdx.x] = a; sh[threadIdx.x + 256] = b;
    double acc[NACC];
    double v[8] = {a, b, c, a + b, a - b, b - c, c + a, a * b};
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
            if (MODE & 2) v[i & 7] = __builtin_fma(v[i & 7], b, c);          // one FP64 VALU op per matrix instruction
            if (MODE & 4) v[i & 7] += sh[(threadIdx.x + 8 * i + it) & 2047];  // one LDS read per matrix instruction
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int NACC>
void run(const char *name, double *out, long long *cyc)
{
    long long h;
    k<MODE, NACC><<<256, 256>>>(out, cyc, 1.0, 2.0);
    k<MODE, NACC><<<256, 256>>>(out, cyc, 1.0, 2.0);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-46s %6.1f ticks per matrix instruction\n", name, (double)h / (N_IT * (double)NACC));
}
int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * 256 * 256);
    hipMalloc(&cyc, 64);
    run<0, 32>("VGPR accumulators, 32 independent", out, cyc);
    run<1, 32>("AGPR accumulators, 32 independent", out, cyc);
    run<1, 96>("AGPR accumulators, 96 independent", out, cyc);
    run<2, 32>("VGPR acc + 1 v_fma_f64 each", out, cyc);
    run<3, 32>("AGPR acc + 1 v_fma_f64 each", out, cyc);
    run<4, 32>("VGPR acc + 1 ds_read_b64 each", out, cyc);
    run<5, 32>("AGPR acc + 1 ds_read_b64 each", out, cyc);
    return 0;
}
