// Microbenchmark (GPU box): issue rate of the FP64 matrix instructions and FP64 VALU on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_bench.hip -o /tmp/mfma_bench && /tmp/mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define N_IT 2048

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, double a0, double b0)
{
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    d4 acc4[NACC];
    double acc1[NACC];
    for (int i = 0; i < NACC; ++i) { acc4[i] = (d4){0, 0, 0, 0}; acc1[i] = 0; }
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 0) acc4[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc4[i], 0, 0, 0);
            if (MODE == 1) acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
            if (MODE == 2) acc1[i] = __builtin_fma(a, b, acc1[i]);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += (MODE == 0) ? acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3] : acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// one wave per SIMD does MFMA, another does FP64 VALU: do they overlap?
__global__ __launch_bounds__(512) void mix(double *out, long long *cyc, double a0, double b0)
{
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    const int w = threadIdx.x >> 6;
    d4 acc4[4] = {(d4){0, 0, 0, 0}, (d4){0, 0, 0, 0}, (d4){0, 0, 0, 0}, (d4){0, 0, 0, 0}};
    double acc1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    if (w < 4) {
        for (int it = 0; it < N_IT; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc4[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc4[i], 0, 0, 0);
    } else {
        for (int it = 0; it < N_IT; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc1[i] = __builtin_fma(a, b, acc1[i]);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc4[0][0] + acc4[1][1] + acc4[2][2] + acc4[3][3];
    for (int i = 0; i < 8; ++i) s += acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}

int main()
{
    double *out; long long *cyc, h[8];
    hipMalloc(&out, sizeof(double) * 1024 * 1024);
    hipMalloc(&cyc, 64);
#define RUN(MODE, NACC, WAVES, name, ops)                                                            \
    do {                                                                                             \
        hipLaunchKernelGGL((k<MODE, NACC>), dim3(1), dim3(64 * WAVES), 0, 0, out, cyc, 1.0, 2.0);    \
        hipLaunchKernelGGL((k<MODE, NACC>), dim3(1), dim3(64 * WAVES), 0, 0, out, cyc, 1.0, 2.0);    \
        hipDeviceSynchronize();                                                                      \
        hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);                                                 \
        printf("%-34s waves/CU=%d acc=%d : %.1f cycles per instruction per wave\n", name, WAVES, NACC, \
               (double)h[0] / (N_IT * NACC));                                                        \
    } while (0)
    RUN(0, 1, 4, "v_mfma_f64_16x16x4 (dependent)", 0);
    RUN(0, 4, 4, "v_mfma_f64_16x16x4", 0);
    RUN(0, 4, 8, "v_mfma_f64_16x16x4", 0);
    RUN(1, 1, 4, "v_mfma_f64_4x4x4_4b (dependent)", 0);
    RUN(1, 4, 4, "v_mfma_f64_4x4x4_4b", 0);
    RUN(1, 8, 4, "v_mfma_f64_4x4x4_4b", 0);
    RUN(1, 8, 8, "v_mfma_f64_4x4x4_4b", 0);
    RUN(2, 1, 4, "v_fma_f64 (dependent)", 0);
    RUN(2, 8, 4, "v_fma_f64", 0);
    RUN(2, 8, 8, "v_fma_f64", 0);
    // absolute throughput: whole chip, wall clock (HIP events)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define WALL(MODE, NACC, WAVES, name, flop_per_instr)                                                          \
    do {                                                                                                        \
        const int nblk = 256 * 8;                                                                               \
        hipLaunchKernelGGL((k<MODE, NACC>), dim3(nblk), dim3(64 * WAVES), 0, 0, out, cyc, 1.0, 2.0);            \
        hipEventRecord(e0, 0);                                                                                  \
        hipLaunchKernelGGL((k<MODE, NACC>), dim3(nblk), dim3(64 * WAVES), 0, 0, out, cyc, 1.0, 2.0);            \
        hipEventRecord(e1, 0); hipEventSynchronize(e1);                                                         \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                             \
        double fl = (double)nblk * WAVES * N_IT * NACC * flop_per_instr;                                        \
        printf("%-26s %d waves/block: %.3f ms -> %.1f TFLOP/s\n", name, WAVES, ms, fl / ms / 1e9);             \
    } while (0)
    WALL(0, 4, 4, "mfma_f64_16x16x4", 2048.0);
    WALL(0, 4, 8, "mfma_f64_16x16x4", 2048.0);
    WALL(1, 8, 4, "mfma_f64_4x4x4_4b", 512.0);
    WALL(1, 8, 8, "mfma_f64_4x4x4_4b", 512.0);
    WALL(2, 8, 4, "v_fma_f64", 128.0);
    WALL(2, 8, 8, "v_fma_f64", 128.0);
    hipLaunchKernelGGL(mix, dim3(1), dim3(512), 0, 0, out, cyc, 1.0, 2.0);
    hipLaunchKernelGGL(mix, dim3(1), dim3(512), 0, 0, out, cyc, 1.0, 2.0);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("mix (waves 0-3 mfma 16x16x4, 4-7 v_fma_f64): mfma %.1f cyc/instr, fma %.1f cyc/instr\n",
           (double)h[0] / (N_IT * 4), (double)h[4] / (N_IT * 8));
    return 0;
}
