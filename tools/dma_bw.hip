// Microbenchmark (GPU box): per-CU read bandwidth of bursts of global_load_lds_dwordx4 ("LDS DMA") versus plain
// global_load_dwordx4, 1 workgroup per CU, NW issuing waves, TOT KB per step, with a wait + barrier per step
// (the access pattern of the fused direct kernel's input side).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int NW, int PERW>   // MODE 0: DMA, 1: VGPR loads (+ ds_write), PERW: 1 KB instructions per wave per step
__global__ __launch_bounds__(512, 2) void k(const double2 *__restrict__ g, long tile_d2, int steps, double *out, long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double acc = 0.0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        const double2 *src = g + ((long)blockIdx.x + (long)s * gridDim.x) * tile_d2 + (long)w * PERW * 64 + lane;
        if (w < NW) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < PERW; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 64 * i),
                                                     (__attribute__((address_space(3))) void *)(lds + (w * PERW + i) * 128), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (MODE == 2 || MODE == 3) {
                // one 768-byte row per instruction (LDS row stride 97 doubles); MODE 2: lanes >= 48 fetch the next
                // row's first 31 doubles (8-byte misaligned), MODE 3: lanes >= 48 masked off
                const char *t0 = reinterpret_cast<const char *>(g + ((long)blockIdx.x + (long)s * gridDim.x) * tile_d2) + (long)w * PERW * 768;
                const int lo = MODE == 2 ? 16 * lane - (lane >= 48 ? 8 : 0) : 16 * lane;
                if (MODE == 2 || lane < 48) {
#pragma unroll
                    for (int i = 0; i < PERW; ++i)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(t0 + i * 768 + lo),
                                                         (__attribute__((address_space(3))) void *)(lds + (w * PERW + i) * 97), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                double2 v[PERW];
#pragma unroll
                for (int i = 0; i < PERW; ++i) v[i] = src[64 * i];
#pragma unroll
                for (int i = 0; i < PERW; ++i) reinterpret_cast<double2 *>(lds)[(w * PERW + i) * 64 + lane] = v[i];
            }
        }
        __syncthreads();
        acc += lds[(threadIdx.x * 7 + s) & 4095];
        __syncthreads();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 1.2345) out[0] = acc;
}

template <int MODE, int NW, int PERW> void run(const char *name, const double2 *g, long n_d2, double *out, long long *cyc)
{
    const int wgs = 256, lds_bytes = 150 * 1024;
    const long tile_d2 = (MODE >= 2) ? (long)NW * PERW * 48 : (long)NW * PERW * 64;
    const int steps = (int)(n_d2 / (tile_d2 * wgs));
    CHECK(hipFuncSetAttribute((const void *)k<MODE, NW, PERW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<MODE, NW, PERW><<<wgs, 512, lds_bytes>>>(g, tile_d2, steps, out, cyc);
    CHECK(hipEventRecord(e0));
    k<MODE, NW, PERW><<<wgs, 512, lds_bytes>>>(g, tile_d2, steps, out, cyc);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long c0; CHECK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
    const double bytes = (double)steps * wgs * tile_d2 * 16;
    printf("%-28s NW=%d x %2d KB/step: %7.1f GB/s  %6.0f memtime ticks/step (%d steps)\n", name, NW, PERW, bytes / ms / 1e6, (double)c0 / steps, steps);
}

int main()
{
    const long n_d2 = (long)1 << 27;   // 2 GiB
    double2 *g; double *out; long long *cyc;
    CHECK(hipMalloc(&g, n_d2 * 16)); CHECK(hipMemset(g, 0, n_d2 * 16)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&cyc, 8 * 256));
    run<0, 1, 72>("dma, 1 wave", g, n_d2, out, cyc);
    run<0, 3, 24>("dma, 3 waves", g, n_d2, out, cyc);
    run<0, 6, 12>("dma, 6 waves", g, n_d2, out, cyc);
    run<0, 8, 9>("dma, 8 waves", g, n_d2, out, cyc);
    run<0, 8, 16>("dma, 8 waves 128 KB", g, n_d2, out, cyc);
    run<2, 1, 96>("dma rows+overlap, 1 wave", g, n_d2, out, cyc);
    run<3, 1, 96>("dma rows masked, 1 wave", g, n_d2, out, cyc);
    run<2, 3, 32>("dma rows+overlap, 3 waves", g, n_d2, out, cyc);
    run<3, 3, 32>("dma rows masked, 3 waves", g, n_d2, out, cyc);
    run<1, 6, 12>("vgpr, 6 waves", g, n_d2, out, cyc);
    run<1, 8, 9>("vgpr, 8 waves", g, n_d2, out, cyc);
    run<1, 8, 16>("vgpr, 8 waves 128 KB", g, n_d2, out, cyc);
    return 0;
}
