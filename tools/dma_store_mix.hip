// Microbenchmark (GPU box): does the fused direct kernel's output pattern throttle its input stream?
// 1 workgroup per CU; wave 7 streams 72 KB per step by LDS-DMA (as tools/dma_bw.hip), waves 0-3 write the
// step's 31.7 KB spectral tile either as the MFMA D layout dictates (SCATTER: 8-byte stores, 16 contiguous
// bytes per (n, m), rows 496 B apart) or row-contiguous (16 B per lane, whole tile contiguous).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int STORE>   // 0 none, 1 scattered, 2 coalesced
__global__ __launch_bounds__(512, 2) void k(const double2 *__restrict__ g, double *__restrict__ spec, int steps, long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        const long tile = (long)blockIdx.x + (long)s * gridDim.x;
        if (w == 7) {
            const double2 *src = g + tile * (72 * 64) + lane;
#pragma unroll
            for (int i = 0; i < 72; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 64 * i),
                                                 (__attribute__((address_space(3))) void *)(lds + i * 128), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (w < 4 && STORE == 1) {
            const int blk = (lane >> 2) & 3, col = lane & 3, drow = lane >> 4, part = col & 1;
            double *tb = spec + tile * (2 * 992 * 2) + (col >> 1) * (992 * 2) + part;
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const int m = (sl & 1) ? 4 * sl + 3 - w : 4 * sl + w;
                if (m < 31) {
                    double *ob = tb + 2 * m;
                    const int n0 = 2 * (4 * blk + drow);
                    ob[2 * n0 * 31] = (double)lane;
                    ob[2 * (n0 + 1) * 31] = (double)sl;
                }
            }
        } else if (w < 4 && STORE == 2) {
            double2 *tb = reinterpret_cast<double2 *>(spec + tile * (2 * 992 * 2));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = (w * 8 + i) * 64 + lane;
                if (c < 1984) tb[c] = make_double2((double)lane, (double)i);
            }
        }
        __syncthreads();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int STORE> void run(const char *name, const double2 *g, double *spec, long long *cyc)
{
    const int wgs = 256, lds_bytes = 150 * 1024, steps = 100;
    CHECK(hipFuncSetAttribute((const void *)k<STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<STORE><<<wgs, 512, lds_bytes>>>(g, spec, steps, cyc);
    CHECK(hipEventRecord(e0));
    k<STORE><<<wgs, 512, lds_bytes>>>(g, spec, steps, cyc);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long c0; CHECK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
    const double rd = (double)steps * wgs * 72 * 1024, wr = STORE ? (double)steps * wgs * 1984 * 16 : 0.0;
    printf("%-22s read %7.1f GB/s + write %7.1f GB/s = %7.1f GB/s   %6.0f ticks/step\n", name, rd / ms / 1e6, wr / ms / 1e6, (rd + wr) / ms / 1e6, (double)c0 / steps);
}

int main()
{
    const long n_d2 = (long)100 * 256 * 72 * 64;
    double2 *g; double *spec; long long *cyc;
    CHECK(hipMalloc(&g, n_d2 * 16)); CHECK(hipMemset(g, 0, n_d2 * 16));
    CHECK(hipMalloc(&spec, (long)100 * 256 * 1984 * 16)); CHECK(hipMalloc(&cyc, 8 * 256));
    run<0>("dma only", g, spec, cyc);
    run<1>("dma + scattered", g, spec, cyc);
    run<2>("dma + coalesced", g, spec, cyc);
    run<1>("dma + scattered", g, spec, cyc);
    run<2>("dma + coalesced", g, spec, cyc);
    return 0;
}
