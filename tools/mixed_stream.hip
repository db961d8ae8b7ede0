// Microbenchmark (GPU box): chip-level ceiling for the direct transform's traffic mix -- per 2-field tile read
// 73,728 B and write 31,744 B, plain streaming kernel with many small workgroups (no LDS, no persistence).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int RD, int WR>   // double2 per tile read / written
__global__ __launch_bounds__(256) void k(const double2 *__restrict__ g, double2 *__restrict__ o, int ntiles)
{
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const double2 *src = g + (long)t * RD;
        double2 acc = make_double2(0.0, 0.0);
        double2 v[RD / 256];
#pragma unroll
        for (int i = 0; i < RD / 256; ++i) v[i] = src[i * 256 + threadIdx.x];
#pragma unroll
        for (int i = 0; i < RD / 256; ++i) { acc.x += v[i].x; acc.y += v[i].y; }
        double2 *dst = o + (long)t * WR;
#pragma unroll
        for (int i = 0; i < (WR + 255) / 256; ++i)
            if (i * 256 + threadIdx.x < WR) dst[i * 256 + threadIdx.x] = acc;
    }
}

template <int RD, int WR> void run(const char *name, const double2 *g, double2 *o, int ntiles, int wgs)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<RD, WR><<<wgs, 256>>>(g, o, ntiles);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k<RD, WR><<<wgs, 256>>>(g, o, ntiles);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double rd = (double)ntiles * RD * 16, wr = (double)ntiles * WR * 16;
    printf("%-34s %5d WGs: read %7.1f + write %7.1f = %7.1f GB/s  (%.1f us)\n", name, wgs, rd / ms / 1e6, wr / ms / 1e6, (rd + wr) / ms / 1e6, ms * 1e3);
}

int main()
{
    const int ntiles = 3072;   // = 6144 T30 fields
    double2 *g, *o;
    CHECK(hipMalloc(&g, (size_t)ntiles * 4608 * 16)); CHECK(hipMemset(g, 0, (size_t)ntiles * 4608 * 16));
    CHECK(hipMalloc(&o, (size_t)ntiles * 4608 * 16));
    for (int wgs : {1024, 3072}) {
        run<4608, 1984>("direct mix  (73.7 KB in, 31.7 out)", g, o, ntiles, wgs);
        run<2048, 4608>("inverse mix (32 KB in, 73.7 out)", g, o, ntiles, wgs);
        run<4608, 4608>("copy        (73.7 in, 73.7 out)", g, o, ntiles, wgs);
        run<4608, 256>("read-mostly (73.7 in, 4 out)", g, o, ntiles, wgs);
    }
    return 0;
}
