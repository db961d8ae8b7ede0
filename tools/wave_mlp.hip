// Microbenchmark (GPU box): how many 1 KB vector loads (global_load_dwordx4, 64 lanes) can ONE wave keep in
// flight?  One wave per CU issues NL loads back to back, timestamps the end of issue and the arrival of the data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NL, int NW> __global__ __launch_bounds__(512) void k(const double2 *__restrict__ g, int steps, double *out, long long *cyc)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double acc = 0.0;
    long long issue = 0, total = 0;
    for (int s = 0; s < steps; ++s) {
        const double2 *src = g + (((long)blockIdx.x + (long)s * gridDim.x) * NW + w) * (NL * 64) + lane;
        double2 v[NL];
        long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < NL; ++i) v[i] = src[64 * i];
        asm volatile("" ::: "memory");
        long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long t2 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < NL; ++i) acc += v[i].x + v[i].y;
        issue += t1 - t0; total += t2 - t0;
    }
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = issue; cyc[2 * blockIdx.x + 1] = total; }
    if (acc == 1.2345) out[0] = acc;
}

template <int NL, int NW> void run(const double2 *g, double *out, long long *cyc, int wgs)
{
    const int steps = 50;
    k<NL, NW><<<wgs, 64 * NW>>>(g, steps, out, cyc);
    k<NL, NW><<<wgs, 64 * NW>>>(g, steps, out, cyc);
    CHECK(hipDeviceSynchronize());
    long long c[2]; CHECK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
    printf("%d wave(s) x %2d loads, %3d WGs: issue %6.0f ticks, data back %6.0f ticks  -> %5.1f B/tick/CU\n", NW, NL, wgs, (double)c[0] / steps,
           (double)c[1] / steps, (double)NW * NL * 1024.0 * steps / c[1]);
}

int main()
{
    const size_t n_d2 = (size_t)50 * 256 * 8 * 48 * 64;
    double2 *g; double *out; long long *cyc;
    CHECK(hipMalloc(&g, n_d2 * 16)); CHECK(hipMemset(g, 0, n_d2 * 16)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&cyc, 16 * 256));
    for (int wgs : {8, 256}) {
        run<8, 1>(g, out, cyc, wgs); run<16, 1>(g, out, cyc, wgs); run<24, 1>(g, out, cyc, wgs); run<48, 1>(g, out, cyc, wgs);
        run<24, 3>(g, out, cyc, wgs); run<12, 6>(g, out, cyc, wgs); run<9, 8>(g, out, cyc, wgs);
    }
    return 0;
}
