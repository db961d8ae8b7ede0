// Microbenchmark (GPU box): issue cost of the direct kernel's spectral stores.  One workgroup of 4 waves per CU on
// only NWG CUs (HBM far from saturated), each wave issues the store pattern of a Legendre wave (8-byte stores,
// 16 B contiguous per (n, m), rows 496 B apart) or the same bytes as contiguous 16-byte stores, in a tight loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE> __global__ __launch_bounds__(256) void k(double *__restrict__ spec, int steps, long long *cyc)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        const long tile = (long)blockIdx.x + (long)(s & 63) * gridDim.x;
        if (MODE == 0) {
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
        } else {
            double2 *tb = reinterpret_cast<double2 *>(spec + tile * (2 * 992 * 2));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = (w * 8 + i) * 64 + lane;
                if (c < 1984) tb[c] = make_double2((double)lane, (double)i);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, double *spec, long long *cyc, int wgs)
{
    const int steps = 2000;
    k<MODE><<<wgs, 256>>>(spec, steps, cyc);
    k<MODE><<<wgs, 256>>>(spec, steps, cyc);
    CHECK(hipDeviceSynchronize());
    long long c0; CHECK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-12s %3d WGs: %7.0f ticks per 31.7 KB tile per CU  (%5.1f B/tick/CU)\n", name, wgs, (double)c0 / steps, 31744.0 * steps / c0);
}

int main()
{
    double *spec; long long *cyc;
    CHECK(hipMalloc(&spec, (size_t)64 * 256 * 1984 * 16)); CHECK(hipMalloc(&cyc, 8 * 256));
    for (int wgs : {8, 64, 256}) { run<0>("scattered", spec, cyc, wgs); run<1>("coalesced", spec, cyc, wgs); }
    return 0;
}
