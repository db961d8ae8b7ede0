// GPU box: does an FP64-MFMA wave disturb the lane swaps / FP64 vector ops of the wave that shares its SIMD?
// (Behind the T63_ROLE_MIX observation, DESIGN s4.3.)  One 512-thread workgroup per CU with the fused T63 kernels' LDS footprint.
// "worker" waves run self-checking sequences (every result is known exactly: small integers held in doubles):
//    test 0: 4x4 lane transposes with v_permlane32_swap / v_permlane16_swap, forward then backward = identity
//    test 1: the same with FP64 multiply-adds between the swaps
//    test 2: FP64 multiply-adds under 16-lane-row exec masks (if (h == k)), checked against integer arithmetic
// "matrix" waves run back-to-back v_mfma_f64_4x4x4_4b.  placement 0: matrix waves 0-3, workers 4-7 (every SIMD hosts one of
// each: hardware waves w and w + 4 share a SIMD); placement 1: matrix waves {0,1,4,5}, workers {2,3,6,7} (roles by SIMD);
// placement 2: no matrix waves at all.   hipcc --offload-arch=gfx950 -O3 tools/simd_share_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void swap32(double &x, double &y)
{
    const unsigned long long a = __double_as_longlong(x), b = __double_as_longlong(y);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(a >> 32), (unsigned)(b >> 32), false, false);
    x = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]);
    y = __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ void swap16(double &x, double &y)
{
    const unsigned long long a = __double_as_longlong(x), b = __double_as_longlong(y);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(a >> 32), (unsigned)(b >> 32), false, false);
    x = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]);
    y = __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}

template <int TEST>
__global__ __launch_bounds__(512) void probe(int placement, int iters, unsigned long long *errs, double *sink)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 4;
    if (threadIdx.x == 0) lds[0] = 0.0;
    const bool matrix = placement == 0 ? (w < 4) : placement == 1 ? ((w & 2) == 0) : false;
    const bool worker = placement == 0 ? (w >= 4) : ((w & 2) != 0);
    if (matrix) {
        double a = 1.0 + 0.001 * lane, b = 1.0000001, c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
        for (int i = 0; i < iters * 40; ++i) {
            c0 = mfma4(a, b, c0); c1 = mfma4(b, a, c1); c2 = mfma4(a, a, c2); c3 = mfma4(b, b, c3); c4 = mfma4(a, b, c4); c5 = mfma4(b, a, c5);
        }
        sink[blockIdx.x * 512 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5;
    } else if (worker) {
        unsigned long long bad = 0;
        double x[48];
        for (int k = 0; k < 48; ++k) x[k] = (double)(lane * 64 + k);
        for (int i = 0; i < iters; ++i) {
            if (TEST == 0 || TEST == 1) {
#pragma unroll
                for (int r = 0; r < 12; ++r) {
                    double r0 = x[4 * r], r1 = x[4 * r + 1], r2 = x[4 * r + 2], r3 = x[4 * r + 3];
                    if (TEST == 1) { r0 = r0 * 3.0 + 1.0; r1 = r1 * 3.0 + 1.0; r2 = r2 * 3.0 + 1.0; r3 = r3 * 3.0 + 1.0; }
                    swap32(r0, r2); swap32(r1, r3); swap16(r0, r1); swap16(r2, r3);
                    if (TEST == 1) { r0 = r0 + 2.0; r1 = r1 + 2.0; r2 = r2 + 2.0; r3 = r3 + 2.0; }
                    swap16(r2, r3); swap16(r0, r1); swap32(r1, r3); swap32(r0, r2);
                    if (TEST == 1) { r0 = (r0 - 3.0) * (1.0 / 3.0); r1 = (r1 - 3.0) * (1.0 / 3.0); r2 = (r2 - 3.0) * (1.0 / 3.0); r3 = (r3 - 3.0) * (1.0 / 3.0);
                                     r0 = __builtin_rint(r0); r1 = __builtin_rint(r1); r2 = __builtin_rint(r2); r3 = __builtin_rint(r3); }
                    x[4 * r] = r0; x[4 * r + 1] = r1; x[4 * r + 2] = r2; x[4 * r + 3] = r3;
                }
#pragma unroll
                for (int k = 0; k < 48; ++k) bad += x[k] != (double)(lane * 64 + k);
            } else {
                int salt = i & 7;
                asm volatile("" : "+v"(salt));
#pragma unroll
                for (int k = 0; k < 48; ++k) {
                    double y;
                    if (h == (k & 3)) y = x[k] * 3.0 + (double)salt;          // 16-lane-row exec masks
                    else y = x[k] * 5.0 - (double)salt;
                    const long long want = h == (k & 3) ? (long long)(lane * 64 + k) * 3 + salt : (long long)(lane * 64 + k) * 5 - salt;
                    bad += (long long)y != want;
                }
            }
        }
        if (bad) atomicAdd(errs, bad);
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, lds_bytes = 150800;
    unsigned long long *errs; double *sink;
    hipMalloc(&errs, 8); hipMalloc(&sink, 256 * 512 * 8);
    const void *k[3] = {(const void *)probe<0>, (const void *)probe<1>, (const void *)probe<2>};
    for (int t = 0; t < 3; ++t) hipFuncSetAttribute(k[t], hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    for (int test = 0; test < 3; ++test)
        for (int placement = 0; placement < 3; ++placement)
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(errs, 0, 8);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (test == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), lds_bytes, 0, placement, iters, errs, sink);
                else if (test == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), lds_bytes, 0, placement, iters, errs, sink);
                else hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), lds_bytes, 0, placement, iters, errs, sink);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                unsigned long long h = 0; float ms = 0;
                hipMemcpy(&h, errs, 8, hipMemcpyDeviceToHost); hipEventElapsedTime(&ms, e0, e1);
                printf("test %d placement %d (%s): %llu wrong values of %.3g checked  (%.1f ms) %s\n", test, placement,
                       placement == 0 ? "matrix + worker on every SIMD" : placement == 1 ? "roles by SIMD" : "no matrix waves", h,
                       256.0 * 4 * 64 * 48 * iters, ms, hipGetErrorString(hipGetLastError()));
            }
    return 0;
}
