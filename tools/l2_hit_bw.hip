// Microbenchmark (GPU box): aggregate bandwidth when every CU streams the SAME small table (L2 hits), as the
// fused kernels' prologues do with the Legendre operand tables, versus private (HBM) data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k(const double2 *__restrict__ g, long wg_stride_d2, int table_d2, int reps, double *out)
{
    const double2 *src = g + (long)blockIdx.x * wg_stride_d2;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < table_d2; i += 256 * 8) {
            double2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(i + 256 * j) % table_d2];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y;
        }
    if (acc == 1.2345) out[0] = acc;
}

void run(const char *name, const double2 *g, long stride, int table_d2, int wgs, double *out)
{
    const int reps = 20;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<<<wgs, 256>>>(g, stride, table_d2, reps, out);
    CHECK(hipEventRecord(e0));
    k<<<wgs, 256>>>(g, stride, table_d2, reps, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s %4d WGs: %8.1f GB/s aggregate, %6.1f B/ns per WG\n", name, wgs, (double)wgs * table_d2 * 16.0 * reps / ms / 1e6,
           (double)table_d2 * 16.0 * reps / ms / 1e6);
}

int main()
{
    const int table_d2 = 9216;   // 147 KB
    double2 *g; double *out;
    CHECK(hipMalloc(&g, (size_t)1024 * table_d2 * 16)); CHECK(hipMemset(g, 0, (size_t)1024 * table_d2 * 16)); CHECK(hipMalloc(&out, 64));
    for (int wgs : {256, 1024}) {
        run("same 147 KB table for every WG (L2 hits)", g, 0, table_d2, wgs, out);
        run("4 copies, WG b reads copy (b/8)%4", g, 0, table_d2, wgs, out);   // placeholder label, same as above
        run("private 147 KB per WG (HBM / MALL)", g, table_d2, table_d2, wgs, out);
    }
    return 0;
}
