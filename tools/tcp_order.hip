// Microbenchmark (GPU box): does a wave's L2-hit stream wait behind ANOTHER wave's HBM misses in the CU's vector memory path?
//   hipcc --offload-arch=gfx950 -O3 tools/tcp_order.hip -o /tmp/tcp_order && /tmp/tcp_order
// Block = 2..4 waves on one CU (one block per CU, 256 blocks).  Wave 0 streams an L2-resident 0.9 MB table with a 12-deep ring
// of 1 KB loads (what a Legendre wave of the fused T63 kernels does with its A operands) and reports ticks per 38-load pass;
// the other waves are idle (mode 0), stream distinct HBM data in bursts of 18 x 1 KB loads (mode 1: plain, mode 2: nt),
// or store 1 KB lines (mode 3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define PASSES 200
template <int MODE>
__global__ __launch_bounds__(256) void k(const double2 *__restrict__ tab, const double2 *__restrict__ big, double2 *__restrict__ sink,
                                         long long *ticks, long big_elems)
{
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double2 acc = make_double2(0, 0);
    if (w == 0) {
        const double2 *p = tab + lane;
        double2 ring[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) ring[i] = p[i * 64];
        const long long t0 = __builtin_amdgcn_s_memtime();
        for (int pass = 0; pass < PASSES; ++pass) {
            const double2 *q = tab + ((pass * 38 * 64) % (57000 - 64 * 64)) + lane;
#pragma unroll
            for (int s = 0; s < 38; ++s) {
                const double2 a = ring[s % 12];
                ring[s % 12] = q[(s + 12) * 64];
                acc.x += a.x; acc.y += a.y;
            }
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) ticks[blockIdx.x] = (t1 - t0) / PASSES;
    } else if (MODE != 0) {
        // distinct HBM lines per block and wave; bursts of 18 KB per wave
        const long stride = 64;                         // double2 per 1 KB wave-load
        const long off = (((long)blockIdx.x * 4 + w) * 18 * stride * 1024) % (big_elems - 18 * stride * 1024);   // 900 bursts stay inside
        for (int it = 0; it < 900; ++it) {
            const double2 *q = big + off + (long)it * 18 * stride + lane;
            if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 18; ++i) sink[off + (long)it * 18 * stride + i * stride + lane] = make_double2(it, i);
                __builtin_amdgcn_s_sleep(20);
            } else {
                double2 v[18];
#pragma unroll
                for (int i = 0; i < 18; ++i)
                    v[i] = MODE == 2 ? make_double2(__builtin_nontemporal_load(&q[i * stride].x), __builtin_nontemporal_load(&q[i * stride].y))
                                     : q[i * stride];
#pragma unroll
                for (int i = 0; i < 18; ++i) { acc.x += v[i].x; acc.y += v[i].y; }
            }
        }
    }
    if (acc.x == 12345.678) sink[0] = acc;
}

int main()
{
    const long big_elems = (long)1 << 27;   // 2 GB of double2
    double2 *tab, *big, *sink; long long *ticks;
    hipMalloc(&tab, 57000 * sizeof(double2) + (1 << 20));
    hipMalloc(&big, big_elems * sizeof(double2));
    if (hipMalloc(&sink, big_elems * sizeof(double2)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&ticks, 256 * sizeof(long long));
    hipMemset(tab, 0, 57000 * sizeof(double2) + (1 << 20));
    hipMemset(big, 0, big_elems * sizeof(double2));
    std::vector<long long> h(256);
#define RUN(MODE, WAVES, name)                                                                          \
    for (int rep = 0; rep < 2; ++rep) {                                                                 \
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(64 * WAVES), 0, 0, tab, big, sink, ticks, big_elems); \
        hipDeviceSynchronize();                                                                         \
        hipMemcpy(h.data(), ticks, 256 * 8, hipMemcpyDeviceToHost);                                     \
        double s = 0; long long mx = 0; for (auto v : h) { s += v; if (v > mx) mx = v; }                \
        if (rep) printf("%-44s waves=%d : %7.0f ticks per 38 KB pass (max %lld)\n", name, WAVES, s / 256, mx); \
    }
    RUN(0, 1, "table stream alone");
    RUN(0, 4, "table stream + 3 idle waves");
    RUN(1, 2, "+ 1 wave streaming HBM (plain loads)");
    RUN(1, 4, "+ 3 waves streaming HBM (plain loads)");
    RUN(2, 2, "+ 1 wave streaming HBM (nt loads)");
    RUN(2, 4, "+ 3 waves streaming HBM (nt loads)");
    RUN(3, 2, "+ 1 wave storing 1 KB lines");
    RUN(3, 4, "+ 3 waves storing 1 KB lines");
    return 0;
}
