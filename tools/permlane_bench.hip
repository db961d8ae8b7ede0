// GPU box: issue cost of v_permlane32_swap / v_permlane16_swap (independent and dependent chains), next to v_add_f64 and
// ds_bpermute.  hipcc --offload-arch=gfx950 -O3 tools/permlane_bench.hip -o /tmp/permlane_bench && /tmp/permlane_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long *out, int mode)
{
    unsigned a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 3 + i; b[i] = threadIdx.x * 7 + i; }
    double d[16];
    for (int i = 0; i < 16; ++i) d[i] = threadIdx.x + i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { auto r = __builtin_amdgcn_permlane32_swap(a[i], b[i], false, false); a[i] = r[0]; b[i] = r[1]; }
        } else if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { auto r = __builtin_amdgcn_permlane16_swap(a[i], b[i], false, false); a[i] = r[0]; b[i] = r[1]; }
        } else if (mode == 2) {   // dependent chain
#pragma unroll
            for (int i = 0; i < 16; ++i) { auto r = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false); a[0] = r[1]; b[0] = r[0] + 1; }
        } else if (mode == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = d[i] + 1.5;
        } else if (mode == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 32) * 4, a[i]);
        } else if (mode == 5) {   // alternate swap32 -> swap16 on the same registers (the transpose pattern)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                auto r = __builtin_amdgcn_permlane32_swap(a[i], b[i], false, false);
                auto q = __builtin_amdgcn_permlane16_swap(r[0], r[1], false, false);
                a[i] = q[0]; b[i] = q[1];
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0; double sd = 0;
    for (int i = 0; i < 16; ++i) { s += a[i] + b[i]; sd += d[i]; }
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (s == 0x12345 && sd == 1.25) out[1] = s;
}
int main()
{
    long long *o; hipMalloc(&o, 4096);
    const char *names[] = {"permlane32_swap indep", "permlane16_swap indep", "permlane32_swap dependent", "v_add_f64 indep", "ds_bpermute indep", "swap32->swap16 pairs"};
    for (int waves = 1; waves <= 2; ++waves)
        for (int m = 0; m < 6; ++m) {
            k<<<1, 64 * waves * 4>>>(o, m); hipDeviceSynchronize();
            k<<<1, 64 * waves * 4>>>(o, m); hipDeviceSynchronize();
            long long t; hipMemcpy(&t, o, 8, hipMemcpyDeviceToHost);
            printf("%d wave(s)/SIMD  %-28s %6.2f ticks/instr\n", waves, names[m], (double)t / (64 * 16));
        }
    return 0;
}
