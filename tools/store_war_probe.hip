// GPU box: is the data of a 128-bit buffer store safe from a LATER LDS load that targets the same registers?
//
// The sequence the compiler emits in the T63 inverse kernel's copy-out (ROCm 7.2, csrc/spdy_fused_t63.inc phase B):
//      buffer_store_dwordx4 v[106:109], v118, s[24:27], s1 offen offset:768 nt
//      ds_read2_b64         v[104:107], v99 offset1:1           <- overwrites half of the store's data registers
// In program order the store reads its data first.  This probe runs exactly that pair in "worker" waves -- LDS holds pattern A
// at one address and pattern B at another; a worker loads A, stores it, and immediately loads B into the same registers -- and
// checks global memory for B values, with "loader" waves streaming buffer loads through the same CU: placement 0 = a loader and
// a worker on every SIMD (hardware waves w and w + 4 share a SIMD), 1 = roles by SIMD, 2 = no loaders.
//   hipcc --offload-arch=gfx950 -O3 tools/store_war_probe.hip -o /tmp/warp && /tmp/warp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define SLOTS 1024

__global__ __launch_bounds__(512) void probe(int placement, int iters, int nt, u4 *out, const u4 *table, unsigned long long *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // pattern A at dwords [0, 256), pattern B at dwords [256, 512): lane l's four dwords
    for (int e = threadIdx.x; e < 512; e += 512) lds[e] = (e < 256 ? 0xA0000000u : 0xB0000000u) | (unsigned)(e & 255);
    __syncthreads();
    const bool loader = placement == 0 ? (w < 4) : placement == 1 ? ((w & 2) == 0) : false;
    const bool worker = placement == 0 ? (w >= 4) : ((w & 2) != 0);
    if (loader) {
        unsigned long long acc = 0;
        const u4 *src = table + (size_t)blockIdx.x * 4096 + lane;
        for (int i = 0; i < iters * 4; ++i) {
            u4 a = src[((i * 8 + 0) & 63) * 64], b = src[((i * 8 + 1) & 63) * 64], c = src[((i * 8 + 2) & 63) * 64], d = src[((i * 8 + 3) & 63) * 64];
            u4 e = src[((i * 8 + 4) & 63) * 64], f = src[((i * 8 + 5) & 63) * 64], g = src[((i * 8 + 6) & 63) * 64], h = src[((i * 8 + 7) & 63) * 64];
            acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
        }
        sink[blockIdx.x * 512 + threadIdx.x] = acc;
    } else if (worker) {
        const int wk = placement == 0 ? w - 4 : ((w & 1) | ((w >> 2) << 1));            // 0..3
        // destination: per workgroup, worker and iteration (every store has its own 1 KB: iters <= SLOTS)
        u4 *dst = out + ((size_t)blockIdx.x * 4 + wk) * SLOTS * 64;
        const unsigned ldsA = 16u * lane, ldsB = 1024u + 16u * lane;
        const unsigned long long a = (unsigned long long)dst;
        // buffer resource: base, stride 0, num_records = max, flags as the product uses
        u4 rsrc;
        rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)a);
        rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
        rsrc.z = 0x7fffffffu;
        rsrc.w = 0x00020000u;
        const unsigned voff = 16u * lane;
        for (int i = 0; i < iters / 4; ++i) {
            unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((4 * i) & (SLOTS - 1)) * 1024u);
            u4 d0, d1, d2, d3;
            // four register sets: load pattern A into all, then store each and immediately reload it with pattern B (the burst of
            // a copy-out: several 1 KB stores queued behind each other, every one followed by a load into its data registers)
            if (nt >= 2) {
                // the OTHER overwrite the compiler emits: a VALU write to one data dword in the instruction after the store.
                // nt = 2: soffset in an SGPR (the compiler inserts no wait state for this form); nt = 3: the offset in the
                // VGPR and soffset = 0 (the form for which it inserts one s_nop -- written out here)
                const unsigned vo2 = voff + soff;
                if (nt == 2)
                    asm volatile("ds_read_b128 v[20:23], %0\n\ts_waitcnt lgkmcnt(0)\n\t"
                                 "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n\tv_add_u32 v22, 0xB0000000, %1\n\t"
                                 "ds_read_b128 v[24:27], %0\n\ts_waitcnt lgkmcnt(0)\n\t"
                                 "buffer_store_dwordx4 v[24:27], %1, %2, %3 offen offset:1024\n\tv_add_u32 v24, 0xB0000000, %1\n\t"
                                 "ds_read_b128 v[20:23], %0\n\ts_waitcnt lgkmcnt(0)\n\t"
                                 "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen offset:2048\n\tv_add_u32 v23, 0xB0000000, %1\n\t"
                                 "ds_read_b128 v[24:27], %0\n\ts_waitcnt lgkmcnt(0)\n\t"
                                 "buffer_store_dwordx4 v[24:27], %1, %2, %3 offen offset:3072\n\tv_add_u32 v25, 0xB0000000, %1"
                                 :: "v"(ldsA), "v"(voff), "s"(rsrc), "s"(soff) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory");
                else
                    asm volatile("ds_read_b128 v[20:23], %0\n\ts_waitcnt lgkmcnt(0)\n\t"
                                 "buffer_store_dwordx4 v[20:23], %1, %2, 0 offen\n\ts_nop 0\n\tv_add_u32 v22, 0xB0000000, %1\n\t"
                                 "ds_read_b128 v[24:27], %0\n\ts_waitcnt lgkmcnt(0)\n\t"
                                 "buffer_store_dwordx4 v[24:27], %1, %2, 0 offen offset:1024\n\ts_nop 0\n\tv_add_u32 v24, 0xB0000000, %1\n\t"
                                 "ds_read_b128 v[20:23], %0\n\ts_waitcnt lgkmcnt(0)\n\t"
                                 "buffer_store_dwordx4 v[20:23], %1, %2, 0 offen offset:2048\n\ts_nop 0\n\tv_add_u32 v23, 0xB0000000, %1\n\t"
                                 "ds_read_b128 v[24:27], %0\n\ts_waitcnt lgkmcnt(0)\n\t"
                                 "buffer_store_dwordx4 v[24:27], %1, %2, 0 offen offset:3072\n\ts_nop 0\n\tv_add_u32 v25, 0xB0000000, %1"
                                 :: "v"(ldsA), "v"(vo2), "s"(rsrc) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory");
                d0 = d1 = d2 = d3 = u4{0, 0, 0, 0};
            } else if (nt)
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4\n\tds_read_b128 %3, %4\n\ts_waitcnt lgkmcnt(0)\n\t"
                             "buffer_store_dwordx4 %0, %5, %6, %7 offen nt\n\tds_read_b128 %0, %8\n\t"
                             "buffer_store_dwordx4 %1, %5, %6, %7 offen offset:1024 nt\n\tds_read_b128 %1, %8\n\t"
                             "buffer_store_dwordx4 %2, %5, %6, %7 offen offset:2048 nt\n\tds_read_b128 %2, %8\n\t"
                             "buffer_store_dwordx4 %3, %5, %6, %7 offen offset:3072 nt\n\tds_read_b128 %3, %8\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(ldsA), "v"(voff), "s"(rsrc), "s"(soff), "v"(ldsB) : "memory");
            else
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4\n\tds_read_b128 %3, %4\n\ts_waitcnt lgkmcnt(0)\n\t"
                             "buffer_store_dwordx4 %0, %5, %6, %7 offen\n\tds_read_b128 %0, %8\n\t"
                             "buffer_store_dwordx4 %1, %5, %6, %7 offen offset:1024\n\tds_read_b128 %1, %8\n\t"
                             "buffer_store_dwordx4 %2, %5, %6, %7 offen offset:2048\n\tds_read_b128 %2, %8\n\t"
                             "buffer_store_dwordx4 %3, %5, %6, %7 offen offset:3072\n\tds_read_b128 %3, %8\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(ldsA), "v"(voff), "s"(rsrc), "s"(soff), "v"(ldsB) : "memory");
            if (d0.x == 0x12345678u) sink[0] = d0.y + d1.y + d2.y + d3.y;            // keep them alive
        }
    }
}

int main(int argc, char **argv)
{
    const int iters = SLOTS, lds_bytes = 150800, nwg = 256;
    (void)argc; (void)argv;
    u4 *out, *table; unsigned long long *sink;
    const size_t nout = (size_t)nwg * 4 * SLOTS * 64;
    hipMalloc(&out, nout * sizeof(u4)); hipMalloc(&table, (size_t)nwg * 4096 * sizeof(u4) + 65536); hipMalloc(&sink, nwg * 512 * 8);
    hipMemset(table, 1, (size_t)nwg * 4096 * sizeof(u4));
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    std::vector<u4> h(nout);
    for (int nt = 0; nt < 4; ++nt)
        for (int placement = 0; placement < 3; ++placement)
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(out, 0, nout * sizeof(u4));
                hipLaunchKernelGGL(probe, dim3(nwg), dim3(512), lds_bytes, 0, placement, iters, nt, out, table, sink);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), out, nout * sizeof(u4), hipMemcpyDeviceToHost);
                size_t bad = 0, badB = 0, lanes[64] = {0};
                for (size_t i = 0; i < nout; ++i) {
                    const unsigned lane = (unsigned)(i & 63);
                    const unsigned e[4] = {h[i].x, h[i].y, h[i].z, h[i].w};
                    for (int j = 0; j < 4; ++j) {
                        const unsigned want = 0xA0000000u | (4 * lane + j);
                        if (e[j] != want) { ++bad; if ((e[j] >> 28) == 0xB) ++badB; ++lanes[lane]; }
                    }
                }
                printf("%s, placement %d (%s): %zu wrong dwords of %zu (%zu hold the LATER load's pattern) %s",
                       nt == 0 ? "store + LDS load, plain stores" : nt == 1 ? "store + LDS load, nt stores" : nt == 2 ? "store (SGPR soffset) + VALU write" : "store (soffset 0) + s_nop 0 + VALU write", placement, placement == 0 ? "loader + worker on every SIMD" : placement == 1 ? "roles by SIMD" : "no loaders",
                       bad, nout * 4, badB, hipGetErrorString(hipGetLastError()));
                if (bad) { printf("  lanes:"); for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d", l); }
                printf("\n");
            }
    return 0;
}
