// GPU box: how many wait states does gfx950 need between a 128-bit vector-memory store and a VALU write to one of its data
// registers?  (Root cause of the T63_ROLE_MIX observation, DESIGN s4.3: the compiler -- ROCm 7.2 -- inserts none for a buffer
// store with an SGPR soffset and one for the other forms.)
// Every "worker" wave repeats:  v[20:23] <- pattern A (LDS);  store v[20:23];  [s_nop N];  v_add_u32 v22 <- pattern B;  each store to
// its own 1 KB of global memory, checked on the host for B.  All eight waves of a 512-thread workgroup are workers (two per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/store_valu_hazard.hip -o /tmp/svh && /tmp/svh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define SLOTS 512
#define STR2(x) #x
#define STR(x) STR2(x)
#define LOADA "ds_read_b128 v[20:23], %0\n\ts_waitcnt lgkmcnt(0)\n\t"
#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory"

// KIND 0: buffer store, SGPR soffset; 1: buffer store, soffset 0 (offset in the VGPR); 2: global store.  NOPS < 0: no s_nop at all
template <int KIND, int NOPS>
__global__ __launch_bounds__(512) void probe(u4 *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = (threadIdx.x < 256 ? 0xA0000000u : 0xB0000000u) | (unsigned)(threadIdx.x & 255);
    __syncthreads();
    u4 *dst = out + ((size_t)blockIdx.x * 8 + w) * SLOTS * 64;
    const unsigned ldsA = 16u * lane, voff = 16u * lane;
    const unsigned long long a = (unsigned long long)dst;
    u4 rsrc;
    rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
    rsrc.z = 0x7fffffffu; rsrc.w = 0x00020000u;
    for (int i = 0; i < SLOTS; ++i) {
        const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)i * 1024u);
        const unsigned vo2 = voff + soff;
        u4 *gp = dst + (size_t)i * 64 + lane;
#define NOPSTR(n) "s_nop " STR(n) "\n\t"
#define BODY(storeasm, ...)                                                                                                   \
        if (NOPS < 0) asm volatile(LOADA storeasm "\n\tv_add_u32 v22, 0xB0000000, %1" :: __VA_ARGS__ : CLOB);                 \
        else if (NOPS == 0) asm volatile(LOADA storeasm "\n\t" NOPSTR(0) "v_add_u32 v22, 0xB0000000, %1" :: __VA_ARGS__ : CLOB); \
        else if (NOPS == 1) asm volatile(LOADA storeasm "\n\t" NOPSTR(1) "v_add_u32 v22, 0xB0000000, %1" :: __VA_ARGS__ : CLOB); \
        else if (NOPS == 2) asm volatile(LOADA storeasm "\n\t" NOPSTR(2) "v_add_u32 v22, 0xB0000000, %1" :: __VA_ARGS__ : CLOB); \
        else if (NOPS == 3) asm volatile(LOADA storeasm "\n\t" NOPSTR(3) "v_add_u32 v22, 0xB0000000, %1" :: __VA_ARGS__ : CLOB); \
        else if (NOPS == 5) asm volatile(LOADA storeasm "\n\t" NOPSTR(5) "v_add_u32 v22, 0xB0000000, %1" :: __VA_ARGS__ : CLOB); \
        else if (NOPS == 7) asm volatile(LOADA storeasm "\n\t" NOPSTR(7) "v_add_u32 v22, 0xB0000000, %1" :: __VA_ARGS__ : CLOB); \
        else asm volatile(LOADA storeasm "\n\t" NOPSTR(15) "v_add_u32 v22, 0xB0000000, %1" :: __VA_ARGS__ : CLOB);
        if (KIND == 4) {
            // the LDS analogue: ds_write_b128 of v[20:23], N wait states, VALU write to v22; the cell is read back and copied out
            const unsigned ldsD = 4096u + 16u * threadIdx.x;
#define LDSBODY(nopstr) asm volatile(LOADA "ds_write_b128 %1, v[20:23]\n\t" nopstr "v_add_u32 v22, 0xB0000000, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(ldsA), "v"(ldsD) : "v20", "v21", "v22", "v23", "memory")
            if (NOPS < 0) LDSBODY(""); else if (NOPS == 0) LDSBODY("s_nop 0\n\t"); else if (NOPS == 1) LDSBODY("s_nop 1\n\t"); else LDSBODY("s_nop 3\n\t");
            const u4 back = *reinterpret_cast<const u4 *>(reinterpret_cast<const char *>(lds) + ldsD);
            gp[0] = back;
        } else
        if (KIND == 3) {
            // burst: four 1 KB stores behind each other (two register sets), each followed by N wait states and a VALU write to
            // one of its data dwords -- the copy-out of a transform kernel
            if ((i & 3) == 0) {
#define GRP(regs, r, off, nopstr) "ds_read_b128 " regs ", %0\n\ts_waitcnt lgkmcnt(0)\n\tbuffer_store_dwordx4 " regs ", %1, %2, %3 offen offset:" off "\n\t" nopstr "v_add_u32 " r ", 0xB0000000, %1\n\t"
#define BURST(nopstr) asm volatile(GRP("v[20:23]", "v22", "0", nopstr) GRP("v[24:27]", "v24", "1024", nopstr) GRP("v[20:23]", "v23", "2048", nopstr) GRP("v[24:27]", "v25", "3072", nopstr) \
                                   :: "v"(ldsA), "v"(voff), "s"(rsrc), "s"(soff) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory")
                if (NOPS < 0) BURST("");
                else if (NOPS == 0) BURST("s_nop 0\n\t");
                else if (NOPS == 1) BURST("s_nop 1\n\t");
                else if (NOPS == 2) BURST("s_nop 2\n\t");
                else if (NOPS == 3) BURST("s_nop 3\n\t");
                else if (NOPS == 5) BURST("s_nop 5\n\t");
                else if (NOPS == 7) BURST("s_nop 7\n\t");
                else BURST("s_nop 15\n\t");
            }
        } else
        if (KIND == 0) { BODY("buffer_store_dwordx4 v[20:23], %1, %2, %3 offen", "v"(ldsA), "v"(voff), "s"(rsrc), "s"(soff)) }
        else if (KIND == 1) { BODY("buffer_store_dwordx4 v[20:23], %1, %2, 0 offen", "v"(ldsA), "v"(vo2), "s"(rsrc)) }
        else { BODY("global_store_dwordx4 %2, v[20:23], off", "v"(ldsA), "v"(voff), "v"(gp)) }
    }
}

template <int KIND, int NOPS>
static void run(u4 *out, std::vector<u4> &h, size_t nout)
{
    size_t bad = 0, lanes[64] = {0};
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(out, 0, nout * sizeof(u4));
        // the transform kernels' footprint: one workgroup per CU, two waves per SIMD
        hipFuncSetAttribute((const void *)probe<KIND, NOPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
        hipLaunchKernelGGL((probe<KIND, NOPS>), dim3(256), dim3(512), 150000, 0, out);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, nout * sizeof(u4), hipMemcpyDeviceToHost);
        for (size_t i = 0; i < nout; ++i) {
            const unsigned lane = (unsigned)(i & 63), e[4] = {h[i].x, h[i].y, h[i].z, h[i].w};
            for (int j = 0; j < 4; ++j) if (e[j] != (0xA0000000u | (4 * lane + j))) { ++bad; ++lanes[lane]; }
        }
    }
    printf("%-38s %-22s: %8zu wrong dwords of %zu", KIND == 0 ? "buffer store, SGPR soffset" : KIND == 1 ? "buffer store, soffset 0" : KIND == 2 ? "global store" : KIND == 3 ? "buffer stores, SGPR soffset, bursts of 4" : "LDS store (ds_write_b128)",
           NOPS < 0 ? "VALU write next" : NOPS == 0 ? "s_nop 0 (1 wait state)" : NOPS == 1 ? "s_nop 1" : NOPS == 2 ? "s_nop 2" : NOPS == 3 ? "s_nop 3" : NOPS == 5 ? "s_nop 5" : NOPS == 7 ? "s_nop 7" : "s_nop 15", bad, 2 * nout * 4);
    if (bad) { printf("   lanes:"); for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d", l); }
    printf("\n");
}

int main()
{
    const size_t nout = (size_t)256 * 8 * SLOTS * 64;
    u4 *out; hipMalloc(&out, nout * sizeof(u4));
    std::vector<u4> h(nout);
#define ALLN(K) run<K, -1>(out, h, nout); run<K, 0>(out, h, nout); run<K, 1>(out, h, nout); run<K, 2>(out, h, nout); run<K, 3>(out, h, nout); run<K, 5>(out, h, nout); run<K, 7>(out, h, nout); run<K, 15>(out, h, nout);
    ALLN(0) ALLN(1) ALLN(2) ALLN(3)
    run<4, -1>(out, h, nout); run<4, 0>(out, h, nout); run<4, 1>(out, h, nout);
    return 0;
}
