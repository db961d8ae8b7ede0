// What does a kernel boundary cost as a function of the NEXT kernel's shape?  A captured graph of [small kernel, probe kernel] x 50;
// the probe kernel does nothing but exist with G workgroups of T threads and L bytes of dynamic LDS.  Prints us per pair.
//   hipcc --offload-arch=gfx950 -O2 tools/boundary_cost.hip -o /tmp/boundary_cost && /tmp/boundary_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void small_kernel(double *out) { out[blockIdx.x * blockDim.x + threadIdx.x] = 1.0; }
__global__ void probe_kernel(double *out)
{
    extern __shared__ double lds[];
    if (threadIdx.x == 0) { lds[0] = 1.0; out[blockIdx.x] = lds[0]; }
}
int main()
{
    double *d;
    CK(hipMalloc(&d, 1 << 22));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int grids[] = {1, 37, 74, 148, 256, 512}, ldss[] = {0, 64 * 1024, 152 * 1024}, thr[] = {64, 448};
    for (int T : thr)
        for (int L : ldss)
            for (int G : grids) {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int i = 0; i < 50; ++i) {
                    hipLaunchKernelGGL(small_kernel, dim3(288), dim3(128), 0, s, d);
                    hipLaunchKernelGGL(probe_kernel, dim3(G), dim3(T), L, s, d);
                }
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("threads %3d  LDS %6d B  grid %3d : %.2f us per [small, probe] pair\n", T, L, G, ms * 1e3 / (10 * 50));
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
    return 0;
}
