// Discover the operand lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 (GPU box).
// For every pair (la, lb): A = 1 only in lane la, B = 1 only in lane lb, C = 0 -> which D lanes light up.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned long long *mask)
{
    const int pair = blockIdx.x, la = pair >> 6, lb = pair & 63, lane = threadIdx.x;
    double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    unsigned long long m = __ballot(d != 0.0);
    if (lane == 0) mask[pair] = m;
}
int main()
{
    unsigned long long *dm; std::vector<unsigned long long> h(4096);
    hipMalloc(&dm, 4096 * 8);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, dm);
    hipMemcpy(h.data(), dm, 4096 * 8, hipMemcpyDeviceToHost);
    // D[blk][i][j] = sum_k A[blk][i][k] B[blk][k][j].  A lane la pairs with B lane lb iff same blk and same k.
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d pairs with B lanes:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb]) {
            int dl = __builtin_ctzll(h[la * 64 + lb]);
            printf(" %d->D%d%s", lb, dl, __builtin_popcountll(h[la * 64 + lb]) > 1 ? "+" : "");
        }
        printf("\n");
    }
    return 0;
}
