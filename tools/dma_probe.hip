// Probe: where does global_load_lds_dwordx{3,4} put each lane's bytes in LDS on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#define KERNEL(NAME, SZ) __global__ void NAME(const unsigned* g, unsigned* out){ \
  extern __shared__ unsigned lds[]; \
  int lane = threadIdx.x & 63; \
  for (int i = lane; i < 512; i += 64) lds[i] = 0xffffffffu; \
  __syncthreads(); \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)g + lane*SZ), (__attribute__((address_space(3))) void*)(lds), SZ, 0, 0); \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
  __syncthreads(); \
  for (int i=threadIdx.x;i<512;i+=64) out[i] = lds[i]; \
}
KERNEL(k3, 12)
KERNEL(k4, 16)
int main(){ unsigned *g,*o; (void)hipMalloc(&g,4096); (void)hipMalloc(&o,2048); unsigned h[1024]; for(int i=0;i<1024;++i)h[i]=i; (void)hipMemcpy(g,h,sizeof h,hipMemcpyHostToDevice);
 unsigned r[512];
 k3<<<1,64,2048>>>(g,o); (void)hipMemcpy(r,o,sizeof r,hipMemcpyDeviceToHost); printf("x3:"); for(int i=0;i<272;++i) printf(" %d",(int)r[i]); printf("\n");
 k4<<<1,64,2048>>>(g,o); (void)hipMemcpy(r,o,sizeof r,hipMemcpyDeviceToHost); printf("x4:"); for(int i=0;i<272;++i) printf(" %d",(int)r[i]); printf("\n");
 return 0;}
