// Probe of gfx950 ds_read_b64_tr_b16 lane semantics (used to design dsw_wgrad_x3.hip).
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/tr_probe tools/probe_ds_read_tr16.hip && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(s4* out){
  __shared__ __attribute__((aligned(16))) unsigned short sm[32*32];
  for (int e = threadIdx.x; e < 1024; e += 64) sm[e] = (unsigned short)((e / 32) * 100 + (e % 32));   // [n][c] = n*100 + c
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  const unsigned short* p = sm + (8*(g>>1) + (i>>2)) * 32 + (g&1)*16 + (i&3)*4;
  out[lane] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
}
int main(){
  s4* d; hipMalloc(&d, 64*sizeof(s4));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  s4 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l][0], h[l][1], h[l][2], h[l][3]);
  return 0;
}
