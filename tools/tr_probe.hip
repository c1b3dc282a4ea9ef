// Probe of ds_read_b64_tr_b16 lane semantics on gfx950: LDS tile [8 rows][16 cols] of 16-bit values row*100+col.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(s4* out) {
  __shared__ short lds[64 * 40];
  const int PITCH = 40;  // elements
  for (int i = threadIdx.x; i < 64 * PITCH; i += 64) lds[i] = (short)((i / PITCH) * 100 + (i % PITCH));
  __syncthreads();
  int l = threadIdx.x, g = l >> 4, j = l & 15, q = j >> 2, p = j & 3;
  // group g reads rows 8g+q (q=0..3), cols 4p..4p+3
  const short* a = lds + (8 * g + q) * PITCH + 4 * p;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)a);
  out[l] = v;
}
int main() {
  s4* d; hipMalloc(&d, 64 * sizeof(s4));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  s4 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %d %d %d %d\n", l, h[l][0], h[l][1], h[l][2], h[l][3]);
  return 0;
}
