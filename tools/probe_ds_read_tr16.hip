// Probe of gfx950's transposing LDS read (ds_read_b64_tr_b16), the instruction behind the row-contiguous GEMM operands
// (csrc/gemm_glds.hip: TrStage / tr_fragment).  LDS holds element index = row*stride + col; every lane of a 16-lane group
// supplies the address of 4 consecutive bf16 of a [4 rows][16 cols] block (lane i: row i/4, cols 4*(i%4)..+3).  Output
// on MI355X: lane c of the group receives (row 0..3, col c) -- the block's column c, i.e. 4 consecutive "k" values of
// one "n".  Build + run on a GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probe_ds_read_tr16.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(uint16_t* out, int stride_elems) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  // lane l: row (l>>2)&3 ... try: address = base + ((l & 15) >> 2) * stride + (l & 3) * 4 + (l >> 4) * 16   (a [4][16] block per 16-lane group, groups side by side along columns)
  const int a = ((l & 15) >> 2) * stride_elems + (l & 3) * 4 + (l >> 4) * 16;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + a));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int stride : {64, 128}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d (element index = row*stride + col):\n", stride);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" r%d c%2d", h[l*4+j] / stride, h[l*4+j] % stride); printf("\n"); }
  }
  return 0;
}
