// Lane layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (measurement helper): 16 independent 4x4 blocks per instruction, block b = lanes 4b .. 4b+3.
// Prints, for a few blocks, D[vgpr i][lane j] after one instruction with A = 1 + row id, B = 10 * (1 + col id) encoded per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x, b = l >> 2, j = l & 3;
  const float a = (float)(1 + j) + 100.f * b;          // "row j of block b"
  const float bb = 10.f * (1 + j);                      // "col j"
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bb, acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = acc[i];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b : {0, 1, 15}) {
    printf("block %d: A row r = %d + r + 1, B col c = 10 (c + 1)\n", b, 100 * b);
    for (int j = 0; j < 4; ++j) { printf("  lane %2d:", 4 * b + j); for (int i = 0; i < 4; ++i) printf(" %8.0f", h[(4 * b + j) * 4 + i]); printf("\n"); }
  }
  return 0;
}
