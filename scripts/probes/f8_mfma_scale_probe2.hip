// Round 5: WHICH 32 K-elements does a lane's E8M0 scale cover in v_mfma_scale_f32_32x32x64_f8f6f4?  A = B = ones; the A scale is x2 on ONE lane
// (5 or 37, both row 5); one 16-byte group of row 5's A operand is zeroed at a time: a zeroed group that WAS scaled lowers D[5][*] by 32, an
// unscaled one by 16.  Same for B (column 5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int int8v __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
__global__ void probe(const uint8_t *A, const uint8_t *B, const int *sa, const int *sb, float *D) {
  const int lane = threadIdx.x;
  int8v a = *(const int8v *)(A + lane * 32), b = *(const int8v *)(B + lane * 32);
  float16v c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[lane], 0, sb[lane]);
  for (int i = 0; i < 16; ++i) D[lane * 16 + i] = c[i];
}
int main() {
  uint8_t *dA, *dB; int *dsa, *dsb; float *dD;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 4096);
  const int one = 0x7F7F7F7F, two = 0x80808080;
  for (int operand = 0; operand < 2; ++operand)
    for (int sl : {5, 37})
      for (int grp = -1; grp < 4; ++grp) {
        std::vector<uint8_t> A(2048, 0x38), B(2048, 0x38);
        std::vector<int> sa(64, one), sb(64, one);
        (operand ? sb : sa)[sl] = two;
        if (grp >= 0) { const int lane = (grp >> 1) ? 37 : 5, off = (grp & 1) * 16; for (int j = 0; j < 16; ++j) (operand ? B : A)[lane * 32 + off + j] = 0; }
        hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        std::vector<float> D(1024);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        // operand A: D[row 5][col 0] = lane 32, register i with (i & 3) + 8 (i >> 2) + 4 = 5 -> i = 1;  operand B: D[row 0][col 5] = lane 5, i = 0
        const float v = operand ? D[5 * 16 + 0] : D[32 * 16 + 1];
        printf("%s scale x2 on lane %2d, zeroed group %s: %5.1f\n", operand ? "B" : "A", sl, grp < 0 ? "none            " : grp == 0 ? "lane  5 bytes  0-15" : grp == 1 ? "lane  5 bytes 16-31" : grp == 2 ? "lane 37 bytes  0-15" : "lane 37 bytes 16-31", v);
      }
  return 0;
}
