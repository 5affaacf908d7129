// Round 5: which lane supplies the E8M0 block scale of which K block in v_mfma_scale_f32_32x32x64_f8f6f4?  (exact-lite, conv_mfma_kernel
// NSPLIT == 6, wants DIFFERENT scales for K 0..31 and K 32..63 of one operand row.)  A = B = all ones (e4m3 0x38): every output = 64 with
// unit scales; the scale VGPR is set per lane / per byte and the output says which K block it multiplied.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/f8_mfma_scale_probe.hip -o /tmp/f8s && /tmp/f8s
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int int8v __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
template <int OPA, int OPB>
__global__ void probe(const int *sa, const int *sb, float *D) {
  const int lane = threadIdx.x;
  int8v a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
  float16v c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPA, sa[lane], OPB, sb[lane]);
  for (int i = 0; i < 16; ++i) D[lane * 16 + i] = c[i];
}
int main() {
  int *dsa, *dsb; float *dD;
  hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 64 * 16 * 4);
  auto run = [&](const char *what, std::vector<int> sa, std::vector<int> sb, int opa, int opb) {
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    if (opa == 0 && opb == 0) hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, dsa, dsb, dD);
    else if (opa == 1) hipLaunchKernelGGL((probe<1, 0>), dim3(1), dim3(64), 0, 0, dsa, dsb, dD);
    else hipLaunchKernelGGL((probe<0, 1>), dim3(1), dim3(64), 0, 0, dsa, dsb, dD);
    std::vector<float> D(1024);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    // D[lane][i]: column = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
    printf("%-70s row0 col0 %6.1f  row0 col5 %6.1f  row5 col0 %6.1f  row5 col5 %6.1f\n", what, D[0], D[5 * 16], D[32 * 16 + 1], D[(32 + 5) * 16 + 1]);
  };
  const int one = 0x7F7F7F7F, two = 0x80808080;
  std::vector<int> u(64, one);
  run("unit scales", u, u, 0, 0);
  { auto s = u; for (int l = 0; l < 32; ++l) s[l] = two; run("A scale x2 on lanes 0-31 only (per-lane-half: 96; row-lane only: 128)", s, u, 0, 0); }
  { auto s = u; for (int l = 32; l < 64; ++l) s[l] = two; run("A scale x2 on lanes 32-63 only (per-lane-half: 96; ignored: 64)", s, u, 0, 0); }
  { auto s = u; s[5] = two; run("A scale x2 on lane 5 only (row 5, K 0-31 -> row5: 96)", s, u, 0, 0); }
  { auto s = u; s[37] = two; run("A scale x2 on lane 37 only (row 5, K 32-63 -> row5: 96)", s, u, 0, 0); }
  { auto s = u; for (auto &v : s) v = 0x7F7F807F; run("A scale byte1 = x2 everywhere, opsel 0 (64 if byte 0 is used)", s, u, 0, 0); }
  { auto s = u; for (auto &v : s) v = 0x7F7F807F; run("A scale byte1 = x2 everywhere, opsel 1 (128 if byte 1 is selected)", s, u, 1, 0); }
  { auto s = u; for (int l = 0; l < 32; ++l) s[l] = two; run("B scale x2 on lanes 0-31 only", u, s, 0, 0); }
  { auto s = u; for (int l = 32; l < 64; ++l) s[l] = two; run("B scale x2 on lanes 32-63 only", u, s, 0, 0); }
  { auto s = u; s[5] = two; run("B scale x2 on lane 5 only (col 5, K 0-31 -> col5: 96)", u, s, 0, 0); }
  { auto s = u; s[37] = two; run("B scale x2 on lane 37 only (col 5, K 32-63 -> col5: 96)", u, s, 0, 0); }
  return 0;
}
