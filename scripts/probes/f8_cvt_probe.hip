// Round 5: what v_cvt_scalef32_pk_fp8_f16 does with negative, tiny and large fp16 inputs (exact-lite converts SIGNED lo parts).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));
__global__ void probe(const _Float16 *in, uint8_t *out, int n) {
  const int i = threadIdx.x;
  if (2 * i + 1 >= n + 1) return;
  half2v h = {in[2 * i], in[2 * i + 1]};
  short2v r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, h, 1.0f, false);
  out[2 * i] = (uint8_t)(r[0] & 0xFF);
  out[2 * i + 1] = (uint8_t)((r[0] >> 8) & 0xFF);
}
static float e4m3_to_f(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? m * 0.001953125f : (1.f + m / 8.f) * (float)(1 << e) / 128.f;
  return s ? -v : v;
}
int main() {
  std::vector<float> vals = {0.3f, -0.3f, 1.0f, -1.0f, 0.07f, -0.07f, 0.01f, -0.01f, 0.002f, -0.002f, 100.f, -100.f, 448.f, -448.f, 0.29f, 0.31f, 0.33f, -0.33f, 3.3f, -3.3f, 0.0f, -0.0f};
  std::vector<_Float16> h(vals.size());
  for (size_t i = 0; i < vals.size(); ++i) h[i] = (_Float16)vals[i];
  _Float16 *d; uint8_t *o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, h.size());
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, (int)h.size());
  std::vector<uint8_t> r(h.size());
  hipMemcpy(r.data(), o, r.size(), hipMemcpyDeviceToHost);
  for (size_t i = 0; i < vals.size(); ++i) printf("%9.4f -> 0x%02x = %9.4f\n", vals[i], r[i], e4m3_to_f(r[i]));
  return 0;
}
