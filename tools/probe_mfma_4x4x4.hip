// layout probe for v_mfma_f32_4x4x4_16b_{f16,bf16} and 4x4x1_16b_f32 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) _Float16 h4;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(4))) float f4;
// wave w: A one-hot at (lane l0 = w / 4, elem e0 = w % 4); B(l, e) = 1 + 4*l + e
__global__ void probe_f16(float* out) {
  const int w = blockIdx.x, l = threadIdx.x;
  h4 a = {0, 0, 0, 0};
  if (l == w / 4) a[w % 4] = (_Float16)1.0f;
  h4 b;
  for (int e = 0; e < 4; ++e) b[e] = (_Float16)(float)(1 + 4 * l + e);
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(w * 64 + l) * 4 + r] = acc[r];
}
__global__ void probe_bf16(float* out) {
  const int w = blockIdx.x, l = threadIdx.x;
  s4 a = {0, 0, 0, 0};
  if (l == w / 4) a[w % 4] = (short)0x3f80;
  s4 b;
  for (int e = 0; e < 4; ++e) { float f = (float)(1 + 4 * l + e); b[e] = (short)(__float_as_uint(f) >> 16); }
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(w * 64 + l) * 4 + r] = acc[r];
}
__global__ void probe_f32(float* out) {   // one element per lane: wave w: A one-hot at lane w
  const int w = blockIdx.x, l = threadIdx.x;
  float a = l == w ? 1.f : 0.f, b = (float)(1 + l);
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(w * 64 + l) * 4 + r] = acc[r];
}
int main() {
  float* d; hipMalloc(&d, 256 * 64 * 4 * sizeof(float));
  std::vector<float> h(256 * 64 * 4);
  const char* names[3] = {"f16", "bf16", "f32"};
  for (int t = 0; t < 3; ++t) {
    int waves = t == 2 ? 64 : 256;
    hipMemset(d, 0, h.size() * 4);
    if (t == 0) hipLaunchKernelGGL(probe_f16, dim3(waves), dim3(64), 0, 0, d);
    if (t == 1) hipLaunchKernelGGL(probe_bf16, dim3(waves), dim3(64), 0, 0, d);
    if (t == 2) hipLaunchKernelGGL(probe_f32, dim3(waves), dim3(64), 0, 0, d);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    printf("== %s\n", names[t]);
    for (int w = 0; w < waves; ++w) {
      if (!(w < 20 || w % 37 == 0)) continue;
      if (t == 2) printf("A one-hot lane %d ->", w); else printf("A one-hot (lane %d, elem %d) ->", w / 4, w % 4);
      for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        float v = h[(w * 64 + l) * 4 + r];
        if (v != 0.f) { int code = (int)v - 1; if (t == 2) printf(" D[lane %d reg %d]=B(lane %d)", l, r, code); else printf(" D[lane %d reg %d]=B(lane %d,e %d)", l, r, code / 4, code % 4); }
      }
      printf("\n");
    }
  }
  return 0;
}
