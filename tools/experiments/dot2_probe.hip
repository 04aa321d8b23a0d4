// Is v_dot2c_f32_bf16 (D = A.lo * B.lo + A.hi * B.hi + D) usable as an EXACT per-channel fma when one half of B is zero?
//   fma(a.lo, b.lo, c)  vs  dot2(a, {b.lo, 0}, c)   and   fma(a.hi, b.hi, c)  vs  dot2(a, {0, b.hi}, c)
// bitwise, over random normal values, subnormal inputs, products in the subnormal range, zeros of both signs, huge values.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/experiments/dot2_probe.hip -o /tmp/dot2_probe && /tmp/dot2_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;

__global__ void probe(uint32_t* r_fma_lo, uint32_t* r_dot_lo, uint32_t* r_fma_hi, uint32_t* r_dot_hi, uint32_t* r_dot_both, uint32_t* r_fma_both,
                      const uint32_t* a, const uint32_t* b, const float* c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t aw = a[i], bw = b[i];
  const float alo = __uint_as_float(aw << 16), ahi = __uint_as_float(aw & 0xffff0000u);
  const float blo = __uint_as_float(bw << 16), bhi = __uint_as_float(bw & 0xffff0000u);
  r_fma_lo[i] = __float_as_uint(__builtin_fmaf(alo, blo, c[i]));
  r_fma_hi[i] = __float_as_uint(__builtin_fmaf(ahi, bhi, c[i]));
  r_dot_lo[i] = __float_as_uint(__builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, aw), __builtin_bit_cast(bf2, bw & 0x0000ffffu), c[i], false));
  r_dot_hi[i] = __float_as_uint(__builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, aw), __builtin_bit_cast(bf2, bw & 0xffff0000u), c[i], false));
  r_dot_both[i] = __float_as_uint(__builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, aw), __builtin_bit_cast(bf2, bw), c[i], false));
  r_fma_both[i] = __float_as_uint(__builtin_fmaf(ahi, bhi, __builtin_fmaf(alo, blo, c[i])));
}

static uint16_t rnd_bf16(int cls) {
  const uint16_t sign = (rand() & 1) << 15;
  switch (cls) {
    case 0: return sign | (uint16_t)((100 + rand() % 56) << 7) | (rand() & 0x7f);      // normal, moderate exponent
    case 1: return sign | (uint16_t)(rand() & 0x7f);                                       // subnormal (or zero)
    case 2: return sign;                                                                   // zero
    case 3: return sign | (uint16_t)((1 + rand() % 20) << 7) | (rand() & 0x7f);           // tiny normal: products underflow
    case 4: return sign | (uint16_t)((230 + rand() % 24) << 7) | (rand() & 0x7f);         // huge: products overflow
    default: return sign | (uint16_t)((1 + rand() % 254) << 7) | (rand() & 0x7f);         // any finite normal
  }
}

int main() {
  const int n = 1 << 20;
  uint32_t *a = (uint32_t*)malloc(n * 4), *b = (uint32_t*)malloc(n * 4);
  float* c = (float*)malloc(n * 4);
  int* cls = (int*)malloc(n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) {
    const int k = i % 36, ka = k / 6, kb = k % 6;
    cls[i] = k;
    a[i] = (uint32_t)rnd_bf16(ka) | ((uint32_t)rnd_bf16(rand() % 6) << 16);
    b[i] = (uint32_t)rnd_bf16(kb) | ((uint32_t)rnd_bf16(rand() % 6) << 16);
    const int kc = (i / 36) % 4;
    float cv = kc == 0 ? 0.f : kc == 1 ? (float)(rand() % 2001 - 1000) / 37.f : kc == 2 ? 1e-39f * (rand() % 100) : -0.f;
    c[i] = cv;
  }
  uint32_t *da, *db, *o[6];
  float* dc;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4);
  for (int k = 0; k < 6; ++k) hipMalloc(&o[k], n * 4);
  hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c, n * 4, hipMemcpyHostToDevice);
  probe<<<n / 256, 256>>>(o[0], o[1], o[2], o[3], o[4], o[5], da, db, dc, n);
  uint32_t* h[6];
  for (int k = 0; k < 6; ++k) { h[k] = (uint32_t*)malloc(n * 4); hipMemcpy(h[k], o[k], n * 4, hipMemcpyDeviceToHost); }
  long bad_lo[36] = {0}, bad_hi = 0, bad_both = 0, tot[36] = {0};
  int shown = 0;
  for (int i = 0; i < n; ++i) {
    tot[cls[i]]++;
    const bool nan_lo = (h[0][i] & 0x7fffffff) > 0x7f800000 && (h[1][i] & 0x7fffffff) > 0x7f800000;
    if (h[0][i] != h[1][i] && !nan_lo) {
      bad_lo[cls[i]]++;
      if (shown < 12) { printf("lo mismatch cls a=%d b=%d: a=%08x b=%08x c=%08x fma=%08x dot=%08x\n", cls[i] / 6, cls[i] % 6, a[i], b[i], *(uint32_t*)&c[i], h[0][i], h[1][i]); ++shown; }
    }
    bad_hi += h[2][i] != h[3][i] && !((h[2][i] & 0x7fffffff) > 0x7f800000 && (h[3][i] & 0x7fffffff) > 0x7f800000);
    bad_both += h[4][i] != h[5][i];
  }
  const char* names[6] = {"normal", "subnormal", "zero", "tiny", "huge", "any"};
  for (int k = 0; k < 36; ++k) printf("lo half: a %-9s x b %-9s: %ld of %ld differ\n", names[k / 6], names[k % 6], bad_lo[k], tot[k]);
  printf("hi half: %ld of %d differ; both halves vs two chained fmas: %ld of %d differ\n", bad_hi, n, bad_both, n);
  return 0;
}
