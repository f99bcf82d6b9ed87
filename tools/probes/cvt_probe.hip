// Rounding of v_cvt_pk_f16_f32 and denormal handling of the f16 MFMA on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, float* out) {
  f32x2 v = {in[0], in[1]};
  f16x2 h = __builtin_convertvector(v, f16x2);
  f32x2 b = __builtin_convertvector(h, f32x2);
  out[0] = b.x; out[1] = b.y;
  // MFMA with a subnormal f16 operand: A = [s, 0, ...] (row 0), B = [1, 0, ...]
  const int lane = threadIdx.x;
  _Float16 s = (_Float16)in[2];           // subnormal value
  f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, bb = {0, 0, 0, 0, 0, 0, 0, 0};
  if (lane == 0) { a[0] = s; bb[0] = (_Float16)1.0f; }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bb, acc, 0, 0, 0);
  if (lane == 0) out[2] = acc.x;
}
int main() {
  float h_in[3] = {1.0f + 0x1p-11f + 0x1p-13f, 1.0f + 0x1p-11f - 0x1p-13f, 0x1p-20f};
  float *d_in, *d_out, h_out[3];
  hipMalloc(&d_in, 12); hipMalloc(&d_out, 12);
  hipMemcpy(d_in, h_in, 12, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d_in, d_out);
  hipMemcpy(h_out, d_out, 12, hipMemcpyDeviceToHost);
  printf("cvt(1+2^-11+2^-13) = 1 + %g * 2^-10 (RNE: 1, RTZ: 0)\n", (h_out[0] - 1.0f) / 0x1p-10f);
  printf("cvt(1+2^-11-2^-13) = 1 + %g * 2^-10 (RNE: 0, RTZ: 0)\n", (h_out[1] - 1.0f) / 0x1p-10f);
  printf("mfma(subnormal 2^-20 * 1) = %g (expected %g; 0 means f16 denormals are flushed)\n", h_out[2], 0x1p-20);
  return 0;
}
