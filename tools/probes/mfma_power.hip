// Energy per FLOP of the two bf16 MFMA shapes: back-to-back MFMAs on registers (no memory traffic), two waves per SIMD on every CU,
// a few seconds each; run beside `rocm-smi --showclocks --showpower` (tools/probes/mfma_power.sh).  Prints TFLOP/s per form.
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_power.hip -o /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int FORM>
__global__ __launch_bounds__(512) void k(float* sink, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
  if constexpr (FORM == 0) {
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i].x;
    sink[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    f32x16 c[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i) s += c[i][0];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
  }
}
int main(int argc, char** argv) {
  const int form = argc > 1 ? atoi(argv[1]) : 0;
  const double seconds = argc > 2 ? atof(argv[2]) : 6.0;
  float* sink; (void)hipMalloc(&sink, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200000;   // per launch: 8 x 16x16x32 or 4 x 32x32x16 MFMAs per iteration = 131,072 FLOP x 2 per wave either way
  double flop = 0, ms_total = 0;
  while (ms_total < seconds * 1e3) {
    hipEventRecord(e0);
    if (form == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, sink, iters);
    else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms_total += ms;
    flop += 256.0 * 8 * iters * 8 * 16384.0;   // workgroups x waves x iterations x MFMAs x FLOP (32x32x16: 4 x 32768, the same)
  }
  printf("%s: %.1f TFLOP/s over %.1f s\n", form == 0 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_32x32x16_bf16", flop / ms_total / 1e9, ms_total / 1e3);
  return 0;
}
