// VALU issue cost on gfx950, per wave-instruction: plain fp32 FMA, packed fp32 FMA / MUL, the transcendentals
// (v_exp_f32, v_rcp_f32) and v_med3_f32 -- sixteen independent register chains per lane, 1 or 2 waves per SIMD.
// Answers what bounds a GEMM epilogue's activation: shader-clock ticks per instruction as seen by
// one wave (s_memtime), i.e. 4.0 = full rate for a wave alone on its SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void valu_loop(int iters, float* sink, unsigned long long* ticks) {
  const float t = (float)(threadIdx.x + 1) * 1e-3f;
  float r[16];
  f32x2 p[16];
  for (int k = 0; k < 16; ++k) { r[k] = t + k * 0.01f; p[k] = f32x2{t + k * 0.01f, t - k * 0.01f}; }
  const float a = 0.999f, b = 1e-4f;
  const f32x2 a2 = {0.999f, 0.998f}, b2 = {1e-4f, 2e-4f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b));
      if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(a2), "v"(b2));
      if constexpr (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(a2));
      if constexpr (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(r[k]));
      if constexpr (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[k]));
      if constexpr (KIND == 5) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(b), "v"(a));
      if constexpr (KIND == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[k]) : "v"(a));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int k = 0; k < 16; ++k) s += r[k] + p[k].x + p[k].y;
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int KIND>
static void run(const char* name, float* sink, unsigned long long* ticks) {
  const int iters = 400000;   // tens of milliseconds per launch: the clocks have ramped and hold
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(valu_loop<KIND>, dim3(256), dim3(256 * waves_per_simd), 0, 0, 10, sink, ticks);
    hipEventRecord(e0);
    hipLaunchKernelGGL(valu_loop<KIND>, dim3(256), dim3(256 * waves_per_simd), 0, 0, iters, sink, ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long tk = 0;
    hipMemcpy(&tk, ticks, sizeof tk, hipMemcpyDeviceToHost);
    const double per = (double)tk / ((double)iters * 16);
    const double ns = ms * 1e6 / ((double)iters * 16 * waves_per_simd);   // per instruction issued by one SIMD
    printf("%-18s %d wave(s)/SIMD: %6.3f ns per instruction and SIMD (%5.2f cycles at 2.4 GHz), kernel %.2f ms, %6.2f s_memtime ticks per wave-instruction\n",
           name, waves_per_simd, ns, ns * 2.4, ms, per);
  }
}

int main() {
  float* sink;
  unsigned long long* ticks;
  hipMalloc(&sink, 4);
  hipMalloc(&ticks, 8);
  hipLaunchKernelGGL(valu_loop<0>, dim3(256), dim3(512), 0, 0, 2000000, sink, ticks);   // warm-up: ~0.3 s of VALU work
  hipDeviceSynchronize();
  run<0>("v_fma_f32", sink, ticks);
  run<1>("v_pk_fma_f32", sink, ticks);
  run<2>("v_pk_mul_f32", sink, ticks);
  run<3>("v_exp_f32", sink, ticks);
  run<4>("v_rcp_f32", sink, ticks);
  run<5>("v_med3_f32", sink, ticks);
  run<6>("v_cvt_pk_bf16_f32", sink, ticks);
  return 0;
}
