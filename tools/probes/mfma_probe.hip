// Pure-MFMA ceiling on gfx950: register-resident v_mfma_f32_16x16x32_bf16 streams, random or zero operands.
// Reports TFLOP/s and the effective shader clock (s_memtime ticks / wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mfma_loop(int iters, uint32_t seed, float* sink, unsigned long long* ticks) {
  const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  u32x4 a[4], b[8];
  for (int i = 0; i < 4; ++i)
    a[i] = u32x4{(t * 2654435761u + i * 40503u) * seed & 0x3fff3fffu, (t * 97u + i) * seed * 31u & 0x3fff3fffu,
                 (t + i * 7u) * seed * 2246822519u & 0x3fff3fffu, (t ^ (i * 13u)) * seed * 3266489917u & 0x3fff3fffu};
  for (int i = 0; i < 8; ++i)
    b[i] = u32x4{(t * 374761393u + i * 668265263u) * seed & 0x3fff3fffu, (t * 11u + i) * seed * 17u & 0x3fff3fffu,
                 (t + i * 5u) * seed * 1274126177u & 0x3fff3fffu, (t ^ (i * 29u)) * seed * 2654435761u & 0x3fff3fffu};
  f32x4 acc[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                            acc[i][j], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) r += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
  if (r == 123.456f) sink[0] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// same FLOPs per iteration with v_mfma_f32_32x32x16_bf16: 8 accumulators of 16 registers (wave tile 128 x 64 = 4 x 2)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mfma_loop32(int iters, uint32_t seed, float* sink) {
  const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  u32x4 a[2], b[4];
  for (int i = 0; i < 2; ++i)
    a[i] = u32x4{(t * 2654435761u + i * 40503u) * seed & 0x3fff3fffu, (t * 97u + i) * seed * 31u & 0x3fff3fffu,
                 (t + i * 7u) * seed * 2246822519u & 0x3fff3fffu, (t ^ (i * 13u)) * seed * 3266489917u & 0x3fff3fffu};
  for (int i = 0; i < 4; ++i)
    b[i] = u32x4{(t * 374761393u + i * 668265263u) * seed & 0x3fff3fffu, (t * 11u + i) * seed * 17u & 0x3fff3fffu,
                 (t + i * 5u) * seed * 1274126177u & 0x3fff3fffu, (t ^ (i * 29u)) * seed * 2654435761u & 0x3fff3fffu};
  f32x16 acc[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)      // 2 x (2 x 4) MFMAs of 32x32x16 = the FLOPs of 32 MFMAs of 16x16x32
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                              acc[i][j], 0, 0, 0);
  }
  float r = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) r += acc[i][j][k];
  if (r == 123.456f) sink[0] = r;
}

template <int WAVES>
void run32(const char* name, uint32_t seed, int wgs) {
  float* sink; hipMalloc(&sink, 64);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop32<WAVES><<<wgs, WAVES * 64>>>(100, seed, sink);
  hipEventRecord(e0);
  mfma_loop32<WAVES><<<wgs, WAVES * 64>>>(iters, seed, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)wgs * WAVES * iters * 16 * 32768.0;
  printf("%-40s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
}

template <int WAVES>
void run(const char* name, uint32_t seed, int wgs) {
  float* sink; unsigned long long* ticks;
  hipMalloc(&sink, 64); hipMalloc(&ticks, 64);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<WAVES><<<wgs, WAVES * 64>>>(100, seed, sink, ticks);
  hipEventRecord(e0);
  mfma_loop<WAVES><<<wgs, WAVES * 64>>>(iters, seed, sink, ticks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long tk; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
  const double flop = (double)wgs * WAVES * iters * 32 * 16384.0;
  printf("%-40s %8.3f ms  %8.1f TFLOP/s   s_memtime ticks/us = %.1f\n", name, ms, flop / ms / 1e9, tk / (ms * 1e3));
}

int main() {
  run<8>("8 waves/CU, random operands", 0x9E3779B9u, 256);
  run<8>("8 waves/CU, zero operands", 0u, 256);
  run<4>("4 waves/CU, random operands", 0x9E3779B9u, 256);
  run32<8>("32x32x16: 8 waves/CU, random operands", 0x9E3779B9u, 256);
  run32<8>("32x32x16: 8 waves/CU, zero operands", 0u, 256);
  run32<4>("32x32x16: 4 waves/CU, random operands", 0x9E3779B9u, 256);
  return 0;
}
