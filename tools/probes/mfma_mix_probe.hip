// What does each ingredient of a GEMM main loop cost next to the MFMA stream?  gfx950, 8 waves per CU (512-thread
// workgroup, one per CU), 32 x v_mfma_f32_16x16x32_bf16 per iteration on 32 accumulators (wave tile 128 x 64), plus
//   DS : 12 ds_read_b128 per iteration feeding the MFMA operands (double-buffered fragment sets)
//   BAR: one s_barrier per iteration
//   DMA: 4 global_load_lds_dwordx4 (1 KiB per wave each) per iteration from an L2-resident buffer, counted vmcnt wait
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <bool DS, bool BAR, bool DMA>
__global__ __launch_bounds__(512, 2) void mix(int iters, const char* src, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // random-ish LDS contents
  for (int i = tid; i < 32768; i += 512) reinterpret_cast<uint32_t*>(smem)[i] = (i * 2654435761u) & 0x3f7f3f7fu;
  __syncthreads();
  u32x4 faA[4], fbA[8], faB[4], fbB[8];
  for (int i = 0; i < 4; ++i) { faA[i] = *reinterpret_cast<u32x4*>(smem + (i * 64 + lane) * 16); faB[i] = *reinterpret_cast<u32x4*>(smem + ((4 + i) * 64 + lane) * 16); }
  for (int i = 0; i < 8; ++i) { fbA[i] = *reinterpret_cast<u32x4*>(smem + 8192 + (i * 64 + lane) * 16); fbB[i] = *reinterpret_cast<u32x4*>(smem + 8192 + ((8 + i) * 64 + lane) * 16); }
  f32x4 acc[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const char* gsrc = src + (size_t)blockIdx.x * 65536 + tid * 16;
  auto step = [&](int it, u32x4 (&ca)[4], u32x4 (&cb)[8], u32x4 (&na)[4], u32x4 (&nb)[8]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ca[i]), __builtin_bit_cast(bf16x8, cb[j]), acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (BAR) __builtin_amdgcn_s_barrier();
    if (DMA) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + ((it * 4 + r) & 7) * 8192),
                                         (lds_ptr_t)(smem + 65536 + ((it & 3) * 4 + r) * 8192 + wave * 1024), 16, 0, 0);
    }
    if (DS) {
      const int o = (it & 7) * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i) na[i] = *reinterpret_cast<u32x4*>(smem + o + (i * 64 + lane) * 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) nb[i] = *reinterpret_cast<u32x4*>(smem + 8192 + o + (i * 64 + lane) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 2; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ca[i]), __builtin_bit_cast(bf16x8, cb[j]), acc[i][j], 0, 0, 0);
  };
  for (int it = 0; it < iters; it += 2) {
    step(it, faA, fbA, faB, fbB);
    step(it + 1, faB, fbB, faA, fbA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float r = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) r += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
  if (r == 123.456f) sink[0] = r;
}

template <bool DS, bool BAR, bool DMA>
void run(const char* name, const char* src, float* sink) {
  const int iters = 20000, wgs = 256;
  hipFuncSetAttribute((const void*)mix<DS, BAR, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mix<DS, BAR, DMA><<<wgs, 512, 160 * 1024>>>(200, src, sink);
  hipEventRecord(e0);
  mix<DS, BAR, DMA><<<wgs, 512, 160 * 1024>>>(iters, src, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)wgs * 8 * iters * 32 * 16384.0;
  printf("%-44s %8.3f ms  %8.1f TFLOP/s  %6.0f ns per 32-MFMA iteration\n", name, ms, flop / ms / 1e9, ms * 1e6 / iters);
}

int main() {
  char* src; float* sink;
  hipMalloc(&src, (size_t)256 * 65536); hipMalloc(&sink, 64);
  hipMemset(src, 0x3c, (size_t)256 * 65536);
  run<false, false, false>("MFMA only (LDS-random operands)", src, sink);
  run<true, false, false>("+ 12 ds_read_b128", src, sink);
  run<false, true, false>("+ s_barrier", src, sink);
  run<true, true, false>("+ 12 ds_read_b128 + s_barrier", src, sink);
  run<false, false, true>("+ 4 LDS-DMA pieces (vmcnt(8))", src, sink);
  run<true, true, true>("+ ds_read + s_barrier + LDS-DMA (= ring loop)", src, sink);
  return 0;
}
