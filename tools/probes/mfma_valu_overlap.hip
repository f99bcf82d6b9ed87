// Does one wave overlap its own MFMAs with its own (independent) VALU instructions?  One wave per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define REP16(X) X X X X X X X X X X X X X X X X
#define MF(D) "v_mfma_f32_16x16x32_f16 " D ", %8, %9, " D "\n\t"
template <int PAT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(uint64_t* t, float* sink) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float s0 = 1, s1 = 2, s2 = 3, s3 = 4, sw = 0.5f, u = 1.5f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x + i); b[i] = (_Float16)(i); }
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  for (int it = 0; it < 64; ++it) {
    if constexpr (PAT == 0)   // 64 MFMAs
      asm volatile(REP16(MF("%0") MF("%1") MF("%2") MF("%3")) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(a), "v"(b), "v"(sw), "v"(u));
    if constexpr (PAT == 1)   // 64 MFMAs, each followed by 2 plain FMAs
      asm volatile(REP16(MF("%0") "v_fmac_f32 %4, %10, %11\n\tv_fmac_f32 %5, %10, %11\n\t" MF("%1") "v_fmac_f32 %6, %10, %11\n\tv_fmac_f32 %7, %10, %11\n\t"
                         MF("%2") "v_fmac_f32 %4, %10, %11\n\tv_fmac_f32 %5, %10, %11\n\t" MF("%3") "v_fmac_f32 %6, %10, %11\n\tv_fmac_f32 %7, %10, %11\n\t")
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(a), "v"(b), "v"(sw), "v"(u));
    if constexpr (PAT == 2)   // 64 MFMAs, each followed by 4 plain FMAs
      asm volatile(REP16(MF("%0") "v_fmac_f32 %4, %10, %11\n\tv_fmac_f32 %5, %10, %11\n\tv_fmac_f32 %6, %10, %11\n\tv_fmac_f32 %7, %10, %11\n\t"
                         MF("%1") "v_fmac_f32 %4, %10, %11\n\tv_fmac_f32 %5, %10, %11\n\tv_fmac_f32 %6, %10, %11\n\tv_fmac_f32 %7, %10, %11\n\t"
                         MF("%2") "v_fmac_f32 %4, %10, %11\n\tv_fmac_f32 %5, %10, %11\n\tv_fmac_f32 %6, %10, %11\n\tv_fmac_f32 %7, %10, %11\n\t"
                         MF("%3") "v_fmac_f32 %4, %10, %11\n\tv_fmac_f32 %5, %10, %11\n\tv_fmac_f32 %6, %10, %11\n\tv_fmac_f32 %7, %10, %11\n\t")
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(a), "v"(b), "v"(sw), "v"(u));
    if constexpr (PAT == 3)   // the FMAs of pattern 2 alone (256)
      asm volatile(REP16("v_fmac_f32 %4, %10, %11\n\tv_fmac_f32 %5, %10, %11\n\tv_fmac_f32 %6, %10, %11\n\tv_fmac_f32 %7, %10, %11\n\t"
                         "v_fmac_f32 %4, %10, %11\n\tv_fmac_f32 %5, %10, %11\n\tv_fmac_f32 %6, %10, %11\n\tv_fmac_f32 %7, %10, %11\n\t"
                         "v_fmac_f32 %4, %10, %11\n\tv_fmac_f32 %5, %10, %11\n\tv_fmac_f32 %6, %10, %11\n\tv_fmac_f32 %7, %10, %11\n\t"
                         "v_fmac_f32 %4, %10, %11\n\tv_fmac_f32 %5, %10, %11\n\tv_fmac_f32 %6, %10, %11\n\tv_fmac_f32 %7, %10, %11\n\t")
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(a), "v"(b), "v"(sw), "v"(u));
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  if (threadIdx.x == 0 && blockIdx.x == 0) t[PAT] = t1 - t0;
  sink[threadIdx.x] = c0.x + c1.x + c2.y + c3.y + s0 + s1 + s2 + s3;
}
template <int WAVES>
void run(uint64_t* t, float* sink) {
  hipLaunchKernelGGL((k<0, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, t, sink);
  hipLaunchKernelGGL((k<1, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, t, sink);
  hipLaunchKernelGGL((k<2, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, t, sink);
  hipLaunchKernelGGL((k<3, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, t, sink);
  uint64_t h[4];
  (void)hipMemcpy(h, t, 4 * 8, hipMemcpyDeviceToHost);
  printf("%d wave(s) per SIMD, ticks per group of one MFMA (+ its FMAs):  MFMA alone %.1f | + 2 FMAs %.1f | + 4 FMAs %.1f | the 4 FMAs alone %.1f\n",
         WAVES / 4, h[0] / 4096.0, h[1] / 4096.0, h[2] / 4096.0, h[3] / 4096.0);
}
int main() {
  uint64_t* t; float* sink;
  (void)hipMalloc(&t, 16 * 8); (void)hipMalloc(&sink, 4096);
  run<4>(t, sink);
  run<8>(t, sink);
  return 0;
}
