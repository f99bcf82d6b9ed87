// Issue cost (cycles per wave-instruction, one wave per SIMD) of the VALU patterns perceiver_out_kernel's combine is made of.
// hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP16(X) X X X X X X X X X X X X X X X X
template <int PAT>
__global__ __launch_bounds__(256) void k(uint64_t* t, float* sink) {
  f32x2 a0 = {1, 2}, a1 = {3, 4}, a2 = {5, 6}, a3 = {7, 8}, w = {0.5f, 0.25f}, w2 = {0.5f, 0.25f}, u0 = {1, 1}, u1 = {2, 2}, u2 = {3, 3}, u3 = {4, 4}, p = {threadIdx.x * 1.f, 2.f};
  float s0 = 1, s1 = 2, s2 = 3, s3 = 4, sw = 0.5f;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  for (int it = 0; it < 64; ++it) {
    if constexpr (PAT == 0)   // 64 independent-ish packed FMAs (4 accumulators in turn), no op_sel
      asm volatile(REP16("v_pk_fma_f32 %0, %5, %4, %0\n\tv_pk_fma_f32 %1, %6, %4, %1\n\tv_pk_fma_f32 %2, %7, %4, %2\n\tv_pk_fma_f32 %3, %8, %4, %3\n\t")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
    if constexpr (PAT == 1)   // with op_sel_hi:[0,1,1]
      asm volatile(REP16("v_pk_fma_f32 %0, %5, %4, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %6, %4, %1 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %2, %7, %4, %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %8, %4, %3 op_sel_hi:[0,1,1]\n\t")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
    if constexpr (PAT == 2)   // plain FMAs
      asm volatile(REP16("v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %6\n\tv_fmac_f32 %2, %4, %7\n\tv_fmac_f32 %3, %4, %8\n\t")
                   : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(sw) : "v"(u0.x), "v"(u1.x), "v"(u2.x), "v"(u3.x));
    if constexpr (PAT == 3)   // 64-bit DPP broadcasts
      asm volatile(REP16("v_mov_b64_dpp %0, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t")
                   : "+v"(w), "+v"(w2) : "v"(p));
    if constexpr (PAT == 4)   // 32-bit DPP broadcasts
      asm volatile(REP16("v_mov_b32_dpp %0, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t")
                   : "+v"(sw), "+v"(s0) : "v"(p.x));
    if constexpr (PAT == 5)   // the group as shipped: broadcast, then four packed FMAs that depend on it (x16 = 80 instructions)
      asm volatile(REP16("v_mov_b64_dpp %4, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_pk_fma_f32 %0, %5, %4, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %6, %4, %1 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %2, %7, %4, %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %8, %4, %3 op_sel_hi:[0,1,1]\n\t")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w) : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(p));
    if constexpr (PAT == 6)   // the group with the NEXT broadcast in front of the FMAs (two scratch pairs in turn)
      asm volatile(REP16("v_mov_b64_dpp %10, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_pk_fma_f32 %0, %5, %4, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %6, %4, %1 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %2, %7, %4, %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %8, %4, %3 op_sel_hi:[0,1,1]\n\t"
                         "v_mov_b64_dpp %4, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_pk_fma_f32 %0, %5, %10, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %6, %10, %1 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %2, %7, %10, %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %8, %10, %3 op_sel_hi:[0,1,1]\n\t")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w) : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(p), "v"(w2));
    if constexpr (PAT == 7)   // 32-bit group: broadcast + four dependent plain FMAs
      asm volatile(REP16("v_mov_b32_dpp %4, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %6\n\tv_fmac_f32 %2, %4, %7\n\tv_fmac_f32 %3, %4, %8\n\t")
                   : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(sw) : "v"(u0.x), "v"(u1.x), "v"(u2.x), "v"(u3.x), "v"(p.x));
    if constexpr (PAT == 8)   // packed multiplies by an SGPR pair?  (scalar weights: v_pk_fma_f32 with s[..] source)
      asm volatile(REP16("v_pk_fma_f32 %0, %5, s[4:5], %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %6, s[4:5], %1 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %2, %7, s[4:5], %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %8, s[4:5], %3 op_sel_hi:[0,1,1]\n\t")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  if (threadIdx.x == 0 && blockIdx.x == 0) t[PAT] = t1 - t0;
  sink[threadIdx.x] = a0.x + a1.x + a2.y + a3.y + w.x + w2.y + s0 + s1 + s2 + s3 + sw;
}
int main() {
  uint64_t* t; float* sink;
  hipMalloc(&t, 16 * 8); hipMalloc(&sink, 4096);
  hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, t, sink); hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, t, sink);
  hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, t, sink); hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, t, sink);
  hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, t, sink); hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, t, sink);
  hipLaunchKernelGGL(k<6>, dim3(256), dim3(256), 0, 0, t, sink); hipLaunchKernelGGL(k<7>, dim3(256), dim3(256), 0, 0, t, sink);
  hipLaunchKernelGGL(k<8>, dim3(256), dim3(256), 0, 0, t, sink);
  uint64_t h[16]; hipMemcpy(h, t, 16 * 8, hipMemcpyDeviceToHost);
  const int n[9] = {64, 64, 64, 64, 64, 80, 160, 80, 64};
  const char* name[9] = {"v_pk_fma_f32", "v_pk_fma_f32 op_sel_hi", "v_fmac_f32", "v_mov_b64_dpp", "v_mov_b32_dpp", "bcast64 + 4 pk (dependent)",
                         "bcast64 one step ahead + 4 pk", "bcast32 + 4 fmac (dependent)", "v_pk_fma_f32 with an SGPR pair"};
  for (int i = 0; i < 9; ++i) printf("%-36s %7.2f memtime ticks per instruction\n", name[i], (double)h[i] / (64.0 * n[i]));
  return 0;
}
