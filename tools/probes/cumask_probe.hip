// Can an HBM-bound kernel (LayerNorm / window attention) run BESIDE an MFMA-bound GEMM if it is confined to a few CUs?
// The ping-pong GEMM holds a whole CU (8 waves x ~250 VGPRs, 128 KiB LDS): nothing co-resides with one of its workgroups, and
// on two plain streams a LayerNorm's small workgroups take every CU a finished tile frees until the LayerNorm is done --
// the GEMM stands still meanwhile (profiles/r05_ab_row_chunks.log).  hipExtStreamCreateWithCUMask confines a stream's
// kernels to a CU subset; this probe measures, through the C ABI of libaurora_hip.so,
//   (1) where the mask's bits land (XCC / SE / CU of every workgroup of a census kernel),
//   (2) the LayerNorm's rate on n CUs per XCD (stage-1 / stage-2 shapes of the 0.25-degree step),
//   (3) GEMM and LayerNorm one after the other against GEMM (all CUs) beside LayerNorm (masked stream).
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o tools/probes/cumask_probe -Laurora_amd/_lib -laurora_hip \
//         -Wl,-rpath,'$ORIGIN/../../aurora_amd/_lib'
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <vector>

#include "../../include/aurora_hip.h"

#define HIP_OK(x)                                                                    \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                       \
    }                                                                                \
  } while (0)
#define A_OK(x)                                                                  \
  do {                                                                           \
    int r_ = (x);                                                                \
    if (r_ != 0) { fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #x, r_, aurora_hip_last_error()); exit(3); } \
  } while (0)

__global__ void census_kernel(uint32_t* out) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the workgroup alive for a moment so that the launch spreads over every CU it may use
  const uint64_t t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 200000) {}
  if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xffffu) | ((xcc & 0xfu) << 16);
}

__global__ void fill_kernel(uint16_t* p, int64_t n, uint32_t seed) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    const float f = ((float)(x & 0xffff) / 32768.0f - 1.0f) * 0.5f;
    p[i] = (uint16_t)(__float_as_uint(f) >> 16);
  }
}
__global__ void fill_f32(float* p, int64_t n, uint32_t seed) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    p[i] = (float)(x & 0xffff) / 65536.0f - 0.5f;
  }
}


// ---- a LayerNorm that is meant to run BESIDE a GEMM workgroup on the same CU ----
// The ping-pong GEMM takes 2 waves x 208 VGPRs per SIMD and 128 KiB of LDS: 96 VGPRs per SIMD, 32 KiB of LDS and six wave
// slots per SIMD stay free.  This kernel is persistent -- G workgroups of 4 waves (one per SIMD), rows taken round-robin
// -- so that it never holds more than it started with: a GEMM workgroup always finds its 416 VGPRs per SIMD beside it.
typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));
__device__ inline float wsum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int NC>   // chunks of 256 features
__global__ __launch_bounds__(256) void ln_bg_kernel(const uint16_t* y, const float* gain, const float* shift, const float* res,
                                                    float* out_f32, uint16_t* out_b, int64_t M, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += n_waves) {
    u32x2_ yv[NC];
    f32x4_ rv[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      yv[i] = *reinterpret_cast<const u32x2_*>(y + row * D + (lane + 64 * i) * 4);
      rv[i] = *reinterpret_cast<const f32x4_*>(res + row * D + (lane + 64 * i) * 4);
    }
    float v[NC][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      v[i][0] = __uint_as_float(yv[i].x << 16); v[i][1] = __uint_as_float(yv[i].x & 0xffff0000u);
      v[i][2] = __uint_as_float(yv[i].y << 16); v[i][3] = __uint_as_float(yv[i].y & 0xffff0000u);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wsum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q = fmaf(d, d, q); }
    const float rstd = rsqrtf(wsum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int e = (lane + 64 * i) * 4;
      const f32x4_ g = *reinterpret_cast<const f32x4_*>(gain + e), sh = *reinterpret_cast<const f32x4_*>(shift + e);
      f32x4_ o;
      o.x = (v[i][0] - mean) * rstd * g.x + sh.x + rv[i].x; o.y = (v[i][1] - mean) * rstd * g.y + sh.y + rv[i].y;
      o.z = (v[i][2] - mean) * rstd * g.z + sh.z + rv[i].z; o.w = (v[i][3] - mean) * rstd * g.w + sh.w + rv[i].w;
      *reinterpret_cast<f32x4_*>(out_f32 + row * D + e) = o;
      u32x2_ b;
      b.x = (__float_as_uint(o.x) >> 16) | (__float_as_uint(o.y) & 0xffff0000u);   // (truncation: a probe)
      b.y = (__float_as_uint(o.z) >> 16) | (__float_as_uint(o.w) & 0xffff0000u);
      *reinterpret_cast<u32x2_*>(out_b + row * D + e) = b;
    }
  }
}

static hipStream_t masked_stream(int per_xcd, int n_xcd) {
  // bit i of the mask: XCC i % n_xcd, then round-robin over its shader engines (checked by the census below)
  std::vector<uint32_t> mask(16, 0u);
  for (int i = 0; i < per_xcd * n_xcd; ++i) mask[i / 32] |= 1u << (i % 32);
  hipStream_t s;
  HIP_OK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  return s;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, n_xcd = 8;
  printf("device: %s, %d CUs\n", prop.name, cus);
  uint32_t* d_census;
  HIP_OK(hipMalloc(&d_census, 4096 * 4));
  for (int per : {0, 4, 8, 16}) {
    hipStream_t s;
    if (per == 0) HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else s = masked_stream(per, n_xcd);
    HIP_OK(hipMemsetAsync(d_census, 0xff, 4096 * 4, s));
    census_kernel<<<2048, 64, 0, s>>>(d_census);
    HIP_OK(hipStreamSynchronize(s));
    std::vector<uint32_t> h(2048);
    HIP_OK(hipMemcpy(h.data(), d_census, 2048 * 4, hipMemcpyDeviceToHost));
    std::map<int, std::map<int, int>> by_xcc;   // xcc -> (se, cu) -> workgroups
    for (uint32_t v : h) by_xcc[(v >> 16) & 0xf][((v >> 13) & 7) * 16 + ((v >> 8) & 0xf)]++;
    printf("census, %s: ", per ? "mask of the first `per` x 8 bits" : "no mask");
    if (per) printf("per = %d: ", per);
    int total = 0;
    for (auto& x : by_xcc) { printf("xcc%d:%zu ", x.first, x.second.size()); total += (int)x.second.size(); }
    printf("= %d CUs\n", total);
    if (per == 4)
      for (auto& x : by_xcc) {
        printf("   xcc%d (se.cu):", x.first);
        for (auto& c : x.second) printf(" %d.%d", c.first / 16, c.first % 16);
        printf("\n");
      }
    HIP_OK(hipStreamDestroy(s));
  }

  // ---- LayerNorm rate by CU count ----
  struct Shape { const char* name; int64_t M; int D, N, K; };
  const Shape shapes[] = {{"s1 (LN 64800 x 1024 | fc1 64800 x 4096 x 1024)", 64800, 1024, 4096, 1024},
                          {"s2 (LN 16200 x 2048 | fc1 16200 x 8192 x 2048)", 16200, 2048, 8192, 2048},
                          {"s0 (LN 259200 x 512 | fc1 259200 x 2048 x 512)", 259200, 512, 2048, 512}};
  hipStream_t plain;
  HIP_OK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  hipEvent_t e0, e1, e2;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreate(&e2));
  for (const Shape& sh : shapes) {
    const int64_t M = sh.M;
    const int D = sh.D;
    uint16_t *y, *xb, *A, *W, *C;
    float *xf, *gain, *shift, *bias;
    HIP_OK(hipMalloc(&y, M * D * 2)); HIP_OK(hipMalloc(&xb, M * D * 2)); HIP_OK(hipMalloc(&xf, M * D * 4));
    HIP_OK(hipMalloc(&gain, D * 4)); HIP_OK(hipMalloc(&shift, D * 4)); HIP_OK(hipMalloc(&bias, sh.N * 4));
    HIP_OK(hipMalloc(&A, M * sh.K * 2)); HIP_OK(hipMalloc(&W, (int64_t)sh.N * sh.K * 2)); HIP_OK(hipMalloc(&C, M * sh.N * 2));
    fill_kernel<<<1024, 256>>>(y, M * D, 1); fill_f32<<<1024, 256>>>(xf, M * D, 2);
    fill_f32<<<4, 256>>>(gain, D, 3); fill_f32<<<4, 256>>>(shift, D, 4); fill_f32<<<16, 256>>>(bias, sh.N, 5);
    fill_kernel<<<1024, 256>>>(A, M * sh.K, 6); fill_kernel<<<1024, 256>>>(W, (int64_t)sh.N * sh.K, 7);
    HIP_OK(hipDeviceSynchronize());
    const double ln_bytes = (double)M * D * 12, flop = 2.0 * M * sh.N * sh.K;
    auto ln = [&](hipStream_t s) { A_OK(aurora_hip_layernorm(y, D, gain, shift, xf, D, 0, xf, D, xb, D, M, D, 1e-5f, AURORA_BF16, s)); };
    auto gemm = [&](hipStream_t s) {
      A_OK(aurora_hip_linear_ex(A, sh.K, W, sh.K, bias, C, sh.N, nullptr, 0, nullptr, 0, M, sh.N, sh.K, AURORA_BF16, AURORA_ACT_GELU, -1,
                                nullptr, 0.f, s));
    };
    printf("%s\n", sh.name);
    const int reps = 20;
    float t_ln_full = 0, t_gemm = 0;
    for (int per : {0, 2, 4, 6, 8, 12, 16, 24}) {
      hipStream_t s = per ? masked_stream(per, n_xcd) : plain;
      for (int i = 0; i < 3; ++i) ln(s);
      HIP_OK(hipEventRecord(e0, s));
      for (int i = 0; i < reps; ++i) ln(s);
      HIP_OK(hipEventRecord(e1, s));
      HIP_OK(hipEventSynchronize(e1));
      float ms;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      if (!per) t_ln_full = ms;
      printf("  LayerNorm on %3d CUs: %8.1f us  %5.2f TB/s\n", per ? per * n_xcd : cus, ms * 1e3, ln_bytes / ms / 1e9);
      if (per) HIP_OK(hipStreamDestroy(s));
    }
    {
      for (int i = 0; i < 3; ++i) gemm(plain);
      HIP_OK(hipEventRecord(e0, plain));
      for (int i = 0; i < reps; ++i) gemm(plain);
      HIP_OK(hipEventRecord(e1, plain));
      HIP_OK(hipEventSynchronize(e1));
      HIP_OK(hipEventElapsedTime(&t_gemm, e0, e1));
      t_gemm /= reps;
      printf("  GEMM alone: %8.1f us  %6.0f TFLOP/s\n", t_gemm * 1e3, flop / t_gemm / 1e9);
    }
    // one after the other on one stream, `pairs` times: (GEMM, LayerNorm) -- what the step does today
    const int pairs = 20;
    {
      HIP_OK(hipEventRecord(e0, plain));
      for (int i = 0; i < pairs; ++i) { gemm(plain); ln(plain); }
      HIP_OK(hipEventRecord(e1, plain));
      HIP_OK(hipEventSynchronize(e1));
      float ms;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      printf("  serial GEMM + LayerNorm: %8.1f us per pair (sum of the parts %.1f)\n", ms / pairs * 1e3, (t_gemm + t_ln_full) * 1e3);
    }
    // beside each other: the GEMMs on the plain stream, the LayerNorms on a second stream (plain, or masked to n CUs per XCD)
    for (int per : {0, 2, 4, 6, 8, 12}) {
      hipStream_t s = per ? masked_stream(per, n_xcd) : nullptr;
      if (!per) HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      HIP_OK(hipDeviceSynchronize());
      HIP_OK(hipEventRecord(e0, plain));
      HIP_OK(hipStreamWaitEvent(s, e0, 0));
      for (int i = 0; i < pairs; ++i) { gemm(plain); ln(s); }
      HIP_OK(hipEventRecord(e2, s));
      HIP_OK(hipStreamWaitEvent(plain, e2, 0));
      HIP_OK(hipEventRecord(e1, plain));
      HIP_OK(hipEventSynchronize(e1));
      float ms;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      printf("  two streams, LayerNorm on %3d CUs: %8.1f us per pair\n", per ? per * n_xcd : cus, ms / pairs * 1e3);
      HIP_OK(hipStreamDestroy(s));
    }

    // ---- co-residency: the persistent LayerNorm (G workgroups) beside the GEMM ----
    if (D == 1024 || D == 2048 || D == 512) {
      hipStream_t s2;
      HIP_OK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
      auto bg = [&](hipStream_t st, int G) {
        if (D == 512) ln_bg_kernel<2><<<G, 256, 0, st>>>(y, gain, shift, xf, xf, xb, M, D, 1e-5f);
        else if (D == 1024) ln_bg_kernel<4><<<G, 256, 0, st>>>(y, gain, shift, xf, xf, xb, M, D, 1e-5f);
        else ln_bg_kernel<8><<<G, 256, 0, st>>>(y, gain, shift, xf, xf, xb, M, D, 1e-5f);
      };
      for (int G : {256, 512, 1024}) {
        for (int i = 0; i < 2; ++i) bg(plain, G);
        HIP_OK(hipEventRecord(e0, plain));
        for (int i = 0; i < reps; ++i) bg(plain, G);
        HIP_OK(hipEventRecord(e1, plain));
        HIP_OK(hipEventSynchronize(e1));
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        printf("  persistent LayerNorm alone, %4d workgroups: %8.1f us  %5.2f TB/s\n", G, ms * 1e3, ln_bytes / ms / 1e9);
        // beside the GEMMs: `pairs` GEMMs on one stream, `pairs` LayerNorms on the other, started together
        HIP_OK(hipDeviceSynchronize());
        hipEvent_t l0, l1;
        HIP_OK(hipEventCreate(&l0)); HIP_OK(hipEventCreate(&l1));
        HIP_OK(hipEventRecord(e0, plain));
        HIP_OK(hipStreamWaitEvent(s2, e0, 0));
        HIP_OK(hipEventRecord(l0, s2));
        for (int i = 0; i < pairs; ++i) { gemm(plain); bg(s2, G); }
        HIP_OK(hipEventRecord(l1, s2));
        HIP_OK(hipEventRecord(e2, plain));          // the GEMMs alone are done here
        HIP_OK(hipStreamWaitEvent(plain, l1, 0));
        HIP_OK(hipEventRecord(e1, plain));
        HIP_OK(hipEventSynchronize(e1));
        float ms_all, ms_gemm, ms_ln;
        HIP_OK(hipEventElapsedTime(&ms_all, e0, e1));
        HIP_OK(hipEventElapsedTime(&ms_gemm, e0, e2));
        HIP_OK(hipEventElapsedTime(&ms_ln, l0, l1));
        printf("  beside the GEMM, %4d workgroups: %8.1f us per pair (GEMM stream %.1f, LayerNorm stream %.1f; serial %.1f)\n", G,
               ms_all / pairs * 1e3, ms_gemm / pairs * 1e3, ms_ln / pairs * 1e3, (t_gemm + t_ln_full) * 1e3);
      }
      HIP_OK(hipStreamDestroy(s2));
    }
    for (void* p : {(void*)y, (void*)xb, (void*)xf, (void*)gain, (void*)shift, (void*)bias, (void*)A, (void*)W, (void*)C}) HIP_OK(hipFree(p));
  }
  return 0;
}
