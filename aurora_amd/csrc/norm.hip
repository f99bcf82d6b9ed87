// Row-wise layer normalisations of the backbone, all HBM-bound streaming kernels:
//   * layernorm      : out = res + LN(y) * gain + shift        (AdaLN / affine LN + residual)
//   * merge_ln       : 2x2 spatial gather -> LN(4D)            (front half of PatchMerging3D)
//   * split_ln       : pixel shuffle + crop -> LN(D/2)          (middle of PatchSplitting3D)
// One wavefront owns one row: the row lives in registers (8-element / 16- or 32-byte pieces per
// lane, lane-interleaved so every load instruction covers a contiguous 1-2 KiB), statistics are
// two-pass fp32 (mean, then centred second moment) reduced with xor-shuffles.  4 rows per block.
#include <stdlib.h>

#include "common.h"

namespace aurora {
namespace {

constexpr int ROWS_PER_BLOCK = 4;

// Normalise a row held as v[c][0..7] for pieces c = lane + 64*i, i < MAXC.
template <int MAXC>
__device__ __forceinline__ void row_stats(const float (&v)[MAXC][8], int lane, int n_pieces, float inv_d, float eps,
                                          float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (lane + 64 * i < n_pieces) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    }
  mean = wave_sum(s) * inv_d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (lane + 64 * i < n_pieces) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[i][j] - mean;
        q = fmaf(d, d, q);
      }
    }
  rstd = rsqrtf(wave_sum(q) * inv_d + eps);
}

struct LnArgs {
  const void* y; int64_t ldy; const float* gain; const float* shift;
  const float* res; int64_t ldr; int64_t res_mod;
  float* out_f32; int64_t ldo; void* out_t; int64_t ldt;
  int64_t M; int D; float eps;
  int split_t;   // fp32 kernel: out_t receives the fp16-pair layout of the two-term GEMMs (aurora_hip_layernorm_split)
  int split_res; // fp32 kernel: the residual rows are in that layout (value = high half + remainder)
};

// PRE: the residual row is requested together with y (twice the registers; picked for launches with few rows per CU)
template <typename T, int NC, bool PRE = false>   // chunks of 256 features
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const LnArgs p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int n_pieces = p.D >> 2;
  const T* yr = reinterpret_cast<const T*>(p.y) + row * p.ldy;
  const float* rr = nullptr;
  // (a 64-bit remainder is ~100 instructions of a software division: rows fit 32 bits in every launch of the step)
  if (p.res) rr = p.res + (p.res_mod > 0 ? (int64_t)(((p.M | p.res_mod) >> 32) == 0 ? (uint32_t)row % (uint32_t)p.res_mod : row % p.res_mod) : row) * p.ldr;
  const bool res_plain = PRE && rr != nullptr && !p.split_res;   // (uniform)
  float v[NC][4], r4[PRE ? NC : 1][4];
  // Both operands of a row are requested before anything waits: with few rows per CU (a latitude band's coarse stages:
  // 2,160 rows on 256 CUs) the kernel is a latency chain, and the residual fetched only after the two reductions made it
  // two HBM round trips long.
#pragma unroll
  for (int i = 0; i < NC; ++i)
    if (lane + 64 * i < n_pieces) {
      load4(yr + (lane + 64 * i) * 4, v[i]);
      if constexpr (PRE) {
        if (res_plain) load4(rr + (lane + 64 * i) * 4, r4[i]);
      }
    }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
    if (lane + 64 * i < n_pieces) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float inv_d = 1.0f / p.D;
  const float mean = wave_sum(s) * inv_d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
    if (lane + 64 * i < n_pieces) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[i][j] - mean;
        q = fmaf(d, d, q);
      }
    }
  const float rstd = rsqrtf(wave_sum(q) * inv_d + p.eps);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int e = (lane + 64 * i) * 4;
    if (lane + 64 * i < n_pieces) {
      float gn[4], sh[4], o[4];
      if (p.gain) load4(p.gain + e, gn);
      if (p.shift) load4(p.shift + e, sh);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = (v[i][j] - mean) * rstd;
        if (p.gain) t *= gn[j];
        if (p.shift) t += sh[j];
        o[j] = t;
      }
      if (rr && p.split_res) {
        // features e..e+3 of a pair-layout row: four fp16 high halves, their remainders 64 bytes further on
        const char* s = reinterpret_cast<const char*>(rr) + (e & ~31) * 4 + (e & 31) * 2;
        const u32x2 h = *reinterpret_cast<const u32x2*>(s);
        const u32x2 l = *reinterpret_cast<const u32x2*>(s + 64);
        // (halves are taken out of the words by hand: bit-casting the ELEMENTS of the loaded vector to fp16 pairs made hipcc
        // read the first word twice)
        auto half = [](uint32_t w, int hi) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(hi ? w >> 16 : w & 0xffffu)); };
        const uint32_t hw[2] = {h.x, h.y}, lw[2] = {l.x, l.y};
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += half(hw[j >> 1], j & 1) + half(lw[j >> 1], j & 1);
      } else if (rr) {
        if constexpr (PRE) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += r4[i][j];
        } else {
          float r[4];
          load4(rr + e, r);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += r[j];
        }
      }
      if (p.out_f32) store4(p.out_f32 + row * p.ldo + e, o);
      if (p.out_t && p.split_t) {
        // features e..e+3 of group e / 32: four high halves (8 bytes), their remainders 64 bytes further on
        uint32_t h0, h1, l0, l1;
        split_pair_f16(o[0], o[1], h0, l0);
        split_pair_f16(o[2], o[3], h1, l1);
        // Lanes 2i and 2i+1 (features e..e+7 between them) trade halves -- the even lane ends up with the eight high halves,
        // the odd lane with the eight remainders, 16 bytes each -- so that ONE store instruction covers every 128-byte
        // group completely (four lanes the high half of the line, four the low half) instead of two instructions
        // writing half a line each.
        const bool odd = (lane & 1) != 0;
        auto swap1 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true); };   // quad_perm [1,0,3,2]
        const uint32_t r0 = swap1(odd ? h0 : l0), r1 = swap1(odd ? h1 : l1);
        const int e8 = e & ~7;
        char* d = reinterpret_cast<char*>(p.out_t) + (row * p.ldt + (e8 & ~31)) * 4 + (e8 & 31) * 2 + (odd ? 64 : 0);
        *reinterpret_cast<u32x4*>(d) = odd ? u32x4{r0, r1, l0, l1} : u32x4{h0, h1, r0, r1};
      } else if (p.out_t) {
        store4(reinterpret_cast<T*>(p.out_t) + row * p.ldt + e, o);
      }
    }
  }
}

struct MergeArgs {
  const float* x; const float* w; const float* b; void* out;
  int B, C, H, W, D, H2, W2; float eps;
};

template <typename T, int MAXC>
__global__ __launch_bounds__(256) void merge_ln_kernel(const MergeArgs p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int64_t rows = (int64_t)p.B * p.C * p.H2 * p.W2;
  if (row >= rows) return;
  const int w2 = (int)(row % p.W2);
  const int h2 = (int)((row / p.W2) % p.H2);
  const int64_t bc = row / ((int64_t)p.W2 * p.H2);
  const int D4 = 4 * p.D, n_pieces = D4 >> 3;
  float v[MAXC][8];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int e = (lane + 64 * i) * 8;
    if (lane + 64 * i < n_pieces) {
      const int seg = e / p.D, d = e - seg * p.D;  // seg = dh*2 + dw
      const int hh = 2 * h2 + (seg >> 1), ww = 2 * w2 + (seg & 1);
      if (hh < p.H && ww < p.W) {
        load8(p.x + ((bc * p.H + hh) * p.W + ww) * p.D + d, v[i]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = 0.f;  // bottom/right zero padding of odd grids
      }
    }
  }
  float mean, rstd;
  row_stats<MAXC>(v, lane, n_pieces, 1.0f / D4, p.eps, mean, rstd);
  T* orow = reinterpret_cast<T*>(p.out) + row * D4;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int e = (lane + 64 * i) * 8;
    if (lane + 64 * i < n_pieces) {
      float gn[8], sh[8], o[8];
      load8(p.w + e, gn);
      load8(p.b + e, sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * gn[j] + sh[j];
      store8(orow + e, o);
    }
  }
}

struct SplitArgs {
  const void* y; const float* w; const float* b; void* out;
  int B, C, H, W, Dq, Ho, Wo; float eps;
};

template <typename T, int MAXC>
__global__ __launch_bounds__(256) void split_ln_kernel(const SplitArgs p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int64_t rows = (int64_t)p.B * p.C * p.Ho * p.Wo;
  if (row >= rows) return;
  const int wo = (int)(row % p.Wo);
  const int ho = (int)((row / p.Wo) % p.Ho);
  const int64_t bc = row / ((int64_t)p.Wo * p.Ho);
  const int n_pieces = p.Dq >> 3;
  const int seg = (ho & 1) * 2 + (wo & 1);
  const T* src = reinterpret_cast<const T*>(p.y) + ((bc * p.H + (ho >> 1)) * p.W + (wo >> 1)) * (4 * (int64_t)p.Dq) +
                 (int64_t)seg * p.Dq;
  float v[MAXC][8];
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (lane + 64 * i < n_pieces) load8(src + (lane + 64 * i) * 8, v[i]);
  float mean, rstd;
  row_stats<MAXC>(v, lane, n_pieces, 1.0f / p.Dq, p.eps, mean, rstd);
  T* orow = reinterpret_cast<T*>(p.out) + row * p.Dq;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int e = (lane + 64 * i) * 8;
    if (lane + 64 * i < n_pieces) {
      float gn[8], sh[8], o[8];
      load8(p.w + e, gn);
      load8(p.b + e, sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * gn[j] + sh[j];
      store8(orow + e, o);
    }
  }
}

inline unsigned row_blocks(int64_t rows) { return (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK); }

}  // namespace
}  // namespace aurora

using namespace aurora;

// Pick the register-tile instantiation for a row of D elements: MAXC pieces of 8 per lane.
#define AURORA_DISPATCH_ROW(KERNEL, T, D, ...)                                                        \
  do {                                                                                                \
    if ((D) <= 512) hipLaunchKernelGGL((KERNEL<T, 1>), __VA_ARGS__);                                  \
    else if ((D) <= 1024) hipLaunchKernelGGL((KERNEL<T, 2>), __VA_ARGS__);                            \
    else if ((D) <= 2048) hipLaunchKernelGGL((KERNEL<T, 4>), __VA_ARGS__);                            \
    else hipLaunchKernelGGL((KERNEL<T, 8>), __VA_ARGS__);                                             \
  } while (0)

static int layernorm_impl(const void* y, int64_t ldy, const float* gain, const float* shift, const float* res, int64_t ldr,
                          int64_t res_mod, float* out_f32, int64_t ldo, void* out_t, int64_t ldt, int64_t M, int D,
                          float eps, int dtype, int split_t, int split_res, void* stream);

extern "C" int aurora_hip_layernorm(const void* y, int64_t ldy, const float* gain, const float* shift,
                                    const float* res, int64_t ldr, int64_t res_mod, float* out_f32,
                                    int64_t ldo, void* out_t, int64_t ldt, int64_t M, int D, float eps,
                                    int dtype, void* stream) {
  return layernorm_impl(y, ldy, gain, shift, res, ldr, res_mod, out_f32, ldo, out_t, ldt, M, D, eps, dtype, 0, 0, stream);
}

extern "C" int aurora_hip_layernorm_split(const float* y, int64_t ldy, const float* gain, const float* shift,
                                          const void* res, int64_t ldr, int64_t res_mod, int res_is_split, float* out_f32,
                                          int64_t ldo, void* out_split, int64_t ld_split, int64_t M, int D, float eps,
                                          void* stream) {
  AURORA_CHECK_ARG(D % 32 == 0, "layernorm_split: D=%d must be a multiple of 32", D);
  AURORA_CHECK_ARG(out_split == nullptr || (ld_split % 32 == 0 && ld_split >= D),
                   "layernorm_split: the pair-layout output stride must be a multiple of 32, >= D");
  AURORA_CHECK_ARG(!res_is_split || (res != nullptr && ldr % 32 == 0 && ldr >= D),
                   "layernorm_split: the pair-layout residual stride must be a multiple of 32, >= D");
  return layernorm_impl(y, ldy, gain, shift, static_cast<const float*>(res), ldr, res_mod, out_f32, ldo, out_split, ld_split,
                        M, D, eps, AURORA_F32, out_split ? 1 : 0, res_is_split ? 1 : 0, stream);
}

static int layernorm_impl(const void* y, int64_t ldy, const float* gain, const float* shift, const float* res, int64_t ldr,
                          int64_t res_mod, float* out_f32, int64_t ldo, void* out_t, int64_t ldt, int64_t M, int D,
                          float eps, int dtype, int split_t, int split_res, void* stream) {
  AURORA_CHECK_ARG(dtype == AURORA_F32 || dtype == AURORA_BF16, "layernorm: bad dtype");
  AURORA_CHECK_ARG(D % 8 == 0 && D > 0 && D <= 4096, "layernorm: D=%d must be a multiple of 8, <= 4096", D);
  const int es = dtype == AURORA_F32 ? 4 : 2;
  AURORA_CHECK_ARG((ldy * es) % 16 == 0 && (uintptr_t)y % 16 == 0, "layernorm: unaligned input rows");
  AURORA_CHECK_ARG((!res || ((ldr * 4) % 16 == 0 && (uintptr_t)res % 16 == 0)) &&
                       (!out_f32 || ((ldo * 4) % 16 == 0 && (uintptr_t)out_f32 % 16 == 0)) &&
                       (!out_t || ((ldt * es) % 16 == 0 && (uintptr_t)out_t % 16 == 0)),
                   "layernorm: unaligned residual/output rows");
  AURORA_CHECK_ARG(out_f32 || out_t, "layernorm: no output");
  if (M <= 0) return AURORA_OK;
  LnArgs p{y, ldy, gain, shift, res, ldr, res_mod, out_f32, ldo, out_t, ldt, M, D, eps, split_t, split_res};
  if (dtype == AURORA_F32) {
    const dim3 grid(row_blocks(M)), block(256);
    if (D <= 256) hipLaunchKernelGGL((layernorm_f32_kernel<float, 1>), grid, block, 0, as_stream(stream), p);
    else if (D <= 512) hipLaunchKernelGGL((layernorm_f32_kernel<float, 2>), grid, block, 0, as_stream(stream), p);
    else if (D <= 1024) hipLaunchKernelGGL((layernorm_f32_kernel<float, 4>), grid, block, 0, as_stream(stream), p);
    else if (D <= 2048) hipLaunchKernelGGL((layernorm_f32_kernel<float, 8>), grid, block, 0, as_stream(stream), p);
    else hipLaunchKernelGGL((layernorm_f32_kernel<float, 16>), grid, block, 0, as_stream(stream), p);
  } else {
    // bf16 rows go through the same 4-features-per-lane kernel: every access of a wave covers whole cache lines (8-byte
    // bf16 pieces, 16-byte fp32 pieces on consecutive lanes; profiles/r02_ab_ln_layout.log)
    const dim3 grid(row_blocks(M)), block(256);
    // few rows per CU (a latitude band's stages, the coarse stage of the un-sharded step): the latency chain of a row
    // bounds the launch, and the variant that requests the residual row up front halves it
    // (A/B inside the step, profiles/r04_ab_gn_lnprefetch_instep.log: forced on everywhere LayerNorm 11.2 -> 11.4 ms per
    // un-sharded step -- twice the registers halve the waves of the big launches --, on a rank of eight 3.05 -> 2.86 ms)
    const bool pre = res != nullptr && M <= (int64_t)40 * device_cus();   // (16,200 rows, the un-sharded stage 2: 74.5 -> 79.0 us with it)
    if (pre) {
      if (D <= 256) hipLaunchKernelGGL((layernorm_f32_kernel<bf16_t, 1, true>), grid, block, 0, as_stream(stream), p);
      else if (D <= 512) hipLaunchKernelGGL((layernorm_f32_kernel<bf16_t, 2, true>), grid, block, 0, as_stream(stream), p);
      else if (D <= 1024) hipLaunchKernelGGL((layernorm_f32_kernel<bf16_t, 4, true>), grid, block, 0, as_stream(stream), p);
      else if (D <= 2048) hipLaunchKernelGGL((layernorm_f32_kernel<bf16_t, 8, true>), grid, block, 0, as_stream(stream), p);
      else hipLaunchKernelGGL((layernorm_f32_kernel<bf16_t, 16>), grid, block, 0, as_stream(stream), p);
    } else {
      if (D <= 256) hipLaunchKernelGGL((layernorm_f32_kernel<bf16_t, 1>), grid, block, 0, as_stream(stream), p);
      else if (D <= 512) hipLaunchKernelGGL((layernorm_f32_kernel<bf16_t, 2>), grid, block, 0, as_stream(stream), p);
      else if (D <= 1024) hipLaunchKernelGGL((layernorm_f32_kernel<bf16_t, 4>), grid, block, 0, as_stream(stream), p);
      else if (D <= 2048) hipLaunchKernelGGL((layernorm_f32_kernel<bf16_t, 8>), grid, block, 0, as_stream(stream), p);
      else hipLaunchKernelGGL((layernorm_f32_kernel<bf16_t, 16>), grid, block, 0, as_stream(stream), p);
    }
  }
  return check_launch("layernorm");
}

extern "C" int aurora_hip_merge_ln(const float* x, const float* ln_w, const float* ln_b, void* out, int B, int C,
                                   int H, int W, int D, float eps, int dtype, void* stream) {
  AURORA_CHECK_ARG(dtype == AURORA_F32 || dtype == AURORA_BF16, "merge_ln: bad dtype");
  AURORA_CHECK_ARG(D % 8 == 0 && 4 * D <= 4096, "merge_ln: D=%d must be a multiple of 8 with 4D <= 4096", D);
  AURORA_CHECK_ARG(H >= 1 && W >= 1, "merge_ln: empty grid %dx%d", H, W);  // a band may own a single (odd) row
  MergeArgs p{x, ln_w, ln_b, out, B, C, H, W, D, (H + 1) / 2, (W + 1) / 2, eps};
  const int64_t rows = (int64_t)B * C * p.H2 * p.W2;
  if (dtype == AURORA_F32)
    AURORA_DISPATCH_ROW(merge_ln_kernel, float, 4 * D, dim3(row_blocks(rows)), dim3(256), 0, as_stream(stream), p);
  else
    AURORA_DISPATCH_ROW(merge_ln_kernel, bf16_t, 4 * D, dim3(row_blocks(rows)), dim3(256), 0, as_stream(stream), p);
  return check_launch("merge_ln");
}

extern "C" int aurora_hip_split_ln(const void* y, const float* ln_w, const float* ln_b, void* out, int B, int C,
                                   int H, int W, int Dq, int crop_h, int crop_w, float eps, int dtype,
                                   void* stream) {
  AURORA_CHECK_ARG(dtype == AURORA_F32 || dtype == AURORA_BF16, "split_ln: bad dtype");
  AURORA_CHECK_ARG(Dq % 8 == 0 && Dq <= 4096, "split_ln: Dq=%d must be a multiple of 8, <= 4096", Dq);
  AURORA_CHECK_ARG(crop_h >= 0 && crop_h <= 1 && crop_w >= 0 && crop_w <= 1, "split_ln: crop must be 0 or 1");
  SplitArgs p{y, ln_w, ln_b, out, B, C, H, W, Dq, 2 * H - crop_h, 2 * W - crop_w, eps};
  const int64_t rows = (int64_t)B * C * p.Ho * p.Wo;
  if (dtype == AURORA_F32)
    AURORA_DISPATCH_ROW(split_ln_kernel, float, Dq, dim3(row_blocks(rows)), dim3(256), 0, as_stream(stream), p);
  else
    AURORA_DISPATCH_ROW(split_ln_kernel, bf16_t, Dq, dim3(row_blocks(rows)), dim3(256), 0, as_stream(stream), p);
  return check_launch("split_ln");
}
