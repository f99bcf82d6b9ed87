// The decoder's level de-aggregation, re-associated (perceiver.py:141-152 as used by decoder.py:156-166, 225-231).
//
// Upstream, per grid column: 13 level queries attend to 3 latent keys (16 heads x 64), the 13 x 1024 attention output goes
// through `to_out` (1024 x 1024).  At 0.25 degree that is a 842,400 x 1024 x 1024 fp32-grade GEMM behind a 3.45 GB round
// trip of the attention output.  But a column has only THREE value rows, and the level queries are the same for every
// column, so
//     to_out(concat_h sum_j p[l,h,j] v[j,h,:])  =  sum_h sum_j p[l,h,j] (W_out[:, h] v[j,h,:])
// i.e. project the 3 values of a column per head once (194,400 rows instead of 842,400: 0.41 instead of 1.77 TFLOP
// fp32-equivalent) and take the 13 x 48 convex combinations per column in registers.  Two kernels:
//
//   perceiver_probs_kernel   per (column, head): the 13 x 3 softmax weights p (fp32) -> P[col][head][l][j (4)], and the
//                            column's value rows re-written in the fp16-pair layout of the two-term GEMMs (gemm.hip,
//                            "The fp16-pair layout") -> Vp[col * 3 + j][inner].  HBM-bound: reads k | v once.
//   perceiver_out_kernel     U_h = Vp_h . W_out_h^T on the matrix pipe (two fp16 terms, three MFMAs per product, weights
//                            pre-split and scaled by 2^6 as everywhere), out[l] += p[l,h,j] U_h[j] on the VALU, 16 heads,
//                            then one store of the fp32 rows LayerNorm 1 reads.  The attention output never exists.
//
// perceiver_out_kernel, per workgroup: 32 columns (128 operand rows: row 4 c + j, j = 3 unused) x 128 output features;
// 8 waves as 4 (column groups of 8) x 2 (64 features).  The MFMA "A" operand is the VALUE tile, so lane (g = lane >> 4,
// i = lane & 15) of a 16 x 16 result holds rows 4 g .. 4 g + 3 = the three keys of ONE column for feature i: the combine
// needs no cross-lane traffic for U.  The weights p of that column arrive through LDS one value per lane and are
// broadcast inside the 16-lane row by DPP (v_fmac_f32_dpp ... row_newbcast): 39 FMAs per U tile, no moves.
// Weight rows are interleaved (LDS row 16 nt + i <-> feature 4 i + nt) so that a lane ends up with 4 CONSECUTIVE features
// per (column, level): the result leaves as 16-byte stores covering 256 contiguous bytes per column and level.
// K runs over the heads: a head is two K-stages of 32 (128 bytes per operand row in the pair layout), staged by LDS-DMA
// into a ring of four stages (two heads) + the head's 8 KiB tile of P, all by counted waits.
// Schedule (ping-pong, as linear_kernel_f32pp): per head four phases  X_A  M_A  X_B  M_B  with a barrier after each;
//   M_A / M_B: 24 MFMAs each -- the products of feature fragments 0, 1 / 2, 3 --, nothing else;
//   X_A: fragment reads for M_A, combine of the PREVIOUS head's fragments 2, 3, LDS-DMA of the next head;
//   X_B: fragment reads for M_B, this head's P, combine of fragments 0, 1, wait for the next head's pieces.
// Waves 4-7 run one phase behind waves 0-3 (their SIMD partners): a SIMD's matrix pipe belongs to one wave while the
// other does the LDS / VALU work.
#include <type_traits>

#include "common.h"

namespace aurora {

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

constexpr int PO_THREADS = 512, PO_COLS = 32, PO_N = 128, PO_ROWB = 128;
constexpr int PO_OPER = 128 * PO_ROWB;      // 16 KiB per operand and K-stage
constexpr int PO_STAGE = 2 * PO_OPER;       // values | weights
constexpr int PO_NST = 4;                   // two heads
constexpr int PO_PS = 64;                   // floats of P per (column, head): [l][4], padded (Lq <= 16)
constexpr int PO_PTILE = PO_COLS * PO_PS * 4;   // 8 KiB: the 32 columns' weights of one head
constexpr int PO_LDS = PO_NST * PO_STAGE + 2 * PO_PTILE;   // 144 KiB

struct PercOutArgs {
  const char* V; int64_t ldv_b;     // value rows in the fp16-pair layout: row col * 3 + j
  const char* W; int64_t ldw_b;     // weights [N][inner] in the fp16-pair layout, scaled by 2^6
  const float* P;                   // [n_cols][heads][PO_PS]
  const float* bias;                // [N] or null
  float* out; int64_t ldo;          // rows col * LQ + l
  int64_t n_cols; int N, heads, tiles_n; int64_t n_blocks;
  const float* guard; float guard_limit;   // runs iff *guard < guard_limit (null: always)
};

__device__ __forceinline__ f32x4 mma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// acc += p[lane K of this lane's row of 16] * u  (one VALU instruction)
template <int K>
__device__ __forceinline__ void fmac_bcast(float& acc, float p, float u) {
  asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(p), "v"(u), "i"(K));
}

// XCD-aware order: each XCD owns a contiguous range of tiles, the n-tiles of a column tile adjacent (its values are then
// fetched from HBM once and re-used out of that XCD's L2).  Bijective for any tile count.
__device__ __forceinline__ uint32_t xcd_logical(uint32_t bid, uint32_t nb) {
  const uint32_t q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
}

template <int LQ>
__global__ __launch_bounds__(PO_THREADS, 2) void perceiver_out_kernel(const PercOutArgs p) {
  static_assert(LQ >= 1 && LQ * 4 <= PO_PS, "level queries per column");
  constexpr int LK = 3;
  constexpr int PREGS = (LQ * 4 + 15) / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (p.guard != nullptr && !(*p.guard < p.guard_limit)) return;   // (uniform)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int late = wave >> 2;          // waves w and w + 4 share a SIMD
  const int wm = wave & 3, wn = wave >> 2;
  const uint32_t logical = xcd_logical(blockIdx.x, (uint32_t)p.n_blocks);
  const uint32_t tile_m = logical / (uint32_t)p.tiles_n, tile_n = logical - tile_m * (uint32_t)p.tiles_n;
  const int64_t col0 = (int64_t)tile_m * PO_COLS;
  const int n0 = (int)tile_n * PO_N;
  const int heads = p.heads;

  // ---- LDS-DMA sources: per K-stage 2 pieces of the value tile, 2 of the weight tile; per head 1 piece of P.
  //      Uniform 64-bit bases (scalar registers) + one 32-bit offset per lane and piece. ----
  const int64_t cols_here = p.n_cols - col0 < PO_COLS ? p.n_cols - col0 : PO_COLS;   // (uniform) columns of this tile that exist
  const char* const base_v = p.V + col0 * LK * p.ldv_b;
  const char* const base_w = p.W + (int64_t)n0 * p.ldw_b;
  const char* const base_p = reinterpret_cast<const char*>(p.P) + col0 * heads * (PO_PS * 4);
  uint32_t vo_v[2], vo_w[2], vo_p;
  // (row 4 c + 3 of a column does not exist: its lanes stay out of the DMA, what LDS holds there is never used.  The same
  //  lanes in both pieces: piece r covers rows 64 r + (tid >> 3))
  const bool v_on = ((tid >> 3) & 3) < LK;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = r * 64 + (tid >> 3), c = tid & 7;
    const int j = row & 3;
    int cl = row >> 2;
    cl = cl < (int)cols_here ? cl : (int)cols_here - 1;
    vo_v[r] = (uint32_t)((cl * LK + (j < LK ? j : 0)) * (int)p.ldv_b + ((c ^ (row & 7)) << 4));
    // LDS row t of the weight tile holds feature 64 (t >> 6) + 4 (t & 15) + ((t >> 4) & 3)
    const int feat = (row & 64) + 4 * (row & 15) + ((row >> 4) & 3);
    vo_w[r] = (uint32_t)(feat * (int)p.ldw_b + ((c ^ (row & 7)) << 4));
  }
  {
    int cl = tid >> 4;
    cl = cl < (int)cols_here ? cl : (int)cols_here - 1;
    vo_p = (uint32_t)(cl * heads * (PO_PS * 4) + (tid & 15) * 16);
  }
  char* const pbase = smem + PO_NST * PO_STAGE;
  auto stage_head = [&](int h) {   // both K-stages of head h and its tile of P: 9 LDS-DMA instructions per lane
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      char* base = smem + (((2 * h + ks) & (PO_NST - 1))) * PO_STAGE;
      const int64_t koff = (int64_t)(2 * h + ks) * PO_ROWB;
      if (v_on) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_v + koff + vo_v[r]),
                                           (lds_ptr_t)(base + (r * PO_THREADS + wave * 64) * 16), 16, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 2; ++r)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_w + koff + vo_w[r]),
                                         (lds_ptr_t)(base + PO_OPER + (r * PO_THREADS + wave * 64) * 16), 16, 0, 0);
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_p + (int64_t)h * (PO_PS * 4) + vo_p),
                                     (lds_ptr_t)(pbase + (h & 1) * PO_PTILE + wave * 1024), 16, 0, 0);
  };

  // ---- fragment read offsets inside a K-stage (pair layout: chunk g = high halves of k = 8g..8g+7, chunk g + 4 = remainders).
  //      The swizzle of a row depends on row & 7 = i16 & 7 only: the fragments of an operand are one address + constants. ----
  const int i16 = lane & 15, g = lane >> 4;
  int off_v[2], off_w[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    off_v[t] = (wm * 32 + i16) * PO_ROWB + (((g + 4 * t) ^ (i16 & 7)) << 4);
    off_w[t] = PO_OPER + (wn * 64 + i16) * PO_ROWB + (((g + 4 * t) ^ (i16 & 7)) << 4);
  }
  // this lane's share of its columns' weights: value 16 c + i16 of column wm * 8 + 4 mt + g
  const int off_p = (wm * 8 + g) * (PO_PS * 4) + i16 * 4;

  float out[2][4][LQ];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int l = 0; l < LQ; ++l) out[mt][nt][l] = 0.f;
  // (head 0's X_A combines "the previous head": zeros times zeros -- a branch around it would make hipcc keep two copies of
  //  the 104 accumulators and move them every iteration)
  f32x4 U[2][4];
  float pr[2][PREGS];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) U[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < PREGS; ++c) pr[mt][c] = 0.f;
  }

  struct Frags { u32x4 vh[2][2], vl[2][2], wh[2][2], wl[2][2]; };   // [mt][ks], [nt - nt0][ks]
  auto read_frags = [&](int h, auto NT0, Frags& f) {
    constexpr int nt0 = decltype(NT0)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const char* buf = smem + ((2 * h + ks) & (PO_NST - 1)) * PO_STAGE;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f.vh[mt][ks] = *reinterpret_cast<const u32x4*>(buf + off_v[0] + mt * 16 * PO_ROWB);
        f.vl[mt][ks] = *reinterpret_cast<const u32x4*>(buf + off_v[1] + mt * 16 * PO_ROWB);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f.wh[q][ks] = *reinterpret_cast<const u32x4*>(buf + off_w[0] + (nt0 + q) * 16 * PO_ROWB);
        f.wl[q][ks] = *reinterpret_cast<const u32x4*>(buf + off_w[1] + (nt0 + q) * 16 * PO_ROWB);
      }
    }
  };
  // 24 MFMAs: smallest terms first; consecutive MFMAs go to different accumulators
  auto matrix = [&](auto NT0, const Frags& f) {
    constexpr int nt0 = decltype(NT0)::value;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 2; ++q) U[mt][nt0 + q] = mma_f16(f.vh[mt][0], f.wl[q][0], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 2; ++q) U[mt][nt0 + q] = mma_f16(f.vl[mt][0], f.wh[q][0], U[mt][nt0 + q]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 2; ++q) U[mt][nt0 + q] = mma_f16(f.vh[mt][1], f.wl[q][1], U[mt][nt0 + q]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 2; ++q) U[mt][nt0 + q] = mma_f16(f.vl[mt][1], f.wh[q][1], U[mt][nt0 + q]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 2; ++q) U[mt][nt0 + q] = mma_f16(f.vh[mt][0], f.wh[q][0], U[mt][nt0 + q]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 2; ++q) U[mt][nt0 + q] = mma_f16(f.vh[mt][1], f.wh[q][1], U[mt][nt0 + q]);
  };
  // out[l] += p[l][j] U[j] for the two feature fragments nt0, nt0 + 1 (weights of the head in `pr`)
  auto combine = [&](auto NT0) {
    constexpr int nt0 = decltype(NT0)::value;
    // (the asm FMAs are invisible to hipcc's hazard recogniser: a matrix result may be read 11 states after its MFMA issued)
    asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int j = 0; j < LK; ++j)   // (key outermost: an accumulator comes back 2 LQ instructions later, not 2)
#pragma unroll
        for (int l = 0; l < LQ; ++l)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int idx = l * 4 + j;
            float& acc = out[mt][nt0 + q][l];
            const float u = U[mt][nt0 + q][j];
            switch (idx & 15) {   // (compile-time after unrolling)
              case 0: fmac_bcast<0>(acc, pr[mt][idx >> 4], u); break;
              case 1: fmac_bcast<1>(acc, pr[mt][idx >> 4], u); break;
              case 2: fmac_bcast<2>(acc, pr[mt][idx >> 4], u); break;
              case 4: fmac_bcast<4>(acc, pr[mt][idx >> 4], u); break;
              case 5: fmac_bcast<5>(acc, pr[mt][idx >> 4], u); break;
              case 6: fmac_bcast<6>(acc, pr[mt][idx >> 4], u); break;
              case 8: fmac_bcast<8>(acc, pr[mt][idx >> 4], u); break;
              case 9: fmac_bcast<9>(acc, pr[mt][idx >> 4], u); break;
              case 10: fmac_bcast<10>(acc, pr[mt][idx >> 4], u); break;
              case 12: fmac_bcast<12>(acc, pr[mt][idx >> 4], u); break;
              case 13: fmac_bcast<13>(acc, pr[mt][idx >> 4], u); break;
              default: fmac_bcast<14>(acc, pr[mt][idx >> 4], u); break;
            }
          }
  };

  const std::integral_constant<int, 0> N0{};
  const std::integral_constant<int, 2> N2{};
  // ---- prologue: heads 0 and 1 on their way; head 0 published ----
  stage_head(0);
  if (heads > 1) stage_head(1);
  // (the wave's own outstanding pieces: counted per lane in issue order; a lane with a skipped value piece has fewer --
  //  the count is the lane's own, the wait below is the conservative one: everything of head 0)
  if (heads > 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (late == 1) __builtin_amdgcn_s_barrier();   // the late half: one phase behind from here on

  for (int h = 0; h < heads; ++h) {
    Frags f;
    // ---- X_A(h) ----
    read_frags(h, N0, f);
    combine(N2);                                  // fragments 2, 3 of head h - 1 (weights still in pr)
    if (h >= 1 && h + 1 < heads) stage_head(h + 1);   // into the slots of head h - 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    // ---- M_A(h) ----
    matrix(N0, f);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    // ---- X_B(h) ----
    read_frags(h, N2, f);
    {
      const char* pb = pbase + (h & 1) * PO_PTILE;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int c = 0; c < PREGS; ++c) pr[mt][c] = *reinterpret_cast<const float*>(pb + off_p + mt * 4 * (PO_PS * 4) + c * 64);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    combine(N0);                                   // fragments 0, 1 of head h
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of head h + 1 (issued two phases ago) have landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    // ---- M_B(h) ----
    matrix(N2, f);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  }
  combine(N2);   // fragments 2, 3 of the last head
  if (late == 0) __builtin_amdgcn_s_barrier();   // (every wave passes the same number of barriers)

  // ---- result: undo the 2^6 weight scale (exact), bias; 16 bytes per (column, level): features n .. n + 3 ----
  const int n = n0 + wn * 64 + 4 * i16;
  f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int64_t col = col0 + wm * 8 + 4 * mt + g;
    if (col >= p.n_cols) continue;
    float* row = p.out + col * LQ * p.ldo + n;
#pragma unroll
    for (int l = 0; l < LQ; ++l) {
      const f32x4 v = {fmaf(out[mt][0][l], 0.015625f, b4.x), fmaf(out[mt][1][l], 0.015625f, b4.y),
                       fmaf(out[mt][2][l], 0.015625f, b4.z), fmaf(out[mt][3][l], 0.015625f, b4.w)};
      *reinterpret_cast<f32x4*>(row + (int64_t)l * p.ldo) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Softmax weights of the level queries over a column's (three) keys, and the column's values as fp16 pairs.
// A group of 16 adjacent lanes owns one (column, head): a lane holds 4 of the head's 64 features (the layout of
// perceiver_attention_kernel, embed.hip: a group reads a key / value row as one contiguous 256-byte run).
// ------------------------------------------------------------------------------------------------------------------
struct PercProbArgs {
  const float* q; const float* kv; float* P; char* Vp;
  int B; int64_t cols_per_b, kv_bstride, kv_lstride; int heads;
  const float* guard; float guard_limit;
};

__device__ __forceinline__ float group16_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror: i <-> 15 - i
  v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror: i <-> 7 - i
  v += dpp(v, std::integral_constant<int, 0x1B>{});    // quad_perm [3,2,1,0]
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  return v;
}

template <int LQ>
__global__ __launch_bounds__(256) void perceiver_probs_kernel(const PercProbArgs p) {
  constexpr int HDIM = 64, LK = 3;
  if (p.guard != nullptr && !(*p.guard < p.guard_limit)) return;   // (uniform)
  const int inner = p.heads * HDIM;
  const int64_t n_cols = (int64_t)p.B * p.cols_per_b;
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int i16 = (int)(threadIdx.x & 15);
  const int d0 = i16 * 4;
  if (grp >= n_cols * p.heads) return;   // (whole groups leave together)
  const int h = (int)(grp % p.heads);
  const int64_t col = grp / p.heads;
  const int b = (int)(col / p.cols_per_b);
  const int64_t l = col - (int64_t)b * p.cols_per_b;
  const float scale = 0.125f;   // 1 / sqrt(64)
  const float* kv0 = p.kv + (b * p.kv_bstride + l) * (2 * (int64_t)inner) + h * HDIM + d0;
  const int64_t kv_step = p.kv_lstride * (2 * (int64_t)inner);
  float kk[LK][4], vv[LK][4];
#pragma unroll
  for (int j = 0; j < LK; ++j) {
    load4(kv0 + j * kv_step, kk[j]);
    load4(kv0 + j * kv_step + inner, vv[j]);
  }
  // ---- the weights: every lane of the group ends up with all of them; lane i keeps those of query i ----
  float mine[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < LQ; ++c) {
    float qv[4];
    load4(p.q + (int64_t)c * inner + h * HDIM + d0, qv);
    float sc[LK];
#pragma unroll
    for (int j = 0; j < LK; ++j)
      sc[j] = group16_sum(fmaf(qv[0], kk[j][0], fmaf(qv[1], kk[j][1], fmaf(qv[2], kk[j][2], qv[3] * kk[j][3])))) * scale;
    const float m = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
    float e[LK];
#pragma unroll
    for (int j = 0; j < LK; ++j) e[j] = __expf(sc[j] - m);
    const float inv = 1.0f / ((e[0] + e[1]) + e[2]);
    if (i16 == c) {
      mine[0] = e[0] * inv;
      mine[1] = e[1] * inv;
      mine[2] = e[2] * inv;
    }
  }
  float* prow = p.P + (col * p.heads + h) * PO_PS;
  if (i16 * 4 < PO_PS) {
    if (i16 >= LQ) mine[0] = mine[1] = mine[2] = 0.f;
    store4(prow + i16 * 4, mine);
  }
  // ---- the values as fp16 pairs: lanes 2i, 2i + 1 hold 8 consecutive features between them and trade halves; the even
  //      lane stores the eight high halves, the odd one the remainders (the layout of aurora_hip_split_f16) ----
  const bool odd = (threadIdx.x & 1) != 0;
  auto swap1 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true); };
  const int f8 = (h * HDIM + d0) & ~7;
#pragma unroll
  for (int j = 0; j < LK; ++j) {
    uint32_t h0, h1, l0, l1;
    split_pair_f16(vv[j][0], vv[j][1], h0, l0);
    split_pair_f16(vv[j][2], vv[j][3], h1, l1);
    const uint32_t s0 = swap1(odd ? h0 : l0), s1 = swap1(odd ? h1 : l1);
    char* d = p.Vp + ((col * LK + j) * inner + (f8 & ~31)) * 4 + (f8 & 31) * 2 + (odd ? 64 : 0);
    *reinterpret_cast<u32x4*>(d) = odd ? u32x4{s0, s1, l0, l1} : u32x4{h0, h1, s0, s1};
  }
}

}  // namespace

}  // namespace aurora

using namespace aurora;

extern "C" int aurora_hip_perceiver_out_supported(int Lq, int Lk, int heads, int head_dim, int N) {
  return (Lq == 3 || Lq == 4 || Lq == 13) && Lk == 3 && head_dim == 64 && heads >= 1 && N > 0 && N % PO_N == 0;
}

extern "C" int aurora_hip_perceiver_probs(const float* q, const float* kv, float* P, void* Vp, int B, int64_t cols_per_b,
                                          int64_t kv_bstride, int64_t kv_lstride, int Lq, int Lk, int heads, int head_dim,
                                          const float* guard, float guard_limit, void* stream) {
  AURORA_CHECK_ARG(aurora_hip_perceiver_out_supported(Lq, Lk, heads, head_dim, PO_N),
                   "perceiver_probs: Lq=%d Lk=%d head_dim=%d (built for Lq in {3, 4, 13}, Lk = 3, head_dim 64)", Lq, Lk, head_dim);
  AURORA_CHECK_ARG(q && kv && P && Vp && B > 0 && cols_per_b > 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)kv % 16) == 0 &&
                       ((uintptr_t)P % 16) == 0 && ((uintptr_t)Vp % 16) == 0,
                   "perceiver_probs: null / unaligned argument");
  PercProbArgs p{q, kv, P, (char*)Vp, B, cols_per_b, kv_bstride, kv_lstride, heads, guard, guard_limit};
  const int64_t items = (int64_t)B * cols_per_b * heads * 16;
  const dim3 grid((unsigned)((items + 255) / 256)), block(256);
  switch (Lq) {
    case 3: hipLaunchKernelGGL(perceiver_probs_kernel<3>, grid, block, 0, as_stream(stream), p); break;
    case 4: hipLaunchKernelGGL(perceiver_probs_kernel<4>, grid, block, 0, as_stream(stream), p); break;
    default: hipLaunchKernelGGL(perceiver_probs_kernel<13>, grid, block, 0, as_stream(stream), p); break;
  }
  return check_launch("perceiver_probs");
}

extern "C" int aurora_hip_perceiver_out(const void* Vp, const void* W_pairs, int64_t ldw, const float* P, const float* bias,
                                        float* out, int64_t ldo, int64_t n_cols, int Lq, int Lk, int heads, int head_dim, int N,
                                        const float* guard, float guard_limit, void* stream) {
  AURORA_CHECK_ARG(aurora_hip_perceiver_out_supported(Lq, Lk, heads, head_dim, N),
                   "perceiver_out: Lq=%d Lk=%d head_dim=%d N=%d (built for Lq in {3, 4, 13}, Lk = 3, head_dim 64, N %% 128 == 0)", Lq,
                   Lk, head_dim, N);
  const int inner = heads * head_dim;
  AURORA_CHECK_ARG(Vp && W_pairs && P && out && n_cols > 0 && ldw >= inner && ldw % 32 == 0 && ldo >= N && ldo % 4 == 0 &&
                       ((uintptr_t)Vp % 16) == 0 && ((uintptr_t)W_pairs % 16) == 0 && ((uintptr_t)P % 16) == 0 &&
                       ((uintptr_t)out % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0),
                   "perceiver_out: strides / alignment");
  PercOutArgs p{(const char*)Vp, (int64_t)inner * 4, (const char*)W_pairs, ldw * 4, P, bias, out, ldo, n_cols, N, heads, N / PO_N, 0,
                guard, guard_limit};
  p.n_blocks = ((n_cols + PO_COLS - 1) / PO_COLS) * p.tiles_n;
  AURORA_CHECK_ARG(p.n_blocks < (int64_t)1 << 31, "perceiver_out: too many tiles");
  static bool attr_done_dev[64] = {false};
  bool& attr_done = attr_done_dev[current_device() & 63];
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)perceiver_out_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, PO_LDS);
    (void)hipFuncSetAttribute((const void*)perceiver_out_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, PO_LDS);
    (void)hipFuncSetAttribute((const void*)perceiver_out_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, PO_LDS);
    attr_done = true;
  }
  const dim3 grid((unsigned)p.n_blocks), block(PO_THREADS);
  switch (Lq) {
    case 3: hipLaunchKernelGGL(perceiver_out_kernel<3>, grid, block, PO_LDS, as_stream(stream), p); break;
    case 4: hipLaunchKernelGGL(perceiver_out_kernel<4>, grid, block, PO_LDS, as_stream(stream), p); break;
    default: hipLaunchKernelGGL(perceiver_out_kernel<13>, grid, block, PO_LDS, as_stream(stream), p); break;
  }
  return check_launch("perceiver_out");
}
