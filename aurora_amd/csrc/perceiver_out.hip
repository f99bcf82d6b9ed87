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
//   perceiver_probs_kernel   per (column, head): the 13 x 3 softmax weights p (fp32) -> P[col][head] (16 pairs, see there), and
//                            the column's value rows re-written in the fp16-pair layout of the two-term GEMMs (gemm.hip,
//                            "The fp16-pair layout") -> Vp[col * 3 + j][inner].  HBM-bound: reads k | v once.
//   perceiver_out_kernel     U_h = Vp_h . W_out_h^T on the matrix pipe (two fp16 terms, three MFMAs per product, weights
//                            pre-split and scaled by 2^6 as everywhere), out[l] += p[l,h,j] U_h[j] on the VALU, 16 heads,
//                            then one store of the fp32 rows LayerNorm 1 reads.  The attention output never exists.
//
// perceiver_out_kernel, per workgroup: 32 columns (128 operand rows: row 4 c + j, j = 3 unused) x 128 output features;
// 8 waves as 4 (column groups of 8) x 2 (64 features).  The MFMA "A" operand is the VALUE tile, so lane (g = lane >> 4,
// i = lane & 15) of a 16 x 16 result holds rows 4 g .. 4 g + 3 = the three keys of ONE column for feature i: the combine
// needs no cross-lane traffic for U.  The weights p of that column arrive through LDS one PAIR (two levels, one key) per
// lane and are broadcast inside the 16-lane row by a 64-bit DPP move; the FMAs are packed (two levels per instruction).
// Weight rows are interleaved (LDS row 16 nt + i <-> feature 4 i + nt) so that a lane ends up with 4 CONSECUTIVE features
// per (column, level): the result leaves as 16-byte stores covering 256 contiguous bytes per column and level.
// K runs over the heads: a head is two K-stages of 32 (128 bytes per operand row in the pair layout), staged by LDS-DMA
// into a ring of four stages (two heads) + the head's 8 KiB tile of P, all by counted waits.
// Schedule: per K-stage two slots with a barrier after each --
//   R: the stage's 12 fragment reads (+ the head's weights), the LDS-DMA of stage st + 3, the counted wait;
//   C: 24 MFMAs, the previous head's combine for one row fragment dealt out between them;
// waves 4-7 run one slot behind waves 0-3 (their SIMD partners): one computes while the other reads.
// What bounds it (tools/probes/mfma_valu_mix.py, mfma_valu_overlap.hip, profiles/r06_perceiver_out_dev.log): an MFMA holds
// the SIMD's VALU issue port for 8 of its 16 cycles, whichever wave the next VALU instruction comes from, so per SIMD and
// K-stage 48 MFMAs x 8 + 2 x 70 combine instructions x ~5.9 ~ 1,200 cycles are the floor of this formulation (the matrix pipe
// alone: 768); the schedule above runs at ~1,700.
#include <type_traits>
#include <utility>

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

#ifdef PO_STAMPS   // probe build (AURORA_BUILD_FLAGS=-DPO_STAMPS): tools/probes/po_stamps.py
__device__ uint32_t po_stamps[8 * 256];
#endif

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

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The combine runs on PACKED fp32 FMAs (v_pk_fma_f32: two FMAs per lane and instruction at the rate of one plain FMA -- the
// VALU, not the matrix pipe, bounds this kernel: 2 x 13 x 4 FMAs per product tile against three MFMAs), two LEVELS per
// instruction:  (a[nt].lo, a[nt].hi) += u[nt][J] * (w.lo, w.hi), nt = 0..3, where (w.lo, w.hi) = the weights of levels 2 lp and
// 2 lp + 1 for key J, held as a pair by lane K of this lane's row of 16 and broadcast by ONE 64-bit DPP move (row_newbcast is
// the one DPP control 64-bit operands support; cross-lane VALU operations run at half rate, so the weights are broadcast
// once, not folded into each FMA).  `u` is the (key 0, key 1) half of a product tile's four registers, op_sel picks key J
// for both halves.  One asm statement: hipcc would pad a state between dependent statements.
// `w` is the caller's ONE scratch pair, read-write in every statement and so live from the first to the last of them:
// as a per-statement output hipcc put it into the dead fourth element of whatever product tile the MFMA in front had just
// been issued into -- the matrix pipe then overwrote the weight between two FMAs (the hazard recogniser does not look
// into asm statements).  A value that is live across an MFMA cannot share a register with its result.
template <int K, int J>
__device__ __forceinline__ void pkfma4_bcast(f32x2& a0, f32x2& a1, f32x2& a2, f32x2& a3, f32x2& w, f32x2 p, f32x2 u0, f32x2 u1,
                                             f32x2 u2, f32x2 u3) {
  if constexpr (J == 0)
    asm volatile("v_mov_b64_dpp %4, %5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_pk_fma_f32 %0, %6, %4, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %7, %4, %1 op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %2, %8, %4, %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %9, %4, %3 op_sel_hi:[0,1,1]"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w)
                 : "v"(p), "v"(u0), "v"(u1), "v"(u2), "v"(u3), "i"(K));
  else
    asm volatile("v_mov_b64_dpp %4, %5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_pk_fma_f32 %0, %6, %4, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %1, %7, %4, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %2, %8, %4, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %3, %9, %4, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w)
                 : "v"(p), "v"(u0), "v"(u1), "v"(u2), "v"(u3), "i"(K));
}

// the odd last level: a[nt] += p[lane K of the row] * u[nt], one 32-bit broadcast and four plain FMAs
template <int K>
__device__ __forceinline__ void fma4_bcast(float& a0, float& a1, float& a2, float& a3, float& w, float p, float u0, float u1,
                                           float u2, float u3) {
  asm volatile("v_mov_b32_dpp %4, %5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32 %0, %4, %6\n\tv_fmac_f32 %1, %4, %7\n\tv_fmac_f32 %2, %4, %8\n\tv_fmac_f32 %3, %4, %9"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w)
               : "v"(p), "v"(u0), "v"(u1), "v"(u2), "v"(u3), "i"(K));
}

template <typename F, int... I>
__device__ __forceinline__ void unroll_idx(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
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
  // P of a (column, head): 2 NLP pairs -- pair j * NLP + lp = the weights of levels 2 lp, 2 lp + 1 for key j (j = 0, 1)
  constexpr int NLP = (LQ + 1) / 2, NFULL = LQ / 2;
  static_assert(2 * NLP <= 16, "one pair of weights per lane of a row of 16");
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
  //      Through buffer descriptors (buffer_load ... lds): one 32-bit offset per lane and operand, everything else -- the
  //      K-stage, the second piece of an operand (64 tile rows further on) -- is a scalar offset: no vector address
  //      arithmetic per piece.  The descriptors end with the arrays: a ragged last tile reads zeros behind the last column. ----
  auto descriptor = [](const char* base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff), 0x00020000);
  };
  const int64_t cols_left = p.n_cols - col0;
  const __amdgpu_buffer_rsrc_t rs_v = descriptor(p.V + col0 * LK * p.ldv_b, cols_left * LK * p.ldv_b);
  const __amdgpu_buffer_rsrc_t rs_w = descriptor(p.W + (int64_t)n0 * p.ldw_b, (int64_t)PO_N * p.ldw_b);
  const __amdgpu_buffer_rsrc_t rs_p = descriptor(reinterpret_cast<const char*>(p.P) + col0 * heads * (PO_PS * 4),
                                                  cols_left * heads * (PO_PS * 4));
  uint32_t vo_v, vo_w, vo_p;
  {
    // (row 4 c + 3 of a column does not exist: its lanes fetch key 2 again -- the same cache lines as their neighbours' --,
    //  what LDS holds there is never used.  NOT masked out of the DMA: hipcc duplicates the code behind a divergent branch
    //  per lane set, and the wave's count of outstanding pieces, which the counted waits below rely on, changes with it.)
    const int row = tid >> 3, c = tid & 7, j = row & 3;
    vo_v = (uint32_t)(((row >> 2) * LK + (j < LK ? j : LK - 1)) * (int)p.ldv_b + ((c ^ (row & 7)) << 4));
    // LDS row t of the weight tile holds feature 64 (t >> 6) + 4 (t & 15) + ((t >> 4) & 3)
    vo_w = (uint32_t)((4 * (row & 15) + ((row >> 4) & 3)) * (int)p.ldw_b + ((c ^ (row & 7)) << 4));
    vo_p = (uint32_t)((tid >> 4) * heads * (PO_PS * 4) + (tid & 15) * 16);
  }
  const int v_half = 16 * LK * (int)p.ldv_b, w_half = 64 * (int)p.ldw_b;   // rows 64.. of a tile: 16 columns / 64 features on
  char* const pbase = smem + PO_NST * PO_STAGE;
  // K-stage s = 2 h + ks of head h: 2 value pieces, 2 weight pieces; the even stage of a head also carries the head's tile of P
  auto stage_in = [&](int st) {
    char* base = smem + (st & (PO_NST - 1)) * PO_STAGE;
    const int koff = st * PO_ROWB;
#pragma unroll
    for (int r = 0; r < 2; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_ptr_t)(base + (r * PO_THREADS + wave * 64) * 16), 16, vo_v, koff + r * v_half, 0, 0);
#pragma unroll
    for (int r = 0; r < 2; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(base + PO_OPER + (r * PO_THREADS + wave * 64) * 16), 16, vo_w,
                                               koff + r * w_half, 0, 0);
    if ((st & 1) == 0) {   // (uniform)
      const int h = st >> 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_p, (lds_ptr_t)(pbase + (h & 1) * PO_PTILE + wave * 1024), 16, vo_p, h * (PO_PS * 4), 0, 0);
    }
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
  const int off_p = (wm * 8 + g) * (PO_PS * 4) + i16 * 8;

  // accumulators: levels in pairs (the operands of the packed FMAs), an odd last level on its own
  f32x2 out2[2][4][NFULL > 0 ? NFULL : 1];
  float outl[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int l = 0; l < NFULL; ++l) out2[mt][nt][l] = f32x2{0.f, 0.f};
      outl[mt][nt] = 0.f;
    }
  // Two sets of products and weights, A and B: an even head combines A (the previous head's) while it multiplies into B, an
  // odd head the other way round (no moves; heads is even).  Head 0 combines zeros times zeros -- a branch around it would
  // make hipcc keep two copies of the 104 accumulators and move them every iteration.
  f32x4 UA[2][4], UB[2][4];
  f32x2 prA[2], prB[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) UA[mt][nt] = UB[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    prA[mt] = prB[mt] = f32x2{0.f, 0.f};
  }

  // The value rows of a column arrive as (v0 - v2, v1 - v2, v2) and the weights as (p0, p1) for level 0, (p0 - p0[0], p1 - p1[0])
  // for the others (perceiver_probs_kernel): with p0 + p1 + p2 = 1,
  //     sum_j p_j U_j  =  U'_2 + p0 U'_0 + p1 U'_1     (U' the products of those rows)
  // -- two FMAs per level instead of three --, level 0 collects the U'_2 of all heads, and the other levels accumulate their
  // DIFFERENCE to level 0, added at the end: no separate accumulator for the level-independent part.
  f32x2 wtmp = {0.f, 0.f};   // the broadcast weights of pkfma4_bcast
  float wtmp1 = 0.f;         // ... of fma4_bcast
  // One step of the combine of row fragment MT: element e of 2 NLP = (key j, level pair lp), key outermost (an accumulator
  // comes back NLP steps later): one broadcast of the pair of weights, four packed FMAs.
  auto lo2 = [](f32x4 u) { return __builtin_shufflevector(u, u, 0, 1); };
  auto pair_one = [&](f32x4 (&U)[2][4], f32x2 (&pr)[2], auto MT, auto E) {
    constexpr int mt = decltype(MT)::value, e = decltype(E)::value;
    constexpr int j = e / NLP, lp = e % NLP;
    if constexpr (lp < NFULL)
      pkfma4_bcast<e, j>(out2[mt][0][lp], out2[mt][1][lp], out2[mt][2][lp], out2[mt][3][lp], wtmp, pr[mt], lo2(U[mt][0]),
                         lo2(U[mt][1]), lo2(U[mt][2]), lo2(U[mt][3]));
    else
      fma4_bcast<e>(outl[mt][0], outl[mt][1], outl[mt][2], outl[mt][3], wtmp1, pr[mt].x, U[mt][0][j], U[mt][1][j], U[mt][2][j],
                    U[mt][3][j]);
  };
  auto base_add = [&](f32x4 (&U)[2][4], auto MT) {
    constexpr int mt = decltype(MT)::value;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      if constexpr (NFULL > 0) out2[mt][nt][0].x += U[mt][nt][2];
      else outl[mt][nt] += U[mt][nt][2];
    }
  };
  // One K-stage (32 of a head's 64) per wave = two slots, R and C (the file header); a barrier ends each.
  // Ring: the waves' slots are numbered s = 2 st (R) and 2 st + 1 (C) for the early half, one later for the late half.
  // Stage st is read in slots 2 st (early) and 2 st + 1 (late); every wave issues its pieces of stage st + 3 in its R of
  // stage st, into the ring slot of stage st - 1 (last read a barrier before), and ends that R waiting until only the pieces
  // of stages st + 2 and st + 3 are in flight: stage st + 1 is complete, for every wave, a barrier before its first reader.
  // P of head h travels with stage 2 h.
  // Measured (s_memtime stamps, 0.25 degree shape, tools/probes/po_stamps.py on a -DPO_STAMPS build of round 6): R ~450-550
  // cycles, C ~800-900 (24 MFMAs alone: ~400; the 70 VALU instructions alone: ~410 -- they add up, see the file header),
  // ~100 per barrier.  Also measured and not kept: the combine in R instead of C (M = MFMAs only: +7 %), the combine's
  // instructions dealt out singly between the MFMAs (+13 %), one slot per stage (+10 %).
  constexpr int NF = 2 * NLP;   // (key, level pair) steps per row fragment
  const int n_st = 2 * heads;
#ifdef PO_STAMPS
  int sidx = 0;
  auto stamp = [&]() {   // slot boundaries of workgroup 4000, per wave, through spare LDS
    if (blockIdx.x == 4000) {
      const uint64_t t = __builtin_amdgcn_s_memtime();
      if (lane == 0) reinterpret_cast<uint32_t*>(smem + PO_LDS)[wave * 256 + sidx] = (uint32_t)t;
    }
    ++sidx;
  };
#else
  auto stamp = [&]() {};
#endif
  auto phase = [&](int st, auto KS, f32x4 (&U)[2][4], f32x2 (&pr)[2], f32x4 (&Un)[2][4], f32x2 (&prn)[2]) {
    constexpr int ks = decltype(KS)::value;
    stamp();
    // ---- R: this stage's fragments (+ the head's weights), the LDS-DMA of stage st + 3; the SIMD partner computes meanwhile ----
    const char* buf = smem + (st & (PO_NST - 1)) * PO_STAGE;
    u32x4 vh[2], vl[2], wh[4], wl[4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      vh[mt] = *reinterpret_cast<const u32x4*>(buf + off_v[0] + mt * 16 * PO_ROWB);
      vl[mt] = *reinterpret_cast<const u32x4*>(buf + off_v[1] + mt * 16 * PO_ROWB);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      wh[nt] = *reinterpret_cast<const u32x4*>(buf + off_w[0] + nt * 16 * PO_ROWB);
      wl[nt] = *reinterpret_cast<const u32x4*>(buf + off_w[1] + nt * 16 * PO_ROWB);
    }
    if constexpr (ks == 0) {   // this head's weights, for the combine that starts with the next head
      const char* pb = pbase + ((st >> 1) & 1) * PO_PTILE;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) prn[mt] = *reinterpret_cast<const f32x2*>(pb + off_p + mt * 4 * (PO_PS * 4));
    }
    const int nx = st + 3;
    if (nx < n_st) stage_in(nx);
    // stage st + 1 has landed when only the pieces of stages st + 2 (and st + 3) are in flight: 4 per stage, 5 with the
    // head's P -- one of two consecutive stages is even
    if (nx < n_st) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
    else if (nx == n_st) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    stamp();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    stamp();
    // ---- C: 24 MFMAs into Un, the steps of the previous head's combine between them; the partner reads ----
    // (the asm FMAs are invisible to hipcc's hazard recogniser -- a matrix result may be read 11 states after its MFMA issued --:
    //  the products they read were finished at least a stage and two barriers ago)
    unroll_idx(std::make_integer_sequence<int, 24>{}, [&](auto I) {
      constexpr int i = decltype(I)::value, term = i / 8, mt = (i % 8) / 4, nt = i % 4;
      const u32x4 a = term == 1 ? vl[mt] : vh[mt];
      const u32x4 b = term == 0 ? wl[nt] : wh[nt];
      if constexpr (ks == 0 && term == 0) Un[mt][nt] = mma_f16(a, b, f32x4{0.f, 0.f, 0.f, 0.f});
      else Un[mt][nt] = mma_f16(a, b, Un[mt][nt]);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int e0 = i * NF / 24, e1 = (i + 1) * NF / 24;
      unroll_idx(std::make_integer_sequence<int, e1 - e0>{}, [&](auto D) {
        pair_one(U, pr, std::integral_constant<int, ks>{}, std::integral_constant<int, e0 + decltype(D)::value>{});
      });
      if constexpr (i == 23) base_add(U, std::integral_constant<int, ks>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    stamp();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  };

  // ---- prologue: stages 0, 1, 2 on their way, stage 0 published; the late half starts one slot later ----
  stage_in(0);
  stage_in(1);
  stage_in(2);
  asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   // stages 1 (4 pieces) and 2 (5) in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (late == 1) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  const std::integral_constant<int, 0> K0{};
  const std::integral_constant<int, 1> K1{};
  // an even head combines (UA, prA) and multiplies into UB, its weights go to prB; an odd head the other way round
  for (int h = 0; h < heads; h += 2) {
    phase(2 * h, K0, UA, prA, UB, prB);
    phase(2 * h + 1, K1, UA, prA, UB, prB);
    phase(2 * h + 2, K0, UB, prB, UA, prA);
    phase(2 * h + 3, K1, UB, prB, UA, prA);
  }
  // the last head's combine; then the level-independent part.  (Its products left the matrix pipe only just now, and the asm
  // FMAs are invisible to hipcc's hazard recogniser: a late wave whose last barrier opens at once would read them early.)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  unroll_idx(std::make_integer_sequence<int, NF>{}, [&](auto E) { pair_one(UA, prA, K0, E); });
  unroll_idx(std::make_integer_sequence<int, NF>{}, [&](auto E) { pair_one(UA, prA, K1, E); });
  base_add(UA, K0);
  base_add(UA, K1);
  auto level = [&](int mt, int nt, int l) -> float { return l < 2 * NFULL ? out2[mt][nt][l >> 1][l & 1] : outl[mt][nt]; };
  if (late == 0) __builtin_amdgcn_s_barrier();   // (every wave passes the same number of barriers)

#ifdef PO_STAMPS
  if (blockIdx.x == 4000) {
    __syncthreads();
    for (int i = tid; i < 8 * 256; i += PO_THREADS) po_stamps[i] = reinterpret_cast<uint32_t*>(smem + PO_LDS)[i];
  }
#endif
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
    for (int l = 0; l < LQ; ++l) {   // level 0 holds the level-independent part, the others their difference to it
      f32x4 v;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        v[nt] = fmaf(l == 0 ? level(mt, nt, 0) : level(mt, nt, l) + level(mt, nt, 0), 0.015625f, b4[nt]);
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
  int64_t ld; int s_off;   // SCORES: a context row is [v (inner) | ... | scores at s_off], ld floats apart (else k | v, 2 inner)
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

// SCORES: the scaled scores q_l . k_j arrive pre-multiplied in the context rows (embed.hip, perceiver_attention_scores_kernel:
// the queries of a first layer are model constants); lane i of a group then owns level i -- its three scores are three
// loads, its softmax needs no other lane.
template <int LQ, bool SCORES>
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
  const int64_t row_len = SCORES ? p.ld : 2 * (int64_t)inner;
  const float* row0 = p.kv + (b * p.kv_bstride + l) * row_len;
  const int64_t kv_step = p.kv_lstride * row_len;
  const float* v0 = row0 + (SCORES ? 0 : inner) + h * HDIM + d0;
  float vv[LK][4];
#pragma unroll
  for (int j = 0; j < LK; ++j) load4(v0 + j * kv_step, vv[j]);
  // ---- the weights.  With w_j[l] = p_j of level 0, the DIFFERENCE to level 0 for the others (the third weight follows from
  //      p0 + p1 + p2 = 1; how perceiver_out_kernel uses them is said there),
  //      P[col][head] = 16 pairs: pair j * NLP + lp = (w_j[2 lp], w_j[2 lp + 1]), j = 0, 1, NLP = ceil(Lq / 2); lane i writes pair i ----
  constexpr int NLP = (LQ + 1) / 2;
  float mine[2] = {0.f, 0.f};
  float* prow = p.P + (col * p.heads + h) * PO_PS;
  auto weights = [](const float (&sc)[LK], float& w0, float& w1) {
    const float m = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
    float e[LK];
#pragma unroll
    for (int j = 0; j < LK; ++j) e[j] = __expf(sc[j] - m);
    const float inv = 1.0f / ((e[0] + e[1]) + e[2]);
    w0 = e[0] * inv;
    w1 = e[1] * inv;
  };
  if constexpr (SCORES) {
    auto dpp = [](float x, auto ctrl) {
      return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    const int lq = i16 < LQ ? i16 : LQ - 1;   // (lanes past the last level repeat it; their weights are zeroed below)
    float sc[LK];
#pragma unroll
    for (int j = 0; j < LK; ++j) sc[j] = row0[j * kv_step + p.s_off + lq * p.heads + h];
    float w0, w1;
    weights(sc, w0, w1);
    const float p00 = dpp(w0, std::integral_constant<int, 0x150>{}), p10 = dpp(w1, std::integral_constant<int, 0x150>{});   // row_newbcast:0
    if (i16 > 0) { w0 -= p00; w1 -= p10; }
    if (i16 >= LQ) w0 = w1 = 0.f;
    const float n0 = dpp(w0, std::integral_constant<int, 0xB1>{}), n1 = dpp(w1, std::integral_constant<int, 0xB1>{});   // lane i ^ 1
    // lane 2 lp writes pairs lp and NLP + lp; the lanes from 2 NLP on zero the pairs of their own number (all 16 pairs that
    // perceiver_out_kernel's lanes read are defined)
    if (i16 >= 2 * NLP) *reinterpret_cast<float2*>(prow + i16 * 2) = make_float2(0.f, 0.f);
    if ((i16 & 1) == 0 && i16 < 2 * NLP) {
      *reinterpret_cast<float2*>(prow + (i16 >> 1) * 2) = make_float2(w0, n0);
      *reinterpret_cast<float2*>(prow + (NLP + (i16 >> 1)) * 2) = make_float2(w1, n1);
    }
  } else {
    // every lane of the group ends up with all the weights (the dot products are reduced across it); lane i keeps pair i
    float kk[LK][4];
#pragma unroll
    for (int j = 0; j < LK; ++j) load4(row0 + h * HDIM + d0 + j * kv_step, kk[j]);
    float p00 = 0.f, p10 = 0.f;
#pragma unroll
    for (int c = 0; c < LQ; ++c) {
      float qv[4];
      load4(p.q + (int64_t)c * inner + h * HDIM + d0, qv);
      float sc[LK];
#pragma unroll
      for (int j = 0; j < LK; ++j)
        sc[j] = group16_sum(fmaf(qv[0], kk[j][0], fmaf(qv[1], kk[j][1], fmaf(qv[2], kk[j][2], qv[3] * kk[j][3])))) * scale;
      float w0, w1;
      weights(sc, w0, w1);
      if (c == 0) { p00 = w0; p10 = w1; }
      else { w0 -= p00; w1 -= p10; }
      if (i16 == c / 2) mine[c & 1] = w0;
      if (i16 == NLP + c / 2) mine[c & 1] = w1;
    }
    // (lanes past the last pair write zeros: the 16 pairs perceiver_out_kernel's lanes read are defined everywhere)
    *reinterpret_cast<float2*>(prow + i16 * 2) = make_float2(mine[0], mine[1]);
  }
  // ---- the values as fp16 pairs: lanes 2i, 2i + 1 hold 8 consecutive features between them and trade halves; the even
  //      lane stores the eight high halves, the odd one the remainders (the layout of aurora_hip_split_f16) ----
  const bool odd = (threadIdx.x & 1) != 0;
  auto swap1 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true); };
  const int f8 = (h * HDIM + d0) & ~7;
#pragma unroll
  for (int j = 0; j < LK; ++j) {
    uint32_t h0, h1, l0, l1;
    // rows (v0 - v2, v1 - v2, v2): see perceiver_out_kernel
    const float d0 = j < 2 ? vv[j][0] - vv[2][0] : vv[2][0], d1 = j < 2 ? vv[j][1] - vv[2][1] : vv[2][1];
    const float d2 = j < 2 ? vv[j][2] - vv[2][2] : vv[2][2], d3 = j < 2 ? vv[j][3] - vv[2][3] : vv[2][3];
    split_pair_f16(d0, d1, h0, l0);
    split_pair_f16(d2, d3, h1, l1);
    const uint32_t s0 = swap1(odd ? h0 : l0), s1 = swap1(odd ? h1 : l1);
    char* d = p.Vp + ((col * LK + j) * inner + (f8 & ~31)) * 4 + (f8 & 31) * 2 + (odd ? 64 : 0);
    *reinterpret_cast<u32x4*>(d) = odd ? u32x4{s0, s1, l0, l1} : u32x4{h0, h1, s0, s1};
  }
}

}  // namespace

}  // namespace aurora

using namespace aurora;

#ifdef PO_STAMPS
extern "C" void aurora_hip_debug_po_stamps(void* dst) { (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(po_stamps), sizeof(po_stamps)); }
constexpr int PO_LDS_LAUNCH = PO_LDS + 8192;
#else
constexpr int PO_LDS_LAUNCH = PO_LDS;
#endif

extern "C" int aurora_hip_perceiver_out_supported(int Lq, int Lk, int heads, int head_dim, int N) {
  return (Lq == 3 || Lq == 4 || Lq == 13) && Lk == 3 && head_dim == 64 && heads >= 2 && heads % 2 == 0 && N > 0 && N % PO_N == 0;
}

namespace {
int launch_probs(const PercProbArgs& p, int Lq, bool scores, void* stream) {
  const int64_t items = (int64_t)p.B * p.cols_per_b * p.heads * 16;
  const dim3 grid((unsigned)((items + 255) / 256)), block(256);
#define AURORA_PROBS(LQ)                                                                                          \
  do {                                                                                                            \
    if (scores) hipLaunchKernelGGL((perceiver_probs_kernel<LQ, true>), grid, block, 0, as_stream(stream), p);     \
    else hipLaunchKernelGGL((perceiver_probs_kernel<LQ, false>), grid, block, 0, as_stream(stream), p);           \
  } while (0)
  switch (Lq) {
    case 3: AURORA_PROBS(3); break;
    case 4: AURORA_PROBS(4); break;
    default: AURORA_PROBS(13); break;
  }
#undef AURORA_PROBS
  return check_launch("perceiver_probs");
}
}  // namespace

extern "C" int aurora_hip_perceiver_probs(const float* q, const float* kv, float* P, void* Vp, int B, int64_t cols_per_b,
                                          int64_t kv_bstride, int64_t kv_lstride, int Lq, int Lk, int heads, int head_dim,
                                          const float* guard, float guard_limit, void* stream) {
  AURORA_CHECK_ARG(aurora_hip_perceiver_out_supported(Lq, Lk, heads, head_dim, PO_N),
                   "perceiver_probs: Lq=%d Lk=%d head_dim=%d (built for Lq in {3, 4, 13}, Lk = 3, head_dim 64)", Lq, Lk, head_dim);
  AURORA_CHECK_ARG(q && kv && P && Vp && B > 0 && cols_per_b > 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)kv % 16) == 0 &&
                       ((uintptr_t)P % 16) == 0 && ((uintptr_t)Vp % 16) == 0,
                   "perceiver_probs: null / unaligned argument");
  PercProbArgs p{q, kv, P, (char*)Vp, B, cols_per_b, kv_bstride, kv_lstride, heads, guard, guard_limit, 0, 0};
  return launch_probs(p, Lq, false, stream);
}

extern "C" int aurora_hip_perceiver_probs_scores(const float* vs, int64_t ld_vs, int s_off, float* P, void* Vp, int B,
                                                 int64_t cols_per_b, int64_t kv_bstride, int64_t kv_lstride, int Lq, int Lk,
                                                 int heads, int head_dim, const float* guard, float guard_limit, void* stream) {
  AURORA_CHECK_ARG(aurora_hip_perceiver_out_supported(Lq, Lk, heads, head_dim, PO_N),
                   "perceiver_probs_scores: Lq=%d Lk=%d head_dim=%d (built for Lq in {3, 4, 13}, Lk = 3, head_dim 64)", Lq, Lk,
                   head_dim);
  AURORA_CHECK_ARG(vs && P && Vp && B > 0 && cols_per_b > 0 && ((uintptr_t)vs % 16) == 0 && ((uintptr_t)P % 16) == 0 &&
                       ((uintptr_t)Vp % 16) == 0 && ld_vs % 4 == 0 && s_off >= heads * head_dim &&
                       ld_vs >= (int64_t)s_off + (int64_t)Lq * heads,
                   "perceiver_probs_scores: a row is [v (heads * head_dim) | ... | scores (Lq * heads) at s_off], ld %% 4 == 0");
  PercProbArgs p{nullptr, vs, P, (char*)Vp, B, cols_per_b, kv_bstride, kv_lstride, heads, guard, guard_limit, ld_vs, s_off};
  return launch_probs(p, Lq, true, stream);
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
    (void)hipFuncSetAttribute((const void*)perceiver_out_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, PO_LDS_LAUNCH);
    (void)hipFuncSetAttribute((const void*)perceiver_out_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, PO_LDS_LAUNCH);
    (void)hipFuncSetAttribute((const void*)perceiver_out_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, PO_LDS_LAUNCH);
    attr_done = true;
  }
  const dim3 grid((unsigned)p.n_blocks), block(PO_THREADS);
  switch (Lq) {
    case 3: hipLaunchKernelGGL(perceiver_out_kernel<3>, grid, block, PO_LDS_LAUNCH, as_stream(stream), p); break;
    case 4: hipLaunchKernelGGL(perceiver_out_kernel<4>, grid, block, PO_LDS_LAUNCH, as_stream(stream), p); break;
    default: hipLaunchKernelGGL(perceiver_out_kernel<13>, grid, block, PO_LDS_LAUNCH, as_stream(stream), p); break;
  }
  return check_launch("perceiver_out");
}
