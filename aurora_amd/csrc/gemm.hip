// Dense linear layers on the gfx950 matrix cores:  C = act(A . W^T + bias) (+ residual).
//
// Kernels in this file (dispatch: aurora_hip_linear at the end):
//   linear_kernel<T>            128 x 128 tile, two workgroups per CU -- small / ragged shapes, few-tile shapes
//   linear_kernel_256<T,4,4>    256 x 256 tile, 4-stage LDS ring     -- the backbone linears (bf16) and native-fp32 mode
//   linear_kernel_256_f32x3<3>  fp32 by three bf16 terms (6 MFMAs)   -- fp32 linears, any input range
//   linear_kernel_256_f32x3<2>  fp32 by two fp16 terms (3 MFMAs)     -- fp32 linears with inputs bounded by construction
//   epilogue_256*               bias / activation / residual; single-dtype results leave as whole rows via LDS
// The description below is the 128 x 128 kernel; the others document their differences in place.
//
// One kernel template serves bf16 (v_mfma_f32_16x16x32_bf16) and fp32
// (v_mfma_f32_16x16x4_f32, exact fp32 FMA chains) because both are tiled in BYTES: a K-tile
// is 128 bytes of every operand row (64 bf16 / 32 fp32), staged by 16-byte LDS-DMA pieces
// (global_load_lds_dwordx4, no VGPR round trip) into two LDS buffers per operand.
//
// Block tile 128 (m) x 128 (n), 256 threads = 4 waves as 2 (m) x 2 (n), wave tile 64 x 64 =
// 4 x 4 MFMA fragments of 16 x 16.  The MFMA "A" operand is the WEIGHT tile and the "B"
// operand the activation tile, i.e. every fragment holds C^T: lane (j = lane & 15, g = lane >> 4)
// owns activation row m = 16*fm + j and 4 consecutive output features per fragment.  Weight rows
// are interleaved over the 4 n-fragments (row = 16*(i>>2) + 4*fn + (i&3) for operand row i), so
// that the lane ends up with 16 CONSECUTIVE output features n = 16*g + 0..15 of one row: the
// epilogue (bias, exact GELU, residual, dual-dtype store) then works on whole 32/64-byte row
// pieces with 16-byte stores.
//
// LDS image of a tile: [128 rows][8 chunks of 16 B], chunk c of row r stored at position
// c ^ f(r).  LDS-DMA writes lane-linearly, so the swizzle is applied to the per-lane SOURCE
// address and again on the fragment read (both sides or neither).  f is chosen per operand so
// that the 16 rows touched by one ds_read_b128 lane group fall on 16 distinct 16-byte bank slots:
//   activations: rows i, i+1, ...          f(r) = r & 7
//   weights    : rows 16a + 4fn + b        f(r) = ((r >> 4) & 3) << 1 | ((r >> 1) & 1)
//
// Workgroup ids are remapped so that each XCD (8 of them, private L2s, block b runs on XCD b % 8)
// owns a contiguous range of tiles with the n-tiles of one m-tile adjacent: the activation tile
// is then fetched from HBM once per XCD and re-used out of that XCD's L2.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

#ifndef F32PP_PARTS   // passes of the two-term fp32 kernel's whole-row epilogue (1, 2 or 4): see linear_kernel_f32pp
#define F32PP_PARTS 4
#endif

namespace aurora {

namespace {

constexpr int BM = 128;       // activation rows per block
constexpr int BN = 128;       // output features per block
constexpr int ROW_BYTES = 128;  // bytes of K per tile row
constexpr int TILE_BYTES = 128 * ROW_BYTES;  // 16 KiB per operand per buffer
constexpr int THREADS = 256;
constexpr int ACT_GELU_FAST = 4;  // internal: fp32 results of the operand-splitting kernels, erf to 1.5e-7 (packed)
constexpr int A4_DEFAULT_MIN_K = 0;   // (0: the four-wave kernel of gemm_a4.hip is off unless AURORA_GEMM_A4_MIN_K says otherwise)

struct LinearArgs {
  const char* A; int64_t lda_b;   // byte strides
  const char* W; int64_t ldw_b;
  const float* bias;
  char* C; int64_t ldc;           // element strides from here on
  char* C2; int64_t ldc2;
  const float* res; int64_t ldr;
  int64_t M; int N; int k_tiles; int act;
  int tiles_n; int64_t n_blocks;
  int vec_store;                  // 1: every C/C2/res row piece is 16-byte aligned
  const float* guard; float guard_limit;   // f32 split kernels (guarded launch): two fp16 terms iff *guard < guard_limit
  int out_split;                  // two-term ping-pong kernel: C is written in the fp16-pair layout (see aurora_hip_split_f16)
  // strided batch (aurora_hip_linear_batched): problem blockIdx.y adds these to A / W / C (bytes) and bias (floats)
  int64_t bs_a, bs_w, bs_c, bs_bias;
  // split-K (linear_kernel_256pp, MODE 1): workgroup b multiplies K-slice b / n_blocks of tile b % n_blocks; slices
  // meet through fp32 slabs (256 KiB per slice and tile) and a ticket per tile -- the last arriver adds up and finishes
  int split; float* slabs; int32_t* tickets;
  // head planes (aurora_hip_linear_planes; 0: rows of ldc elements): the 64-column blocks of the result are q | k | v of
  // the attention heads (block sel * plane_heads + h); head h owns a plane of [M rows][q | k | v = 192 elements],
  // plane_stride elements after the previous head's
  int64_t plane_stride; int plane_heads;
};

// Element (m, n) of the result (n a multiple of 16: a 16-element piece never straddles two 64-column blocks).
template <typename T>
__device__ __forceinline__ T* out_piece(const LinearArgs& p, int64_t m, int n) {
  T* const c = reinterpret_cast<T*>(p.C);
  if (!p.plane_stride) return c + m * p.ldc + n;
  const int blk = n >> 6, sel = blk / p.plane_heads, h = blk - sel * p.plane_heads;
  return c + (int64_t)h * p.plane_stride + m * 192 + sel * 64 + (n & 63);
}

// The problem of a strided batch this workgroup belongs to (blockIdx.y; a plain launch has one problem and zero strides).
__device__ __forceinline__ LinearArgs batch_problem(const LinearArgs& in) {
  LinearArgs p = in;
  const int64_t g = blockIdx.y;
  p.A += g * in.bs_a;
  p.W += g * in.bs_w;
  p.C += g * in.bs_c;
  if (in.bias) p.bias += g * in.bs_bias;
  return p;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ int swz_x(int row) { return row & 7; }
__device__ __forceinline__ int swz_w(int row) { return (((row >> 4) & 3) << 1) | ((row >> 1) & 1); }

// One 16 x 16 x (128 bytes of K / 2) MFMA step on 16-byte operand pieces.
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  __device__ static __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // The K order inside a tile is free as long as both operands agree: lane group g supplies
  // k = 16*chunk + 4*g + s to the s-th of four 16x16x4 steps.
  __device__ static __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
};

template <typename T>
__device__ __forceinline__ void store16(T* dst, const float (&v)[16], bool vec, int n_left);

template <>
__device__ __forceinline__ void store16<float>(float* dst, const float (&v)[16], bool vec, int n_left) {
  if (vec && n_left >= 16) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      reinterpret_cast<f32x4*>(dst)[q] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  } else {
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if (t < n_left) dst[t] = v[t];
  }
}
template <>
__device__ __forceinline__ void store16<bf16_t>(bf16_t* dst, const float (&v)[16], bool vec, int n_left) {
  if (vec && n_left >= 16) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
      reinterpret_cast<u32x4*>(dst)[q] =
          u32x4{pack_bf16x2(v[8 * q], v[8 * q + 1]), pack_bf16x2(v[8 * q + 2], v[8 * q + 3]),
                pack_bf16x2(v[8 * q + 4], v[8 * q + 5]), pack_bf16x2(v[8 * q + 6], v[8 * q + 7])};
  } else {
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if (t < n_left) dst[t] = f32_to_bf16(v[t]);
  }
}

template <uint32_t GN = 8>
__device__ __forceinline__ void tile_of_block(uint32_t bid, uint32_t nb, uint32_t tiles_m, uint32_t tiles_n,
                                              uint32_t& tile_m, uint32_t& tile_n);

template <typename T> struct Other;
template <> struct Other<float> { typedef bf16_t type; };
template <> struct Other<bf16_t> { typedef float type; };

template <typename T>
__global__ __launch_bounds__(THREADS, 2) void linear_kernel(const LinearArgs p_in) {
  const LinearArgs p = batch_problem(p_in);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Per buffer: X tile then W tile; 2 buffers.
  auto lds_x = [&](int buf) { return smem + buf * 2 * TILE_BYTES; };
  auto lds_w = [&](int buf) { return smem + buf * 2 * TILE_BYTES + TILE_BYTES; };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware, L2-friendly tile assignment (tile_of_block, below) ----
  uint32_t tile_m, tile_n_u;
  tile_of_block(blockIdx.x, (uint32_t)p.n_blocks, (uint32_t)(p.n_blocks / p.tiles_n), (uint32_t)p.tiles_n, tile_m, tile_n_u);
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = (int)tile_n_u * BN;

  // ---- per-thread staging addresses: 4 pieces of X and 4 of W per K-tile ----
  const char* src_x[4];
  const char* src_w[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r * 32 + (tid >> 3);
    const int c = tid & 7;
    int64_t gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    int gn = n0 + row;
    gn = gn < p.N ? gn : p.N - 1;
    src_x[r] = p.A + gm * p.lda_b + ((c ^ swz_x(row)) << 4);
    src_w[r] = p.W + (int64_t)gn * p.ldw_b + ((c ^ swz_w(row)) << 4);
  }

  auto stage = [&](int kt, int buf) {
    const int64_t koff = (int64_t)kt * ROW_BYTES;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // wave-uniform LDS base; the hardware adds lane * 16.
      const int base = (r * 256 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src_x[r] + koff),
          (lds_ptr_t)(lds_x(buf) + base), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src_w[r] + koff),
          (lds_ptr_t)(lds_w(buf) + base), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes inside a tile), one per (fragment, k-half) ----
  const int i16 = lane & 15, g = lane >> 4;
  int off_w[4][2], off_x[4][2];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int row_w = wn * 64 + 16 * (i16 >> 2) + 4 * f + (i16 & 3);
    const int row_x = wm * 64 + 16 * f + i16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = g + 4 * ks;
      off_w[f][ks] = row_w * ROW_BYTES + ((c ^ swz_w(row_w)) << 4);
      off_x[f][ks] = row_x * ROW_BYTES + ((c ^ swz_x(row_x)) << 4);
    }
  }

  f32x4 acc[4][4];  // [fn][fm]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < p.k_tiles; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < p.k_tiles) stage(kt + 1, buf ^ 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fw[4], fx[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        fw[f] = *reinterpret_cast<const u32x4*>(lds_w(buf) + off_w[f][ks]);
        fx[f] = *reinterpret_cast<const u32x4*>(lds_x(buf) + off_x[f][ks]);
      }
#pragma unroll
      for (int fn = 0; fn < 4; ++fn)
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) acc[fn][fm] = Mma<T>::run(fw[fn], fx[fm], acc[fn][fm]);
    }
    // The LDS-DMA of tile kt+1 must have landed, and every wave must be done reading tile kt,
    // before the next iteration reads one buffer and overwrites the other.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane owns row m (per fm) x 16 consecutive features ----
  const int nbase = n0 + wn * 64 + 16 * g;
  const int n_left = p.N - nbase;
  if (n_left <= 0) return;
  float bias_v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) bias_v[t] = (p.bias && t < n_left) ? p.bias[nbase + t] : 0.f;
  const bool vec = p.vec_store != 0;
  typedef typename Other<T>::type T2;

#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const int64_t m = m0 + wm * 64 + 16 * fm + i16;
    if (m >= p.M) continue;
    float v[16];
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
      v[4 * fn + 0] = acc[fn][fm].x + bias_v[4 * fn + 0];
      v[4 * fn + 1] = acc[fn][fm].y + bias_v[4 * fn + 1];
      v[4 * fn + 2] = acc[fn][fm].z + bias_v[4 * fn + 2];
      v[4 * fn + 3] = acc[fn][fm].w + bias_v[4 * fn + 3];
    }
    if (p.act == AURORA_ACT_GELU) {
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = gelu_for<T>(v[t]);
    } else if (p.act == ACT_GELU_FAST) {
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = gelu_erf_fast(v[t]);
    } else if (p.act == AURORA_ACT_SILU) {
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = v[t] / (1.0f + expf(-v[t]));
    }
    if (p.res) {
      const float* rp = p.res + m * p.ldr + nbase;
      if (vec && n_left >= 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 rv = reinterpret_cast<const f32x4*>(rp)[q];
          v[4 * q] += rv.x; v[4 * q + 1] += rv.y; v[4 * q + 2] += rv.z; v[4 * q + 3] += rv.w;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t)
          if (t < n_left) v[t] += rp[t];
      }
    }
    store16<T>(out_piece<T>(p, m, nbase), v, vec, n_left);
    if (p.C2) store16<T2>(reinterpret_cast<T2*>(p.C2) + m * p.ldc2 + nbase, v, vec, n_left);
  }
}


// =================================================================================================
// Big-tile kernel for the backbone shapes (M >= 1024, N % 256 == 0): 256 x 256 tile, 512 threads =
// 8 waves as 2 (m) x 4 (n), wave tile 128 x 64 = 8 x 4 fragments.
//
// Why: the 128 x 128 kernel above keeps one 32 KiB K-tile in flight per workgroup; with ~1-2 us of
// HBM/L2 latency and ~0.2 us of MFMA work per tile the short-K GEMMs of stage 0 (K = 512) run at
// ~500 TFLOP/s, latency-bound.  Latency hiding capacity is (bytes in flight per CU) x (FLOP per
// byte of tile).  A 256 x 256 tile doubles the FLOP per byte (128), and K-tiles of 64 BYTES per
// row (32 bf16) make a stage 32 KiB, so a 4-stage LDS ring (128 KiB) keeps THREE tiles = 96 KiB
// in flight per CU: 3x the capacity.  The ring needs counted waits: `s_waitcnt vmcnt(8)` (two
// younger stages x 4 LDS-DMA instructions per lane stay in flight across the barrier) and a raw
// `s_barrier` -- a __syncthreads() would drain the LDS-DMA queue (vmcnt(0)).
//
// One barrier per stage does double duty: (RAW) every wave has waited for its own pieces of
// stage t before arriving, so after the barrier all of stage t is in LDS; (WAR) every wave has
// finished reading stage t-1 (its MFMAs consumed the fragments), so the DMA of stage t+3 may
// overwrite that buffer.
//
// LDS image per operand tile: [256 rows][4 pieces of 16 B]; piece c of row r at position c ^ f(r)
// with f = 0,0,3,3 over (r >> 2) & 3 (activations) / (r >> 4) & 3 (interleaved weight rows): the
// 16 rows of one ds_read_b128 lane group then hit 16 distinct 16-byte slots of the 256-byte bank row.
// =================================================================================================
constexpr int BM2 = 256, BN2 = 256, ROW2 = 64, THREADS2 = 512, NSTAGE2 = 4;
constexpr int OPER2 = 256 * ROW2;      // 16 KiB per operand per stage
constexpr int STAGE2 = 2 * OPER2;      // 32 KiB per stage

__device__ __forceinline__ int swz2(int a) { return ((a >> 1) & 1) * 3; }
__device__ __forceinline__ int swz2_x(int row) { return swz2((row >> 2) & 3); }
__device__ __forceinline__ int swz2_w(int row) { return swz2((row >> 4) & 3); }

// XCD-aware, L2-friendly tile order shared by both kernels: each XCD owns a contiguous range of
// logical ids; inside it n-tiles are visited in groups of `GN` with the m-tile index in between,
// so the workgroups that run together on one XCD share a few activation tiles AND a few weight
// tiles (both then come out of that XCD's 4 MiB L2).
// (GN = 8 for the 256 x 256 bf16 tiles; the fp32 ping-pong kernel's 128 x 256 tiles stage twice the weight bytes per
// activation byte and do best with 16 m-tiles x 2 n-tiles per XCD -- in-step A/B over GN = 2 .. 32, profiles/r02_ab_tile_group.log)
template <uint32_t GN>
__device__ __forceinline__ void tile_of_block(uint32_t bid, uint32_t nb, uint32_t tiles_m, uint32_t tiles_n,
                                              uint32_t& tile_m, uint32_t& tile_n) {
  const uint32_t q8 = nb >> 3, r8 = nb & 7;
  const uint32_t xcd = bid & 7, idx = bid >> 3;
  const uint32_t logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const uint32_t full = (tiles_n / GN) * GN;          // n-tiles covered by complete groups
  const uint32_t per_group = GN * tiles_m;
  if (logical < (full / GN) * per_group) {
    const uint32_t grp = logical / per_group, rem = logical - grp * per_group;
    tile_m = rem / GN;
    tile_n = grp * GN + (rem - tile_m * GN);
  } else {                                             // last, narrower group
    const uint32_t rem = logical - (full / GN) * per_group, gw = tiles_n - full;
    tile_m = rem / gw;
    tile_n = full + (rem - tile_m * gw);
  }
}

// Epilogue of the 256 x 256 kernels: a lane owns a row x 16 consecutive features (same ownership as the
// 128 x 128 kernel) -> bias, activation, fp32 residual, dual-dtype 16-byte stores.
template <typename T>
__device__ __forceinline__ void epilogue_256(const LinearArgs& p, f32x4 (&acc)[4][8], int64_t m0, int n0,
                                             int wm, int wn, int i16, int g) {
  const int nbase = n0 + wn * 64 + 16 * g;
  const int n_left = p.N - nbase;
  if (n_left <= 0) return;
  float bias_v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) bias_v[t] = (p.bias && t < n_left) ? p.bias[nbase + t] : 0.f;
  const bool vec = p.vec_store != 0;
  typedef typename Other<T>::type T2;
#pragma unroll
  for (int fm = 0; fm < 8; ++fm) {
    const int64_t m = m0 + wm * 128 + 16 * fm + i16;
    if (m >= p.M) continue;
    float v[16];
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
      v[4 * fn + 0] = acc[fn][fm].x + bias_v[4 * fn + 0];
      v[4 * fn + 1] = acc[fn][fm].y + bias_v[4 * fn + 1];
      v[4 * fn + 2] = acc[fn][fm].z + bias_v[4 * fn + 2];
      v[4 * fn + 3] = acc[fn][fm].w + bias_v[4 * fn + 3];
    }
    if (p.act == AURORA_ACT_GELU) {
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = gelu_for<T>(v[t]);
    } else if (p.act == ACT_GELU_FAST) {
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = gelu_erf_fast(v[t]);
    } else if (p.act == AURORA_ACT_SILU) {
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = v[t] / (1.0f + expf(-v[t]));
    }
    if (p.res) {
      const float* rp = p.res + m * p.ldr + nbase;
      if (vec && n_left >= 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 rv = reinterpret_cast<const f32x4*>(rp)[q];
          v[4 * q] += rv.x; v[4 * q + 1] += rv.y; v[4 * q + 2] += rv.z; v[4 * q + 3] += rv.w;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t)
          if (t < n_left) v[t] += rp[t];
      }
    }
    store16<T>(out_piece<T>(p, m, nbase), v, vec, n_left);
    if (p.C2) store16<T2>(reinterpret_cast<T2*>(p.C2) + m * p.ldc2 + nbase, v, vec, n_left);
  }
}

// bf16-only outputs (every backbone linear): transpose the wave's 128 x 64 result through LDS so that a
// store instruction writes 8 whole 128-byte row segments with CONSECUTIVE lanes on consecutive 16-byte pieces.
// The direct epilogue above has lane (j, g) write row j, i.e. 64 separate 16-byte requests per instruction,
// and the CU's store path then takes ~8 us per 256 x 256 tile (measured: a K = 512 tile costs 24.7 us with
// its stores and 16.9 us without) -- as long as half the tile's MFMA time.  The ring is dead after the main
// loop, so each wave borrows 16 KiB of it; LDS rows are XOR-swizzled (piece ^ (row & 7)): conflict-free for the
// b128 writes (8 rows per lane group) and reads (4 rows x 4 pieces per lane group).
template <int PARTS>   // 1: a wave's 128 x 64 results in one pass (16 KiB of the dead ring); 2 / 4: passes of 64 / 32 rows (8 / 4 KiB)
__device__ __forceinline__ void epilogue_256_bf16_coalesced(const LinearArgs& p, f32x4 (&acc)[4][8], int64_t m0,
                                                            int n0, int wm, int wn, int wave, int lane, char* smem,
                                                            const float (*bias_pre)[16] = nullptr) {
  const int i16 = lane & 15, g = lane >> 4;
  char* mine = smem + wave * (16384 / PARTS);
  const int nbase = n0 + wn * 64 + 16 * g;
  // (`bias_pre`: the lane's 16 bias values, requested by the caller before its main loop -- requested here, the first use
  //  waits out an L2 round trip with nothing else in flight)
  float bias_v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) bias_v[t] = bias_pre ? (*bias_pre)[t] : (p.bias ? p.bias[nbase + t] : 0.f);
  const int rr = lane >> 3, cc = lane & 7;
  // (head planes: the wave's 64 columns are q, k or v of ONE head: 128-byte pieces of its plane's 384-byte rows)
  bf16_t* cbase = out_piece<bf16_t>(p, 0, n0 + wn * 64) + cc * 8;
  const int64_t ld_rows = p.plane_stride ? 192 : p.ldc;
#pragma unroll
  for (int part = 0; part < PARTS; ++part) {
#pragma unroll
    for (int f = 0; f < 8 / PARTS; ++f) {
      const int fm = part * (8 / PARTS) + f;
      float v[16];
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) {
        v[4 * fn + 0] = acc[fn][fm].x + bias_v[4 * fn + 0];
        v[4 * fn + 1] = acc[fn][fm].y + bias_v[4 * fn + 1];
        v[4 * fn + 2] = acc[fn][fm].z + bias_v[4 * fn + 2];
        v[4 * fn + 3] = acc[fn][fm].w + bias_v[4 * fn + 3];
      }
      if (p.act == AURORA_ACT_GELU) {   // (packed form of gelu_for<bf16_t>: same operations, same bits)
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
          const f32x2_hw r = gelu_sig2(f32x2_hw{v[t], v[t + 1]});
          v[t] = r.x;
          v[t + 1] = r.y;
        }
      } else if (p.act == AURORA_ACT_SILU) {
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = v[t] / (1.0f + expf(-v[t]));
      }
      const int row = 16 * f + i16, sw = row & 7;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        *reinterpret_cast<u32x4*>(mine + row * 128 + (((2 * g + q) ^ sw) << 4)) =
            u32x4{pack_bf16x2(v[8 * q], v[8 * q + 1]), pack_bf16x2(v[8 * q + 2], v[8 * q + 3]),
                  pack_bf16x2(v[8 * q + 4], v[8 * q + 5]), pack_bf16x2(v[8 * q + 6], v[8 * q + 7])};
    }
#pragma unroll
    for (int it = 0; it < 16 / PARTS; ++it) {
      const int row = it * 8 + rr;
      const u32x4 d = *reinterpret_cast<const u32x4*>(mine + row * 128 + ((cc ^ rr) << 4));
      const int64_t m = m0 + wm * 128 + part * (128 / PARTS) + row;
      if (m < p.M) __builtin_nontemporal_store(d, reinterpret_cast<u32x4*>(cbase + m * ld_rows));
    }
  }
}

// fp32 outputs, same idea in two halves (a wave's 128 x 64 fp32 results are 32 KiB, its share of the dead ring
// 16 KiB): rows of 256 bytes, pieces XOR-swizzled by (row & 7); a store instruction writes 4 whole row segments.
__device__ __forceinline__ void epilogue_256_f32_coalesced(const LinearArgs& p, f32x4 (&acc)[4][8], int64_t m0,
                                                           int n0, int wm, int wn, int wave, int lane, char* smem) {
  const int i16 = lane & 15, g = lane >> 4;
  char* mine = smem + wave * 16384;
  const int nbase = n0 + wn * 64 + 16 * g;
  float bias_v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) bias_v[t] = p.bias ? p.bias[nbase + t] : 0.f;
  const int rr = lane >> 4, cc = lane & 15;
  float* cbase = reinterpret_cast<float*>(p.C) + n0 + wn * 64 + cc * 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int fm = 4 * half + f;
      float v[16];
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) {
        v[4 * fn + 0] = acc[fn][fm].x + bias_v[4 * fn + 0];
        v[4 * fn + 1] = acc[fn][fm].y + bias_v[4 * fn + 1];
        v[4 * fn + 2] = acc[fn][fm].z + bias_v[4 * fn + 2];
        v[4 * fn + 3] = acc[fn][fm].w + bias_v[4 * fn + 3];
      }
      if (p.act == AURORA_ACT_GELU) {
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = gelu_for<float>(v[t]);
      } else if (p.act == ACT_GELU_FAST) {
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
          const f32x2_hw r = gelu_erf_fast2(f32x2_hw{v[t], v[t + 1]});
          v[t] = r.x;
          v[t + 1] = r.y;
        }
      } else if (p.act == AURORA_ACT_SILU) {
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = v[t] / (1.0f + expf(-v[t]));
      }
      if (p.res) {
        const int64_t m = m0 + wm * 128 + 16 * fm + i16;
        const float* rp = p.res + (m < p.M ? m : p.M - 1) * p.ldr + nbase;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 rv = reinterpret_cast<const f32x4*>(rp)[q];
          v[4 * q] += rv.x; v[4 * q + 1] += rv.y; v[4 * q + 2] += rv.z; v[4 * q + 3] += rv.w;
        }
      }
      const int row = 16 * f + i16, sw = row & 7;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(mine + row * 256 + (((4 * g + q) ^ sw) << 4)) =
            f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = it * 4 + rr;
      const f32x4 d = *reinterpret_cast<const f32x4*>(mine + row * 256 + ((cc ^ (row & 7)) << 4));
      const int64_t m = m0 + wm * 128 + 64 * half + row;
      if (m < p.M) __builtin_nontemporal_store(d, reinterpret_cast<f32x4*>(cbase + m * p.ldc));
    }
  }
}

// WN = number of wave columns: 4 -> 256 x 256 tile, 512 threads, one workgroup per CU (4-stage ring, 128 KiB) -- the
// one in use.  (WN = 2, NST = 3 is a 256 x 128 tile with two workgroups per CU; measured 5-15 % slower on every
// backbone shape -- co-resident workgroups start together and stay in phase -- and not instantiated.)
// (Two co-resident 8-wave workgroups per CU -- a two-stage 64 KiB ring each -- do not fit: the 128 x 64 wave tile alone
// holds 128 accumulator registers, the kernel needs ~250 of the 256 a wave gets at two waves per SIMD.)
// PRIO: static s_setprio(1) for the second-dispatched half of the waves (the arbitration loser of every K-stage).
template <typename T, int WN, int NST, int PRIO = 0>
__global__ __launch_bounds__(128 * WN, 2) void linear_kernel_256(const LinearArgs p_in) {
  const LinearArgs p = batch_problem(p_in);
  constexpr int NTHR = 128 * WN;
  constexpr int XP = (BM2 * 4) / NTHR;            // 16-byte pieces of the activation tile per thread and stage
  constexpr int WP = (64 * WN * 4) / NTHR;        // ... of the weight tile (= 2)
  constexpr int LPS = XP + WP;                    // LDS-DMA instructions per lane and stage
  constexpr int OPER_X = BM2 * ROW2, OPER_W = 64 * WN * ROW2, STAGE = OPER_X + OPER_W;
  static_assert((NST == 4 && LPS == 4) || (NST == 3 && LPS == 6), "waitcnt immediates below assume these");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  if constexpr (PRIO != 0) {
    if (wave >= 2 * WN) __builtin_amdgcn_s_setprio(1);
  }

  uint32_t tile_m, tile_n;
  tile_of_block(blockIdx.x, (uint32_t)p.n_blocks, (uint32_t)(p.n_blocks / p.tiles_n), (uint32_t)p.tiles_n, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * BM2;
  const int n0 = (int)tile_n * 64 * WN;

  const char* src_x[XP];
  const char* src_w[WP];
#pragma unroll
  for (int r = 0; r < XP; ++r) {
    const int id = r * NTHR + tid;
    const int row = id >> 2, c = id & 3;
    int64_t gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    src_x[r] = p.A + gm * p.lda_b + ((c ^ swz2_x(row)) << 4);
  }
#pragma unroll
  for (int r = 0; r < WP; ++r) {
    const int id = r * NTHR + tid;
    const int row = id >> 2, c = id & 3;
    int gn = n0 + row;
    gn = gn < p.N ? gn : p.N - 1;
    src_w[r] = p.W + (int64_t)gn * p.ldw_b + ((c ^ swz2_w(row)) << 4);
  }
  auto stage = [&](int kt) {
    const int64_t koff = (int64_t)kt * ROW2;
    char* base = smem + (kt % NST) * STAGE;
#pragma unroll
    for (int r = 0; r < XP; ++r)  // wave-uniform LDS address; the hardware adds lane * 16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_x[r] + koff),
                                       (lds_ptr_t)(base + (r * NTHR + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int r = 0; r < WP; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_w[r] + koff),
                                       (lds_ptr_t)(base + OPER_X + (r * NTHR + wave * 64) * 16), 16, 0, 0);
  };

  const int i16 = lane & 15, g = lane >> 4;
  int off_x[8], off_w[4];
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const int row = wm * 128 + 16 * f + i16;
    off_x[f] = row * ROW2 + ((g ^ swz2_x(row)) << 4);
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int row = wn * 64 + 16 * (i16 >> 2) + 4 * f + (i16 & 3);
    off_w[f] = OPER_X + row * ROW2 + ((g ^ swz2_w(row)) << 4);
  }

  f32x4 acc[4][8];  // [fn][fm]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline (nt is even: K * sizeof(T) is a multiple of 128 bytes):
  //   LDS ring   : stages kt+2 .. kt+NST in flight while stage kt is multiplied
  //   registers  : the fragments of stage kt+1 are read from LDS while the 32 MFMAs of stage kt
  //                run on the other fragment set -- the matrix pipe never waits for a ds_read.
  const int nt = p.k_tiles;
  auto read_frags = [&](int kt, u32x4 (&fw)[4], u32x4 (&fx)[8]) {
    const char* buf = smem + (kt % NST) * STAGE;
#pragma unroll
    for (int f = 0; f < 4; ++f) fw[f] = *reinterpret_cast<const u32x4*>(buf + off_w[f]);
#pragma unroll
    for (int f = 0; f < 8; ++f) fx[f] = *reinterpret_cast<const u32x4*>(buf + off_x[f]);
  };
  // One pipeline step.  Order matters: the first MFMAs of stage kt need only registers (their
  // ds_reads were issued a whole step ago, so the compiler's lgkmcnt(0) in front of them is free);
  // then stage kt+1 is made visible (counted vmcnt + raw barrier), the ring is refilled and the
  // fragments of stage kt+1 are requested; the remaining MFMAs of stage kt cover that latency.
  auto mma_rows = [&](u32x4 (&cw)[4], u32x4 (&cx)[8], int fm_lo, int fm_hi) {
#pragma unroll
    for (int fm = fm_lo; fm < fm_hi; ++fm)
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = Mma<T>::run(cw[fn], cx[fm], acc[fn][fm]);
  };
  auto step = [&](int kt, u32x4 (&cw)[4], u32x4 (&cx)[8], u32x4 (&nw)[4], u32x4 (&nx)[8]) {
    mma_rows(cw, cx, 0, 2);
    __builtin_amdgcn_sched_barrier(0);
    // my pieces of stage kt+1 have landed once only the younger stages (NST - 2 of them) remain outstanding
    if constexpr (NST == 4) {
      if (kt + 3 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (kt + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (kt + 2 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // RAW: stage kt+1 complete.  WAR: everyone has read stage kt.
    asm volatile("" ::: "memory");
    if (kt + NST < nt) stage(kt + NST);  // into stage kt's buffer
    read_frags(kt + 1, nw, nx);
    __builtin_amdgcn_sched_barrier(0);
    mma_rows(cw, cx, 2, 8);
  };

  stage(0);
  stage(1);  // nt >= 2
  if (nt > 2) stage(2);
  if constexpr (NST == 4) {
    if (nt > 3) stage(3);
    if (nt > 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (nt > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    if (nt > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  u32x4 fwA[4], fxA[8], fwB[4], fxB[8];
  read_frags(0, fwA, fxA);
  for (int kt = 0; kt + 2 < nt; kt += 2) {
    step(kt, fwA, fxA, fwB, fxB);
    step(kt + 1, fwB, fxB, fwA, fxA);
  }
  step(nt - 2, fwA, fxA, fwB, fxB);  // fetches the last stage
  mma_rows(fwB, fxB, 0, 8);

  if constexpr (sizeof(T) == 2) {
    if (p.C2 == nullptr && p.res == nullptr && p.vec_store) {   // (uniform)
      __syncthreads();   // every wave is done with the ring
      // (in parts, as in linear_kernel_256pp below)
      if (p.act == AURORA_ACT_GELU) epilogue_256_bf16_coalesced<4>(p, acc, m0, n0, wm, wn, wave, lane, smem);
      else epilogue_256_bf16_coalesced<2>(p, acc, m0, n0, wm, wn, wave, lane, smem);
      return;
    }
  } else if constexpr (WN == 4) {
    if (p.C2 == nullptr && p.vec_store) {
      __syncthreads();
      epilogue_256_f32_coalesced(p, acc, m0, n0, wm, wn, wave, lane, smem);
      return;
    }
  }
  epilogue_256<T>(p, acc, m0, n0, wm, wn, i16, g);
}


// =================================================================================================
// Ping-pong form of the 256 x 256 ring kernel (bf16): the two waves that share a SIMD never issue MFMAs at the same time.
//
// In linear_kernel_256 all eight waves run the same stream in phase -- 8 MFMAs, wait, barrier, DMA issue, 12 fragment
// reads, 24 MFMAs -- and the two waves of a SIMD compete for its matrix pipe and issue slots (measured in round 1:
// 1668 cycles per K-stage for 1024 cycles of MFMA work, one wave of each pair parked ~580 cycles per stage).  Here every
// wave alternates a LOAD phase L(s) (12 fragment reads of stage s, DMA issue of stage s+3, counted waits) and a MATRIX
// phase M(s) (32 back-to-back MFMAs), with a barrier after each -- and waves 4-7 (the SIMD partners of waves 0-3) run
// ONE PHASE BEHIND, by a single extra barrier in front (balanced by one for waves 0-3 at the end).  So in every
// wall-clock phase a SIMD's matrix pipe belongs to exactly one wave while its partner does the LDS / DMA work; the
// instruction stream is the same for all waves, and fragments are read one phase before they are multiplied (one
// register set instead of two).
// Ring bookkeeping (per wave, iteration s): stage s+1 is published by the barrier that ends L(s) -- every wave has by
// then waited for ITS pieces of it (vmcnt(8): stages s+2, s+3 are younger) -- and is read no earlier than L(s+1); the
// DMA of stage s+3 in L(s) overwrites the buffer of stage s-1, whose last reader (a late wave's L(s-1), finished with
// lgkmcnt(0) before its barrier) is at least one barrier in the past for early and late waves alike.
// =================================================================================================
// 16 bytes written THROUGH to memory (sc1): the payload of an in-launch hand-off needs no release fence then, only the
// writing wave's own `s_waitcnt vmcnt(0)` before the flag (MI355X guide, "publish-large": 3.0 against 8.2 us per 64 KiB).
// hipcc does not count an asm store: the callers drain explicitly.
__device__ __forceinline__ void store_through(float* dst, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}

constexpr int PP_LDS = NSTAGE2 * STAGE2;            // the ring: 128 KiB

// SPLIT: split-K.  Launches with fewer tiles than CUs and a long K (a latitude band's coarse stages: 72 tiles, K = 8192)
//   cut every tile's K range into `split` slices, one workgroup each.  A slice publishes its fp32 accumulators to its
//   slab (write-through stores, drained), takes a ticket of the tile; whoever draws the last ticket -- no workgroup ever
//   waits for another, so nothing depends on dispatch order or co-residency -- adds the slices up IN SLICE ORDER (two
//   slices: its registers + the other slab, commutative; more: all slabs from memory, its own included, so that the sum
//   does not depend on who came last), resets the ticket for the next launch and runs the ordinary epilogue.
// Built, measured and deleted again in round 4 (profiles/r04_ab_gemm_variants_isolated.log, r04_ab_gemm_persistent_instep.json):
//   * a persistent form (workgroups walk over tiles; the next tile's bias row and first K-stages are requested before the
//     epilogue, which keeps the buffer of stage 3 for its transposition; counted waits raised by the 16 result stores that
//     sit behind the prologue in the in-order counter): -0.8 % over the step's shapes in isolation (-2...-4 % on the
//     K = 512 ones), 134.6 against 134.2 ms inside the step -- the cold start of a tile is not what its fixed cost is;
//   * a register-only epilogue (lanes i16 and i16 ^ 8 trade 16-byte pieces by a DPP rotation, stores cover whole 128-byte
//     rows): +1 % -- a 16-lane group then writes eight 32-byte pieces, the LDS-transposed form two whole rows;
//   * narrower n-groups of the tile order (4 or 2 n-tiles instead of 8, so that the weight panels of a round might survive
//     in L2 at K >= 1024): L2 -> fabric reads unchanged (-4 %; the activation panels streaming through evict them anyway),
//     time unchanged (profiles/r04_pmc_gemm_fetch_by_group_width.txt).
template <bool SPLIT>
__global__ __launch_bounds__(THREADS2, 2) void linear_kernel_256pp(const LinearArgs p_in) {
  const LinearArgs p = batch_problem(p_in);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // waves w and w+4 share a SIMD; wm = 1 runs one phase behind
  const uint32_t nb = (uint32_t)p.n_blocks;

  uint32_t item = blockIdx.x;   // tile, in launch order
  int kb = 0, nt = p.k_tiles, part = 0;
  if constexpr (SPLIT) {
    part = (int)(item / nb);
    item -= (uint32_t)part * nb;
    kb = (int)((int64_t)part * p.k_tiles / p.split);
    nt = (int)((int64_t)(part + 1) * p.k_tiles / p.split) - kb;   // >= 4 (dispatch)
  }
  uint32_t tile_m, tile_n;
  tile_of_block(item, nb, nb / (uint32_t)p.tiles_n, (uint32_t)p.tiles_n, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * BM2;
  const int n0 = (int)tile_n * BN2;

  const char* src_x[2];
  const char* src_w[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int id = r * THREADS2 + tid;
    const int row = id >> 2, c = id & 3;
    int64_t gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    int gn = n0 + row;
    gn = gn < p.N ? gn : p.N - 1;
    src_x[r] = p.A + gm * p.lda_b + ((c ^ swz2_x(row)) << 4) + (int64_t)kb * ROW2;
    src_w[r] = p.W + (int64_t)gn * p.ldw_b + ((c ^ swz2_w(row)) << 4) + (int64_t)kb * ROW2;
  }
  auto stage = [&](int kt) {
    const int64_t koff = (int64_t)kt * ROW2;
    char* base = smem + (kt & (NSTAGE2 - 1)) * STAGE2;
#pragma unroll
    for (int r = 0; r < 2; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_x[r] + koff),
                                       (lds_ptr_t)(base + (r * THREADS2 + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int r = 0; r < 2; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_w[r] + koff),
                                       (lds_ptr_t)(base + OPER2 + (r * THREADS2 + wave * 64) * 16), 16, 0, 0);
  };
  const int i16 = lane & 15, g = lane >> 4;
  int off_x[8], off_w[4];
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const int row = wm * 128 + 16 * f + i16;
    off_x[f] = row * ROW2 + ((g ^ swz2_x(row)) << 4);
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int row = wn * 64 + 16 * (i16 >> 2) + 4 * f + (i16 & 3);
    off_w[f] = OPER2 + row * ROW2 + ((g ^ swz2_w(row)) << 4);
  }
  f32x4 acc[4][8];  // [fn][fm]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The lane's 16 bias values, for the epilogue: requested NOW, in front of (and so older than) every piece the counted waits
  // below leave in flight; 16 registers the main loop does not need (201 of 256 before).  In the step 127.6 -> 127.3 ms
  // (profiles/r06_ab_epilogue_parts.log).
  float bias_pre[16];
  {
    const int nb16 = n0 + wn * 64 + 16 * g;
#pragma unroll
    for (int t = 0; t < 16; ++t) bias_pre[t] = p.bias ? p.bias[nb16 + t] : 0.f;   // (N is a multiple of the tile width here)
  }
  asm volatile("" ::: "memory");
  stage(0);
  stage(1);
  stage(2);
  stage(3);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // stage 0 is complete
  asm volatile("" ::: "memory");
  if (wm == 1) __builtin_amdgcn_s_barrier();   // the late half: one phase behind from here on

  for (int s = 0; s < nt; ++s) {
    // ---- L(s): fragments of stage s, refill the ring, settle what the next barrier publishes ----
    u32x4 fw[4], fx[8];
    {
      const char* buf = smem + (s & (NSTAGE2 - 1)) * STAGE2;
#pragma unroll
      for (int f = 0; f < 4; ++f) fw[f] = *reinterpret_cast<const u32x4*>(buf + off_w[f]);
#pragma unroll
      for (int f = 0; f < 8; ++f) fx[f] = *reinterpret_cast<const u32x4*>(buf + off_x[f]);
    }
    if (s >= 1 && s + 3 < nt) stage(s + 3);   // into the buffer of stage s-1
    if (s + 3 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // own pieces of stage s+1 have landed
    else if (s + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- M(s): the matrix pipe is this wave's alone ----
#pragma unroll
    for (int fm = 0; fm < 8; ++fm)
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = Mma<bf16_t>::run(fw[fn], fx[fm], acc[fn][fm]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();   // the early half waits for the late half's last phase: the ring is dead
  asm volatile("" ::: "memory");

  if constexpr (SPLIT) {
    // ---- publish this slice, take a ticket; the last arriver combines ----
    float* const slab0 = p.slabs + (int64_t)item * p.split * (BM2 * BN2);
    const int64_t mine_off = (int64_t)(wave * 32 * 64 + lane) * 4;
    {
      float* const mine = slab0 + (int64_t)part * (BM2 * BN2) + mine_off;
#pragma unroll
      for (int fn = 0; fn < 4; ++fn)
#pragma unroll
        for (int fm = 0; fm < 8; ++fm) store_through(mine + (fn * 8 + fm) * 256, acc[fn][fm]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its own stores ...
    __syncthreads();                                    // ... before ONE lane takes the ticket
    int* const s_ticket = reinterpret_cast<int*>(smem);
    if (tid == 0) *s_ticket = __hip_atomic_fetch_add(p.tickets + item, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = *s_ticket;
    if (ticket != p.split - 1) return;   // (uniform) someone else will finish this tile
    if (tid == 0) __hip_atomic_store(p.tickets + item, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
    if (wave == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop this CU's stale L1 lines, once
    __syncthreads();
    const float* const src = slab0 + mine_off;
    // Two slices: registers + the other slab (commutative).  More: every slab from memory, this one's own included,
    // in slice order -- the sum must not depend on who drew the last ticket.  Eight loads in flight per lane (the
    // accumulators hold 128 registers; the fence keeps hipcc from hoisting a slab's 32 loads above the first add).
    const bool two = p.split == 2;
    if (!two) {
#pragma unroll
      for (int fn = 0; fn < 4; ++fn)
#pragma unroll
        for (int fm = 0; fm < 8; ++fm) acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int j = 0; j < p.split; ++j) {
      if (two && j == part) continue;
      const float* const o = src + (int64_t)j * (BM2 * BN2);
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) {
        f32x4 t[8];
#pragma unroll
        for (int fm = 0; fm < 8; ++fm) t[fm] = *reinterpret_cast<const f32x4*>(o + (fn * 8 + fm) * 256);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int fm = 0; fm < 8; ++fm) acc[fn][fm] += t[fm];
      }
    }
  }
  if (p.C2 == nullptr && p.res == nullptr && p.vec_store) {   // (uniform)
    // In parts (64 / 32 rows of the wave's 128 at a time): the stores of one part are in flight while the next part's bias,
    // activation and packing run on the VALU -- in one pass the GELU of a stage-0 fc1 tile (11 VALU instructions per two
    // values, 9 K cycles per SIMD) ran with no memory operation in flight.  In the step: 129.7 -> 128.2 ms
    // (profiles/r06_ab_epilogue_parts.log); the same values either way.
    if (p.act == AURORA_ACT_GELU) epilogue_256_bf16_coalesced<4>(p, acc, m0, n0, wm, wn, wave, lane, smem, &bias_pre);
    else epilogue_256_bf16_coalesced<2>(p, acc, m0, n0, wm, wn, wave, lane, smem, &bias_pre);
    return;
  }
  epilogue_256<bf16_t>(p, acc, m0, n0, wm, wn, i16, g);
}

constexpr int MID_LDS = 3 * (BM2 + 128) * ROW2;   // 256 x 128 tiles: three stages of 24 KiB, two workgroups per CU

// =================================================================================================
// fp32 linear layers on the bf16 matrix pipe: "3 x bf16" operand splitting.
//
// gfx950 multiplies bf16 sixteen times faster than fp32 on the matrix cores (v_mfma_f32_16x16x32_bf16:
// 16 Ki FLOP in 16 cycles; v_mfma_f32_16x16x4_f32: 2 Ki FLOP in 32 cycles).  An fp32 number is EXACTLY the
// sum of three bf16 numbers (8 + 8 + 8 significand bits, by truncation): a = a_h + a_m + a_l.  Then
//     a.b = a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_l b_h + a_m b_m) + O(2^-24 |a||b|)
// and every bf16 x bf16 product is exact in the fp32 accumulator, so six bf16 MFMAs reproduce the fp32
// product to ~1.2e-7 relative (the three dropped terms), the same order as the 2^-24 rounding an fp32 FMA
// chain commits per step: an fp32-grade GEMM at 16/6 = 2.7x the fp32 MFMA rate.  (The encoder and decoder of
// Aurora are fp32 upstream, outside autocast; this keeps them fp32-accurate.  tests/test_gpu_ops.py measures
// both this kernel and the native-fp32 one against an fp64 product.)
//
// Same 256 x 256 tile, LDS-DMA staging, swizzles and epilogue as linear_kernel_256<float>; a K-stage is 16
// fp32 per row, so two stages (a "pair") make the K = 32 of one bf16 MFMA: lane (row, g) holds fp32
// k = 4g..4g+3 of both stages, which become its 8 bf16 k-slots (the k order is free as long as both operands
// agree).  Splitting is done on the fragments in registers: ~36 VALU ops per 8-value fragment, 12 fragments
// per pair and wave against 192 MFMAs (3072 matrix-pipe cycles), so the VALU work hides under the MFMAs.
// Ring: pair j is consumed while pair j+1 (64 KiB) is in flight; one barrier per pair.
// =================================================================================================
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct Split3 { u32x4 h, m, l; };

__device__ __forceinline__ void split_pair(uint32_t a0, uint32_t a1, uint32_t& h, uint32_t& m, uint32_t& l) {
  // top 16 bits of two fp32 -> one packed bf16x2 word (truncation), remainder exact in fp32
  constexpr uint32_t SEL = 0x07060302u;
  h = __builtin_amdgcn_perm(a1, a0, SEL);
  const float r0 = __uint_as_float(a0) - __uint_as_float(a0 & 0xffff0000u);
  const float r1 = __uint_as_float(a1) - __uint_as_float(a1 & 0xffff0000u);
  const uint32_t q0 = __float_as_uint(r0), q1 = __float_as_uint(r1);
  m = __builtin_amdgcn_perm(q1, q0, SEL);
  const float s0 = r0 - __uint_as_float(q0 & 0xffff0000u);
  const float s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), SEL);
}

__device__ __forceinline__ Split3 split8(u32x4 a, u32x4 b) {
  uint32_t h[4], m[4], l[4];
  split_pair(a.x, a.y, h[0], m[0], l[0]);
  split_pair(a.z, a.w, h[1], m[1], l[1]);
  split_pair(b.x, b.y, h[2], m[2], l[2]);
  split_pair(b.z, b.w, h[3], m[3], l[3]);
  return Split3{u32x4{h[0], h[1], h[2], h[3]}, u32x4{m[0], m[1], m[2], m[3]}, u32x4{l[0], l[1], l[2], l[3]}};
}

__device__ __forceinline__ f32x4 mma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                 c, 0, 0, 0);
}

// ---- second variant: two fp16 terms ----
// With round-to-nearest, a = a_h + a_l where a_h = fp16(a) and a_l = fp16(a - a_h) reproduces a to 2^-24 |a| (the
// remainder is exact in fp32, has <= 14 significant bits and loses at most its last three to the 11-bit fp16
// significand), so  a.b = a_h b_h + a_h b_l + a_l b_h + O(2^-24 |a||b|)  needs THREE MFMAs (fp16 x fp16 products are
// exact in the fp32 accumulator) and 5 VALU operations per operand pair instead of six MFMAs and 9.  The price is
// fp16's range: a_h overflows at |a| >= 65520, and a_l is a subnormal for |a| < 0.25, i.e. carries an ABSOLUTE error
// of up to 3e-8.  The weight operand is therefore scaled by 2^6 on the fly (nn.Linear weights are O(1e-2); the
// accumulators are scaled back exactly in the epilogue) and the variant is only used where the caller vouches for
// activations that are bounded by construction (f32_gemm = 2 of aurora_hip_linear_ex: LayerNorm outputs and their GELU'd linears).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
struct Split2 { u32x4 h, l; };

template <bool SCALE>
__device__ __forceinline__ Split2 split8_f16(u32x4 a, u32x4 b) {
  constexpr float S = SCALE ? 64.0f : 1.0f;
  uint32_t h[4], l[4];
  split_pair_f16(__uint_as_float(a.x) * S, __uint_as_float(a.y) * S, h[0], l[0]);
  split_pair_f16(__uint_as_float(a.z) * S, __uint_as_float(a.w) * S, h[1], l[1]);
  split_pair_f16(__uint_as_float(b.x) * S, __uint_as_float(b.y) * S, h[2], l[2]);
  split_pair_f16(__uint_as_float(b.z) * S, __uint_as_float(b.w) * S, h[3], l[3]);
  return Split2{u32x4{h[0], h[1], h[2], h[3]}, u32x4{l[0], l[1], l[2], l[3]}};
}
__device__ __forceinline__ f32x4 mma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <int TERMS>   // 3: three bf16 terms, six MFMAs;  2: two fp16 terms, three MFMAs
__global__ __launch_bounds__(THREADS2, 2) void linear_kernel_256_f32x3(const LinearArgs p_in) {
  const LinearArgs p = batch_problem(p_in);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Guarded launch: the host has launched BOTH variants; the word the caller left in device memory (max |activation|
  // or a bound of it) decides which one does the work -- two fp16 terms inside the safe range, three bf16 terms
  // otherwise -- and the other one retires at once.  Uniform: every workgroup reads the same word.
  if (p.guard != nullptr && (*p.guard < p.guard_limit) != (TERMS == 2)) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  uint32_t tile_m, tile_n;
  tile_of_block(blockIdx.x, (uint32_t)p.n_blocks, (uint32_t)(p.n_blocks / p.tiles_n), (uint32_t)p.tiles_n, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * BM2;
  const int n0 = (int)tile_n * BN2;

  const char* src_x[2];
  const char* src_w[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int id = r * THREADS2 + tid;
    const int row = id >> 2, c = id & 3;
    int64_t gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    int gn = n0 + row;
    gn = gn < p.N ? gn : p.N - 1;
    src_x[r] = p.A + gm * p.lda_b + ((c ^ swz2_x(row)) << 4);
    src_w[r] = p.W + (int64_t)gn * p.ldw_b + ((c ^ swz2_w(row)) << 4);
  }
  auto stage = [&](int kt) {
    const int64_t koff = (int64_t)kt * ROW2;
    char* base = smem + (kt & (NSTAGE2 - 1)) * STAGE2;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int off = (r * THREADS2 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_x[r] + koff),
                                       (lds_ptr_t)(base + off), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_w[r] + koff),
                                       (lds_ptr_t)(base + OPER2 + off), 16, 0, 0);
    }
  };

  const int i16 = lane & 15, g = lane >> 4;
  int off_x[8], off_w[4];
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const int row = wm * 128 + 16 * f + i16;
    off_x[f] = row * ROW2 + ((g ^ swz2_x(row)) << 4);
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int row = wn * 64 + 16 * (i16 >> 2) + 4 * f + (i16 & 3);
    off_w[f] = OPER2 + row * ROW2 + ((g ^ swz2_w(row)) << 4);
  }

  f32x4 acc[4][8];  // [fn][fm]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int np = p.k_tiles >> 1;  // pairs of stages (k_tiles is even)
  stage(0);
  stage(1);
  if (np > 1) {
    stage(2);
    stage(3);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  constexpr bool use_two = TERMS == 2;
  auto pair3 = [&](const char* bufa, const char* bufb) {
      Split3 w[4];
#pragma unroll
      for (int f = 0; f < 4; ++f)
        w[f] = split8(*reinterpret_cast<const u32x4*>(bufa + off_w[f]), *reinterpret_cast<const u32x4*>(bufb + off_w[f]));
      u32x4 ra = *reinterpret_cast<const u32x4*>(bufa + off_x[0]);
      u32x4 rb = *reinterpret_cast<const u32x4*>(bufb + off_x[0]);
#pragma unroll
      for (int fm = 0; fm < 8; ++fm) {
        const Split3 x = split8(ra, rb);
        if (fm + 1 < 8) {
          ra = *reinterpret_cast<const u32x4*>(bufa + off_x[fm + 1]);
          rb = *reinterpret_cast<const u32x4*>(bufb + off_x[fm + 1]);
        }
        // smallest terms first; consecutive MFMAs go to different accumulators
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = mma_bf16(w[fn].l, x.h, acc[fn][fm]);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = mma_bf16(w[fn].h, x.l, acc[fn][fm]);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = mma_bf16(w[fn].m, x.m, acc[fn][fm]);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = mma_bf16(w[fn].m, x.h, acc[fn][fm]);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = mma_bf16(w[fn].h, x.m, acc[fn][fm]);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = mma_bf16(w[fn].h, x.h, acc[fn][fm]);
      }
  };
  auto pair2 = [&](const char* bufa, const char* bufb) {
      Split2 w[4];
#pragma unroll
      for (int f = 0; f < 4; ++f)
        w[f] = split8_f16<true>(*reinterpret_cast<const u32x4*>(bufa + off_w[f]),
                                *reinterpret_cast<const u32x4*>(bufb + off_w[f]));
      u32x4 ra = *reinterpret_cast<const u32x4*>(bufa + off_x[0]);
      u32x4 rb = *reinterpret_cast<const u32x4*>(bufb + off_x[0]);
#pragma unroll
      for (int fm = 0; fm < 8; ++fm) {
        const Split2 x = split8_f16<false>(ra, rb);
        if (fm + 1 < 8) {
          ra = *reinterpret_cast<const u32x4*>(bufa + off_x[fm + 1]);
          rb = *reinterpret_cast<const u32x4*>(bufb + off_x[fm + 1]);
        }
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = mma_f16(w[fn].l, x.h, acc[fn][fm]);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = mma_f16(w[fn].h, x.l, acc[fn][fm]);
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = mma_f16(w[fn].h, x.h, acc[fn][fm]);
      }
  };
  for (int j = 0; j < np; ++j) {
    const char* bufa = smem + ((2 * j) & (NSTAGE2 - 1)) * STAGE2;
    const char* bufb = smem + ((2 * j + 1) & (NSTAGE2 - 1)) * STAGE2;
    if constexpr (TERMS == 3) pair3(bufa, bufb);
    else pair2(bufa, bufb);
    if (j + 1 < np) {
      // RAW: my pieces of pair j+1 (issued a whole pair ago) have landed; WAR: everyone has read pair j.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (j + 2 < np) {
        stage(2 * j + 4);
        stage(2 * j + 5);
      }
    }
  }
  if constexpr (use_two) {   // undo the 2^6 weight scale (exact)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] *= 0.015625f;
  }
  if (p.C2 == nullptr && p.vec_store) {   // (uniform)
    __syncthreads();   // every wave is done with the ring
    epilogue_256_f32_coalesced(p, acc, m0, n0, wm, wn, wave, lane, smem);
    return;
  }
  epilogue_256<float>(p, acc, m0, n0, wm, wn, i16, g);
}


// =================================================================================================
// fp32 linears by two fp16 terms, ping-pong form: 128 x 256 tile, whole-line K-stages, three-stage ring.
//
// linear_kernel_256_f32x3<2> above runs all eight waves in phase through "read fragments, split, 96 MFMAs" with one
// barrier and a full drain (vmcnt(0)) per K = 32: the two waves of a SIMD fight over its matrix pipe and its VALU issue
// (the split costs ~240 VALU instructions per wave and K = 32 against 96 MFMAs), and the kernel sits at 290 TFLOP/s
// fp32-equivalent = 0.87 PFLOP/s of fp16 MFMA work where the bf16 kernels reach 1.15-1.4.  The ping-pong schedule of
// linear_kernel_256pp needs every LDS read of a stage inside the LOAD phase (the partner's DMA refills the ring during
// the MATRIX phase) and a ring at least three K-steps deep; fp32 operands of a 256 x 256 tile are 64 KiB per K = 32,
// i.e. two steps.  Hence this geometry:
//   * tile 128 x 256, 8 waves as 2 (m) x 4 (n), wave tile 64 x 64 = 4 x 4 fragments (64 accumulator registers);
//   * a K-stage is 128 BYTES of every operand row (32 fp32 = one fp16 MFMA of K = 32): 16 KiB of activations + 32 KiB of
//     weights, staged by LDS-DMA in whole cache lines (8 rows x 128 B per wave instruction -- the pattern the vector
//     memory front end moves 4x faster than 16 rows x 64 B), XOR-swizzled as in the 128 x 128 kernel; 3 stages = 144 KiB;
//   * L(s): 16 fragment reads (raw fp32: 32 registers of activations; the weights are split to fp16 pairs at once),
//     DMA of stage s+2, counted wait for the wave's pieces of stage s+1;  M(s): per activation fragment one split
//     (20 VALU) + 12 MFMAs, the VALU work overlapping the wave's own matrix instructions;
//   * waves 4-7 run one phase behind waves 0-3 (one extra barrier in front, one behind for the others): a SIMD's matrix
//     pipe always belongs to exactly one wave.
// Same range contract and guard as the kernel above (which remains the fallback for K % 32 != 0).
// =================================================================================================
constexpr int VM = 128, VN = 256, VROW = 128, VTHREADS = 512, VNST = 3;
constexpr int VOPER_X = VM * VROW, VOPER_W = VN * VROW, VSTAGE = VOPER_X + VOPER_W;   // 16 + 32 = 48 KiB

// A_PRE / W_PRE: operand already in the fp16-pair layout: its split (all of its VALU work) disappears.
// TALL: the tile is 256 (m) x 128 (n) -- waves 4 x 2, the same 64 x 64 wave tile, the same 48 KiB stage (32 KiB of
// activations + 16 KiB of weights) and the same six LDS-DMA instructions per lane and stage -- for the narrow linears: the
// decoder's output heads have 80 real columns (5 variables x 16 pixels), which the 256-wide tile pads to 256 (69 % of
// its MFMAs on zeros), this one to 128.
template <bool A_PRE, bool W_PRE, bool TALL = false>
__global__ __launch_bounds__(VTHREADS, 2) void linear_kernel_f32pp(const LinearArgs p_in) {
  constexpr int TM = TALL ? 256 : VM, TN = TALL ? 128 : VN, OPX = TM * VROW, XP = TM / 64, WP = TN / 64;
  static_assert(OPX + TN * VROW == VSTAGE && XP + WP == 6, "stage size / DMA count the waitcnt immediates assume");
  const LinearArgs p = batch_problem(p_in);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (p.guard != nullptr && !(*p.guard < p.guard_limit)) return;   // guarded launch: the three-term kernel does the work
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int late = wave >> 2;   // waves w and w+4 share a SIMD; the late half runs one phase behind
  const int wm = TALL ? wave >> 1 : wave >> 2, wn = TALL ? wave & 1 : wave & 3;

  uint32_t tile_m, tile_n;
  tile_of_block<2>(blockIdx.x, (uint32_t)p.n_blocks, (uint32_t)(p.n_blocks / p.tiles_n), (uint32_t)p.tiles_n, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * TM;
  const int n0 = (int)tile_n * TN;

  const char* src_x[XP];
  const char* src_w[WP];
#pragma unroll
  for (int r = 0; r < XP; ++r) {
    const int id = r * VTHREADS + tid;
    const int row = id >> 3, c = id & 7;
    int64_t gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    src_x[r] = p.A + gm * p.lda_b + ((c ^ swz_x(row)) << 4);
  }
#pragma unroll
  for (int r = 0; r < WP; ++r) {
    const int id = r * VTHREADS + tid;
    const int row = id >> 3, c = id & 7;
    src_w[r] = p.W + (int64_t)(n0 + row) * p.ldw_b + ((c ^ swz_w(row)) << 4);
  }
  auto stage = [&](int kt) {
    const int64_t koff = (int64_t)kt * VROW;
    char* base = smem + (kt % VNST) * VSTAGE;
#pragma unroll
    for (int r = 0; r < XP; ++r)   // wave-uniform LDS address; the hardware adds lane * 16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_x[r] + koff),
                                       (lds_ptr_t)(base + (r * VTHREADS + wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int r = 0; r < WP; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_w[r] + koff),
                                       (lds_ptr_t)(base + OPX + (r * VTHREADS + wave * 64) * 16), 16, 0, 0);
  };
  // fragment read offsets.  Lane group g multiplies k = 8g..8g+7 of the stage: as fp32 that is chunks 2g and 2g + 1 of the
  // row, in the fp16-pair layout chunk g (high halves) and chunk g + 4 (remainders) -- the same k order either way, so
  // an operand may arrive split or not without changing a bit of the result.
  const int i16 = lane & 15, g = lane >> 4;
  int off_x[4][2], off_w[4][2];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int row_x = wm * 64 + 16 * f + i16;
    const int row_w = wn * 64 + 16 * (i16 >> 2) + 4 * f + (i16 & 3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cx = A_PRE ? g + 4 * ks : 2 * g + ks, cw = W_PRE ? g + 4 * ks : 2 * g + ks;
      off_x[f][ks] = row_x * VROW + ((cx ^ swz_x(row_x)) << 4);
      off_w[f][ks] = OPX + row_w * VROW + ((cw ^ swz_w(row_w)) << 4);
    }
  }
  f32x4 acc[4][4];  // [fn][fm]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt = p.k_tiles;   // K / 32, >= 3 (dispatch)
  stage(0);
  stage(1);
  stage(2);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // stage 0 is complete
  asm volatile("" ::: "memory");
  if (late == 1) __builtin_amdgcn_s_barrier();   // the late half: one phase behind from here on

  for (int s = 0; s < nt; ++s) {
    // ---- L(s) ----
    const char* buf = smem + (s % VNST) * VSTAGE;
    u32x4 xa[4], xb[4];
    Split2 w[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      xa[f] = *reinterpret_cast<const u32x4*>(buf + off_x[f][0]);
      xb[f] = *reinterpret_cast<const u32x4*>(buf + off_x[f][1]);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
      if constexpr (W_PRE)   // chunk g = high halves of k = 8g..8g+7, chunk g + 4 = their remainders
        w[f] = Split2{*reinterpret_cast<const u32x4*>(buf + off_w[f][0]), *reinterpret_cast<const u32x4*>(buf + off_w[f][1])};
      else
        w[f] = split8_f16<true>(*reinterpret_cast<const u32x4*>(buf + off_w[f][0]), *reinterpret_cast<const u32x4*>(buf + off_w[f][1]));
    if (s >= 1 && s + 2 < nt) stage(s + 2);   // into the buffer of stage s-1
    if (s + 2 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // own pieces of stage s+1 have landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    // ---- M(s): smallest terms first; consecutive MFMAs go to different accumulators ----
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
      Split2 x;
      if constexpr (A_PRE) x = Split2{xa[fm], xb[fm]};
      else x = split8_f16<false>(xa[fm], xb[fm]);
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = mma_f16(w[fn].l, x.h, acc[fn][fm]);
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = mma_f16(w[fn].h, x.l, acc[fn][fm]);
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) acc[fn][fm] = mma_f16(w[fn].h, x.h, acc[fn][fm]);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  }
  if (late == 0) __builtin_amdgcn_s_barrier();   // the early half waits for the late half's last phase
  asm volatile("" ::: "memory");

  // ---- epilogue: lane owns row m (per fm) x 16 consecutive features; undo the 2^6 weight scale (exact) ----
  const int nbase = n0 + wn * 64 + 16 * g;
  float bias_v[16];   // (requested before the main loop instead, as in linear_kernel_256pp: measured, no gain here -- the tiles are long)
#pragma unroll
  for (int t = 0; t < 16; ++t) bias_v[t] = p.bias ? p.bias[nbase + t] : 0.f;
  const bool vec = p.vec_store != 0;
  if (vec && p.C2 == nullptr && p.res == nullptr) {   // (uniform)
    // Plain result (fp32 or fp16 pairs): through LDS, so that a store instruction writes four whole 256-byte row
    // segments with consecutive lanes on consecutive 16-byte pieces.  The direct form below has lane (i16, g) write 16
    // bytes of row i16 at a 64-byte stride -- every instruction touches 32 cache lines, 32 bytes each, and the CU's
    // store path takes as long over a tile's 128 KiB as 6-8 K-stages of MFMAs.  The ring is dead after the main loop;
    // each wave takes 16 KiB of it for its 64 x 64 results, stored as the exact bytes of the output rows (the pair
    // layout keeps a wave's 64 features in 256 contiguous bytes too: two groups of 32 high halves + 32 remainders),
    // 16-byte pieces XOR-swizzled by row & 7.
    // (both halves are past their last LDS read: the barrier above is the late half's last in-loop one)
    // In F32PP_PARTS parts (32 / 16 of the wave's 64 rows at a time): a part's stores are in flight while the next part's
    // scaling, activation and splitting run on the VALU (as epilogue_256_bf16_coalesced; profiles/r06_ab_epilogue_parts.log).
    char* mine = smem + wave * 16384;
    const int rr = lane >> 4, cc = lane & 15;
    float* cbase = reinterpret_cast<float*>(p.C) + n0 + wn * 64 + cc * 4;
#pragma unroll
    for (int part = 0; part < F32PP_PARTS; ++part) {
#pragma unroll
    for (int fm = part * (4 / F32PP_PARTS); fm < (part + 1) * (4 / F32PP_PARTS); ++fm) {
      float v[16];
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) {
        v[4 * fn + 0] = fmaf(acc[fn][fm].x, 0.015625f, bias_v[4 * fn + 0]);
        v[4 * fn + 1] = fmaf(acc[fn][fm].y, 0.015625f, bias_v[4 * fn + 1]);
        v[4 * fn + 2] = fmaf(acc[fn][fm].z, 0.015625f, bias_v[4 * fn + 2]);
        v[4 * fn + 3] = fmaf(acc[fn][fm].w, 0.015625f, bias_v[4 * fn + 3]);
      }
      if (p.act == AURORA_ACT_GELU || p.act == ACT_GELU_FAST) {
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
          const f32x2_hw r = gelu_erf_fast2(f32x2_hw{v[t], v[t + 1]});
          v[t] = r.x;
          v[t + 1] = r.y;
        }
      } else if (p.act == AURORA_ACT_SILU) {
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = v[t] / (1.0f + expf(-v[t]));
      }
      const int row = 16 * fm + i16, sw = row & 7;
      char* lrow = mine + row * 256;
      if (p.out_split) {
        uint32_t h[8], l[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) split_pair_f16(v[2 * t], v[2 * t + 1], h[t], l[t]);
        // features 16g..16g+15 = halves 16 (g & 1).. of group g >> 1: pieces 8 (g >> 1) + 2 (g & 1) + {0, 1}, remainders + 4
        const int pc = 8 * (g >> 1) + 2 * (g & 1);
        *reinterpret_cast<u32x4*>(lrow + (((pc + 0) ^ sw) << 4)) = u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4*>(lrow + (((pc + 1) ^ sw) << 4)) = u32x4{h[4], h[5], h[6], h[7]};
        *reinterpret_cast<u32x4*>(lrow + (((pc + 4) ^ sw) << 4)) = u32x4{l[0], l[1], l[2], l[3]};
        *reinterpret_cast<u32x4*>(lrow + (((pc + 5) ^ sw) << 4)) = u32x4{l[4], l[5], l[6], l[7]};
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(lrow + (((4 * g + q) ^ sw) << 4)) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      }
    }
#pragma unroll
    for (int it = part * (16 / F32PP_PARTS); it < (part + 1) * (16 / F32PP_PARTS); ++it) {
      const int row = it * 4 + rr;
      const f32x4 d = *reinterpret_cast<const f32x4*>(mine + row * 256 + ((cc ^ (row & 7)) << 4));
      const int64_t m = m0 + wm * 64 + row;
      if (m < p.M) *reinterpret_cast<f32x4*>(cbase + m * p.ldc) = d;
    }
    }
    return;
  }
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const int64_t m = m0 + wm * 64 + 16 * fm + i16;
    if (m >= p.M) continue;
    float v[16];
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
      v[4 * fn + 0] = fmaf(acc[fn][fm].x, 0.015625f, bias_v[4 * fn + 0]);
      v[4 * fn + 1] = fmaf(acc[fn][fm].y, 0.015625f, bias_v[4 * fn + 1]);
      v[4 * fn + 2] = fmaf(acc[fn][fm].z, 0.015625f, bias_v[4 * fn + 2]);
      v[4 * fn + 3] = fmaf(acc[fn][fm].w, 0.015625f, bias_v[4 * fn + 3]);
    }
    if (p.act == AURORA_ACT_GELU || p.act == ACT_GELU_FAST) {
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        const f32x2_hw r = gelu_erf_fast2(f32x2_hw{v[t], v[t + 1]});
        v[t] = r.x;
        v[t + 1] = r.y;
      }
    } else if (p.act == AURORA_ACT_SILU) {
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = v[t] / (1.0f + expf(-v[t]));
    }
    if (p.res) {
      const float* rp = p.res + m * p.ldr + nbase;
      if (vec) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 rv = reinterpret_cast<const f32x4*>(rp)[q];
          v[4 * q] += rv.x; v[4 * q + 1] += rv.y; v[4 * q + 2] += rv.z; v[4 * q + 3] += rv.w;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] += rp[t];
      }
    }
    if (p.out_split) {
      // fp16-pair layout: the 16 features nbase.. are halves (nbase % 32) .. +15 of group nbase / 32 -- 32 bytes of high
      // halves, and 32 bytes of remainders 64 bytes further on
      uint32_t h[8], l[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) split_pair_f16(v[2 * t], v[2 * t + 1], h[t], l[t]);
      char* dst = p.C + (m * p.ldc + (nbase & ~31)) * 4 + (nbase & 31) * 2;
      reinterpret_cast<u32x4*>(dst)[0] = u32x4{h[0], h[1], h[2], h[3]};
      reinterpret_cast<u32x4*>(dst)[1] = u32x4{h[4], h[5], h[6], h[7]};
      reinterpret_cast<u32x4*>(dst + 64)[0] = u32x4{l[0], l[1], l[2], l[3]};
      reinterpret_cast<u32x4*>(dst + 64)[1] = u32x4{l[4], l[5], l[6], l[7]};
      continue;
    }
    store16<float>(reinterpret_cast<float*>(p.C) + m * p.ldc + nbase, v, vec, 16);
    if (p.C2) store16<bf16_t>(reinterpret_cast<bf16_t*>(p.C2) + m * p.ldc2 + nbase, v, vec, 16);
  }
}

// fp32 rows -> the fp16-pair layout the two-term kernels can take directly: per 32 features 128 bytes, the 32 fp16 high
// halves  h = fp16(s x)  followed by the 32 remainders  l = fp16(s x - h).  One lane per 8 features.
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ src, int64_t ld_src, char* __restrict__ dst,
                                                        int64_t ld_dst, int64_t rows, int K, float scale) {
  const int per_row = K >> 3;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= rows * per_row) return;
  const int64_t r = id / per_row;
  const int c = (int)(id - r * per_row) * 8;
  const f32x4 a = *reinterpret_cast<const f32x4*>(src + r * ld_src + c);
  const f32x4 b = *reinterpret_cast<const f32x4*>(src + r * ld_src + c + 4);
  uint32_t h[4], l[4];
  split_pair_f16(a.x * scale, a.y * scale, h[0], l[0]);
  split_pair_f16(a.z * scale, a.w * scale, h[1], l[1]);
  split_pair_f16(b.x * scale, b.y * scale, h[2], l[2]);
  split_pair_f16(b.z * scale, b.w * scale, h[3], l[3]);
  char* d = dst + (r * ld_dst + (c & ~31)) * 4 + (c & 31) * 2;
  *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<u32x4*>(d + 64) = u32x4{l[0], l[1], l[2], l[3]};
}


// =================================================================================================
// bf16 linear + (adaptive) LayerNorm + residual in one launch, for D = 512 (stage 0 of the backbone):
//     x_out = x_in + LN(A W^T + bias) * gain + shift,   shadow = bf16(x_out)
// i.e. `x = shortcut + norm(proj(...), c)` / `x = x + norm(mlp(x), c)` of a Swin block (swin3d.py:507-508, film.py:38-49)
// without the bf16 round trip of the linear's result through HBM (2 of the 14 bytes per element the linear + LayerNorm
// pair moves) and without the second launch.  A workgroup must own whole rows: the tile is 128 x 512 -- 8 waves as
// 2 (m) x 4 (n), wave tile 64 x 128 (128 accumulator registers, the same as the 128 x 64 tile of the square kernels),
// K-stages of 64 bytes per row (8 KiB of activations + 32 KiB of weights), three-stage ring, ping-pong schedule.
// Epilogue: bias, rounding to bf16 (the reference's linear yields bf16 under autocast; statistics are taken of the
// rounded values, as the separate kernels do), two-pass fp32 row statistics (lane -> 4 lane groups by permlane swaps ->
// 4 waves through LDS), then 16 rows at a time through LDS so that every global access of the residual stream covers
// whole cache lines: a lane reads 4 consecutive features of a row, normalises, adds the fp32 residual, writes fp32 and
// bf16.  D = 1024 / 2048 would need 64 / 32-row tiles (fetch-bound) or a cross-workgroup statistics exchange: not built.
// =================================================================================================
constexpr int FM = 128, FN = 512, FTHREADS = 512, FNST = 3;
constexpr int FOPER_X = FM * ROW2, FOPER_W = FN * ROW2, FSTAGE = FOPER_X + FOPER_W;   // 8 + 32 = 40 KiB

struct LinearLnArgs {
  const char* A; int64_t lda_b; const char* W; int64_t ldw_b;
  const float* bias; const float* gain; const float* shift;
  const float* x_in; int64_t ldx; float* x_out; int64_t ldo; bf16_t* xb; int64_t ldb;
  int64_t M; int k_tiles; float eps;
  int64_t tile0;   // first tile of this launch
};

__device__ __forceinline__ float group4_sum(float v) {   // over the 4 lane groups (lanes l, l^16, l^32, l^48)
  typedef uint32_t u32x2_sw __attribute__((ext_vector_type(2)));
  u32x2_sw r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

template <bool FULL>   // FULL: every row of every tile of the launch exists
__global__ __launch_bounds__(FTHREADS, 2) void linear_ln512_kernel(const LinearLnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // waves w and w+4 share a SIMD; wm = 1 runs one phase behind
  const int64_t m0 = (p.tile0 + blockIdx.x) * FM;
#ifdef LN_PROBE_TIMES
  uint64_t ts[8];
  ts[0] = wall_clock64();
#define LN_TS(i) ts[i] = wall_clock64()
#else
#define LN_TS(i)
#endif
  // (Do the CUs of a launch run in lockstep -- every main loop at once with HBM idle, then every epilogue at once?  Holding
  // the first-round workgroups of every other CU back by 8 ... 55 us changed nothing but the delay itself,
  // profiles/r04_ab_ln512_stagger.log: a CU's epilogue is as fast as the bytes it keeps in flight allow, whatever its
  // neighbours do.)
  const char* src_x;
  const char* src_w[4];
  {
    const int row = tid >> 2, c = tid & 3;
    int64_t gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    src_x = p.A + gm * p.lda_b + ((c ^ swz2_x(row)) << 4);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int id = r * FTHREADS + tid;
    const int row = id >> 2, c = id & 3;
    src_w[r] = p.W + (int64_t)row * p.ldw_b + ((c ^ swz2_w(row)) << 4);
  }
  auto stage = [&](int kt) {
    const int64_t koff = (int64_t)kt * ROW2;
    char* base = smem + (kt % FNST) * FSTAGE;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_x + koff),
                                     (lds_ptr_t)(base + (wave * 64) * 16), 16, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_w[r] + koff),
                                       (lds_ptr_t)(base + FOPER_X + (r * FTHREADS + wave * 64) * 16), 16, 0, 0);
  };
  const int i16 = lane & 15, g = lane >> 4;
  int off_x[4], off_w[8];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int row = wm * 64 + 16 * f + i16;
    off_x[f] = row * ROW2 + ((g ^ swz2_x(row)) << 4);
  }
#pragma unroll
  for (int f = 0; f < 8; ++f) {   // weight rows interleaved so that a lane ends up with 32 CONSECUTIVE output features
    const int row = wn * 128 + 32 * (i16 >> 2) + 4 * f + (i16 & 3);
    off_w[f] = FOPER_X + row * ROW2 + ((g ^ swz2_w(row)) << 4);
  }
  f32x4 acc[8][4];  // [fn][fm]: features wn*128 + 32g + 4fn .. +3 of row wm*64 + 16fm + i16
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nt = p.k_tiles;   // >= 3 (dispatch)
  stage(0);
  stage(1);
  stage(2);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // stage 0 is complete
  asm volatile("" ::: "memory");
  if (wm == 1) __builtin_amdgcn_s_barrier();   // the late half: one phase behind from here on
  LN_TS(1);

  for (int s = 0; s < nt; ++s) {
    u32x4 fw[8], fx[4];
    {
      const char* buf = smem + (s % FNST) * FSTAGE;
#pragma unroll
      for (int f = 0; f < 8; ++f) fw[f] = *reinterpret_cast<const u32x4*>(buf + off_w[f]);
#pragma unroll
      for (int f = 0; f < 4; ++f) fx[f] = *reinterpret_cast<const u32x4*>(buf + off_x[f]);
    }
    if (s >= 1 && s + 2 < nt) stage(s + 2);   // into the buffer of stage s-1
    if (s + 2 < nt) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // own pieces of stage s+1 have landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int fm = 0; fm < 4; ++fm)
#pragma unroll
      for (int fn = 0; fn < 8; ++fn) acc[fn][fm] = Mma<bf16_t>::run(fw[fn], fx[fm], acc[fn][fm]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();   // the early half waits for the late half's last phase: the ring is dead
  asm volatile("" ::: "memory");
  LN_TS(2);

  // ---- epilogue.  Where a tile's time goes (profiles/r04_ln512_phases.log, K = 512: 40 us): prologue 3.3, main loop
  // 15.6, residual requests + bias + rounding 4-6, statistics 2-4, the four passes 12-13.  Every global address below is
  // a UNIFORM base (scalar arithmetic: tile, wave, pass, row pair) plus one per-lane 32-bit offset computed once, and whole
  // tiles run without row predicates: 64-bit per-row multiplies and clamps were a quarter of the epilogue's ~3,000
  // instructions per wave (profiles/r04_ln512_pmc.log).  That bought 1 % (r04_ab_ln512_addressing.log): the epilogue
  // waits for memory, not for the VALU -- without the residual reads a K = 512 launch takes 318 instead of 385 us, without
  // the stores 272, without both 234 (r04_ln512_probe_no_residual_no_store.log).  Normalising in the MFMA layout with
  // packed arithmetic (a lane holds its rows' statistics there) needs ~40 registers more than the 256 there are.
  const int L = lane & 31, half = lane >> 5;
  const int col = wn * 128 + 4 * L;
  // rows of this tile that exist, counted from this wave's first row (uniform; FULL: all of them, nothing is predicated):
  // row r of the wave (r = 16 fm + 2 j + half) exists iff r < wave_rows
  const int wave_rows = FULL ? 64 : (int)(p.M - m0 < FM ? p.M - m0 : FM) - wm * 64;
  const uint32_t lane_x = (uint32_t)((half * p.ldx + col) * 4);
  const uint32_t lane_o = (uint32_t)((half * p.ldo + col) * 4);
  const uint32_t lane_b = (uint32_t)((half * p.ldb + col) * 2);
  const char* const x_wave = reinterpret_cast<const char*>(p.x_in) + (m0 + wm * 64) * p.ldx * 4;
  char* const o_wave = reinterpret_cast<char*>(p.x_out) + (m0 + wm * 64) * p.ldo * 4;
  char* const b_wave = reinterpret_cast<char*>(p.xb) + (m0 + wm * 64) * p.ldb * 2;
  // The residual rows of the first two 16-row passes are requested NOW, before bias / rounding / the two statistics passes:
  // nothing they need depends on the product, and the statistics hide their HBM round trip.
  f32x4 xr[3][8];
  auto fetch_x = [&](int fm, f32x4 (&dst)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 16 * fm + 2 * j;   // (uniform: scalar address arithmetic)
      if constexpr (!FULL) dst[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (FULL || r + half < wave_rows) dst[j] = *reinterpret_cast<const f32x4*>(x_wave + (int64_t)(r * (int)p.ldx) * 4 + lane_x);
    }
  };
  fetch_x(0, xr[0]);
  fetch_x(1, xr[1]);
  // ---- bias, rounding to bf16 ----
  const int nb = wn * 128 + 32 * g;
#pragma unroll
  for (int fn = 0; fn < 8; ++fn) {
    const f32x4 b4 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nb + 4 * fn) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
      const uint32_t lo = pack_bf16x2(acc[fn][fm].x + b4.x, acc[fn][fm].y + b4.y);
      const uint32_t hi = pack_bf16x2(acc[fn][fm].z + b4.z, acc[fn][fm].w + b4.w);
      acc[fn][fm] = f32x4{__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                          __uint_as_float(hi & 0xffff0000u)};
    }
  }
  LN_TS(3);
  // ---- row statistics: two passes over the registers; partial sums of the four n-waves meet in LDS ----
  float* const st_sum = reinterpret_cast<float*>(smem + 65536);   // [128 rows][4 n-waves]
  float* const st_sq = st_sum + 512;
  float* const st_mr = st_sq + 512 + wave * 128;                  // this wave's own copy: [64 rows][mean, rstd]
  float mean[4], rstd[4];
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    float t = 0.f;
#pragma unroll
    for (int fn = 0; fn < 8; ++fn) t += (acc[fn][fm].x + acc[fn][fm].y) + (acc[fn][fm].z + acc[fn][fm].w);
    t = group4_sum(t);
    if (g == 0) st_sum[(wm * 64 + 16 * fm + i16) * 4 + wn] = t;
  }
  __syncthreads();
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(st_sum + (wm * 64 + 16 * fm + i16) * 4);
    mean[fm] = ((t.x + t.y) + (t.z + t.w)) * (1.0f / FN);
    float q = 0.f;
#pragma unroll
    for (int fn = 0; fn < 8; ++fn) {
      const float d0 = acc[fn][fm].x - mean[fm], d1 = acc[fn][fm].y - mean[fm], d2 = acc[fn][fm].z - mean[fm],
                  d3 = acc[fn][fm].w - mean[fm];
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    q = group4_sum(q);
    if (g == 0) st_sq[(wm * 64 + 16 * fm + i16) * 4 + wn] = q;
  }
  __syncthreads();
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(st_sq + (wm * 64 + 16 * fm + i16) * 4);
    rstd[fm] = rsqrtf(((t.x + t.y) + (t.z + t.w)) * (1.0f / FN) + p.eps);
    if (g == 0) {
      st_mr[(16 * fm + i16) * 2] = mean[fm];
      st_mr[(16 * fm + i16) * 2 + 1] = rstd[fm];
    }
  }
  // ---- 16 rows at a time through this wave's 8 KiB: [16 rows][32 pieces of 16 B], piece P of row r at P ^ c(r) with
  //      c(r) = r ^ 2 (r >> 2): conflict-free for the b128 writes (a lane writes pieces 8g..8g+7 of row i16) and for the
  //      row-major b128 reads (two rows per instruction) under gfx950's 16-lane service groups ----
  LN_TS(4);
  char* const mine = smem + wave * 8192;
  f32x4 gn = f32x4{1.f, 1.f, 1.f, 1.f}, sh = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.gain) gn = *reinterpret_cast<const f32x4*>(p.gain + col);
  if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + col);
  const int cw = (i16 ^ ((i16 >> 2) << 1)) & 31;
  // The residual rows of a 16-row pass are fetched ahead of it, all eight loads of a lane at once: x_out may alias x_in, so
  // a load written behind the previous row's store would have to wait for it -- 32 exposed round trips per tile.  TWO
  // passes ahead (round 4; one before): the 32 accumulator registers a pass has parked in LDS are free from there on, so
  // the third buffer costs no register the main loop needs, and the epilogue is a latency chain on 8 waves -- the bytes
  // in flight are what its bandwidth is made of.
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x4*>(mine + i16 * 512 + (((8 * g + q) ^ cw) << 4)) = acc[q][fm];
    if (fm + 2 < 4) fetch_x(fm + 2, xr[(fm + 2) % 3]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r16 = 2 * j + half;
      const int cr = (r16 ^ ((r16 >> 2) << 1)) & 31;
      const f32x4 v = *reinterpret_cast<const f32x4*>(mine + r16 * 512 + ((L ^ cr) << 4));
      const float mu = st_mr[(16 * fm + r16) * 2], rs = st_mr[(16 * fm + r16) * 2 + 1];
      const f32x4 x = xr[fm % 3][j];
      f32x4 o;
      o.x = fmaf((v.x - mu) * rs, gn.x, sh.x) + x.x;
      o.y = fmaf((v.y - mu) * rs, gn.y, sh.y) + x.y;
      o.z = fmaf((v.z - mu) * rs, gn.z, sh.z) + x.z;
      o.w = fmaf((v.w - mu) * rs, gn.w, sh.w) + x.w;
      const int rr = 16 * fm + 2 * j;   // (uniform: scalar address arithmetic)
      if (FULL || rr + half < wave_rows) {
        *reinterpret_cast<f32x4*>(o_wave + (int64_t)(rr * (int)p.ldo) * 4 + lane_o) = o;
        if (p.xb)
          *reinterpret_cast<u32x2*>(b_wave + (int64_t)(rr * (int)p.ldb) * 2 + lane_b) = u32x2{pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w)};
      }
    }
#ifdef LN_PROBE_TIMES
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ts[5 + (fm & 1)] = wall_clock64();   // (5: passes 0 / 2 done, 6: passes 1 / 3 done -- the last two survive)
#endif
  }
#ifdef LN_PROBE_TIMES
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ts[7] = wall_clock64();
  if (p.xb && (tid & 63) == 0) {   // one record per wave in the (unused by the probe) bf16 shadow: row m0 + wave
    uint64_t* rec = reinterpret_cast<uint64_t*>(p.xb + (m0 + wave) * p.ldb);
#pragma unroll
    for (int i = 0; i < 8; ++i) rec[i] = ts[i];
  }
#endif
}

}  // namespace

}  // namespace aurora

// (gemm_a4.hip includes this file for the device code above -- LinearArgs, the LDS images, the epilogues -- and stops here)
#ifndef AURORA_GEMM_DEVICE_ONLY

// the four-wave tile with the hand-scheduled main loop (gemm_a4.hip, its own translation unit)
extern "C" __attribute__((visibility("hidden"))) int aurora_a4_launch(const void* linear_args, unsigned n_blocks, unsigned batch,
                                                                     void* stream);

using namespace aurora;

namespace {
// Process default of how large fp32 linears are multiplied (read once from AURORA_F32_GEMM, never changed afterwards):
// 0: native fp32 MFMA (v_mfma_f32_16x16x4_f32); 1: 3 x bf16 operand splitting (default); 2: 2 x fp16 splitting.
int default_f32_mode() {
  static const int mode = [] {
    const char* e = getenv("AURORA_F32_GEMM");
    return !e ? 1 : (e[0] == 'n' || e[0] == '0') ? 0 : (e[0] == 'f' || e[0] == '2') ? 2 : 1;   // native | bf16 | f16
  }();
  return mode;
}
}  // namespace

extern "C" int aurora_hip_default_f32_gemm(void) { return default_f32_mode(); }

namespace {
// Smallest K (elements) of a plain bf16 linear on 256 x 256 tiles that takes the four-wave kernel with the hand-scheduled
// main loop (gemm_a4.hip) instead of the eight-wave ping-pong one.  A read-once process default like AURORA_F32_GEMM
// (AURORA_GEMM_A4_MIN_K; 0 = never): both kernels accumulate K in the same 32-wide steps -- the same bits either way.
int a4_min_k() {
  static const int v = [] {
    const char* e = getenv("AURORA_GEMM_A4_MIN_K");
    const int k = e ? atoi(e) : A4_DEFAULT_MIN_K;
    return k <= 0 ? 0x7fffffff : k;
  }();
  return v;
}
}  // namespace

namespace {
// Scratch of a split-K launch, owned by the caller: fp32 slabs (split x 256 KiB per tile; contents do not matter) and one
// ticket per tile (zero on entry, left zero).
struct SplitWs { float* slabs; int64_t slab_bytes; int32_t* tickets; int n_tickets; int split; };


// K split of a plain bf16 linear on 256 x 256 tiles (1 = none).  Only launches that leave most of the chip idle and
// have K to spare: tiles * split <= CUs (one round), >= 64 K-steps (K = 2048) per slice -- below that the slab round trip
// costs more than the idle CUs were worth (measured: 72 tiles, K = 2048: 31.6 -> 37 us; K = 8192: 108 -> 80 us) --, at most 8.
int choose_split(int64_t M, int N, int K) {
  if (N % BN2 != 0 || M < 1024 || K % 32 != 0) return 1;
  const int64_t tiles = ((M + BM2 - 1) / BM2) * (N / BN2), cus = device_cus();
  const int kt = K / 32;
  int s = (int)std::min<int64_t>(std::min<int64_t>(8, cus / tiles), kt / 64);
  return s < 1 ? 1 : s;
}
}  // namespace

extern "C" int64_t aurora_hip_linear_workspace(int64_t M, int N, int K, int dtype) {
  if (dtype != AURORA_BF16 || M <= 0 || N <= 0 || K <= 0) return 0;
  const int s = choose_split(M, N, K);
  return s <= 1 ? 0 : ((M + BM2 - 1) / BM2) * (N / BN2) * s * (int64_t)(BM2 * BN2 * 4);
}

extern "C" int aurora_hip_linear(const void* A, int64_t lda, const void* W, int64_t ldw,
                                 const float* bias, void* C, int64_t ldc, void* C2, int64_t ldc2,
                                 const float* residual, int64_t ldr, int64_t M, int N, int K,
                                 int dtype, int act, void* stream) {
  return aurora_hip_linear_ex(A, lda, W, ldw, bias, C, ldc, C2, ldc2, residual, ldr, M, N, K, dtype, act, -1, nullptr,
                              0.f, stream);
}

namespace {
int linear_impl(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc, void* C2,
                int64_t ldc2, const float* residual, int64_t ldr, int64_t M, int N, int K, int dtype, int act, int f32_gemm,
                const float* guard, float guard_limit, int batch, int64_t stride_a, int64_t stride_w, int64_t stride_bias,
                int64_t stride_c, void* stream, const SplitWs* ws = nullptr, int64_t plane_stride = 0, int plane_heads = 0);
}

extern "C" int aurora_hip_linear_ex(const void* A, int64_t lda, const void* W, int64_t ldw,
                                    const float* bias, void* C, int64_t ldc, void* C2, int64_t ldc2,
                                    const float* residual, int64_t ldr, int64_t M, int N, int K,
                                    int dtype, int act, int f32_gemm, const float* guard, float guard_limit,
                                    void* stream) {
  return linear_impl(A, lda, W, ldw, bias, C, ldc, C2, ldc2, residual, ldr, M, N, K, dtype, act, f32_gemm, guard, guard_limit,
                     1, 0, 0, 0, 0, stream);
}

extern "C" int aurora_hip_linear_planes(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C,
                                        int64_t plane_stride, int heads, int64_t M, int N, int K, int dtype, void* stream) {
  AURORA_CHECK_ARG(plane_stride > 0 && heads > 0, "linear_planes: plane_stride=%lld heads=%d", (long long)plane_stride, heads);
  return linear_impl(A, lda, W, ldw, bias, C, N, nullptr, 0, nullptr, 0, M, N, K, dtype, AURORA_ACT_NONE, -1, nullptr, 0.f, 1, 0,
                     0, 0, 0, stream, nullptr, plane_stride, heads);
}

extern "C" int aurora_hip_linear_ws(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C,
                                    int64_t ldc, void* C2, int64_t ldc2, const float* residual, int64_t ldr, int64_t M, int N,
                                    int K, int dtype, int act, void* workspace, int64_t workspace_bytes, int32_t* tickets,
                                    int n_tickets, int split, void* stream) {
  AURORA_CHECK_ARG(workspace_bytes >= 0 && n_tickets >= 0 && split >= 0 && ((uintptr_t)workspace % 16) == 0,
                   "linear_ws: bad workspace arguments");
  const SplitWs ws{(float*)workspace, workspace ? workspace_bytes : 0, tickets, tickets ? n_tickets : 0, split};
  return linear_impl(A, lda, W, ldw, bias, C, ldc, C2, ldc2, residual, ldr, M, N, K, dtype, act, -1, nullptr, 0.f, 1, 0, 0, 0, 0,
                     stream, &ws);
}

extern "C" int aurora_hip_linear_batched(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C,
                                         int64_t ldc, int64_t M, int N, int K, int dtype, int act, int f32_gemm,
                                         const float* guard, float guard_limit, int batch, int64_t stride_a,
                                         int64_t stride_w, int64_t stride_bias, int64_t stride_c, void* stream) {
  AURORA_CHECK_ARG(batch >= 1 && batch <= 65535, "linear_batched: 1 <= batch <= 65535 (got %d)", batch);
  const int es = dtype == AURORA_F32 ? 4 : 2;
  AURORA_CHECK_ARG((stride_a * es) % 16 == 0 && (stride_w * es) % 16 == 0 && (stride_c * es) % 16 == 0 && stride_bias % 4 == 0,
                   "linear_batched: batch strides must keep every problem 16-byte aligned");
  return linear_impl(A, lda, W, ldw, bias, C, ldc, nullptr, 0, nullptr, 0, M, N, K, dtype, act, f32_gemm, guard, guard_limit,
                     batch, stride_a, stride_w, stride_bias, stride_c, stream);
}

namespace {
int linear_impl(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc, void* C2,
                int64_t ldc2, const float* residual, int64_t ldr, int64_t M, int N, int K, int dtype, int act, int f32_gemm,
                const float* guard, float guard_limit, int batch, int64_t stride_a, int64_t stride_w, int64_t stride_bias,
                int64_t stride_c, void* stream, const SplitWs* ws, int64_t plane_stride, int plane_heads) {
  AURORA_CHECK_ARG(dtype == AURORA_F32 || dtype == AURORA_BF16, "linear: bad dtype %d", dtype);
  AURORA_CHECK_ARG(plane_stride == 0 || (dtype == AURORA_BF16 && plane_heads > 0 && N % (64 * plane_heads) == 0 &&
                                         N <= 192 * plane_heads && plane_stride >= M * 192 && plane_stride % 8 == 0 &&
                                         C2 == nullptr && residual == nullptr && batch == 1),
                   "linear: head planes need bf16, N = 64 heads x (1, 2 or 3), planes of >= M rows, one output, no residual");
  const int pre = f32_gemm < 0 ? 0 : f32_gemm & (AURORA_F32_A_SPLIT | AURORA_F32_W_SPLIT | AURORA_F32_C_SPLIT);
  if (pre) f32_gemm &= ~pre;
  AURORA_CHECK_ARG(f32_gemm >= -1 && f32_gemm <= 2, "linear: bad fp32 GEMM mode %d", f32_gemm);
  const int mode = f32_gemm < 0 ? default_f32_mode() : f32_gemm;
  // pre-split operands / output: the two-term ping-pong kernel only.  An A-split launch cannot fall back to three terms
  // (they need the fp32 values), so it takes no guard; a W-split launch with a guard runs iff the guard holds and the
  // caller pairs it with a mode-1 launch on the fp32 weights carrying the same guard, which runs iff it does not.
  AURORA_CHECK_ARG(!pre || (dtype == AURORA_F32 && mode == 2 && N % 128 == 0 && K % 32 == 0 && K >= 96),
                   "linear: fp16-pair operands need fp32, mode 2, N %% 128 == 0, K %% 32 == 0, K >= 96 (N=%d K=%d)", N, K);
  // (an A-split launch WITH a guard is for buffers whose format was itself decided by that guard on the device -- written
  // as pairs by a guarded two-term producer iff it holds, as fp32 by its three-term twin otherwise)
  AURORA_CHECK_ARG(!(pre & AURORA_F32_A_SPLIT) || (pre & AURORA_F32_W_SPLIT),
                   "linear: a pre-split activation operand needs pre-split weights");
  AURORA_CHECK_ARG(!(pre & AURORA_F32_C_SPLIT) || (C2 == nullptr && ldc % 32 == 0 && ((uintptr_t)C % 16) == 0),
                   "linear: fp16-pair output needs ldc %% 32 == 0, 16-byte alignment and no second output");
  const float* const g_guard = (mode == 2 || (mode == 1 && dtype == AURORA_F32)) ? guard : nullptr;
  const float g_guard_limit = guard_limit;
  AURORA_CHECK_ARG(M > 0 && N > 0 && K > 0, "linear: empty problem M=%lld N=%d K=%d", (long long)M, N, K);
  const int es = dtype == AURORA_F32 ? 4 : 2, es2 = dtype == AURORA_F32 ? 2 : 4;
  AURORA_CHECK_ARG(((int64_t)K * es) % ROW_BYTES == 0,
                   "linear: K=%d must be a multiple of %d elements", K, ROW_BYTES / es);
  AURORA_CHECK_ARG((lda * es) % 16 == 0 && (ldw * es) % 16 == 0 && lda >= K && ldw >= K,
                   "linear: operand strides must be 16-byte multiples >= K");
  AURORA_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "linear: unaligned operand");
  AURORA_CHECK_ARG(act >= AURORA_ACT_NONE && act <= AURORA_ACT_SILU, "linear: bad activation %d", act);
  AURORA_CHECK_ARG(C != nullptr && ldc >= N && (!C2 || ldc2 >= N) && (!residual || ldr >= N || ldr == 0),
                   "linear: bad output strides");

  // Big backbone shapes take the 256 x 256 ring kernel; everything else the 128 x 128 one.
  // (fp32 in split mode: the kernel choice must not depend on M, or a latitude band of a sharded model
  // would round differently from the same rows of the un-sharded one.)
  const bool split = dtype == AURORA_F32 && mode >= 1;
  const bool tall = pre != 0 && N % VN != 0;   // pre-split operands, N a multiple of 128 only: the 256 x 128 two-term tiles
  bool big = (M >= 1024 || split) && (N % BN2 == 0 || tall);
  bool mid = false;   // bf16 only: 256 x 128 tiles of the ring kernel, two 4-wave workgroups per CU
  // split-K on 256 x 256 tiles when the caller brought scratch for it (aurora_hip_linear_ws) and the launch would
  // otherwise leave most of the chip idle
  int ksplit = 1;
  if (big && dtype == AURORA_BF16 && batch == 1 && ws != nullptr) {
    ksplit = ws->split > 0 ? std::min(ws->split, K / 128) : choose_split(M, N, K);
    const int64_t tiles = ((M + BM2 - 1) / BM2) * (N / BN2);
    if (ksplit > 1 && (tiles > ws->n_tickets || tiles * ksplit * (int64_t)(BM2 * BN2 * 4) > ws->slab_bytes)) ksplit = 1;
  }
  if (big && !split && ksplit <= 1) {
    // Few tiles (a latitude band of a sharded model, the coarse stages): 256 x 256 tiles leave CUs idle or end in a
    // thin last round.  Smaller tiles fill the chip: 256 x 128 tiles of the same ring kernel (two 4-wave workgroups per
    // CU, each with the full 128 x 64 wave tile) or 128 x 128 tiles.  The choice is a cost model fitted to measurements
    // (profiles/r02_ab_gemm_tiles.log): a round of tiles costs a + b K microseconds -- 256^2: 10.1 + 0.0266 K (256 per
    // round), 256 x 128: 7.7 + 0.0317 K (512 per round), 128^2: 7.6 + 0.0153 K (512 per round) -- and a last round that
    // leaves every CU with at most one of its two workgroups runs in ~0.65 of that.  (Results do not depend on the
    // tiling: every kernel accumulates K in the same 32-wide steps.)
    const int64_t cus = device_cus();
    const int64_t nb_big = ((M + BM2 - 1) / BM2) * (N / BN2), nb_small = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int64_t nb_mid = ((M + BM2 - 1) / BM2) * (N / 128);
    if (dtype == AURORA_BF16) {
      auto cost = [&](int64_t nb, int64_t slots, double a, double b) {
        const int64_t full = nb / slots, rem = nb % slots;
        const double part = rem == 0 ? 0.0 : (slots > cus && rem <= cus) ? 0.65 : 1.0;
        return ((double)full + part) * (a + b * (double)K);
      };
      const double t_big = cost(nb_big, cus, 10.1, 0.0266), t_mid = cost(nb_mid, 2 * cus, 7.7, 0.0317),
                   t_small = cost(nb_small, 2 * cus, 7.6, 0.0153);
      if (t_mid < 0.97 * t_big && t_mid <= t_small) mid = true;
      else if (t_small < 0.97 * t_big) big = false;
    } else {
      auto fill = [&](int64_t nb, int64_t slots) { return (double)nb / (double)(((nb + slots - 1) / slots) * slots); };
      if (0.8 * fill(nb_small, 2 * cus) > fill(nb_big, cus)) big = false;
    }
  }
  const int bm = big ? BM2 : BM, bn = mid ? 128 : big ? BN2 : BN, rowb = big ? ROW2 : ROW_BYTES;
  LinearArgs p;
  p.A = (const char*)A; p.lda_b = lda * es;
  p.W = (const char*)W; p.ldw_b = ldw * es;
  p.bias = bias;
  p.C = (char*)C; p.ldc = ldc; p.C2 = (char*)C2; p.ldc2 = ldc2;
  p.res = residual; p.ldr = ldr;
  p.M = M; p.N = N; p.k_tiles = (int)(((int64_t)K * es) / rowb); p.act = act;
  p.tiles_n = (N + bn - 1) / bn;
  p.n_blocks = ((M + bm - 1) / bm) * p.tiles_n;
  bool vec = ((uintptr_t)C % 16) == 0 && (ldc * es) % 16 == 0;
  if (C2) vec = vec && ((uintptr_t)C2 % 16) == 0 && (ldc2 * es2) % 16 == 0;
  if (residual) vec = vec && ((uintptr_t)residual % 16) == 0 && (ldr * 4) % 16 == 0;
  p.vec_store = vec ? 1 : 0;
  if (split && big && act == AURORA_ACT_GELU) p.act = ACT_GELU_FAST;
  p.guard = split ? g_guard : nullptr;
  p.guard_limit = g_guard_limit;
  p.out_split = (pre & AURORA_F32_C_SPLIT) ? 1 : 0;
  AURORA_CHECK_ARG(p.n_blocks < (int64_t)1 << 31, "linear: too many tiles");

  p.bs_a = stride_a * es; p.bs_w = stride_w * es; p.bs_c = stride_c * es; p.bs_bias = stride_bias;
  p.split = ksplit; p.slabs = ksplit > 1 ? ws->slabs : nullptr; p.tickets = ksplit > 1 ? ws->tickets : nullptr;
  p.plane_stride = plane_stride; p.plane_heads = plane_heads;
  if (plane_stride) p.vec_store = ((uintptr_t)C % 16) == 0 ? 1 : 0;   // (rows of a plane are 128 bytes: ldc plays no part)
  dim3 grid((unsigned)p.n_blocks, (unsigned)batch);
  static bool attr_done_dev[64] = {false};   // function attributes are per device
  bool& attr_done = attr_done_dev[current_device() & 63];
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)linear_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    (void)hipFuncSetAttribute((const void*)linear_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256<float, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE2 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256<bf16_t, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE2 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256<bf16_t, 2, 3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256pp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256pp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256_f32x3<3>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE2 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256_f32x3<2>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE2 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_f32pp<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, VNST * VSTAGE);
    (void)hipFuncSetAttribute((const void*)linear_kernel_f32pp<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, VNST * VSTAGE);
    (void)hipFuncSetAttribute((const void*)linear_kernel_f32pp<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, VNST * VSTAGE);
    (void)hipFuncSetAttribute((const void*)linear_kernel_f32pp<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, VNST * VSTAGE);
    (void)hipFuncSetAttribute((const void*)linear_kernel_f32pp<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, VNST * VSTAGE);
    attr_done = true;
  }
  // two fp16 terms: the ping-pong kernel (128 x 256 tiles, K-stages of 32) when K allows, else the in-phase 256 x 256 one
  const bool f32pp = big && split && mode == 2 && K % 32 == 0 && K >= 96;
  auto launch_f32pp = [&]() {
    LinearArgs q = p;
    q.k_tiles = K / 32;
    q.tiles_n = tall ? N / 128 : N / VN;
    q.n_blocks = tall ? ((M + 255) / 256) * q.tiles_n : ((M + VM - 1) / VM) * q.tiles_n;
    const dim3 g((unsigned)q.n_blocks, (unsigned)batch), b(VTHREADS);
    if (tall && (pre & AURORA_F32_A_SPLIT)) hipLaunchKernelGGL((linear_kernel_f32pp<true, true, true>), g, b, VNST * VSTAGE, as_stream(stream), q);
    else if (tall) hipLaunchKernelGGL((linear_kernel_f32pp<false, true, true>), g, b, VNST * VSTAGE, as_stream(stream), q);
    else if (pre & AURORA_F32_A_SPLIT) hipLaunchKernelGGL((linear_kernel_f32pp<true, true>), g, b, VNST * VSTAGE, as_stream(stream), q);
    else if (pre & AURORA_F32_W_SPLIT) hipLaunchKernelGGL((linear_kernel_f32pp<false, true>), g, b, VNST * VSTAGE, as_stream(stream), q);
    else hipLaunchKernelGGL((linear_kernel_f32pp<false, false>), g, b, VNST * VSTAGE, as_stream(stream), q);
  };
  if (mid) {
    hipLaunchKernelGGL((linear_kernel_256<bf16_t, 2, 3, 0>), grid, dim3(256), MID_LDS, as_stream(stream), p);
  } else if (big) {
    if (pre) {
      launch_f32pp();
    } else if (split && mode == 1 && g_guard != nullptr) {   // the three-term half of a guarded pair (see above)
      hipLaunchKernelGGL(linear_kernel_256_f32x3<3>, grid, dim3(THREADS2), NSTAGE2 * STAGE2, as_stream(stream), p);
    } else if (split && mode == 2 && g_guard != nullptr) {   // both variants; the device word picks one
      if (f32pp) launch_f32pp();
      else hipLaunchKernelGGL(linear_kernel_256_f32x3<2>, grid, dim3(THREADS2), NSTAGE2 * STAGE2, as_stream(stream), p);
      hipLaunchKernelGGL(linear_kernel_256_f32x3<3>, grid, dim3(THREADS2), NSTAGE2 * STAGE2, as_stream(stream), p);
    } else if (f32pp)
      launch_f32pp();
    else if (split && mode == 2)
      hipLaunchKernelGGL(linear_kernel_256_f32x3<2>, grid, dim3(THREADS2), NSTAGE2 * STAGE2, as_stream(stream), p);
    else if (split)
      hipLaunchKernelGGL(linear_kernel_256_f32x3<3>, grid, dim3(THREADS2), NSTAGE2 * STAGE2, as_stream(stream), p);
    else if (dtype == AURORA_F32)
      hipLaunchKernelGGL((linear_kernel_256<float, 4, 4>), grid, dim3(THREADS2), NSTAGE2 * STAGE2, as_stream(stream), p);
    else if (ksplit > 1)       // ping-pong main loop, one workgroup per K-slice of a tile
      hipLaunchKernelGGL(linear_kernel_256pp<true>, dim3((unsigned)(p.n_blocks * ksplit)), dim3(THREADS2), PP_LDS, as_stream(stream), p);
    else if (p.k_tiles >= 8 && p.k_tiles % 2 == 0 && K >= a4_min_k())
      (void)aurora_a4_launch(&p, (unsigned)p.n_blocks, (unsigned)batch, stream);   // four waves, hand-scheduled loop (gemm_a4.hip)
    else if (p.k_tiles >= 4)   // ping-pong main loop, one workgroup per tile
      hipLaunchKernelGGL(linear_kernel_256pp<false>, grid, dim3(THREADS2), PP_LDS, as_stream(stream), p);
    else
      hipLaunchKernelGGL((linear_kernel_256<bf16_t, 4, 4>), grid, dim3(THREADS2), NSTAGE2 * STAGE2, as_stream(stream), p);
  } else {
    if (dtype == AURORA_F32)
      hipLaunchKernelGGL(linear_kernel<float>, grid, dim3(THREADS), 4 * TILE_BYTES, as_stream(stream), p);
    else
      hipLaunchKernelGGL(linear_kernel<bf16_t>, grid, dim3(THREADS), 4 * TILE_BYTES, as_stream(stream), p);
  }
  return check_launch("linear");
}
}  // namespace

extern "C" int aurora_hip_split_f16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int K,
                                    float scale, void* stream) {
  AURORA_CHECK_ARG(src != nullptr && dst != nullptr && rows > 0 && K > 0 && K % 32 == 0, "split_f16: K=%d must be a positive multiple of 32", K);
  AURORA_CHECK_ARG(ld_src >= K && ld_dst >= K && ld_src % 4 == 0 && ld_dst % 32 == 0 && ((uintptr_t)src % 16) == 0 &&
                   ((uintptr_t)dst % 16) == 0, "split_f16: strides / alignment");
  const int64_t n = rows * (K >> 3);
  hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), src, ld_src,
                     (char*)dst, ld_dst, rows, K, scale);
  return check_launch("split_f16");
}

extern "C" int aurora_hip_linear_layernorm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                           const float* gain, const float* shift, const float* x_in, int64_t ldx,
                                           float* x_out, int64_t ldo, void* x_bf16, int64_t ldb, int64_t M, int N, int K,
                                           float eps, void* stream) {
  AURORA_CHECK_ARG(N == FN, "linear_layernorm: N=%d (only D = 512 rows are owned by one workgroup)", N);
  AURORA_CHECK_ARG(M > 0 && K % 32 == 0 && K >= 96, "linear_layernorm: K=%d must be a multiple of 32, >= 96", K);
  AURORA_CHECK_ARG(A && W && x_in && x_out && lda >= K && ldw >= K && (lda * 2) % 16 == 0 && (ldw * 2) % 16 == 0 &&
                       ((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0,
                   "linear_layernorm: operand strides / alignment");
  AURORA_CHECK_ARG(ldx >= N && ldo >= N && ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x_in % 16) == 0 &&
                       ((uintptr_t)x_out % 16) == 0 && (!x_bf16 || (ldb >= N && ldb % 4 == 0 && ((uintptr_t)x_bf16 % 8) == 0)),
                   "linear_layernorm: residual / output strides / alignment");
  AURORA_CHECK_ARG((!bias || ((uintptr_t)bias % 16) == 0) && (!gain || ((uintptr_t)gain % 16) == 0) &&
                       (!shift || ((uintptr_t)shift % 16) == 0), "linear_layernorm: unaligned bias / gain / shift");
  LinearLnArgs p{(const char*)A, lda * 2, (const char*)W, ldw * 2, bias, gain, shift, x_in, ldx, x_out, ldo, (bf16_t*)x_bf16, ldb,
                 M, K / 32, eps, 0};
  static bool attr_done_dev[64] = {false};
  bool& attr_done = attr_done_dev[current_device() & 63];
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)linear_ln512_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FNST * FSTAGE);
    (void)hipFuncSetAttribute((const void*)linear_ln512_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FNST * FSTAGE);
    attr_done = true;
  }
  // whole tiles by the kernel without row predicates; the ragged last tile, if any, by its own one-workgroup launch
  const int64_t whole = M / FM;
  AURORA_CHECK_ARG(whole < (int64_t)1 << 31, "linear_layernorm: too many tiles");
  if (whole > 0)
    hipLaunchKernelGGL(linear_ln512_kernel<true>, dim3((unsigned)whole), dim3(FTHREADS), FNST * FSTAGE, as_stream(stream), p);
  if (M % FM != 0) {
    p.tile0 = whole;
    hipLaunchKernelGGL(linear_ln512_kernel<false>, dim3(1), dim3(FTHREADS), FNST * FSTAGE, as_stream(stream), p);
  }
  return check_launch("linear_layernorm");
}

#endif  // AURORA_GEMM_DEVICE_ONLY
