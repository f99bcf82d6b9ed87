// Dense linear layers on the gfx950 matrix cores:  C = act(A . W^T + bias) (+ residual).
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

#include "common.h"

namespace aurora {

namespace {

constexpr int BM = 128;       // activation rows per block
constexpr int BN = 128;       // output features per block
constexpr int ROW_BYTES = 128;  // bytes of K per tile row
constexpr int TILE_BYTES = 128 * ROW_BYTES;  // 16 KiB per operand per buffer
constexpr int THREADS = 256;

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
};

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

__device__ __forceinline__ void tile_of_block(uint32_t bid, uint32_t nb, uint32_t tiles_m, uint32_t tiles_n,
                                              uint32_t& tile_m, uint32_t& tile_n);

template <typename T> struct Other;
template <> struct Other<float> { typedef bf16_t type; };
template <> struct Other<bf16_t> { typedef float type; };

template <typename T>
__global__ __launch_bounds__(THREADS, 2) void linear_kernel(const LinearArgs p) {
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
    store16<T>(reinterpret_cast<T*>(p.C) + m * p.ldc + nbase, v, vec, n_left);
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
constexpr int HALF_LDS = 3 * (OPER2 + OPER2 / 2);  // 256 x 128 variant: 3 stages of 24 KiB

__device__ __forceinline__ int swz2(int a) { return ((a >> 1) & 1) * 3; }
__device__ __forceinline__ int swz2_x(int row) { return swz2((row >> 2) & 3); }
__device__ __forceinline__ int swz2_w(int row) { return swz2((row >> 4) & 3); }

// XCD-aware, L2-friendly tile order shared by both kernels: each XCD owns a contiguous range of
// logical ids; inside it n-tiles are visited in groups of `GN` with the m-tile index in between,
// so the workgroups that run together on one XCD share a few activation tiles AND a few weight
// tiles (both then come out of that XCD's 4 MiB L2).
__device__ __forceinline__ void tile_of_block(uint32_t bid, uint32_t nb, uint32_t tiles_m, uint32_t tiles_n,
                                              uint32_t& tile_m, uint32_t& tile_n) {
  const uint32_t q8 = nb >> 3, r8 = nb & 7;
  const uint32_t xcd = bid & 7, idx = bid >> 3;
  const uint32_t logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  constexpr uint32_t GN = 8;
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
    store16<T>(reinterpret_cast<T*>(p.C) + m * p.ldc + nbase, v, vec, n_left);
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
__device__ __forceinline__ void epilogue_256_bf16_coalesced(const LinearArgs& p, f32x4 (&acc)[4][8], int64_t m0,
                                                            int n0, int wm, int wn, int wave, int lane, char* smem) {
  const int i16 = lane & 15, g = lane >> 4;
  char* mine = smem + wave * 16384;
  const int nbase = n0 + wn * 64 + 16 * g;
  float bias_v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) bias_v[t] = p.bias ? p.bias[nbase + t] : 0.f;
#pragma unroll
  for (int fm = 0; fm < 8; ++fm) {
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
      for (int t = 0; t < 16; ++t) v[t] = gelu_for<bf16_t>(v[t]);
    } else if (p.act == AURORA_ACT_SILU) {
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = v[t] / (1.0f + expf(-v[t]));
    }
    const int row = 16 * fm + i16, sw = row & 7;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      *reinterpret_cast<u32x4*>(mine + row * 128 + (((2 * g + q) ^ sw) << 4)) =
          u32x4{pack_bf16x2(v[8 * q], v[8 * q + 1]), pack_bf16x2(v[8 * q + 2], v[8 * q + 3]),
                pack_bf16x2(v[8 * q + 4], v[8 * q + 5]), pack_bf16x2(v[8 * q + 6], v[8 * q + 7])};
  }
  const int rr = lane >> 3, cc = lane & 7;
  bf16_t* cbase = reinterpret_cast<bf16_t*>(p.C) + n0 + wn * 64 + cc * 8;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 8 + rr;
    const u32x4 d = *reinterpret_cast<const u32x4*>(mine + row * 128 + ((cc ^ rr) << 4));
    const int64_t m = m0 + wm * 128 + row;
    if (m < p.M) __builtin_nontemporal_store(d, reinterpret_cast<u32x4*>(cbase + m * p.ldc));
  }
}

// WN = number of wave columns: 4 -> 256 x 256 tile, 512 threads, one workgroup per CU (4-stage ring, 128 KiB);
//                              2 -> 256 x 128 tile, 256 threads, TWO workgroups per CU (3-stage ring, 72 KiB each):
// the two workgroups run out of phase, so one's prologue / epilogue / barrier waits are covered by the other's
// MFMAs -- the better choice for short K, where a tile is mostly prologue and epilogue.
template <typename T, int WN, int NST>
__global__ __launch_bounds__(128 * WN, 2) void linear_kernel_256(const LinearArgs p) {
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
      epilogue_256_bf16_coalesced(p, acc, m0, n0, wm, wn, wave, lane, smem);
      return;
    }
  }
  epilogue_256<T>(p, acc, m0, n0, wm, wn, i16, g);
}


// =================================================================================================
// Persistent "streamed-weights" kernel for bf16 linears with a plain bf16 output (every backbone linear).
//
// What the measurements say about the ring kernel above (tools/probes/l1_probe.hip, tools/gemm_ksweep.py):
//   * its main loop delivers 32 KiB of operands per K-stage in ~1900 cycles = 17 B/clk/CU, against 1024 cycles
//     of MFMA work: it is bound by the vector-memory front end, because a K-stage row is 64 bytes, HALF a cache
//     line -- LDS-DMA instructions that fetch 16 rows x 64 B run at 15 B/clk/CU, instructions that fetch 8 rows
//     x 128 B (whole lines) at 59 B/clk/CU, both from L2;
//   * with 128-byte rows a 256 x 256 tile needs 64 KiB per stage, so only two stages fit in LDS and the ring is
//     too shallow to hide L2 latency (tried: 1.0 PFLOP/s at K = 4096 against 1.15 for the 64-byte ring);
//   * a tile's fixed cost (workgroup launch, un-hidden prologue, epilogue + store drain) is ~9.5 us, as much as
//     the MFMA time of a K = 512 tile.
// So: weights do not go through LDS at all.  They are static per call, so a pre-pass (pack_w_kernel) rewrites
// W[N][K] into MFMA-fragment order -- the 4 KiB a wave needs per K-stage become one contiguous run, fetched
// with four 1 KiB global_load_dwordx4 straight into registers one stage ahead.  LDS then holds only the
// activation tile, 256 rows x 128 B = 32 KiB per stage of K = 64: whole lines AND a 4-slot ring (3 stages in
// flight).  The 8 waves sit side by side along N (wave tile 256 x 32, acc[2][16]); every wave reads all 256
// activation rows from LDS (ds_read_b128 runs at 256 B/clk/CU, 50 % busy) and no weight fragment is loaded twice.
// One persistent workgroup per CU walks over its tiles; the ring never drains between tiles.
//
// vmcnt bookkeeping: operations are issued in groups of 4 per lane (a W stage = 4 loads, an A stage = 4 LDS-DMA
// pieces).  Loads return in order among loads, so "everything older than the g youngest groups has landed" is
// `s_waitcnt vmcnt(4 g)`; the epilogue's stores are not counted in g, which only makes a wait stronger.
// =================================================================================================
constexpr int SW_STAGE = 256 * 128;                 // activation tile per stage (K = 64 bf16)
constexpr int SW_SLOTS = 4;
constexpr int SW_LDS = SW_SLOTS * SW_STAGE;         // 128 KiB ring
constexpr int SW_LDS_TOTAL = SW_LDS + 2 * 16384;    // + two 16 KiB epilogue slabs = 160 KiB

// Wp[((tn * KT + kt) * 8 + w) * 4 + ks * 2 + fn][lane] (16 bytes each)
//   = W[256 tn + 32 w + 8 (i >> 2) + 4 fn + (i & 3)][64 kt + 32 ks + 8 g .. + 8],   lane = (i = lane & 15, g = lane >> 4)
// i.e. MFMA A-operand fragments, with the rows interleaved such that lane (j, g) of the C^T fragments ends up
// with the 8 consecutive features 8 g .. 8 g + 7 of its wave's 32 (fn = 0, 1 x 4 registers).
__global__ __launch_bounds__(256) void pack_w_kernel(const bf16_t* __restrict__ W, int64_t ldw, u32x4* __restrict__ Wp,
                                                     int KT, int64_t n_pieces) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= n_pieces) return;
  const int lane = (int)(id & 63);
  const int frag = (int)((id >> 6) & 3), w = (int)((id >> 8) & 7);
  const int64_t st = id >> 11;                     // tn * KT + kt
  const int kt = (int)(st % KT);
  const int64_t tn = st / KT;
  const int i = lane & 15, g = lane >> 4, ks = frag >> 1, fn = frag & 1;
  const int64_t n = 256 * tn + 32 * w + 8 * (i >> 2) + 4 * fn + (i & 3);
  Wp[id] = *reinterpret_cast<const u32x4*>(W + n * ldw + 64 * kt + 32 * ks + 8 * g);
}

__global__ __launch_bounds__(THREADS2, 2) void linear_kernel_sw(const LinearArgs p, const u32x4* __restrict__ Wp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  const uint32_t n_tiles = (uint32_t)p.n_blocks, tiles_n = (uint32_t)p.tiles_n, tiles_m = n_tiles / tiles_n;
  const int nt = p.k_tiles;   // stages of K = 64 per tile
  const uint32_t my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;   // >= 1 (grid <= n_tiles)
  const int S = (int)my_tiles * nt;

  // ---- activation stream (LDS-DMA), 4 pieces per lane and stage ----
  const char* a_base = nullptr;   // p.A + m0 * lda (uniform)
  int last_row = 0;
  const int row0 = tid >> 3;      // piece r covers tile row 64 r + row0, 16-byte chunk tid & 7 (swizzled at the source)
  const uint32_t cx = (uint32_t)(((tid & 7) ^ swz_x(row0)) << 4);
  uint32_t a_tile = blockIdx.x;
  int a_kt = 0, a_issued = 0;
  auto a_set = [&](uint32_t t) {
    uint32_t tm, tn;
    tile_of_block(t, n_tiles, tiles_m, tiles_n, tm, tn);
    const int64_t m0 = (int64_t)tm * BM2;
    a_base = p.A + m0 * p.lda_b;
    last_row = (int)(p.M - 1 - m0 < BM2 - 1 ? p.M - 1 - m0 : BM2 - 1);   // rows past M re-read row M-1
  };
  auto a_piece = [&](int r) {
    const int row = row0 + 64 * r;
    const uint32_t vo = (uint32_t)(row < last_row ? row : last_row) * (uint32_t)p.lda_b + cx;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(a_base + (int64_t)a_kt * 128 + vo),
        (lds_ptr_t)(smem + (a_issued & (SW_SLOTS - 1)) * SW_STAGE + (r * THREADS2 + wave * 64) * 16), 16, 0, 0);
  };
  auto a_done = [&]() {
    ++a_issued;
    if (++a_kt == nt) {
      a_kt = 0;
      a_tile += gridDim.x;
      if (a_tile < n_tiles) a_set(a_tile);
    }
  };

  // ---- weight stream (global -> registers), one stage = 4 fragments [ks * 2 + fn] ----
  uint32_t w_tile = blockIdx.x;
  int w_kt = 0;
  const u32x4* w_ptr = nullptr;   // fragments of (w_tile, kt = 0) for this wave and lane
  auto w_set = [&](uint32_t t) {
    uint32_t tm, tn;
    tile_of_block(t, n_tiles, tiles_m, tiles_n, tm, tn);
    w_ptr = Wp + ((int64_t)tn * nt * 8 + wave) * 256 + lane;
  };
  auto w_load = [&](u32x4 (&w)[4]) {   // next stage of the stream (caller guarantees there is one)
    const u32x4* q = w_ptr + (int64_t)w_kt * (8 * 256);
#pragma unroll
    for (int f = 0; f < 4; ++f) w[f] = q[f * 64];
    if (++w_kt == nt) {
      w_kt = 0;
      w_tile += gridDim.x;
      if (w_tile < n_tiles) w_set(w_tile);
    }
  };
  auto wait_groups = [&](int younger) {   // everything older than the `younger` youngest groups has landed
    if (younger >= 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (younger == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (younger == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---- multiply side ----
  const int off_x0 = i16 * 128 + ((g ^ swz_x(i16)) << 4);   // fragment row 16 fm + i16: + fm * 2048; ks: ^ 64
  auto lds_x = [&](int s, int ks, int fm) {
    return *reinterpret_cast<const u32x4*>(smem + (s & (SW_SLOTS - 1)) * SW_STAGE + (off_x0 ^ (ks << 6)) + fm * 2048);
  };
  f32x4 acc[2][16];  // [fn][fm]
  u32x4 X[8], Wa[4], Wb[4];
  // A stage is multiplied in four phases of 16 MFMAs: (ks, half) = (0, lo) (0, hi) (1, lo) (1, hi), where lo / hi
  // are fragment rows 0-7 / 8-15.  The activation fragments live in ONE set of 8 registers-quads that is refilled
  // row by row: once the two MFMAs of a row are issued, the same row of the NEXT phase is requested into the same
  // registers (needed a whole phase later).  Weights: 4 fragments per stage, double-buffered across stages.
  auto mma_row = [&](u32x4 (&w)[4], int ks, int r, int fm) {
    acc[0][fm] = Mma<bf16_t>::run(w[2 * ks], X[r], acc[0][fm]);
    acc[1][fm] = Mma<bf16_t>::run(w[2 * ks + 1], X[r], acc[1][fm]);
  };
  auto phase = [&](u32x4 (&w)[4], int s, int ks, int hi) {   // phases 0..2: the next phase is in the same slot
    const int nks = hi ? ks + 1 : ks, nhi = hi ^ 1;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      mma_row(w, ks, r, 8 * hi + r);
      X[r] = lds_x(s, nks, 8 * nhi + r);
      if ((r & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
  };
  // Last phase (1, hi): refill from (s + 1, 0, lo) in the next slot.  Rows 0-1 need only registers and cover the
  // synchronisation: stage s+1 has landed (issued three stages ago), every wave is done reading stage s (its last
  // reads were the refills of phase 2), whose slot takes stage s+4 -- the 4 pieces go between MFMA rows.  `reload`
  // is false on a tile's last stage (the next tile's first fragments are read after the epilogue).
  auto phase_sync = [&](u32x4 (&w)[4], int s) {
    mma_row(w, 1, 0, 8);
    mma_row(w, 1, 1, 9);
    __builtin_amdgcn_sched_barrier(0);
    bool refill = false;
    if (s + 1 < S) {
      // Groups issued after A(s+1) [in stage s-3]: W(s-1), A(s+2), W(s), A(s+3), W(s+1); the first SW_SLOTS stages
      // were drained completely in the prologue.
      if (s + 1 >= SW_SLOTS) wait_groups(3 + (s + 2 < S) + (s + 3 < S));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      refill = a_issued < S;
    }
    // (after the very last stage the refills read a stale slot; the values are never used)
    X[0] = lds_x(s + 1, 0, 0);
    X[1] = lds_x(s + 1, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 2; r < 8; ++r) {
      mma_row(w, 1, r, 8 + r);
      if (refill && r < 6) a_piece(r - 2);
      X[r] = lds_x(s + 1, 0, r);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (refill) a_done();
    if (s + 1 < S) wait_groups(refill ? 1 : 0);   // W(s+1) [issued in phase 0] has landed; only A(s+4) may be younger
  };
  auto load_first = [&](int s) {
#pragma unroll
    for (int r = 0; r < 8; ++r) X[r] = lds_x(s, 0, r);
  };

  a_set(a_tile);
  w_set(w_tile);
  for (int st = 0; st < SW_SLOTS && st < S; ++st) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a_piece(r);
    a_done();
  }
  w_load(Wa);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  load_first(0);

  int s = 0;
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b2 = 0; b2 < 16; ++b2) acc[a][b2] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nt; ++kt, ++s) {
      if (s + 1 < S) w_load(Wb);   // weights of the next stage, one stage ahead
      phase(Wa, s, 0, 0);
      phase(Wa, s, 0, 1);
      phase(Wa, s, 1, 0);
      phase_sync(Wa, s);
#pragma unroll
      for (int f = 0; f < 4; ++f) Wa[f] = Wb[f];
    }

    // ---- epilogue ----
    // Lane (j, g) owns rows 16 fm + j, features 8 g .. 8 g + 7 of its wave's 32: written directly that is 16 rows x
    // 64 bytes per store instruction, a pattern the CU's store path runs at 15 B/clk (3.6 us per tile even on an
    // otherwise idle chip; tools/probes/l1_probe.hip) against 56 B/clk for whole contiguous rows.  So the 8 waves
    // assemble 32 rows x 512 B at a time in a spare LDS slab (two slabs alternate: one barrier per pass) and
    // every store instruction then writes 2 whole rows of the tile.  Slab pieces are XOR-swizzled (piece ^ (row & 7)):
    // conflict-free for the b128 writes (8 rows per lane group) and reads (pieces 0..31 of one row).
    uint32_t tm, tn;
    tile_of_block(tile, n_tiles, tiles_m, tiles_n, tm, tn);
    const int64_t m0 = (int64_t)tm * BM2;
    const int n0 = (int)tn * BN2;
    float bias_v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) bias_v[t] = p.bias ? p.bias[n0 + wave * 32 + 8 * g + t] : 0.f;
    bf16_t* cbase = reinterpret_cast<bf16_t*>(p.C) + n0 + (lane & 31) * 8;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      char* slab = smem + SW_LDS + (pass & 1) * 16384;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int fm = 2 * pass + h;
        float v[8];
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) {
          v[4 * fn + 0] = acc[fn][fm].x + bias_v[4 * fn + 0];
          v[4 * fn + 1] = acc[fn][fm].y + bias_v[4 * fn + 1];
          v[4 * fn + 2] = acc[fn][fm].z + bias_v[4 * fn + 2];
          v[4 * fn + 3] = acc[fn][fm].w + bias_v[4 * fn + 3];
        }
        if (p.act == AURORA_ACT_GELU) {
#pragma unroll
          for (int t = 0; t < 8; ++t) v[t] = gelu_for<bf16_t>(v[t]);
        } else if (p.act == AURORA_ACT_SILU) {
#pragma unroll
          for (int t = 0; t < 8; ++t) v[t] = v[t] / (1.0f + expf(-v[t]));
        }
        const int row = 16 * h + i16;
        *reinterpret_cast<u32x4*>(slab + row * 512 + (((4 * wave + g) ^ (row & 7)) << 4)) =
            u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int row = 4 * wave + 2 * q + (lane >> 5);
        const u32x4 d = *reinterpret_cast<const u32x4*>(slab + row * 512 + (((lane & 31) ^ (row & 7)) << 4));
        const int64_t m = m0 + 32 * pass + row;
        // (non-temporal: the result is not re-read by this kernel and should not displace the operand panels in L2)
        if (m < p.M) __builtin_nontemporal_store(d, reinterpret_cast<u32x4*>(cbase + m * p.ldc));
      }
    }
  }
}

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

__global__ __launch_bounds__(THREADS2, 2) void linear_kernel_256_f32x3(const LinearArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
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

  for (int j = 0; j < np; ++j) {
    const char* bufa = smem + ((2 * j) & (NSTAGE2 - 1)) * STAGE2;
    const char* bufb = smem + ((2 * j + 1) & (NSTAGE2 - 1)) * STAGE2;
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
  epilogue_256<float>(p, acc, m0, n0, wm, wn, i16, g);
}

}  // namespace

}  // namespace aurora

using namespace aurora;

namespace {
// 0: native fp32 MFMA (v_mfma_f32_16x16x4_f32); 1: 3 x bf16 operand splitting (default).
int g_f32_mode = -1;
int f32_mode() {
  if (g_f32_mode < 0) {
    const char* e = getenv("AURORA_F32_GEMM");
    g_f32_mode = (e && e[0] == 'n') ? 0 : 1;   // AURORA_F32_GEMM=native
  }
  return g_f32_mode;
}

// Workspace for the fragment-ordered copy of a weight matrix (rewritten by every call; stream-ordered: the pack
// kernel and the GEMM that reads it are launched back to back on the caller's stream, so calls on ONE stream --
// what the engine does -- are safe; concurrent streams would race on it).  The first call allocates it, which is
// not legal inside a stream capture: callers that capture warm up eagerly first (the engine does).
void* packed_weight_buffer(size_t bytes) {
  static void* buf = nullptr;
  static bool tried = false;
  constexpr size_t CAP = (size_t)128 << 20;   // 8192 x 8192 bf16; allocated once and never moved, so that a
  if (!tried) {                               // captured hipGraph can never hold a stale pointer
    tried = true;
    if (hipMalloc(&buf, CAP) != hipSuccess) buf = nullptr;
  }
  return bytes <= CAP ? buf : nullptr;
}
int device_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n;
}
}  // namespace

extern "C" int aurora_hip_set_f32_gemm(int mode) {
  const int prev = f32_mode();
  if (mode == 0 || mode == 1) g_f32_mode = mode;
  return prev;
}

extern "C" int aurora_hip_linear(const void* A, int64_t lda, const void* W, int64_t ldw,
                                 const float* bias, void* C, int64_t ldc, void* C2, int64_t ldc2,
                                 const float* residual, int64_t ldr, int64_t M, int N, int K,
                                 int dtype, int act, void* stream) {
  AURORA_CHECK_ARG(dtype == AURORA_F32 || dtype == AURORA_BF16, "linear: bad dtype %d", dtype);
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
  const bool split = dtype == AURORA_F32 && f32_mode() == 1;
  const bool big = (M >= 1024 || split) && N % BN2 == 0 && getenv("AURORA_GEMM_SMALL_ONLY") == nullptr;
  // 256 x 128 tiles with two workgroups per CU when K is short (few stages per tile) -- measured, tools/gemm_bench.py
  const bool half = false;
  const int bm = big ? BM2 : BM, bn = big ? BN2 : BN, rowb = big ? ROW2 : ROW_BYTES;
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
  AURORA_CHECK_ARG(p.n_blocks < (int64_t)1 << 31, "linear: too many tiles");

  dim3 grid((unsigned)p.n_blocks);
  // Streamed-weights persistent kernel: bf16, plain bf16 output (AURORA_GEMM_SW=0 falls back to the ring kernel).
  static const char* sw_env = getenv("AURORA_GEMM_SW");
  // Measured (tools/gemm_bench.py): it wins for K <= 512, where a tile is mostly prologue / epilogue; for longer K the
  // ring kernel's wider wave tile (fewer LDS reads per MFMA) wins.  AURORA_GEMM_SW=0 / 1 forces never / always.
  bool sw = big && dtype == AURORA_BF16 && C2 == nullptr && residual == nullptr && vec && K % 64 == 0 &&
            lda * es * 256 < ((int64_t)1 << 31) && !(sw_env && sw_env[0] == '0') &&
            (K <= 512 || (sw_env && sw_env[0] == '1'));
  u32x4* Wp = nullptr;
  if (sw) {
    Wp = reinterpret_cast<u32x4*>(packed_weight_buffer((size_t)N * K * 2));
    sw = Wp != nullptr;
  }
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)linear_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    (void)hipFuncSetAttribute((const void*)linear_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256<float, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE2 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256<bf16_t, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE2 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256_f32x3, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE2 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_sw, hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS_TOTAL);
    attr_done = true;
  }
  if (big) {
    if (split)
      hipLaunchKernelGGL(linear_kernel_256_f32x3, grid, dim3(THREADS2), NSTAGE2 * STAGE2, as_stream(stream), p);
    else if (sw) {
      const int KT = K / 64;
      const int64_t n_pieces = (int64_t)(N / 256) * KT * 8 * 4 * 64;   // 16-byte pieces = N * K / 8
      hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((n_pieces + 255) / 256)), dim3(256), 0, as_stream(stream),
                         (const bf16_t*)W, ldw, Wp, KT, n_pieces);
      p.k_tiles = KT;
      const unsigned cus = (unsigned)device_cus();
      hipLaunchKernelGGL(linear_kernel_sw, dim3((unsigned)(p.n_blocks < cus ? p.n_blocks : cus)), dim3(THREADS2), SW_LDS_TOTAL,
                         as_stream(stream), p, (const u32x4*)Wp);
    }
    else if (dtype == AURORA_F32)
      hipLaunchKernelGGL((linear_kernel_256<float, 4, 4>), grid, dim3(THREADS2), NSTAGE2 * STAGE2, as_stream(stream), p);
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
