// 3D shifted-window attention core:  O = softmax(Q K^T / 8 + mask) V  per (window, head).
//
// One workgroup per (batch element, window, head).  A window holds up to 144 tokens
// (2 levels x 6 x 12), head_dim = 64.  The cyclic shift, zero padding, window partition and
// their inverses (reference swin3d.py:471-505) are pure index permutations and arrive as the
// per-window token table `tok` (host geometry, aurora_amd/engine/geometry.py): this kernel
// gathers q/k/v rows of the token-ordered qkv buffer and scatters O back to token order, so
// none of the reference's 4-6 full-tensor shuffle copies per block exists here.  Padded window
// positions (tok < 0) carry q = k = v = bias (what Linear(0) yields upstream) and are not stored.
//
// bf16 kernel (the hot one, HBM-bound: 4 * 144 * 64 * 2 B = 72 KiB of traffic per workgroup
// against ~10.6 MFLOP): 3 waves, wave w owns query tiles {w, w+3, w+6} of 16 queries.
//   * K and V are staged row-major in LDS (128 B rows, 16-byte pieces XOR-swizzled by row & 7),
//     Q fragments come straight from global memory; all 18 16-byte loads of a thread are issued
//     before anything waits.
//   * S^T = K Q^T with v_mfma_f32_16x16x32_bf16: the C fragment then gives every lane ONE
//     query (lane & 15) and the keys 16*kt + 4*(lane>>4) + r.  The softmax row reduction is 36
//     in-lane values + two xor-shuffles (lanes l, l^16, l^32, l^48 share a query).
//   * That same register layout IS the B operand of v_mfma_f32_16x16x16_bf16 for
//     O^T = V^T P^T (k-slot 4g + j <-> key 16*kt + 4g + j), so P never leaves registers and
//     144 = 9 * 16 needs no key padding.  The A operand (V^T) comes out of the row-major V image
//     through gfx950's transposing LDS read ds_read_b64_tr_b16 -- no transposed copy of V exists.
//   * The kernel was VALU-bound before it was HBM-bound (4.2 k VALU instructions per wave in the
//     first version): conversions use v_cvt_pk_bf16_f32, the softmax runs in the exp2 domain with the
//     1/8 scale folded into one FMA per score, and the -100 mask is skipped for single-group windows.
//   * O^T's C fragment holds 4 consecutive d per lane and query: 8-byte stores to token order.
//
// fp32 kernel (exact-parity path for fp32 models): one thread per query, K/V broadcast from
// LDS, online softmax in fp32 FMAs.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace aurora {
namespace {

constexpr int HD = 64;          // head dim
constexpr int MAXN = 144;       // max tokens per window
constexpr int MAXT = MAXN / 16; // 9 tiles of 16

struct AttnArgs {
  const void* qkv; const float* bias; void* out;
  const int32_t* tok; const uint8_t* grp;
  int B; int64_t L; int D; int heads; int n_windows; int N;
  int xcd_order;  // 1: every XCD walks a contiguous range of (window, head) items (the heads of a window run side by side)
  int64_t L_out;  // rows of `out` per batch element; tokens >= L_out (halo rows of a band) are not stored
  // bf16 kernel: 0 = `qkv` is (B * L, 3 D) token-major; > 0 = head planes (aurora_hip_linear_planes): head h owns a plane of
  // [B * L rows][q | k | v = 192 elements], planes this many elements apart.  A window's tokens are runs of consecutive
  // rows, so a plane keeps what an item gathers -- 384 bytes per token -- contiguous in DRAM over each run (token-major
  // the 128-byte pieces are D * 2 and 3 D * 2 bytes apart): profiles/r04_ab_attention.log (2).
  int64_t plane_stride;
};

typedef short bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr float LOG2E = 1.4426950408889634f;

// max without the canonicalising `v_max_f32 x, x, x` hipcc puts in front of every operand of fmaxf (it cannot prove that an
// MFMA result is not a signalling NaN): 63 instructions for the maximum of a lane's 36 scores became 18.  The softmax is
// VALU-issue-bound (an item costs ~12.5 k SIMD cycles per wave, ~1.7 k of them MFMA), so every instruction per score counts.
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// The -100 mask (swin3d.py:357-358) of four keys at once: x holds (key group) ^ (query group) per byte, 0 for the same
// group.  Every non-zero byte becomes 1 (the carry-free "has a non-zero byte" trick: bit 7 of ((b & 0x7f) + 0x7f) | b is set
// iff b != 0), so a key of another group gets the reference's literal -100 -- not -100 * (group difference), which round 4
// shipped: identical after rounding for any realistic score range, but an out-of-group key whose raw score exceeds the
// in-group maximum by more than ~90 would have differed (tests/test_gpu_ops.py pins it with such a key).  The mask then
// costs five integer operations per FOUR keys plus a byte conversion and an FMA per score (v_cvt_f32_ubyteN, v_fmac);
// the kernel is HBM-bound, not issue-bound (profiles/r04_ab_attention.log).
__device__ __forceinline__ void mask4(f32x4& s, uint32_t x) {
  x = ((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) >> 7) & 0x01010101u;
  s.x = fmaf((float)(x & 0xffu), -100.0f * 8.0f, s.x);   // pre-scale: multiplied by 1/8 later
  s.y = fmaf((float)((x >> 8) & 0xffu), -100.0f * 8.0f, s.y);
  s.z = fmaf((float)((x >> 16) & 0xffu), -100.0f * 8.0f, s.z);
  s.w = fmaf((float)(x >> 24), -100.0f * 8.0f, s.w);
}
// Row sums on the matrix pipe: O^T = V^T P^T with a V^T tile of ones yields sum_k P[q][k] in every row -- 9 MFMAs of a pipe
// that idles during the softmax instead of 36 adds and two cross-lane steps; the weights that are normalised are then the
// bf16 probabilities the P V product really uses (they sum to one exactly).
__device__ __forceinline__ bf16x4_t ones_bf16x4() { return bf16x4_t{(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80}; }

// Reductions over the four lanes l, l^16, l^32, l^48 that share a query: gfx950's row / half swaps
// (v_permlane16_swap, v_permlane32_swap: one VALU slot each) instead of two ds_bpermute round trips through the LDS
// crossbar.  swap(x, x) leaves {own-or-partner, partner-or-own} in the two results, so op(r.x, r.y) is the pairwise
// reduction in every lane.
typedef unsigned u32x2_sw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float quad_max(float v) {
  u32x2_sw r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = vmax2(__uint_as_float(r.x), __uint_as_float(r.y));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return vmax2(__uint_as_float(r.x), __uint_as_float(r.y));
}
__device__ __forceinline__ float quad_sum(float v) {
  u32x2_sw r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// Lanes g and g^1 (rows of 16 lanes) trade one of two packed d-tiles: even rows end up with d-tile `a` of both rows,
// odd rows with d-tile `b` of both -- 16 consecutive bytes of the output row per lane (see WIDE below).
__device__ __forceinline__ u32x4 pair_rows(u32x2 a, u32x2 b) {
  const u32x2_sw x = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
  const u32x2_sw y = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
  return u32x4{x.x, y.x, x.y, y.y};
}

// FULL: the window has exactly 144 tokens (every stage of the published 0.25 / 0.1 / 0.4 degree
// configurations): tile counts become compile-time constants.
template <bool FULL, int WIDE>   // WIDE: 1 16-byte result stores, 2 16-byte stores covering whole 128-byte rows
__global__ __launch_bounds__(192) void window_attention_bf16(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * MAXN * 128 + MAXN * 4 + MAXN + 16];
  char* const s_k = smem;                 // [144][128 B], 16-byte pieces XOR-swizzled by row & 7
  char* const s_v = smem + MAXN * 128;    // same image for V; transposed on READ (ds_read_b64_tr_b16)
  int32_t* const s_tok = reinterpret_cast<int32_t*>(smem + 2 * MAXN * 128);
  uint8_t* const s_grp = reinterpret_cast<uint8_t*>(s_tok + MAXN);

  const int tid = threadIdx.x;
  const int N = FULL ? MAXN : p.N, nt = FULL ? MAXT : (N + 15) >> 4;
  uint32_t item = blockIdx.x;
  if (p.xcd_order) {   // workgroup b runs on XCD b % 8: give each XCD a contiguous range of items
    const uint32_t nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = item & 7, idx = item >> 3;
    item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  // rows of 3 D: the heads of a window side by side (pieces of the same token rows); head planes: the windows of a head
  // side by side (neighbouring windows continue each other's runs of rows in the same plane)
  const bool head_major = p.plane_stride != 0;   // (6.24 -> 6.14 ms per step against window-major on planes)
  const int h = head_major ? (item / p.n_windows) % p.heads : item % p.heads;
  const int w = head_major ? item % p.n_windows : (item / p.heads) % p.n_windows;
  const int b = item / (p.heads * p.n_windows);

  // Token / group tables of this window; the -100 mask only matters if the window really mixes groups.
  int differs = 0;
  if (tid < MAXN) {
    s_tok[tid] = tid < N ? p.tok[(int64_t)w * N + tid] : -2;
    int gv = 0;
    if (p.grp && tid < N) {
      gv = p.grp[(int64_t)w * N + tid];
      differs = gv != p.grp[(int64_t)w * N];
    }
    s_grp[tid] = (uint8_t)gv;
  }
  const bool masked = __syncthreads_or(differs) != 0;

  const int col_q = h * HD, col_k = p.D + h * HD, col_v = 2 * p.D + h * HD;   // columns of the token-major row (and of the bias)
  // where this head's q / k / v rows start, and how far apart they are
  const bf16_t* const base = reinterpret_cast<const bf16_t*>(p.qkv);
  const bool planes = p.plane_stride != 0;   // (uniform)
  const int64_t row_stride = planes ? 3 * HD : 3 * p.D;
  const bf16_t* const rows0 = base + (int64_t)b * p.L * row_stride + (planes ? (int64_t)h * p.plane_stride : 0);
  const bf16_t* const q_rows = rows0 + (planes ? 0 : col_q);
  const bf16_t* const k_rows = rows0 + (planes ? HD : col_k);
  const bf16_t* const v_rows = rows0 + (planes ? 2 * HD : col_v);

  // 16 bytes (8 bf16) of row `t` of q / k / v, `off` elements into the head.  Loads are issued unconditionally
  // (clamped row) so that all of a thread's loads are in flight together; padded rows (t == -1: bias,
  // what Linear(0) yields) and rows beyond the window (t == -2: zeros) are patched afterwards.
  auto issue = [&](const bf16_t* rows, int t, int off) -> u32x4 {
    return *reinterpret_cast<const u32x4*>(rows + (int64_t)(t < 0 ? 0 : t) * row_stride + off);
  };
  auto patch = [&](u32x4 v, int t, int col) -> u32x4 {
    if (t >= 0) return v;
    if (t == -1 && p.bias) {
      float bb[8];
      load8(p.bias + col, bb);
      return u32x4{pack_bf16x2(bb[0], bb[1]), pack_bf16x2(bb[2], bb[3]), pack_bf16x2(bb[4], bb[5]), pack_bf16x2(bb[6], bb[7])};
    }
    return u32x4{0u, 0u, 0u, 0u};
  };

  const int lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;

  // ---- issue every global load of this thread: 6 K + 6 V pieces, then its Q fragments ----
  constexpr int PIECES = MAXN * 8 / 192;
  u32x4 kreg[PIECES], vreg[PIECES];
  int trow[PIECES];
#pragma unroll
  for (int it = 0; it < PIECES; ++it) {
    const int idx = tid + it * 192, row = idx >> 3, c = idx & 7;
    trow[it] = (FULL || row < nt * 16) ? s_tok[row] : -3;  // -3: row not staged at all
    if (trow[it] != -3) {
      kreg[it] = issue(k_rows, trow[it], c * 8);
      vreg[it] = issue(v_rows, trow[it], c * 8);
    }
  }
  u32x4 qreg[3][2];
  int tq_of[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int qt = wave + 3 * j;
    tq_of[j] = qt < nt ? s_tok[qt * 16 + i16] : -2;
    if (qt < nt) {
      qreg[j][0] = issue(q_rows, tq_of[j], g * 8);
      qreg[j][1] = issue(q_rows, tq_of[j], 32 + g * 8);
    }
  }
#pragma unroll
  for (int it = 0; it < PIECES; ++it) {
    if (trow[it] == -3) continue;
    const int idx = tid + it * 192, row = idx >> 3, c = idx & 7;
    const int off = row * 128 + ((c ^ (row & 7)) << 4);
    *reinterpret_cast<u32x4*>(s_k + off) = patch(kreg[it], trow[it], col_k + c * 8);
    *reinterpret_cast<u32x4*>(s_v + off) = patch(vreg[it], trow[it], col_v + c * 8);
  }
  __syncthreads();

  // Per-lane LDS offsets.  K fragment (A operand of S^T = K Q^T): row 16*kt + i16, piece g + 4*ks.
  const int koff0 = i16 * 128 + ((g ^ (i16 & 7)) << 4), koff1 = i16 * 128 + (((g + 4) ^ (i16 & 7)) << 4);
  // V^T fragment (A operand of O^T = V^T P^T) through the transposing read: within a 16-lane group,
  // lane 4*r + q supplies the 8 bytes V[key 4g + r][16*dt + 4q .. +3]; lane c receives column c.
  int voff[4];
  {
    const int key = 4 * g + (i16 >> 2), q = i16 & 3;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      voff[dt] = key * 128 + (((2 * dt + (q >> 1)) ^ (key & 7)) << 4) + (q & 1) * 8;
  }
  bf16_t* const out = reinterpret_cast<bf16_t*>(p.out) + (int64_t)b * p.L_out * p.D;
  const float c_scale = 0.125f * LOG2E;  // 1/sqrt(64), in the exp2 domain

#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int qt = wave + 3 * j;
    if (qt >= nt) break;
    const int tq = tq_of[j];  // -2 beyond N
    const u32x4 qf0 = patch(qreg[j][0], tq, col_q + g * 8), qf1 = patch(qreg[j][1], tq, col_q + 32 + g * 8);

    // ---- S^T tiles: keys along registers, this lane's query along lanes ----
    f32x4 st[MAXT];
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (FULL || kt < nt) {
        const u32x4 k0 = *reinterpret_cast<const u32x4*>(s_k + kt * 2048 + koff0);
        const u32x4 k1 = *reinterpret_cast<const u32x4*>(s_k + kt * 2048 + koff1);
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, k0),
                                                         __builtin_bit_cast(bf16x8_t, qf0), st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, k1),
                                                         __builtin_bit_cast(bf16x8_t, qf1), st[kt], 0, 0, 0);
      }
    }

    // ---- softmax over the keys of this lane's query, in the exp2 domain ----
    if (masked) {
      const uint32_t gq4 = (uint32_t)s_grp[qt * 16 + i16] * 0x01010101u;
#pragma unroll
      for (int kt = 0; kt < MAXT; ++kt)
        if (FULL || kt < nt) mask4(st[kt], *reinterpret_cast<const uint32_t*>(s_grp + kt * 16 + 4 * g) ^ gq4);
    }
    if (!FULL && N < nt * 16) {  // keys beyond the window in the last tile
      const int kt = nt - 1, k0 = kt * 16 + 4 * g;
#pragma unroll
      for (int t2 = 0; t2 < MAXT; ++t2)
        if (t2 == kt) {
          if (k0 + 0 >= N) st[t2].x = -INFINITY;
          if (k0 + 1 >= N) st[t2].y = -INFINITY;
          if (k0 + 2 >= N) st[t2].z = -INFINITY;
          if (k0 + 3 >= N) st[t2].w = -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt)
      if (FULL || kt < nt) mx = vmax3(vmax3(mx, st[kt].x, st[kt].y), st[kt].z, st[kt].w);
    mx = quad_max(mx);
    const float mxs = mx * c_scale;
    u32x2 pk[MAXT];  // packed bf16 probabilities, kept as dwords (bit-cast at the MFMA)
    f32x4 osum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt) {
      if (FULL || kt < nt) {
        const float e0 = __builtin_amdgcn_exp2f(fmaf(st[kt].x, c_scale, -mxs));
        const float e1 = __builtin_amdgcn_exp2f(fmaf(st[kt].y, c_scale, -mxs));
        const float e2 = __builtin_amdgcn_exp2f(fmaf(st[kt].z, c_scale, -mxs));
        const float e3 = __builtin_amdgcn_exp2f(fmaf(st[kt].w, c_scale, -mxs));
        pk[kt] = u32x2{pack_bf16x2(e0, e1), pack_bf16x2(e2, e3)};
        osum = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ones_bf16x4(), __builtin_bit_cast(bf16x4_t, pk[kt]), osum, 0, 0, 0);
      } else {
        pk[kt] = u32x2{0u, 0u};
      }
    }
    const float inv = __builtin_amdgcn_rcpf(osum.x);   // every row of osum holds this lane's query's row sum

    // ---- O^T = V^T P^T, 4 d-tiles of 16 ----
    auto pv = [&](int dt) -> u32x2 {
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < MAXT; ++kt) {
        if (FULL || kt < nt) {
          const bf16x4_t vf = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4_t*)(s_v + kt * 2048 + voff[dt]));
          o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf, __builtin_bit_cast(bf16x4_t, pk[kt]), o, 0, 0, 0);
        }
      }
      return u32x2{pack_bf16x2(o.x * inv, o.y * inv), pack_bf16x2(o.z * inv, o.w * inv)};
    };
    const bool live = tq >= 0 && tq < p.L_out;
    if constexpr (WIDE == 2) {
      // As WIDE == 1 below, then queries i and i ^ 8 trade one 16-byte piece each (a DPP rotation of the 16-lane row by
      // 8), so that a store instruction writes the WHOLE 128-byte row of eight queries instead of half a row of
      // sixteen: the first covers queries 0-7 of the tile (lanes i16 < 8 their own first half, lanes i16 >= 8 the second
      // half of query i16 - 8), the second queries 8-15.  No cache line is written in two halves.
      const bool odd = (g & 1) != 0, upper = i16 >= 8;
      const u32x4 v0 = pair_rows(pv(0), pv(1)), v1 = pair_rows(pv(2), pv(3));
      const u32x4 give = upper ? v0 : v1;   // what the partner lane stores for me
      auto ror8 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true); };   // row_ror:8
      const u32x4 got = u32x4{ror8(give.x), ror8(give.y), ror8(give.z), ror8(give.w)};
      const int tq_p = (int)ror8((uint32_t)tq);
      const bool live_p = tq_p >= 0 && tq_p < p.L_out;
      const int piece = odd ? 16 + 4 * (g - 1) : 4 * g;
      // first instruction: rows of queries 0-7; second: rows of queries 8-15
      const int64_t t_a = upper ? tq_p : tq, t_b = upper ? tq : tq_p;
      const bool l_a = upper ? live_p : live, l_b = upper ? live : live_p;
      const int col_a = col_q + (upper ? 32 : 0) + piece, col_b = col_a;
      if (l_a) *reinterpret_cast<u32x4*>(out + t_a * p.D + col_a) = upper ? got : v0;
      if (l_b) *reinterpret_cast<u32x4*>(out + t_b * p.D + col_b) = upper ? v1 : got;
    } else {
      static_assert(WIDE == 1, "WIDE is 1 or 2");
      // The C fragment gives a lane 4 consecutive d (8 bytes) per d-tile.  Lanes g, g^1 trade halves of two d-tiles so
      // that every lane stores 16 bytes and four lanes cover 64 contiguous bytes of the output row.
      const bool odd = (g & 1) != 0;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const u32x4 v = pair_rows(pv(2 * a), pv(2 * a + 1));
        const int col = col_q + 32 * a + (odd ? 16 + 4 * (g - 1) : 4 * g);
        if (live) *reinterpret_cast<u32x4*>(out + (int64_t)tq * p.D + col) = v;
      }
    }
  }
}

// ---- fp32: one thread per query -----------------------------------------------------------------
__global__ __launch_bounds__(192) void window_attention_f32(const AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  float* const s_k = reinterpret_cast<float*>(dyn);   // [N][64]
  float* const s_v = s_k + MAXN * HD;                 // [N][64]
  int32_t* const s_tok = reinterpret_cast<int32_t*>(s_v + MAXN * HD);
  uint8_t* const s_grp = reinterpret_cast<uint8_t*>(s_tok + MAXN);

  const int tid = threadIdx.x;
  const int N = p.N;
  const int h = blockIdx.x % p.heads;
  const int w = (blockIdx.x / p.heads) % p.n_windows;
  const int b = blockIdx.x / (p.heads * p.n_windows);
  const bool masked = p.grp != nullptr;

  if (tid < MAXN) {
    s_tok[tid] = tid < N ? p.tok[(int64_t)w * N + tid] : -2;
    s_grp[tid] = (masked && tid < N) ? p.grp[(int64_t)w * N + tid] : 0;
  }
  __syncthreads();

  const float* const qkv = reinterpret_cast<const float*>(p.qkv) + (int64_t)b * p.L * 3 * p.D;
  const int D3 = 3 * p.D;
  const int col_q = h * HD, col_k = p.D + h * HD, col_v = 2 * p.D + h * HD;

  auto fetch4 = [&](int t, int col) -> f32x4 {
    if (t >= 0) return *reinterpret_cast<const f32x4*>(qkv + (int64_t)t * D3 + col);
    if (t == -1 && p.bias) return *reinterpret_cast<const f32x4*>(p.bias + col);
    return f32x4{0.f, 0.f, 0.f, 0.f};
  };

  for (int idx = tid; idx < N * 16; idx += 192) {
    const int row = idx >> 4, c = idx & 15;
    const int t = s_tok[row];
    *reinterpret_cast<f32x4*>(s_k + row * HD + c * 4) = fetch4(t, col_k + c * 4);
    *reinterpret_cast<f32x4*>(s_v + row * HD + c * 4) = fetch4(t, col_v + c * 4);
  }
  __syncthreads();

  if (tid >= N) return;
  const int tq = s_tok[tid];
  if (tq < 0 || tq >= p.L_out) return;  // padded query (cropped upstream) or a halo row of a band
  const int gq = s_grp[tid];
  float q[HD], o[HD];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const f32x4 v = fetch4(tq, col_q + c * 4);
    q[4 * c] = v.x * 0.125f; q[4 * c + 1] = v.y * 0.125f; q[4 * c + 2] = v.z * 0.125f; q[4 * c + 3] = v.w * 0.125f;
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  float mx = -INFINITY, sum = 0.f;
  for (int j = 0; j < N; ++j) {
    const float* kr = s_k + j * HD;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + 4 * c);
      s0 = fmaf(q[4 * c], kv.x, s0); s1 = fmaf(q[4 * c + 1], kv.y, s1);
      s2 = fmaf(q[4 * c + 2], kv.z, s2); s3 = fmaf(q[4 * c + 3], kv.w, s3);
    }
    float s = (s0 + s1) + (s2 + s3);
    if (masked && (int)s_grp[j] != gq) s += -100.0f;
    const float nm = fmaxf(mx, s);
    const float corr = expf(mx - nm), e = expf(s - nm);
    mx = nm;
    sum = sum * corr + e;
    const float* vr = s_v + j * HD;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + 4 * c);
      o[4 * c] = fmaf(e, vv.x, o[4 * c] * corr); o[4 * c + 1] = fmaf(e, vv.y, o[4 * c + 1] * corr);
      o[4 * c + 2] = fmaf(e, vv.z, o[4 * c + 2] * corr); o[4 * c + 3] = fmaf(e, vv.w, o[4 * c + 3] * corr);
    }
  }
  const float inv = 1.0f / sum;
  float* const op = reinterpret_cast<float*>(p.out) + ((int64_t)b * p.L_out + tq) * p.D + col_q;
#pragma unroll
  for (int c = 0; c < 16; ++c)
    *reinterpret_cast<f32x4*>(op + 4 * c) = f32x4{o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv};
}

}  // namespace
}  // namespace aurora

using namespace aurora;

extern "C" int aurora_hip_window_attention(const void* qkv, const float* qkv_bias, void* out,
                                           const int32_t* tok, const uint8_t* grp, int B, int64_t L,
                                           int64_t L_out, int D, int heads, int n_windows, int win_tokens,
                                           int dtype, void* stream) {
  return aurora_hip_window_attention_planes(qkv, 0, qkv_bias, out, tok, grp, B, L, L_out, D, heads, n_windows, win_tokens, dtype,
                                            stream);
}

extern "C" int aurora_hip_window_attention_planes(const void* qkv, int64_t plane_stride, const float* qkv_bias, void* out,
                                                  const int32_t* tok, const uint8_t* grp, int B, int64_t L,
                                                  int64_t L_out, int D, int heads, int n_windows, int win_tokens,
                                                  int dtype, void* stream) {
  AURORA_CHECK_ARG(dtype == AURORA_F32 || dtype == AURORA_BF16, "window_attention: bad dtype");
  AURORA_CHECK_ARG(plane_stride == 0 || (dtype == AURORA_BF16 && plane_stride >= (int64_t)B * L * 3 * HD && plane_stride % 8 == 0),
                   "window_attention: head planes need bf16 and planes of >= B * L rows (plane_stride=%lld)", (long long)plane_stride);
  AURORA_CHECK_ARG(heads > 0 && D == heads * HD, "window_attention: head_dim must be 64 (D=%d heads=%d)", D, heads);
  AURORA_CHECK_ARG(win_tokens >= 1 && win_tokens <= MAXN, "window_attention: window of %d tokens (max %d)", win_tokens, MAXN);
  AURORA_CHECK_ARG(B > 0 && n_windows > 0 && L > 0 && L_out > 0 && L_out <= L, "window_attention: empty problem");
  AURORA_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0 && (!qkv_bias || (uintptr_t)qkv_bias % 16 == 0),
                   "window_attention: unaligned buffer");
  const int64_t blocks = (int64_t)B * n_windows * heads;
  AURORA_CHECK_ARG(blocks < ((int64_t)1 << 31), "window_attention: grid too large");
  // Item order: with many items per launch every XCD walks a contiguous range of (window, head) items, so that the heads of
  // a window -- 128-byte pieces of the same token rows, i.e. of the same DRAM pages -- are requested side by side by the
  // CUs of one XCD instead of by eight XCDs at unrelated times: 4.55 -> 4.94 TB/s at stage 0, 4.18 -> 4.58 at stage 1
  // (isolated); a launch of a few thousand items (stage 2, a latitude band) is latency-bound and keeps the plain order.
  // (head planes, where an XCD's range is whole heads: from 3,000 items -- the un-sharded stage 2, 4,096 items: 6.23 -> 6.08 ms per step)
  const int xcd_order = blocks >= (plane_stride ? 3000 : 6000) ? 1 : 0;
  AttnArgs p{qkv, qkv_bias, out, tok, grp, B, L, D, heads, n_windows, win_tokens, xcd_order, L_out, plane_stride};
  if (dtype == AURORA_BF16) {
    // one workgroup per (window, head); full 144-token windows store whole 128-byte rows
    if (win_tokens != MAXN)
      hipLaunchKernelGGL((window_attention_bf16<false, 1>), dim3((unsigned)blocks), dim3(192), 0, as_stream(stream), p);
    else
      hipLaunchKernelGGL((window_attention_bf16<true, 2>), dim3((unsigned)blocks), dim3(192), 0, as_stream(stream), p);
  } else {
    const size_t lds = 2 * MAXN * HD * 4 + MAXN * 4 + MAXN + 16;
    static bool attr_done_dev[64] = {false};   // function attributes are per device
    bool& attr_done = attr_done_dev[current_device() & 63];
    if (!attr_done) {
      (void)hipFuncSetAttribute((const void*)window_attention_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_done = true;
    }
    hipLaunchKernelGGL(window_attention_f32, dim3((unsigned)blocks), dim3(192), lds, as_stream(stream), p);
  }
  return check_launch("window_attention");
}
