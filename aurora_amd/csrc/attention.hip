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
//   * K is staged row-major in LDS (128 B rows, 16-byte pieces XOR-swizzled by row & 7),
//     V is staged TRANSPOSED (Vt[d][key]), Q fragments come straight from global memory.
//   * S^T = K Q^T with v_mfma_f32_16x16x32_bf16: the C fragment then gives every lane ONE
//     query (lane & 15) and the keys 16*kt + 4*(lane>>4) + r.  The softmax row reduction is 36
//     in-lane values + two xor-shuffles (lanes l, l^16, l^32, l^48 share a query).
//   * That same register layout IS the B operand of v_mfma_f32_16x16x16_bf16 for
//     O^T = V^T P^T (k-slot 4g + j <-> key 16*kt + 4g + j), so P never leaves registers and
//     144 = 9 * 16 needs no key padding.  The A operand is an 8-byte read of Vt.
//   * O^T's C fragment holds 4 consecutive d per lane and query: 8-byte stores to token order.
//
// fp32 kernel (exact-parity path for fp32 models): one thread per query, K/V broadcast from
// LDS, online softmax in fp32 FMAs.
#include <math.h>

#include "common.h"

namespace aurora {
namespace {

constexpr int HD = 64;          // head dim
constexpr int MAXN = 144;       // max tokens per window
constexpr int MAXT = MAXN / 16; // 9 tiles of 16
constexpr int VT_STRIDE = 148;  // elements per Vt row (296 B: 8-byte aligned, off the 256 B bank period)

struct AttnArgs {
  const void* qkv; const float* bias; void* out;
  const int32_t* tok; const uint8_t* grp;
  int B; int64_t L; int D; int heads; int n_windows; int N;
};

typedef short bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(192) void window_attention_bf16(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[MAXN * 128 + HD * VT_STRIDE * 2 + MAXN * 4 + MAXN + 16];
  char* const s_k = smem;                                            // [144][128 B], swizzled
  bf16_t* const s_vt = reinterpret_cast<bf16_t*>(smem + MAXN * 128);  // [64][VT_STRIDE]
  int32_t* const s_tok = reinterpret_cast<int32_t*>(smem + MAXN * 128 + HD * VT_STRIDE * 2);
  uint8_t* const s_grp = reinterpret_cast<uint8_t*>(s_tok + MAXN);

  const int tid = threadIdx.x;
  const int N = p.N, nt = (N + 15) >> 4;
  const int h = blockIdx.x % p.heads;
  const int w = (blockIdx.x / p.heads) % p.n_windows;
  const int b = blockIdx.x / (p.heads * p.n_windows);
  const bool masked = p.grp != nullptr;

  if (tid < MAXN) {
    s_tok[tid] = tid < N ? p.tok[(int64_t)w * N + tid] : -2;
    s_grp[tid] = (masked && tid < N) ? p.grp[(int64_t)w * N + tid] : 0;
  }
  __syncthreads();

  const bf16_t* const qkv = reinterpret_cast<const bf16_t*>(p.qkv) + (int64_t)b * p.L * 3 * p.D;
  const int D3 = 3 * p.D;
  const int col_q = h * HD, col_k = p.D + h * HD, col_v = 2 * p.D + h * HD;

  // 16 bytes (8 bf16) of row `t`'s column block starting at `col`; bias for padded rows.
  auto fetch = [&](int t, int col) -> u32x4 {
    if (t >= 0) return *reinterpret_cast<const u32x4*>(qkv + (int64_t)t * D3 + col);
    if (t == -1) {
      float v[8];
      if (p.bias) load8(p.bias + col, v);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      return u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    }
    return u32x4{0u, 0u, 0u, 0u};
  };

  // ---- stage K (row-major, swizzled) and V (transposed) ----
  for (int idx = tid; idx < nt * 16 * 8; idx += 192) {
    const int row = idx >> 3, c = idx & 7;
    const int t = s_tok[row];
    const u32x4 kv = fetch(t, col_k + c * 8);
    *reinterpret_cast<u32x4*>(s_k + row * 128 + ((c ^ (row & 7)) << 4)) = kv;
    const u32x4 vv = fetch(t, col_v + c * 8);
    bf16_t* dst = s_vt + (c * 8) * VT_STRIDE + row;
    dst[0 * VT_STRIDE] = (bf16_t)(vv.x & 0xffff); dst[1 * VT_STRIDE] = (bf16_t)(vv.x >> 16);
    dst[2 * VT_STRIDE] = (bf16_t)(vv.y & 0xffff); dst[3 * VT_STRIDE] = (bf16_t)(vv.y >> 16);
    dst[4 * VT_STRIDE] = (bf16_t)(vv.z & 0xffff); dst[5 * VT_STRIDE] = (bf16_t)(vv.z >> 16);
    dst[6 * VT_STRIDE] = (bf16_t)(vv.w & 0xffff); dst[7 * VT_STRIDE] = (bf16_t)(vv.w >> 16);
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  bf16_t* const out = reinterpret_cast<bf16_t*>(p.out) + (int64_t)b * p.L * p.D;

  for (int qt = wave; qt < nt; qt += 3) {
    const int qi = qt * 16 + i16;
    const int tq = s_tok[qi];  // -2 beyond N
    const int gq = s_grp[qi];
    u32x4 qf[2];
    qf[0] = fetch(tq, col_q + g * 8);
    qf[1] = fetch(tq, col_q + 32 + g * 8);

    // ---- S^T tiles: keys along registers, this lane's query along lanes ----
    f32x4 st[MAXT];
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < nt) {
        const int row = kt * 16 + i16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int c = g + 4 * ks;
          const u32x4 kf = *reinterpret_cast<const u32x4*>(s_k + row * 128 + ((c ^ (row & 7)) << 4));
          st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf),
                                                           __builtin_bit_cast(bf16x8_t, qf[ks]), st[kt], 0, 0, 0);
        }
      }
    }

    // ---- scale, mask, softmax over the 144 keys of this lane's query ----
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt) {
      if (kt < nt) {
        const int k0 = kt * 16 + 4 * g;
        const uint32_t gk4 = *reinterpret_cast<const uint32_t*>(s_grp + k0);
        float s[4] = {st[kt].x, st[kt].y, st[kt].z, st[kt].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = s[r] * 0.125f;
          if (masked && (int)((gk4 >> (8 * r)) & 0xff) != gq) v += -100.0f;
          if (k0 + r >= N) v = -INFINITY;
          s[r] = v;
          mx = fmaxf(mx, v);
        }
        st[kt] = f32x4{s[0], s[1], s[2], s[3]};
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    bf16x4_t pk[MAXT];
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt) {
      if (kt < nt) {
        const float e0 = __expf(st[kt].x - mx), e1 = __expf(st[kt].y - mx);
        const float e2 = __expf(st[kt].z - mx), e3 = __expf(st[kt].w - mx);
        sum += (e0 + e1) + (e2 + e3);
        const uint32_t lo = pack_bf16x2(e0, e1), hi = pack_bf16x2(e2, e3);
        pk[kt] = __builtin_bit_cast(bf16x4_t, u32x2{lo, hi});
      } else {
        pk[kt] = bf16x4_t{0, 0, 0, 0};
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // ---- O^T = V^T P^T, 4 d-tiles of 16 ----
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
      const bf16_t* vrow = s_vt + (dt * 16 + i16) * VT_STRIDE + 4 * g;
#pragma unroll
      for (int kt = 0; kt < MAXT; ++kt) {
        if (kt < nt) {
          const u32x2 vf = *reinterpret_cast<const u32x2*>(vrow + kt * 16);
          o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4_t, vf), pk[kt], o, 0, 0, 0);
        }
      }
      if (tq >= 0) {
        const float v[4] = {o.x * inv, o.y * inv, o.z * inv, o.w * inv};
        store4(out + (int64_t)tq * p.D + col_q + dt * 16 + 4 * g, v);
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
  if (tq < 0) return;  // padded query: its output is cropped away upstream
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
  float* const op = reinterpret_cast<float*>(p.out) + ((int64_t)b * p.L + tq) * p.D + col_q;
#pragma unroll
  for (int c = 0; c < 16; ++c)
    *reinterpret_cast<f32x4*>(op + 4 * c) = f32x4{o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv};
}

}  // namespace
}  // namespace aurora

using namespace aurora;

extern "C" int aurora_hip_window_attention(const void* qkv, const float* qkv_bias, void* out,
                                           const int32_t* tok, const uint8_t* grp, int B, int64_t L,
                                           int D, int heads, int n_windows, int win_tokens, int dtype,
                                           void* stream) {
  AURORA_CHECK_ARG(dtype == AURORA_F32 || dtype == AURORA_BF16, "window_attention: bad dtype");
  AURORA_CHECK_ARG(heads > 0 && D == heads * HD, "window_attention: head_dim must be 64 (D=%d heads=%d)", D, heads);
  AURORA_CHECK_ARG(win_tokens >= 1 && win_tokens <= MAXN, "window_attention: window of %d tokens (max %d)", win_tokens, MAXN);
  AURORA_CHECK_ARG(B > 0 && n_windows > 0 && L > 0, "window_attention: empty problem");
  AURORA_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0 && (!qkv_bias || (uintptr_t)qkv_bias % 16 == 0),
                   "window_attention: unaligned buffer");
  const int64_t blocks = (int64_t)B * n_windows * heads;
  AURORA_CHECK_ARG(blocks < ((int64_t)1 << 31), "window_attention: grid too large");
  AttnArgs p{qkv, qkv_bias, out, tok, grp, B, L, D, heads, n_windows, win_tokens};
  if (dtype == AURORA_BF16) {
    hipLaunchKernelGGL(window_attention_bf16, dim3((unsigned)blocks), dim3(192), 0, as_stream(stream), p);
  } else {
    const size_t lds = 2 * MAXN * HD * 4 + MAXN * 4 + MAXN + 16;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)window_attention_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr = true;
    }
    hipLaunchKernelGGL(window_attention_f32, dim3((unsigned)blocks), dim3(192), lds, as_stream(stream), p);
  }
  return check_launch("window_attention");
}
