// Encoder front end / decoder back end kernels (all HBM-bound, fp32 pixels on one side):
//   * patchify            : Batch.normalise + pre-encoder transforms + unfold into the patch-embed
//                           GEMM operand (LevelPatchEmbed's conv3d == GEMM over unfolded patches)
//   * perceiver_attention : the tiny (3 x 13 / 13 x 3) per-column cross attentions
//   * assemble_tokens     : surface/latent concat + position/scale/time embeddings
//   * unpatchify          : head outputs -> (B, V, C, H, W) fields, clamp + Batch.unnormalise fused
#include <math.h>

#include <type_traits>

#include "common.h"

namespace aurora {
namespace {

// ------------------------------------------------------------------------------------------------
// patchify
// ------------------------------------------------------------------------------------------------
constexpr int MAX_VARS = 32;

struct PatchVar {
  const float* src; int64_t sb, st, sc, sh, sw;
  const float* loc; const float* inv_scale;
  int32_t transform; float tw0, tw1, tb;
};
struct PatchArgs {
  PatchVar v[MAX_VARS];
  void* out; int64_t Kpad; int k_offset; int K_total;
  int n_vars, B, T, n_lvl, Hp, Wp, P;
  float* absmax;   // nullable: max |value written| is folded into this word (non-negative floats order like their bits)
};

// max over the wave, then at most one atomic per wave -- and only if it can still raise the word (a relaxed look first):
// the guard word of the encoder's operand-split decision (step.hip) costs the producer of the values nothing but this,
// where a separate absmax pass re-read all of them (0.6 GB per step at 0.25 degree).
__device__ __forceinline__ void fold_absmax(float* word, float m) {
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) {
    const unsigned int bits = __float_as_uint(m);
    if (bits > __hip_atomic_load(reinterpret_cast<unsigned int*>(word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(reinterpret_cast<unsigned int*>(word), bits);
  }
}

__device__ __forceinline__ float patch_transform(float z, const PatchVar& d) {
  if (d.transform == 1) return fmaxf(z, 0.f);
  if (d.transform == 2) {
    // reference aurora.py:733-742: Linear(2,1)([clamp(z,0,2.5), (log(max(z,eps)) - log eps) / -log eps])
    z = fmaxf(z, 0.f);  // positive variables are clamped at 0 before the hook (aurora.py:301-317)
    const float eps = 1e-4f, log_eps = -9.210340371976182f;  // log(1e-4)
    const float f0 = fminf(z, 2.5f);
    const float f1 = (logf(fmaxf(z, eps)) - log_eps) / (-log_eps);
    return d.tw0 * f0 + d.tw1 * f1 + d.tb;
  }
  // ocean-wave variant (reference aurora.py:892-912): missing values are NaN in the input
  if (d.transform == 3) return isnan(z) ? 0.f : 1.f;                       // density channel
  if (d.transform == 4) return isnan(z) ? 0.f : z;                         // value, nan_to_num(0)
  if (d.transform == 5) { const float r = sinf(z * 0.017453292519943295f); return isnan(r) ? 0.f : r; }
  if (d.transform == 6) { const float r = cosf(z * 0.017453292519943295f); return isnan(r) ? 0.f : r; }
  return z;
}

// Patch size 4, fp32, every piece 16-byte aligned (the 0.25-degree models).  A workgroup owns a run of 8 * PATCH4_GROUPS
// patches of one patch row at one level; lane = (piece q = (variable, time, image row i) of the output row, patch s of 8):
// the eight lanes of a q read 128 contiguous bytes of one image row of one field, the eight q of an s write 128 contiguous
// bytes of one output row -- whole cache lines on both sides, the position decoded once per PATCH4_GROUPS pieces.  (The
// first form, one thread per piece in output order, decoded a flat 64-bit index with six 64-bit divisions: ~700 VALU
// instructions per 16 bytes, 0.27 ms per launch where the bytes take 0.09.)
constexpr int PATCH4_GROUPS = 4;

__global__ __launch_bounds__(1024) void patchify_kernel(const PatchArgs p) {
  // grid: x = run of 8 * PATCH4_GROUPS patches, y = patch row, z = (level, batch); blockDim = 8 * pieces per row, rounded up
  const int q_per_row = p.n_vars * p.T * 4;
  const int q = (int)(threadIdx.x >> 3);
  float m = 0.f;
  if (q < q_per_row) {
    const int i = q & 3, vt = q >> 2;
    const int v = vt / p.T, t = vt - v * p.T;
    const unsigned hp = blockIdx.y;
    const unsigned c = blockIdx.z / (unsigned)p.B, b = blockIdx.z - c * (unsigned)p.B;
    const int wp = (int)(blockIdx.x * (8u * PATCH4_GROUPS) + (threadIdx.x & 7u));
    const PatchVar& d = p.v[v];
    const float loc = d.loc[c], inv = d.inv_scale[c];
    const float* src = d.src + b * d.sb + t * d.st + c * d.sc + (int64_t)(hp * 4 + i) * d.sh + (int64_t)wp * 4;   // (stride_w == 1)
    const int64_t row = ((int64_t)blockIdx.z * p.Hp + hp) * p.Wp + wp;   // rows are (level, batch, patch)
    float* out_row = reinterpret_cast<float*>(p.out) + row * p.Kpad;
    float* dst = out_row + p.k_offset + q * 4;
    // the K padding of a row is zeroed by the thread of its last piece, if this call ends the row
    const bool pads = q == q_per_row - 1 && p.k_offset + q_per_row * 4 == p.K_total;
    f32x4 z[PATCH4_GROUPS];
#pragma unroll
    for (int g = 0; g < PATCH4_GROUPS; ++g)
      if (wp + 8 * g < p.Wp) z[g] = *reinterpret_cast<const f32x4*>(src + 32 * g);
#pragma unroll
    for (int g = 0; g < PATCH4_GROUPS; ++g)
      if (wp + 8 * g < p.Wp) {
        const f32x4 r4 = f32x4{patch_transform((z[g].x - loc) * inv, d), patch_transform((z[g].y - loc) * inv, d),
                               patch_transform((z[g].z - loc) * inv, d), patch_transform((z[g].w - loc) * inv, d)};
        *reinterpret_cast<f32x4*>(dst + 8 * g * p.Kpad) = r4;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(r4.x), fabsf(r4.y)), fmaxf(fabsf(r4.z), fabsf(r4.w))));
        if (pads)
          for (int64_t k = p.K_total; k < p.Kpad; ++k) out_row[8 * g * p.Kpad + k] = 0.f;
      }
  }
  if (p.absmax) fold_absmax(p.absmax, m);
}

// Any patch size (10 at 0.1 degree, 3 for the air-pollution model): one thread per OUTPUT element, consecutive threads on
// consecutive columns k = (v, t, i, j) of one patch's row of the GEMM operand -- every store instruction of a wave writes
// 256 contiguous bytes; the loads are runs of P pixels per image row (the neighbouring patches, handled by the next rows'
// threads, use the rest of those cache lines out of L1 / L2).  The first form gave a thread one P-value piece and looped
// over it (one 4-byte load per lane and instruction, 64 cache lines each: 1.1 TB/s on the 0.1-degree grid); a
// pixel-per-thread form with coalesced loads and scattered 40-byte stores was no faster: partial-line WRITES are what hurts.
// An LDS-staged form (a workgroup loads span * P pixel runs of all planes of `span` adjacent patches coalesced, parks them in
// LDS, then writes each patch's columns as one contiguous piece through a column -> LDS-word table) measured SLOWER on both
// grids, 5.5 vs 4.4 ms at 0.1 degree and 2.2 vs 1.8 ms at 0.4 degree, with eight loads in flight per lane slower still: the
// two phases of a workgroup do not overlap and 48 KiB of staging leaves three workgroups per CU to hide them.
// What bounded that form was neither: ~200 VALU instructions per 4-byte element (five 32-bit divisions to decode the
// column, a wave reduction for the range word) -- 14 M wave passes of 200 x 4 cycles are the 4.4 ms.  A thread now decodes
// its column ONCE and walks PATCH_SPAN patches along the latitude circle with it (source += P pixels, destination += one
// row): ~12 instructions per element, PATCH_SPAN independent loads in flight, one range-word reduction per thread.
constexpr int PATCH_SPAN = 16;   // 16 patches x 10 pixels = 640 B of every image row: whole cache lines at 0.1 degree

template <typename T>
__global__ __launch_bounds__(256) void patchify_cols_kernel(const PatchArgs p, const int k_end, const int chunks) {
  // grid: x = (group of PATCH_SPAN patch columns, 256-column chunk of the row), y = patch row, z = (level, batch) -- 32-bit
  // index arithmetic only.  k runs over this call's columns [k_offset, k_end): its variables, plus the zero padding if the
  // call ends at K_total.
  const int n_k = k_end - p.k_offset;
  const unsigned g = blockIdx.x / (unsigned)chunks, chunk = blockIdx.x - g * (unsigned)chunks;
  const int kk = (int)(chunk * 256u + threadIdx.x);
  const unsigned wp0 = g * (unsigned)PATCH_SPAN;
  const int n = min(PATCH_SPAN, p.Wp - (int)wp0);
  float m = 0.f;
  if (kk < n_k) {
    const unsigned hp = blockIdx.y;
    const unsigned c = blockIdx.z / (unsigned)p.B, b = blockIdx.z - c * (unsigned)p.B;
    const int64_t row = ((int64_t)blockIdx.z * p.Hp + hp) * p.Wp + wp0;   // rows are (level, batch, patch)
    T* dst = reinterpret_cast<T*>(p.out) + row * p.Kpad + p.k_offset + kk;
    const unsigned PP = (unsigned)(p.P * p.P);
    const unsigned vt = (unsigned)kk / PP;
    if ((int)vt >= p.n_vars * p.T) {   // K padding
      for (int s = 0; s < n; ++s) elem<T>::store(dst + s * p.Kpad, 0.f);
    } else {
      const unsigned ij = (unsigned)kk - vt * PP;
      const unsigned i = ij / (unsigned)p.P, j = ij - i * (unsigned)p.P;
      const unsigned v = vt / (unsigned)p.T, t = vt - v * (unsigned)p.T;
      const PatchVar& d = p.v[v];
      const float* src = d.src + b * d.sb + t * d.st + c * d.sc + (int64_t)(hp * p.P + i) * d.sh + (int64_t)(wp0 * p.P + j) * d.sw;
      const int64_t step = (int64_t)p.P * d.sw;
      const float loc = d.loc[c], inv = d.inv_scale[c];
      float z[PATCH_SPAN];
#pragma unroll
      for (int s = 0; s < PATCH_SPAN; ++s)
        if (s < n) z[s] = src[s * step];
#pragma unroll
      for (int s = 0; s < PATCH_SPAN; ++s)
        if (s < n) {
          const float r = patch_transform((z[s] - loc) * inv, d);
          elem<T>::store(dst + s * p.Kpad, r);
          m = fmaxf(m, fabsf(r));
        }
    }
  }
  if (p.absmax) fold_absmax(p.absmax, m);
}

// ------------------------------------------------------------------------------------------------
// perceiver attention
// ------------------------------------------------------------------------------------------------
struct PercArgs {
  const void* q; int64_t q_col_stride; const void* kv; void* out;
  int B; int64_t cols_per_b, kv_bstride, kv_lstride; int Lq, Lk, heads;
  const float* pair_guard; float pair_limit;   // fp32 results leave as fp16 pairs iff *pair_guard < pair_limit (else fp32)
  const float* skip_guard; float skip_limit;   // the launch retires at once iff *skip_guard < skip_limit (null: never)
};

// A group of HDIM / 4 adjacent lanes owns one (grid column, head): each lane holds 4 of the head's features, so a
// group reads a key / value row as one contiguous run (16 lanes x 16 B = 256 B for head_dim 64) and the four groups
// of a wave -- four adjacent heads -- read 1 KiB contiguous per load instruction.  (One thread per (column, head,
// query) with the whole head in registers reads 16 B per lane at a 256-byte stride: 1.7 TB/s on kv streams that
// should run at HBM speed.)  The q.k dot products are reduced across the group with DPP adds (mirror, half-mirror,
// quad reverse, pair swap); the keys of a column are re-read per query out of L1 (Lk * 2 * HDIM * 4 B <= 7 KiB).
template <int LPG>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(LPG == 4 || LPG == 8 || LPG == 16 || LPG == 32, "lanes per group");
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  if constexpr (LPG == 32) v += __shfl_xor(v, 16, 64);
  if constexpr (LPG >= 16) v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror: i <-> 15 - i
  if constexpr (LPG >= 8) v += dpp(v, std::integral_constant<int, 0x141>{});    // row_half_mirror: i <-> 7 - i
  v += dpp(v, std::integral_constant<int, 0x1B>{});                              // quad_perm [3,2,1,0]
  v += dpp(v, std::integral_constant<int, 0xB1>{});                              // quad_perm [1,0,3,2]
  return v;
}

// QC: queries taken together, so that a column's keys / values are read once per chunk of QC queries, not once per query.
// Chosen from Lq at launch: 3 for the encoder's three latent queries (a chunk of 4 computed a fourth, discarded, query: a
// third more arithmetic), 7 for the decoder's thirteen level queries (7 + 6: two passes over the three keys instead of four,
// and 14 slots for 13 queries instead of 16), 4 otherwise.
// FEWK: at most four keys (the decoder's three latent levels): the column's keys and values are read ONCE, into
// registers, for all query chunks, and a query's softmax is taken over its (<= 4) scores at once -- one exponential per
// score and no running-maximum rescaling of the output (the online form: two exponentials and a rescale per key).
template <typename T, int HDIM, int QC, bool FEWK = false>
__global__ __launch_bounds__(256) void perceiver_attention_kernel(const PercArgs p) {
  constexpr int LPG = HDIM / 4;   // lanes per (column, head)
  if (p.skip_guard != nullptr && *p.skip_guard < p.skip_limit) return;   // (uniform) the re-associated pair does the work
  const int inner = p.heads * HDIM;
  const int64_t n_cols = (int64_t)p.B * p.cols_per_b;
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPG;
  const int d0 = (int)(threadIdx.x % LPG) * 4;
  if (grp >= n_cols * p.heads) return;   // (whole groups leave together)
  const int h = (int)(grp % p.heads);
  const int64_t col = grp / p.heads;
  const int b = (int)(col / p.cols_per_b);
  const int64_t l = col - (int64_t)b * p.cols_per_b;
  const float scale = rsqrtf((float)HDIM);
  const bool pairs = std::is_same<T, float>::value && p.pair_guard != nullptr && *p.pair_guard < p.pair_limit;   // (uniform)
  const T* kv0 = reinterpret_cast<const T*>(p.kv) + (b * p.kv_bstride + l) * (2 * (int64_t)inner) + h * HDIM + d0;
  const int64_t kv_step = p.kv_lstride * (2 * (int64_t)inner);
  float kk[FEWK ? 4 : 1][4], vv[FEWK ? 4 : 1][4];
  if constexpr (FEWK) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T* kp = kv0 + (j < p.Lk ? j : 0) * kv_step;   // (a slot past Lk repeats key 0; its weight is zero)
      load4(kp, kk[j]);
      load4(kp + inner, vv[j]);
    }
  }
  for (int i0 = 0; i0 < p.Lq; i0 += QC) {
    float qv[QC][4], o[QC][4], mx[QC], sum[QC];
#pragma unroll
    for (int c = 0; c < QC; ++c) {
      const int i = i0 + c < p.Lq ? i0 + c : p.Lq - 1;   // (a padded slot repeats the last query; not stored)
      load4(reinterpret_cast<const T*>(p.q) + (col * p.q_col_stride + i) * inner + h * HDIM + d0, qv[c]);
      o[c][0] = o[c][1] = o[c][2] = o[c][3] = 0.f;
      mx[c] = -INFINITY;
      sum[c] = 0.f;
    }
    if constexpr (FEWK) {
#pragma unroll
      for (int c = 0; c < QC; ++c) {
        float sc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = fmaf(qv[c][0], kk[j][0], fmaf(qv[c][1], kk[j][1], fmaf(qv[c][2], kk[j][2], qv[c][3] * kk[j][3])));
          sc[j] = j < p.Lk ? group_sum<LPG>(d) * scale : -INFINITY;
        }
        const float m = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = __expf(sc[j] - m);   // (exp(-inf) = 0 for the slots past Lk)
        sum[c] = (e[0] + e[1]) + (e[2] + e[3]);
#pragma unroll
        for (int d = 0; d < 4; ++d) o[c][d] = fmaf(e[0], vv[0][d], fmaf(e[1], vv[1][d], fmaf(e[2], vv[2][d], e[3] * vv[3][d])));
      }
    }
#pragma unroll 4   // (several keys' loads in flight; the online-softmax chain only orders the arithmetic)
    for (int j = 0; j < (FEWK ? 0 : p.Lk); ++j) {
      const T* kp = kv0 + j * kv_step;
      float k4[4], v4[4];
      load4(kp, k4);
      load4(kp + inner, v4);
#pragma unroll
      for (int c = 0; c < QC; ++c) {
        float sc = fmaf(qv[c][0], k4[0], fmaf(qv[c][1], k4[1], fmaf(qv[c][2], k4[2], qv[c][3] * k4[3])));
        sc = group_sum<LPG>(sc) * scale;
        const float nm = fmaxf(mx[c], sc);
        // (__expf = v_exp_f32 of the scaled argument: 2 instructions against expf's ~10; its argument error, 6e-8 |x|,
        // only matters for terms that are themselves ~e^-|x|)
        const float corr = __expf(mx[c] - nm), e = __expf(sc - nm);
        mx[c] = nm;
        sum[c] = sum[c] * corr + e;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[c][d] = fmaf(e, v4[d], o[c][d] * corr);
      }
    }
#pragma unroll
    for (int c = 0; c < QC; ++c) {
      if (i0 + c < p.Lq) {
        const float inv = 1.0f / sum[c];
        const float r4[4] = {o[c][0] * inv, o[c][1] * inv, o[c][2] * inv, o[c][3] * inv};
        if (pairs) {
          // the fp16-pair layout of the two-term GEMM that reads this next (to_out): lanes 2i, 2i+1 hold 8 consecutive
          // features between them and trade halves, the even lane stores the eight high halves, the odd one the remainders
          uint32_t h0, h1, l0, l1;
          split_pair_f16(r4[0], r4[1], h0, l0);
          split_pair_f16(r4[2], r4[3], h1, l1);
          const bool odd = (threadIdx.x & 1) != 0;
          auto swap1 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true); };
          const uint32_t s0 = swap1(odd ? h0 : l0), s1 = swap1(odd ? h1 : l1);
          const int f8 = (h * HDIM + d0) & ~7;
          char* d = reinterpret_cast<char*>(p.out) + ((col * p.Lq + i0 + c) * inner + (f8 & ~31)) * 4 + (f8 & 31) * 2 + (odd ? 64 : 0);
          *reinterpret_cast<u32x4*>(d) = odd ? u32x4{s0, s1, l0, l1} : u32x4{h0, h1, s0, s1};
        } else {
          store4(reinterpret_cast<T*>(p.out) + (col * p.Lq + i0 + c) * inner + h * HDIM + d0, r4);
        }
      }
    }
  }
}

// The same attention from PRE-MULTIPLIED scores (first Perceiver layer: its queries are model constants, so
//   q_l . (W_k x_j) = (W_k^T q_l) . x_j
// and the key half of to_kv shrinks from `inner` columns to Lq * heads -- the rows W_k^T q_l / sqrt(head_dim) are made when the
// weights are packed, model.hip:score_weights).  A context row of `vs` holds [v (inner) | ... scores at s_off: (query l, head h)
// at l * heads + h].  Lane layout as above; a group's scores are the same address in all its lanes (one broadcast load), the
// softmax needs no cross-lane traffic at all: two passes over a query's Lk scores (maximum, then exponentials), one over
// the values.  QC queries share a pass over the column's values.
struct PercScoreArgs {
  const float* vs; int64_t ld; int s_off; void* out;
  int B; int64_t cols_per_b, kv_bstride, kv_lstride; int Lq, Lk, heads;
  const float* pair_guard; float pair_limit;
  const float* skip_guard; float skip_limit;
};

template <int HDIM, int QC>
__global__ __launch_bounds__(256) void perceiver_attention_scores_kernel(const PercScoreArgs p) {
  constexpr int LPG = HDIM / 4;
  if (p.skip_guard != nullptr && *p.skip_guard < p.skip_limit) return;   // (uniform)
  const int inner = p.heads * HDIM;
  const int64_t n_cols = (int64_t)p.B * p.cols_per_b;
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPG;
  const int d0 = (int)(threadIdx.x % LPG) * 4;
  if (grp >= n_cols * p.heads) return;   // (whole groups leave together)
  const int h = (int)(grp % p.heads);
  const int64_t col = grp / p.heads;
  const int b = (int)(col / p.cols_per_b);
  const int64_t l = col - (int64_t)b * p.cols_per_b;
  const bool pairs = p.pair_guard != nullptr && *p.pair_guard < p.pair_limit;   // (uniform)
  const float* row0 = p.vs + (b * p.kv_bstride + l) * p.ld;
  const int64_t step = p.kv_lstride * p.ld;
  for (int i0 = 0; i0 < p.Lq; i0 += QC) {
    const float* sc[QC];
    float mx[QC], sum[QC], o[QC][4];
#pragma unroll
    for (int c = 0; c < QC; ++c) {
      const int i = i0 + c < p.Lq ? i0 + c : p.Lq - 1;   // (a padded slot repeats the last query; not stored)
      sc[c] = row0 + p.s_off + i * p.heads + h;
      mx[c] = -INFINITY;
      sum[c] = 0.f;
      o[c][0] = o[c][1] = o[c][2] = o[c][3] = 0.f;
    }
#pragma unroll 4
    for (int j = 0; j < p.Lk; ++j)
#pragma unroll
      for (int c = 0; c < QC; ++c) mx[c] = fmaxf(mx[c], sc[c][j * step]);
#pragma unroll 4
    for (int j = 0; j < p.Lk; ++j) {
      float v4[4];
      load4(row0 + j * step + h * HDIM + d0, v4);
#pragma unroll
      for (int c = 0; c < QC; ++c) {
        const float e = __expf(sc[c][j * step] - mx[c]);
        sum[c] += e;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[c][d] = fmaf(e, v4[d], o[c][d]);
      }
    }
#pragma unroll
    for (int c = 0; c < QC; ++c) {
      if (i0 + c < p.Lq) {
        const float inv = 1.0f / sum[c];
        const float r4[4] = {o[c][0] * inv, o[c][1] * inv, o[c][2] * inv, o[c][3] * inv};
        if (pairs) {   // the fp16-pair layout, as perceiver_attention_kernel writes it
          uint32_t h0, h1, l0, l1;
          split_pair_f16(r4[0], r4[1], h0, l0);
          split_pair_f16(r4[2], r4[3], h1, l1);
          const bool odd = (threadIdx.x & 1) != 0;
          auto swap1 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true); };
          const uint32_t s0 = swap1(odd ? h0 : l0), s1 = swap1(odd ? h1 : l1);
          const int f8 = (h * HDIM + d0) & ~7;
          char* d = reinterpret_cast<char*>(p.out) + ((col * p.Lq + i0 + c) * inner + (f8 & ~31)) * 4 + (f8 & 31) * 2 + (odd ? 64 : 0);
          *reinterpret_cast<u32x4*>(d) = odd ? u32x4{s0, s1, l0, l1} : u32x4{h0, h1, s0, s1};
        } else {
          store4(reinterpret_cast<float*>(p.out) + (col * p.Lq + i0 + c) * inner + h * HDIM + d0, r4);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// assemble tokens
// ------------------------------------------------------------------------------------------------
struct AsmArgs {
  const float* surf; const float* agg; const float* pos_scale; const float* time_emb;
  float* out_f32; void* out_t; int B, Cl; int64_t L; int D;
};

template <typename T>
__global__ __launch_bounds__(256) void assemble_kernel(const AsmArgs p) {
  const int pieces = p.D >> 3;
  const int64_t total = (int64_t)p.B * p.Cl * p.L * pieces;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += stride) {
    const int e = (int)(it % pieces) * 8;
    const int64_t tokn = it / pieces;        // (b, c, l)
    const int64_t l = tokn % p.L;
    const int c = (int)((tokn / p.L) % p.Cl);
    const int b = (int)(tokn / (p.L * p.Cl));
    float v[8], a[8];
    if (c == 0) load8(p.surf + ((int64_t)b * p.L + l) * p.D + e, v);
    else load8(p.agg + (((int64_t)b * p.L + l) * (p.Cl - 1) + (c - 1)) * p.D + e, v);
    load8(p.pos_scale + l * p.D + e, a);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += a[j];
    load8(p.time_emb + (int64_t)b * p.D + e, a);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += a[j];
    store8(p.out_f32 + tokn * p.D + e, v);
    if (p.out_t) store8(reinterpret_cast<T*>(p.out_t) + tokn * p.D + e, v);
  }
}

// ------------------------------------------------------------------------------------------------
// unpatchify
// ------------------------------------------------------------------------------------------------
struct UnpatchVar {
  float* dst; const float* loc; const float* scale; int32_t clamp_min0, col0, lvl_stride, mod_col0;
  const float* prev; int64_t prev_sb, prev_sc, prev_sh; const float* inv_scale; uint32_t clamp_max1_levels;
  int32_t angle_col0, dens_col0; const float* mask; int64_t mask_sh; float mask_thresh;
};
struct UnpatchArgs {
  UnpatchVar v[MAX_VARS];
  const float* y; int64_t ldy; int n_vars, B, n_lvl, Hp, Wp, P;
  int vec4;   // P == 4 and every source / destination piece is 16-byte aligned
};

// VEC = 4: patch size 4, every piece 16-byte aligned.  A workgroup owns a run of 8 * UNPATCH_GROUPS patches of one patch
// row at one level; lane = (piece q = (variable, image row i) of the head row, patch s of 8): the eight lanes of a q write
// 128 contiguous bytes of one image row of one field, the eight q of an s read 128 contiguous bytes of one head row --
// whole cache lines on both sides.  (One thread per piece in (variable, ..., patch) order read a 16-byte piece per line and
// instruction and came back for the rest of each line a variable later: the head outputs crossed the fabric up to eight
// times, 0.38 ms per step where the bytes take 0.1, and six 64-bit divisions per thread did not help.)
// VEC = 1: any patch size: a thread owns one pixel COLUMN of a patch row -- consecutive threads on consecutive longitudes,
// so that every store of a wave writes 256 contiguous bytes of the output field -- and walks the P image rows of that
// patch row with it (head outputs += P, field += one image row): its position is decoded once per P pixels, and the
// P x P head outputs of a patch, one contiguous run, are used up by the same wave while they sit in L1.
constexpr int UNPATCH_GROUPS = 4;

template <int VEC>
__global__ __launch_bounds__(VEC == 4 ? 1024 : 256) void unpatchify_kernel(const UnpatchArgs p) {
  int v, wp, j0, i, hp, c, b;
  if constexpr (VEC == 4) {
    // grid: x = run of 8 * UNPATCH_GROUPS patches, y = patch row, z = (batch, level)
    const int q = (int)(threadIdx.x >> 3);
    if (q >= p.n_vars * 4) return;
    v = q >> 2; i = q & 3; j0 = 0;
    wp = (int)(blockIdx.x * (8u * UNPATCH_GROUPS) + (threadIdx.x & 7u));
    hp = (int)blockIdx.y;
    b = (int)(blockIdx.z / (unsigned)p.n_lvl); c = (int)(blockIdx.z - (unsigned)b * (unsigned)p.n_lvl);
  } else {
    // grid: x = 256-pixel chunk of an image row, y = patch row, z = (variable, batch, level): 32-bit index arithmetic only
    const unsigned x = blockIdx.x * 256u + threadIdx.x;
    if (x >= (unsigned)(p.Wp * p.P)) return;
    wp = (int)(x / (unsigned)p.P); j0 = (int)(x - (unsigned)wp * (unsigned)p.P);
    hp = (int)blockIdx.y; i = 0;
    const unsigned vb = blockIdx.z / (unsigned)p.n_lvl;
    c = (int)(blockIdx.z - vb * (unsigned)p.n_lvl);
    v = (int)(vb / (unsigned)p.B); b = (int)(vb - (unsigned)v * (unsigned)p.B);
  }
  const UnpatchVar& d = p.v[v];
  const int64_t L = (int64_t)p.Hp * p.Wp;
  const int64_t W = (int64_t)p.Wp * p.P, H = (int64_t)p.Hp * p.P;
  const float loc = d.loc[c], sc = d.scale[c];
  const bool has_mod = d.mod_col0 >= 0;
  const float inv = has_mod ? d.inv_scale[c] : 0.f;
  const bool cap1 = (d.clamp_max1_levels >> c) & 1u;
  // everything of patch column `wp` (image rows hp * P + i ...)
  auto patch = [&](int wp) {
    const float* row = p.y + (((int64_t)b * L + (int64_t)hp * p.Wp + wp) * p.n_lvl + c) * p.ldy + c * d.lvl_stride + i * p.P;
    const float* src = row + d.col0;
    const int64_t hh = (int64_t)hp * p.P + i, ww = (int64_t)wp * p.P;
    float* dst = d.dst + (((int64_t)b * p.n_lvl + c) * H + hh) * W + ww;
    const float* mod = row + (has_mod ? d.mod_col0 : 0);
    const float* prev = has_mod ? d.prev + b * d.prev_sb + c * d.prev_sc + hh * d.prev_sh + ww : nullptr;
    const float* cosp = d.angle_col0 >= 0 ? row + d.angle_col0 : nullptr;   // wave: col0 = sin head, this = cos head
    const float* dens = d.dens_col0 >= 0 ? row + d.dens_col0 : nullptr;     // wave: density head
    const float* maskp = dens ? d.mask + hh * d.mask_sh + ww : nullptr;     // raw water-body mask plane
    // j indexes the head outputs; the fields' own rows are prev_sh / mask_sh apart, not P (VEC = 1 walks image rows)
    auto finish = [&](float z, int j, int64_t prev_j, int64_t mask_j) -> float {
      if (has_mod) z = z + (1.0f + mod[j]) * ((prev[prev_j] - loc) * inv);
      if (cap1) z = fminf(z, 1.0f);
      if (d.clamp_min0) z = fmaxf(z, 0.f);
      if (cosp) {  // direction from its sin / cos heads: rad2deg(atan2(sin, cos)) mod 360 in [0, 360)
        z = atan2f(z, cosp[j]) * 57.29577951308232f;
        z = fmodf(z, 360.0f);
        if (z < 0.f) z += 360.0f;
      }
      if (dens) {  // present only over water AND where sigmoid(density) >= 0.5 (aurora.py:924-930)
        const bool water = maskp[mask_j] > d.mask_thresh;
        z = (water && !(dens[j] < 0.f)) ? z : __int_as_float(0x7fc00000);
      }
      return z * sc + loc;
    };
    if constexpr (VEC == 4) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(src);
      *reinterpret_cast<f32x4*>(dst) = f32x4{finish(s4.x, 0, 0, 0), finish(s4.y, 1, 1, 1), finish(s4.z, 2, 2, 2), finish(s4.w, 3, 3, 3)};
    } else {
      for (int ii = 0; ii < p.P; ++ii) {
        const int j = ii * p.P + j0;
        dst[ii * W + j0] = finish(src[j], j, has_mod ? ii * d.prev_sh + j0 : 0, dens ? ii * d.mask_sh + j0 : 0);
      }
    }
  };
  if constexpr (VEC == 4) {
#pragma unroll
    for (int g = 0; g < UNPATCH_GROUPS; ++g)
      if (wp + 8 * g < p.Wp) patch(wp + 8 * g);
  } else {
    patch(wp);
  }
}

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace
}  // namespace aurora

using namespace aurora;

extern "C" int aurora_hip_patchify(const aurora_patch_var* desc, int n_vars, void* out, int64_t Kpad,
                                   int k_offset, int K_total, int B, int T, int n_lvl, int Hp, int Wp, int P,
                                   int dtype, void* stream) {
  return aurora_hip_patchify_absmax(desc, n_vars, out, Kpad, k_offset, K_total, B, T, n_lvl, Hp, Wp, P, dtype, nullptr, stream);
}

extern "C" int aurora_hip_patchify_absmax(const aurora_patch_var* desc, int n_vars, void* out, int64_t Kpad,
                                          int k_offset, int K_total, int B, int T, int n_lvl, int Hp, int Wp, int P,
                                          int dtype, float* absmax, void* stream) {
  AURORA_CHECK_ARG(dtype == AURORA_F32 || dtype == AURORA_BF16, "patchify: bad dtype");
  AURORA_CHECK_ARG(n_vars > 0 && n_vars <= MAX_VARS, "patchify: %d variables per call (max %d)", n_vars, MAX_VARS);
  AURORA_CHECK_ARG(k_offset >= 0 && k_offset + n_vars * T * P * P <= K_total && K_total <= Kpad,
                   "patchify: column range does not fit (k_offset=%d K_total=%d Kpad=%lld)", k_offset, K_total,
                   (long long)Kpad);
  PatchArgs p;
  for (int v = 0; v < n_vars; ++v) {
    const aurora_patch_var& s = desc[v];
    p.v[v] = PatchVar{s.src, s.stride_b, s.stride_t, s.stride_c, s.stride_h, s.stride_w,
                      s.loc, s.inv_scale, s.transform, s.tw0, s.tw1, s.tb};
  }
  p.out = out; p.Kpad = Kpad; p.k_offset = k_offset; p.K_total = K_total;
  p.n_vars = n_vars; p.B = B; p.T = T; p.n_lvl = n_lvl; p.Hp = Hp; p.Wp = Wp; p.P = P;
  p.absmax = absmax;
  AURORA_CHECK_ARG(B > 0 && T > 0 && n_lvl > 0 && Hp > 0 && Wp > 0 && P > 0, "patchify: bad problem size");
  bool vec4 = P == 4 && dtype == AURORA_F32 && (uintptr_t)out % 16 == 0 && Kpad % 4 == 0 && k_offset % 4 == 0 &&
              n_vars * T * 32 <= 1024 && Hp <= 65535 && (int64_t)n_lvl * B <= 65535;
  for (int v = 0; v < n_vars && vec4; ++v) {
    const aurora_patch_var& s = desc[v];
    vec4 = s.stride_w == 1 && (uintptr_t)s.src % 16 == 0 && s.stride_b % 4 == 0 && s.stride_t % 4 == 0 && s.stride_c % 4 == 0 &&
           s.stride_h % 4 == 0;
  }
  if (vec4) {
    const int run = 8 * PATCH4_GROUPS;
    const dim3 grid((unsigned)((Wp + run - 1) / run), (unsigned)Hp, (unsigned)(n_lvl * B));
    hipLaunchKernelGGL(patchify_kernel, grid, dim3((unsigned)((n_vars * T * 32 + 63) / 64 * 64)), 0, as_stream(stream), p);
  } else {
    const int k_end = k_offset + n_vars * T * P * P == K_total ? (int)Kpad : k_offset + n_vars * T * P * P;
    const int chunks = (k_end - k_offset + 255) / 256;
    const int groups = (Wp + PATCH_SPAN - 1) / PATCH_SPAN;
    AURORA_CHECK_ARG((int64_t)groups * chunks < ((int64_t)1 << 31) && Hp <= 65535 && (int64_t)n_lvl * B <= 65535,
                     "patchify: grid too large (%d x %d patches, %d levels x %d)", Hp, Wp, n_lvl, B);
    const dim3 grid((unsigned)(groups * chunks), (unsigned)Hp, (unsigned)(n_lvl * B));
    if (dtype == AURORA_F32)
      hipLaunchKernelGGL(patchify_cols_kernel<float>, grid, dim3(256), 0, as_stream(stream), p, k_end, chunks);
    else
      hipLaunchKernelGGL(patchify_cols_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), p, k_end, chunks);
  }
  return check_launch("patchify");
}

extern "C" int aurora_hip_perceiver_attention(const void* q, int64_t q_col_stride, const void* kv, void* out,
                                              int B, int64_t cols_per_b, int64_t kv_bstride, int64_t kv_lstride,
                                              int Lq, int Lk, int heads, int head_dim, int dtype, void* stream) {
  return aurora_hip_perceiver_attention_ex(q, q_col_stride, kv, out, B, cols_per_b, kv_bstride, kv_lstride, Lq, Lk, heads,
                                           head_dim, dtype, nullptr, 0.f, stream);
}

extern "C" int aurora_hip_perceiver_attention_ex(const void* q, int64_t q_col_stride, const void* kv, void* out,
                                                 int B, int64_t cols_per_b, int64_t kv_bstride, int64_t kv_lstride,
                                                 int Lq, int Lk, int heads, int head_dim, int dtype,
                                                 const float* pair_guard, float pair_limit, void* stream) {
  return aurora_hip_perceiver_attention_unless(q, q_col_stride, kv, out, B, cols_per_b, kv_bstride, kv_lstride, Lq, Lk, heads,
                                               head_dim, dtype, pair_guard, pair_limit, nullptr, 0.f, stream);
}

extern "C" int aurora_hip_perceiver_attention_unless(const void* q, int64_t q_col_stride, const void* kv, void* out,
                                                     int B, int64_t cols_per_b, int64_t kv_bstride, int64_t kv_lstride,
                                                     int Lq, int Lk, int heads, int head_dim, int dtype,
                                                     const float* pair_guard, float pair_limit, const float* skip_guard,
                                                     float skip_limit, void* stream) {
  AURORA_CHECK_ARG(dtype == AURORA_F32 || dtype == AURORA_BF16, "perceiver_attention: bad dtype");
  AURORA_CHECK_ARG(Lq > 0 && Lk > 0 && heads > 0 && B > 0 && cols_per_b > 0, "perceiver_attention: empty problem");
  AURORA_CHECK_ARG(pair_guard == nullptr || (dtype == AURORA_F32 && (heads * head_dim) % 32 == 0 && head_dim % 8 == 0),
                   "perceiver_attention: fp16-pair output needs fp32 and heads * head_dim %% 32 == 0");
  PercArgs p{q, q_col_stride, kv, out, B, cols_per_b, kv_bstride, kv_lstride, Lq, Lk, heads, pair_guard, pair_limit, skip_guard,
             skip_limit};
  const int64_t items = (int64_t)B * cols_per_b * heads * (head_dim / 4);   // one lane per 4 features of a head
  const dim3 grid(blocks_for(items, 256)), block(256);
#define AURORA_PERC(TT, HDIM)                                                                                              \
  do {                                                                                                                   \
    if (Lq % 3 == 0 && Lq <= 6) hipLaunchKernelGGL((perceiver_attention_kernel<TT, HDIM, 3>), grid, block, 0, as_stream(stream), p); \
    else if (Lq > 8 && HDIM <= 64 && Lk <= 4) hipLaunchKernelGGL((perceiver_attention_kernel<TT, HDIM, 7, true>), grid, block, 0, as_stream(stream), p); \
    else if (Lq > 8 && HDIM <= 64) hipLaunchKernelGGL((perceiver_attention_kernel<TT, HDIM, 7>), grid, block, 0, as_stream(stream), p); \
    else hipLaunchKernelGGL((perceiver_attention_kernel<TT, HDIM, 4>), grid, block, 0, as_stream(stream), p);              \
  } while (0)
  if (dtype == AURORA_F32) {
    switch (head_dim) {
      case 16: AURORA_PERC(float, 16); break;
      case 32: AURORA_PERC(float, 32); break;
      case 64: AURORA_PERC(float, 64); break;
      case 128: AURORA_PERC(float, 128); break;
      default: AURORA_CHECK_ARG(false, "perceiver_attention: head_dim %d not in {16,32,64,128}", head_dim);
    }
  } else {
    switch (head_dim) {
      case 16: AURORA_PERC(bf16_t, 16); break;
      case 32: AURORA_PERC(bf16_t, 32); break;
      case 64: AURORA_PERC(bf16_t, 64); break;
      case 128: AURORA_PERC(bf16_t, 128); break;
      default: AURORA_CHECK_ARG(false, "perceiver_attention: head_dim %d not in {16,32,64,128}", head_dim);
    }
  }
#undef AURORA_PERC
  return check_launch("perceiver_attention");
}

extern "C" int aurora_hip_perceiver_attention_scores(const float* vs, int64_t ld_vs, int s_off, void* out, int B,
                                                     int64_t cols_per_b, int64_t kv_bstride, int64_t kv_lstride, int Lq, int Lk,
                                                     int heads, int head_dim, const float* pair_guard, float pair_limit,
                                                     const float* skip_guard, float skip_limit, void* stream) {
  AURORA_CHECK_ARG(vs && out && Lq > 0 && Lk > 0 && heads > 0 && B > 0 && cols_per_b > 0, "perceiver_attention_scores: empty problem");
  const int inner = heads * head_dim;
  AURORA_CHECK_ARG(s_off >= inner && ld_vs >= (int64_t)s_off + (int64_t)Lq * heads && ld_vs % 4 == 0 && ((uintptr_t)vs % 16) == 0,
                   "perceiver_attention_scores: a row is [v (heads * head_dim) | ... | scores (Lq * heads) at s_off], ld %% 4 == 0");
  AURORA_CHECK_ARG(pair_guard == nullptr || (inner % 32 == 0 && head_dim % 8 == 0),
                   "perceiver_attention_scores: fp16-pair output needs heads * head_dim %% 32 == 0");
  PercScoreArgs p{vs, ld_vs, s_off, out, B, cols_per_b, kv_bstride, kv_lstride, Lq, Lk, heads, pair_guard, pair_limit, skip_guard,
                  skip_limit};
  const int64_t items = (int64_t)B * cols_per_b * heads * (head_dim / 4);
  const dim3 grid(blocks_for(items, 256)), block(256);
#define AURORA_PERC_S(HDIM)                                                                                                  \
  do {                                                                                                                       \
    if (Lq % 3 == 0 && Lq <= 6) hipLaunchKernelGGL((perceiver_attention_scores_kernel<HDIM, 3>), grid, block, 0, as_stream(stream), p); \
    else if (Lq > 8) hipLaunchKernelGGL((perceiver_attention_scores_kernel<HDIM, 7>), grid, block, 0, as_stream(stream), p);    \
    else hipLaunchKernelGGL((perceiver_attention_scores_kernel<HDIM, 4>), grid, block, 0, as_stream(stream), p);                \
  } while (0)
  switch (head_dim) {
    case 16: AURORA_PERC_S(16); break;
    case 32: AURORA_PERC_S(32); break;
    case 64: AURORA_PERC_S(64); break;
    case 128: AURORA_PERC_S(128); break;
    default: AURORA_CHECK_ARG(false, "perceiver_attention_scores: head_dim %d not in {16,32,64,128}", head_dim);
  }
#undef AURORA_PERC_S
  return check_launch("perceiver_attention_scores");
}

extern "C" int aurora_hip_assemble_tokens(const float* surf, const float* agg, const float* pos_scale,
                                          const float* time_emb, float* out_f32, void* out_t, int B, int Cl,
                                          int64_t L, int D, int dtype, void* stream) {
  AURORA_CHECK_ARG(D % 8 == 0 && Cl >= 2, "assemble_tokens: D=%d must be a multiple of 8, Cl=%d >= 2", D, Cl);
  AsmArgs p{surf, agg, pos_scale, time_emb, out_f32, out_t, B, Cl, L, D};
  const int64_t total = (int64_t)B * Cl * L * (D / 8);
  const unsigned blocks = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (dtype == AURORA_F32) {
    p.out_t = nullptr;  // the fp32 stream is its own GEMM operand
    hipLaunchKernelGGL(assemble_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), p);
  } else {
    hipLaunchKernelGGL(assemble_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), p);
  }
  return check_launch("assemble_tokens");
}

extern "C" int aurora_hip_unpatchify(const float* y, int64_t ldy, const aurora_unpatch_var* desc, int n_vars,
                                     int B, int n_lvl, int Hp, int Wp, int P, void* stream) {
  AURORA_CHECK_ARG(n_vars > 0 && n_vars <= MAX_VARS, "unpatchify: %d variables per call (max %d)", n_vars, MAX_VARS);
  UnpatchArgs p;
  for (int v = 0; v < n_vars; ++v)
    p.v[v] = UnpatchVar{desc[v].dst, desc[v].loc, desc[v].scale, desc[v].clamp_min0, desc[v].col0,
                        desc[v].lvl_stride, desc[v].mod_col0, desc[v].prev, desc[v].prev_sb, desc[v].prev_sc,
                        desc[v].prev_sh, desc[v].inv_scale, desc[v].clamp_max1_levels, desc[v].angle_col0,
                        desc[v].dens_col0, desc[v].mask, desc[v].mask_sh, desc[v].mask_thresh};
  p.y = y; p.ldy = ldy; p.n_vars = n_vars; p.B = B; p.n_lvl = n_lvl; p.Hp = Hp; p.Wp = Wp; p.P = P;
  bool al = P == 4 && (uintptr_t)y % 16 == 0 && ldy % 4 == 0;
  for (int v = 0; v < n_vars && al; ++v)
    al = (uintptr_t)desc[v].dst % 16 == 0 && desc[v].col0 % 4 == 0 && desc[v].lvl_stride % 4 == 0;
  p.vec4 = al ? 1 : 0;
  for (int v = 0; v < n_vars && al; ++v)   // (the optional head columns and planes of the post-decoder hooks as well)
    al = (desc[v].mod_col0 < 0 || (desc[v].mod_col0 % 4 == 0 && (uintptr_t)desc[v].prev % 16 == 0 && desc[v].prev_sb % 4 == 0 &&
                                   desc[v].prev_sc % 4 == 0 && desc[v].prev_sh % 4 == 0)) &&
         (desc[v].angle_col0 < 0 || desc[v].angle_col0 % 4 == 0) &&
         (desc[v].dens_col0 < 0 || (desc[v].dens_col0 % 4 == 0 && (uintptr_t)desc[v].mask % 16 == 0 && desc[v].mask_sh % 4 == 0));
  p.vec4 = al ? 1 : 0;
  AURORA_CHECK_ARG(B > 0 && n_lvl > 0 && Hp > 0 && Wp > 0 && P > 0, "unpatchify: bad problem size");
  if (al) {
    AURORA_CHECK_ARG(Hp <= 65535 && (int64_t)B * n_lvl <= 65535, "unpatchify: grid too large");
    const int run = 8 * UNPATCH_GROUPS;
    const dim3 grid((unsigned)((Wp + run - 1) / run), (unsigned)Hp, (unsigned)(B * n_lvl));
    hipLaunchKernelGGL(unpatchify_kernel<4>, grid, dim3((unsigned)((n_vars * 32 + 63) / 64 * 64)), 0, as_stream(stream), p);
  } else {
    AURORA_CHECK_ARG(Hp <= 65535 && (int64_t)n_vars * B * n_lvl <= 65535, "unpatchify: grid too large");
    const dim3 grid((unsigned)((Wp * P + 255) / 256), (unsigned)Hp, (unsigned)(n_vars * B * n_lvl));
    hipLaunchKernelGGL(unpatchify_kernel<1>, grid, dim3(256), 0, as_stream(stream), p);
  }
  return check_launch("unpatchify");
}
