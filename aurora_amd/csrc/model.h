// The model handle of libaurora_hip.so: shared definitions of model.hip (creation, weights, tables) and step.hip (the
// launch sequence of one forecast step).  Host code only; every launch goes through the operator ABI of this library.
#pragma once

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <array>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "band.h"
#include "common.h"

namespace aurora {

struct Fail {
  int code;
};
#define REQUIRE(cond, ...)               \
  do {                                   \
    if (!(cond)) {                       \
      ::aurora::set_error(__VA_ARGS__);  \
      throw ::aurora::Fail{AURORA_E_ARG}; \
    }                                    \
  } while (0)
inline void ok(int code) {
  if (code != AURORA_OK) throw Fail{code};
}
inline void hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    throw Fail{AURORA_E_LAUNCH};
  }
}
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- device memory -------------------------------------------------------------------------------
struct DevBuf {   // owning, persistent
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  explicit DevBuf(size_t n) : bytes(n) { hip_ok(hipMalloc(&p, n ? n : 16), "hipMalloc"); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      if (p) (void)hipFree(p);
      p = o.p; bytes = o.bytes; o.p = nullptr;
    }
    return *this;
  }
  ~DevBuf() { if (p) (void)hipFree(p); }
  float* f() const { return static_cast<float*>(p); }
};

// Workspace of a step: one slab, stack discipline (mark / release), so that the 48 blocks re-use the same few GB.
// A step is first walked in `dry` mode (no launches) to learn its peak, the slab grows if needed, then it runs.
struct Arena {
  char* base = nullptr;
  size_t cap = 0, top = 0, peak = 0;
  void* take(size_t bytes) {
    const size_t at = (top + 255) & ~size_t(255);
    top = at + bytes;
    if (top > peak) peak = top;
    return base + at;   // (dry runs hand out addresses that are never dereferenced)
  }
  ~Arena() { if (base) (void)hipFree(base); }
};

struct Tensor {
  DevBuf buf;
  std::vector<int64_t> shape;
  int64_t numel = 0;
  float* f() const { return buf.f(); }
};

// ---- the model -----------------------------------------------------------------------------------
struct Block {
  std::string prefix;
  int dim, stage, heads, hidden;
  bool shifted;
  const float *gain1, *shift1, *gain2, *shift2;     // AdaLN modulation (slices of `mod`)
  const float *qkv_b, *proj_b, *fc1_b, *fc2_b;
  const void *fc1_w, *fc2_w;                          // compute dtype
};
struct AttnSet { std::vector<DevBuf> own; std::vector<const void*> qkv, proj; };   // per block, compute dtype
struct Resampler {
  struct Layer {
    const float *to_q, *to_kv, *to_out, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    const float *ln_k_w = nullptr, *ln_k_b = nullptr, *ln_q_w = nullptr, *ln_q_b = nullptr;
    int inner, head_dim, hidden, dim, ctx_dim;
    float v_l1;
    int f16_mode;   // fp32 GEMM mode of this layer's bounded linears: 2 (two fp16 terms) when weights / LN bounds allow
    // the same weights in the fp16-pair layout, scaled by 2^6 (null where mode or shape rule it out): the two-term GEMMs
    // then spend no VALU work on the weight operand, and none at all where the activations arrive split as well
    const void *to_kv_s = nullptr, *to_out_s = nullptr, *fc1_s = nullptr, *fc2_s = nullptr;
  };
  std::vector<Layer> layers;
  std::vector<DevBuf> own;
  // First layer, queries known when the weights are packed (model.hip:score_weights): to_kv replaced by
  //   [W_v (inner rows) | W_k^T q_l / sqrt(head_dim) per (query l, head h) (n_s = Lq * heads rows) | zero rows up to n_vs]
  // -- a context row then carries its values and its SCORES with every query; the key projection does not exist.
  DevBuf vs_w, vs_ws;      // fp32, and the fp16-pair layout scaled by 2^6 (empty: not eligible)
  int n_s = 0, n_vs = 0, vs_lq = 0;
};

// One input channel of a patch embedding: where its pixels come from and how they are transformed (embed.hip patchify).
enum SrcKind { SRC_SURF, SRC_STATIC, SRC_DYN, SRC_ATMOS };
struct Channel {
  std::string name;        // key of the patch-embedding weight (`encoder.*_token_embeds.weights.<name>`)
  SrcKind kind;
  int src;                 // index into the step's surf / static / atmos pointer lists, or the dynamic plane 0..5
  int transform;           // aurora_hip_patchify transform code
  float tw0 = 0.f, tw1 = 0.f, tb = 0.f;
};
// A LevelPatchEmbed weight as GEMM operand for one set of present channels and T history steps.
struct EmbedPack {
  DevBuf w, ws;            // [groups][D][Kpad] fp32, and the same in the fp16-pair layout (scaled by 2^6) when eligible
  int K = 0, Kpad = 0, groups = 1;   // groups: 1, or one weight per pressure level (level-conditioned, levelcond.py)
  float l1 = 0.f;          // largest L1 row norm: |embedding| <= l1 * max|input| + |bias|
  std::vector<int> channels;   // indices into the channel list, in K order
};
// A surface / atmospheric output variable and the head columns it is decoded from (embed.hip unpatchify).
struct HeadGroup {
  std::vector<std::string> names;      // head names in column order (variables, then `<v>_mod`)
  DevBuf w, b;                         // [groups][n * P * P][2D], [groups][round_up(n * P * P, 4)]
  int groups = 1;
  // the same heads for the two-term fp16 GEMM (when the weights allow): rows zero-padded to a multiple of 128 -- the narrow tile
  // width of linear_kernel_f32pp --, weights in the fp16-pair layout scaled by 2^6, bias rows padded likewise
  DevBuf ws, bs;                       // [groups][n_pad][2D] pairs, [groups][n_pad]
  int n_pad = 0;                       // 0: not available
};

struct DevTables { DevBuf tok, grp; int n_windows = 0, n_tok = 0; bool has_grp = false; };
// The attention plan of one block flavour of a band, on the device (band.h BandPlan).
struct DevPlan {
  DevBuf tok, grp, send_idx;   // send_idx: rows for the previous rank, then those for the next
  int n_windows = 0, n_tok = 0, n_own = 0, n_halo = 0, n_interior = 0;
  int recv_off[2] = {0, 0}, recv_cnt[2] = {0, 0}, send_cnt[2] = {0, 0};
  bool has_grp = false;
};

}  // namespace aurora

struct aurora_hip_model {
  // configuration
  int D = 0, P = 0, Cl = 0, perceiver_heads = 0, n_stages = 0;
  int enc_depths[4] = {0}, dec_depths[4] = {0}, enc_heads[4] = {0}, dec_heads[4] = {0}, window[3] = {0};
  int enc_depth = 1, dec_depth = 1, max_history = 2, lora_steps = 40, lora_mode = 0;
  bool stabilise = false, use_lora = false, autocast = false;
  float ln_eps = 1e-5f;
  double timestep_hours = 6;
  std::vector<std::string> surf_vars, static_vars, atmos_vars;
  // variants (aurora.py:86-95, 726-796, 854-932)
  int variant = 0;                                  // 0 base, 1 air pollution, 2 ocean wave
  bool dynamic_vars = false, atmos_static_vars = false, clamp_first = false, index_bug = false;
  std::vector<double> level_condition;
  std::vector<std::string> sep_perceiver, mod_heads, pos_surf, pos_atmos, surf_inputs, density_vars, angle_vars;
  std::map<std::string, int> diff_index;            // variable -> history index its predicted difference refers to
  // derived at creation: input channels of the two patch embeddings, output variables and their heads
  std::vector<aurora::Channel> surf_channels, atmos_channels;
  std::vector<std::string> surf_heads, surf_out;    // decoder head names (columns) and predicted surface variables
  std::vector<std::string> atmos_heads;             // atmos_vars + `<v>_mod`

  // weights
  std::map<std::string, aurora::Tensor> w;          // fp32 masters (aurora_hip_pack_weights / a packed file)
  std::map<std::string, aurora::Tensor> w16;        // bf16-only entries of a packed file (shape kept, data bf16)
  bool finalized = false;
  std::vector<aurora::Block> blocks;
  aurora::DevBuf mod, lead_emb, enc_q0;
  std::vector<aurora::DevBuf> keep;                 // bf16 copies and other derived device arrays
  std::map<int, aurora::AttnSet> attn_sets;         // LoRA key (-1 = base) -> merged qkv / proj weights
  struct Merge { const void* w; const float *ln_w, *ln_b; };
  struct Split { const void *w1, *w2; const float *ln_w, *ln_b; };
  std::vector<Merge> merges;
  std::vector<Split> splits;
  aurora::Resampler enc_rs, dec_rs, dec_rs_alt;
  bool has_alt = false;

  // grid / levels
  bool have_grid = false;
  int n_lat = 0, n_lon = 0, Hp = 0, Wp = 0, n_levels = 0;     // of THIS rank's rows (the whole grid when un-sharded)
  int full_Hp = 0;                                            // patch rows of the whole grid
  std::vector<double> levels;
  aurora::DevBuf pos_scale, enc_bias, dec_queries, dec_q, dec_q_alt, stats;   // stats: loc | scale | inv per variable and level
  std::vector<size_t> surf_stat_off, static_stat_off, atmos_stat_off;   // float offsets into `stats`: loc, then scale, inv
  size_t one_stat_off = 0;                                    // identity statistics (dynamic planes): loc 0, scale 1, inv 1 per level
  std::vector<size_t> static_lvl_stat_off;                    // static variables fed at every level: their surface statistics per level
  std::vector<double> static_loc;                             // (host) for the wave variant's water-body mask threshold
  std::map<std::array<int64_t, 3>, aurora::EmbedPack> embed_packs;   // (kind, T, presence mask) -> weights
  float enc_bias_max = 0.f;                                   // max |atmospheric level bias| (precompute)
  // surface MLP behind the surface patch embedding, for its guarded two-term chain (finalize): pre-split weights, the
  // largest L1 row norm and |bias| of its first linear, max |embedding bias| + max |level encoding|
  aurora::DevBuf surf_w0_s, surf_w2_s;
  float surf_l1_0 = 0.f, surf_b0 = 0.f, surf_c = 0.f;
  bool surf_chain = false;
  aurora::HeadGroup head_surf, head_main, head_alt;
  float dec_out_bound = 0.f, dec_out_bound_alt = 0.f;   // |output of the decoder Perceiver(s)| <= this (LayerNorm bounds + queries)
  std::map<std::pair<int, int>, aurora::DevTables> tables;   // (stage, shifted)
  std::vector<aurora::Res> stage_res;                        // token grids of the WHOLE forecast
  std::vector<std::array<int, 2>> merge_pad;                 // (pad_h, pad_w) after each stage (whole grid)

  // latitude band (aurora_hip_set_band): rank / world, transport callbacks, staging buffers of the host
  aurora_hip_band band{0, 1, nullptr, nullptr, nullptr};
  std::vector<std::vector<std::array<int, 2>>> rows;         // [stage][rank] owned rows
  std::map<std::pair<int, int>, aurora::DevPlan> plans;      // (stage, shifted)
  void* stage_send = nullptr;
  void* stage_recv = nullptr;
  int64_t staging_bytes = 0, staging_need = 0;
  bool sharded() const { return band.world > 1; }
  // how this handle runs its steps: aurora_hip_config.tuning, fixed when the handle is created
  int fuse_ln = 1;
  bool split_attention = false;
  bool qkv_planes = true;   // bf16 blocks: q | k | v leave the qkv linear one attention head per plane (aurora_hip_linear_planes)
  bool reassoc_out = true;  // decoder de-aggregation: to_out of the three value rows per column, combined in registers (perceiver_out.hip)
  bool score_weights = true;   // first Perceiver layers: scores from W_k^T q rows instead of a key projection (model.hip:score_weights)

  // per step
  aurora::DevBuf abs_enc, dyn_planes, ctx_max;
  aurora::DevBuf tickets;   // split-K tickets of aurora_hip_linear_ws: SPLIT_TICKETS zeroed words, left zero by every launch
  // AURORA_SPLIT_K=0 at creation: no linear is ever split along K, so that a band's bf16 arithmetic is the un-sharded step's
  // bit for bit and does not depend on the device's CU count (split-K is chosen from the tile count against the CUs)
  bool split_k = true;
  int abs_B = 0;
  struct Pinned { float* host = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; };
  Pinned pinned[4];     // staging ring of aurora_hip_set_time: an upload never waits for the previous step
  int pinned_next = 0;
  ~aurora_hip_model() {
    for (auto& t : timed) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
    for (auto& e : event_pool) (void)hipEventDestroy(e);
    for (auto& s : pinned) {
      if (s.host) (void)hipHostFree(s.host);
      if (s.done) (void)hipEventDestroy(s.done);
    }
  }
  aurora::Arena arena;
  bool dry = false;
  int64_t generation = 0;   // bumped whenever device memory a captured hipGraph may point at is re-allocated

  // optional per-launch timing (aurora_hip_profile_begin / _end): HIP events on the launch stream
  struct Timed { int kind; double work; hipEvent_t e0, e1; };
  uint32_t profile_mask = 0;
  std::vector<Timed> timed;
  std::vector<hipEvent_t> event_pool;

  const float* W(const std::string& name) const {
    auto it = w.find(name);
    REQUIRE(it != w.end(), "missing weight '%s'", name.c_str());
    return it->second.f();
  }
  const aurora::Tensor& T_(const std::string& name) const {
    auto it = w.find(name);
    REQUIRE(it != w.end(), "missing weight '%s'", name.c_str());
    return it->second;
  }
  bool has(const std::string& name) const { return w.count(name) != 0; }
  int bb() const { return autocast ? AURORA_BF16 : AURORA_F32; }
  size_t bbs() const { return autocast ? 2 : 4; }
  int stage_dim(int s) const { return D << s; }
};

namespace aurora {

typedef aurora_hip_model Model;

// Kernel kinds of the per-launch timing; `work` is the algorithmic work of a launch: FLOPs for the linears, bytes
// (q, k, v read + o written once) for the window attention, 0 elsewhere.
enum Kind { K_LINEAR_BF16, K_LINEAR_F32, K_WINDOW_ATTENTION, K_LAYERNORM, K_MERGE_LN, K_SPLIT_LN, K_PATCHIFY,
            K_PERCEIVER_ATTENTION, K_ASSEMBLE, K_UNPATCHIFY, K_COPY2D, K_ABSMAX, K_LINEAR_LN, K_GATHER, K_PERCEIVER_OUT, K_COUNT };

hipEvent_t take_event(Model& m);

// Runs `fn` (one launch), bracketed by an event pair when this kind is being profiled.
template <typename F>
void timed(Model& m, void* stream, int kind, double work, F&& fn) {
  if (m.dry) return;
  const bool on = (m.profile_mask >> kind) & 1u;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (on) {
    e0 = take_event(m);
    e1 = take_event(m);
    hip_ok(hipEventRecord(e0, (hipStream_t)stream), "hipEventRecord");
  }
  ok(fn());
  if (on) {
    hip_ok(hipEventRecord(e1, (hipStream_t)stream), "hipEventRecord");
    m.timed.push_back({kind, work, e0, e1});
  }
}

constexpr int SPLIT_TICKETS = 4096;

// launches (skipped in a dry run)
struct Launcher {
  Model& m;
  void* stream;
  // batch > 1: `batch` strided problems in one launch (aurora_hip_linear_batched); strides in elements
  void linear(const void* A, int64_t lda, const void* Wt, int64_t ldw, const float* bias, void* C, int64_t ldc, int64_t M,
              int N, int K, int dtype, int act = AURORA_ACT_NONE, void* C2 = nullptr, int64_t ldc2 = 0,
              const float* res = nullptr, int64_t ldr = 0, int f32_gemm = -1, const float* guard = nullptr,
              float limit = 0.f, int batch = 1, int64_t sa = 0, int64_t sw = 0, int64_t sbias = 0, int64_t sc = 0) {
    // plain bf16 linears with few tiles and a long K borrow slab scratch from the arena and split along K (gemm.hip)
    void* ws = nullptr;
    int64_t ws_bytes = 0;
    const size_t mark = m.arena.top;
    if (m.split_k && dtype == AURORA_BF16 && batch == 1 && f32_gemm == -1 && (ws_bytes = aurora_hip_linear_workspace(M, N, K, dtype)) > 0)
      ws = m.arena.take((size_t)ws_bytes);
    timed(m, stream, dtype == AURORA_BF16 ? K_LINEAR_BF16 : K_LINEAR_F32, 2.0 * (double)M * N * K * batch, [&] {
      if (ws)
        return aurora_hip_linear_ws(A, lda, Wt, ldw, bias, C, ldc, C2, ldc2, res, ldr, M, N, K, dtype, act, ws, ws_bytes,
                                    (int32_t*)m.tickets.p, SPLIT_TICKETS, 0, stream);
      if (batch > 1)
        return aurora_hip_linear_batched(A, lda, Wt, ldw, bias, C, ldc, M, N, K, dtype, act, f32_gemm, guard, limit, batch, sa,
                                         sw, sbias, sc, stream);
      return aurora_hip_linear_ex(A, lda, Wt, ldw, bias, C, ldc, C2, ldc2, res, ldr, M, N, K, dtype, act, f32_gemm, guard,
                                  limit, stream);
    });
    m.arena.top = mark;   // (stream order: the next launch that takes this memory runs after this one)
  }
  // the qkv linear of a block with its result in head planes (bf16)
  void linear_planes(const void* A, int64_t lda, const void* Wt, int64_t ldw, const float* bias, void* C, int64_t plane_stride,
                     int heads, int64_t M, int N, int K) {
    timed(m, stream, K_LINEAR_BF16, 2.0 * (double)M * N * K, [&] {
      return aurora_hip_linear_planes(A, lda, Wt, ldw, bias, C, plane_stride, heads, M, N, K, AURORA_BF16, stream);
    });
  }
  void layernorm(const void* y, int64_t ldy, const float* gain, const float* shift, const float* res, int64_t ldr,
                 int64_t res_mod, float* out_f32, int64_t ldo, void* out_t, int64_t ldt, int64_t M, int D, float eps,
                 int dtype) {
    timed(m, stream, K_LAYERNORM, 0.0, [&] {
      return aurora_hip_layernorm(y, ldy, gain, shift, res, ldr, res_mod, out_f32, ldo, out_t, ldt, M, D, eps, dtype, stream);
    });
  }
};

inline void upload(void* dst, const void* src, size_t bytes) {
  hip_ok(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), "upload");
}
inline DevBuf to_device(const std::vector<float>& v) {
  DevBuf b(v.size() * sizeof(float));
  upload(b.p, v.data(), v.size() * sizeof(float));
  return b;
}
inline std::vector<float> to_host(const Tensor& t) {
  std::vector<float> h((size_t)t.numel);
  hip_ok(hipMemcpy(h.data(), t.f(), h.size() * 4, hipMemcpyDeviceToHost), "download");
  return h;
}

constexpr float F16_SAFE = 16384.0f;   // activations below this may take the two-term fp16 operand split
// fp32 GEMM mode of the linears whose input is bounded (by construction or by the device-side guard): the two-term fp16
// split, unless the user pinned a mode through AURORA_F32_GEMM
int bounded_mode();
std::string level_to_str(double level);   // `850`, `0_5` (aurora/normalisation.py:19-32 upstream)

// model.hip
int lora_key(const Model& m, int step);
const AttnSet& attn_weights(Model& m, int key, void* stream);
const EmbedPack& embed_pack(Model& m, int kind, int T, const std::vector<char>& present);
const DevTables& tables_for(Model& m, int stage, bool shifted);
const DevPlan& plan_for(Model& m, int stage, bool shifted);
// step.hip
struct StepIO {
  const aurora_hip_step_io* io;
  int B, T;
};
void run_step(Model& m, const StepIO& s, void* stream);

}  // namespace aurora
