// One forecast step behind a C ABI: the model handle of libaurora_hip.so.
//
// aurora_hip_create / _pack_weights / _finalize / _precompute / _set_time / _step / _destroy (include/aurora_hip.h) own
// everything the reference's `Aurora.forward` (aurora/model/aurora.py:265-392) needs besides the input fields: the
// configuration, the weights (fp32 masters + bf16 backbone copies, LoRA merged per roll-out phase), the tables that
// depend on parameters / grid / levels only (AdaLN modulation, Fourier position / scale / level encodings, window
// token tables), the workspace, and the SEQUENCE of kernel launches of a step -- encoder (encoder.py:198-366), 3D Swin
// U-net (swin3d.py:884-936, 440-509), decoder (decoder.py:168-276) -- on a caller-supplied stream.  No torch, no Python:
// any host language that can call C can run Aurora on an MI355X through these seven functions; aurora_amd's own
// Python `Engine` uses them for the ERA5 model family.
//
// Scope: the model family of BASELINE configs 1-4 (Aurora, AuroraPretrained, AuroraSmallPretrained,
// Aurora12hPretrained, AuroraHighRes: any patch size / depths / history / LoRA mode, stabilised level aggregation,
// batch > 1), one device, no latitude-band sharding.  The air-pollution and ocean-wave variants (level-conditioned
// embeddings, feature combiners, second decoder Perceiver, NaN / angle hooks) and sharded steps are sequenced by the
// Python engine over the same operator entry points.
//
// Host code only; every launch goes through the operator ABI of this same library.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <exception>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "band.h"
#include "common.h"

using namespace aurora;

namespace {

constexpr double PI = 3.14159265358979323846;
constexpr int LORA_RANK = 8;

struct Fail {
  int code;
};
#define REQUIRE(cond, ...)               \
  do {                                   \
    if (!(cond)) {                       \
      ::aurora::set_error(__VA_ARGS__);  \
      throw Fail{AURORA_E_ARG};          \
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

// ---- host-side tables ----------------------------------------------------------------------------
// Fourier features (aurora/model/fourier.py:45-92, 112-126): [sin(2 pi x / lambda_j) | cos(...)], lambda log-spaced,
// evaluated in fp64 and cast to fp32 like `encoding.float()` upstream.
double polygon_area_km2(const double (*poly)[2], int n_in) {   // aurora/area.py:12-48, incl. its way of closing the ring
  std::vector<std::array<double, 2>> pts;
  for (int i = 0; i < n_in; ++i) pts.push_back({poly[i][0], poly[i][1]});
  pts.push_back({poly[n_in - 1][0], poly[n_in - 1][1]});
  const int n = (int)pts.size();
  const double R = 6378137.0 / 1000.0, rad = PI / 180.0;
  double total = 0.0;
  for (int i = 0; i < n; ++i)
    total += (pts[(i + 2) % n][1] * rad - pts[i][1] * rad) * sin(pts[(i + 1) % n][0] * rad);
  return fabs(total * R * R / 2);
}

enum Expansion { POS, SCALE, LEAD_TIME, LEVELS, ABS_TIME };
void expansion_range(Expansion kind, double& lower, double& upper, bool& check) {
  const double delta = 0.01, R = 6378137.0 / 1000.0;
  switch (kind) {
    case POS: lower = delta; upper = 720.0; check = true; break;
    case SCALE: {
      const double poly[4][2] = {{90, 0}, {90, delta}, {90 - delta, delta}, {90 - delta, 0}};
      lower = polygon_area_km2(poly, 4); upper = 4 * PI * R * R; check = true; break;
    }
    case LEAD_TIME: lower = 1.0 / 60; upper = 24.0 * 7 * 3; check = true; break;
    case LEVELS: lower = 0.01; upper = 1e5; check = true; break;
    default: lower = 1.0; upper = 24 * 365.25; check = false; break;
  }
}
void fourier(Expansion kind, const double* x, int64_t n, int d, float* out) {
  double lower, upper;
  bool check;
  expansion_range(kind, lower, upper, check);
  REQUIRE(d % 2 == 0, "The dimensionality must be a multiple of two.");
  const int h = d / 2;
  std::vector<double> w(h);
  const double a = log10(lower), b = log10(upper), step = h > 1 ? (b - a) / (h - 1) : 0.0;
  for (int j = 0; j < h; ++j) w[j] = 2 * PI / pow(10.0, j == h - 1 && h > 1 ? b : a + j * step);
  for (int64_t i = 0; i < n; ++i) {
    const double ax = fabs(x[i]);
    REQUIRE(!check || x[i] == 0 || (lower <= ax && ax <= upper),
            "The input tensor is not within the configured range `[%g, %g]`.", lower, upper);
    for (int j = 0; j < h; ++j) {
      const double pr = x[i] * w[j];
      out[i * d + j] = (float)sin(pr);
      out[i * d + h + j] = (float)cos(pr);
    }
  }
}

// Fourier position / scale features of the patch grid (posencoding.py:61-192): [L][D] each, L = Hp * Wp.
// Patch-mean position and patch root area in fp32 like the reference, the trigonometry in fp64 (the reference's fp32
// torch kernels are not reproducible bit for bit outside torch; callers who need that pass the encodings in).
void pos_scale_tables(const double* lat, const double* lon, int Hp, int Wp, int P, int D, float* pos_out, float* scale_out) {
  const int64_t Lp = (int64_t)Hp * Wp;
  std::vector<double> mid_lat(Hp), mid_lon(Wp), area_lat(Hp), area_lon(Wp);
  const float rad = (float)(PI / 180.0);
  for (int hp = 0; hp < Hp; ++hp) {
    float sum = 0.f, mx = -INFINITY, mn = INFINITY;
    for (int i = 0; i < P; ++i) {
      const float v = (float)lat[hp * P + i];
      for (int j = 0; j < P; ++j) sum += v;   // avg_pool2d sums the P x P window of the broadcast grid in fp32
      mx = fmaxf(mx, v); mn = fminf(mn, v);
    }
    REQUIRE(mx > mn, "latitudes of a patch must differ");
    mid_lat[hp] = (double)(sum / (float)(P * P));
    area_lat[hp] = (double)((float)sin((double)(mx * rad)) - (float)sin((double)(mn * rad)));
  }
  for (int wp = 0; wp < Wp; ++wp) {
    float sum = 0.f, mx = -INFINITY, mn = INFINITY;
    for (int i = 0; i < P; ++i)
      for (int j = 0; j < P; ++j) sum += (float)lon[wp * P + j];
    for (int j = 0; j < P; ++j) {
      const float v = (float)lon[wp * P + j];
      mx = fmaxf(mx, v); mn = fminf(mn, v);
    }
    REQUIRE(mx > mn, "longitudes of a patch must differ");
    mid_lon[wp] = (double)(sum / (float)(P * P));
    area_lon[wp] = (double)(mx * rad - mn * rad);
  }
  std::vector<double> xs(Lp), ys(Lp), ra(Lp);
  for (int hp = 0; hp < Hp; ++hp)
    for (int wp = 0; wp < Wp; ++wp) {
      const int64_t l = (int64_t)hp * Wp + wp;
      // avg_pool2d over a P x P patch of a separable grid: mean over rows of the (constant per row) latitudes
      xs[l] = mid_lat[hp];
      ys[l] = mid_lon[wp];
      const float area = (float)(6371.0 * 6371.0 * PI) * (float)area_lat[hp] * (float)area_lon[wp];
      REQUIRE(area > 0, "patch areas must be positive");
      ra[l] = (double)sqrtf(area);
    }
  std::vector<float> half((size_t)Lp * (D / 2));
  fourier(POS, xs.data(), Lp, D / 2, half.data());
  for (int64_t l = 0; l < Lp; ++l) memcpy(&pos_out[(size_t)l * D], &half[(size_t)l * (D / 2)], (D / 2) * 4);
  fourier(POS, ys.data(), Lp, D / 2, half.data());
  for (int64_t l = 0; l < Lp; ++l) memcpy(&pos_out[(size_t)l * D + D / 2], &half[(size_t)l * (D / 2)], (D / 2) * 4);
  fourier(SCALE, ra.data(), Lp, D, scale_out);
}

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
    int inner, head_dim, hidden, dim;
    float v_l1;
    int f16_mode;   // fp32 GEMM mode of this layer's bounded linears: 2 (two fp16 terms) when weights / LN bounds allow
    // the same weights in the fp16-pair layout, scaled by 2^6 (null where mode or shape rule it out): the two-term GEMMs
    // then spend no VALU work on the weight operand, and none at all where the activations arrive split as well
    const void *to_kv_s = nullptr, *to_out_s = nullptr, *fc1_s = nullptr, *fc2_s = nullptr;
  };
  std::vector<Layer> layers;
  std::vector<DevBuf> own;
};
struct DevTables { DevBuf tok, grp; int n_windows = 0, n_tok = 0; bool has_grp = false; };

}  // namespace

struct aurora_hip_model {
  // configuration
  int D = 0, P = 0, Cl = 0, perceiver_heads = 0, n_stages = 0;
  int enc_depths[4] = {0}, dec_depths[4] = {0}, enc_heads[4] = {0}, dec_heads[4] = {0}, window[3] = {0};
  int enc_depth = 1, dec_depth = 1, max_history = 2, lora_steps = 40, lora_mode = 0;
  bool stabilise = false, use_lora = false, autocast = false;
  float ln_eps = 1e-5f;
  double timestep_hours = 6;
  std::vector<std::string> surf_vars, static_vars, atmos_vars;

  // weights
  std::map<std::string, Tensor> w;                // fp32 masters (aurora_hip_pack_weights / a packed file)
  std::map<std::string, Tensor> w16;              // bf16-only entries of a packed file (shape kept, data bf16)
  bool finalized = false;
  std::vector<Block> blocks;
  DevBuf mod, lead_emb, enc_q0;
  std::vector<DevBuf> keep;                       // bf16 copies and other derived device arrays
  std::map<int, AttnSet> attn_sets;               // LoRA key (-1 = base) -> merged qkv / proj weights
  struct Merge { const void* w; const float *ln_w, *ln_b; };
  struct Split { const void *w1, *w2; const float *ln_w, *ln_b; };
  std::vector<Merge> merges;
  std::vector<Split> splits;
  Resampler enc_rs, dec_rs;

  // grid / levels
  bool have_grid = false;
  int n_lat = 0, n_lon = 0, Hp = 0, Wp = 0, n_levels = 0;
  DevBuf pos_scale, enc_bias, dec_queries, dec_q, stats;   // stats: loc | scale | inv per variable and level
  std::vector<size_t> surf_stat_off, static_stat_off, atmos_stat_off;   // float offsets into `stats`: loc, then scale, inv
  std::map<std::pair<int, int>, DevBuf> embed_w;           // (0 surf / 1 atmos, T) -> (D, Kpad) patch-embed GEMM weight
  std::map<std::pair<int, int>, DevBuf> embed_ws;          // ... the same in the fp16-pair layout (scaled by 2^6), if eligible
  std::map<std::pair<int, int>, float> embed_l1;           // ... its largest L1 row norm: |embedding| <= l1 * max|input| + |bias|
  float enc_bias_max = 0.f;                                // max |atmospheric level bias| (precompute)
  // surface MLP behind the surface patch embedding, for its guarded two-term chain (finalize): pre-split weights, the
  // largest L1 row norm and |bias| of its first linear, max |embedding bias| + max |level encoding|
  DevBuf surf_w0_s, surf_w2_s;
  float surf_l1_0 = 0.f, surf_b0 = 0.f, surf_c = 0.f;
  bool surf_chain = false;
  DevBuf head_surf_w, head_surf_b, head_atmos_w, head_atmos_b;
  std::map<std::pair<int, int>, DevTables> tables;         // (stage, shifted)
  std::vector<Res> stage_res;
  std::vector<std::array<int, 2>> merge_pad;               // (pad_h, pad_w) after each stage

  // per step
  DevBuf abs_enc, ctx_max;
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
  Arena arena;
  bool dry = false;

  // optional per-launch timing (aurora_hip_profile_begin / _end): HIP events on the launch stream
  struct Timed { int kind; double work; hipEvent_t e0, e1; };
  uint32_t profile_mask = 0;
  std::vector<Timed> timed;
  std::vector<hipEvent_t> event_pool;

  const void* dt_ptr(const std::string& name);   // weight in the backbone compute dtype
  const float* W(const std::string& name) const {
    auto it = w.find(name);
    REQUIRE(it != w.end(), "missing weight '%s'", name.c_str());
    return it->second.f();
  }
  const Tensor& T_(const std::string& name) const {
    auto it = w.find(name);
    REQUIRE(it != w.end(), "missing weight '%s'", name.c_str());
    return it->second;
  }
  bool has(const std::string& name) const { return w.count(name) != 0; }
  int bb() const { return autocast ? AURORA_BF16 : AURORA_F32; }
  size_t bbs() const { return autocast ? 2 : 4; }
  int stage_dim(int s) const { return D << s; }
};

namespace {

typedef aurora_hip_model Model;

// Kernel kinds of the per-launch timing; `work` is the algorithmic work of a launch: FLOPs for the linears, bytes
// (q, k, v read + o written once) for the window attention, 0 elsewhere.
enum Kind { K_LINEAR_BF16, K_LINEAR_F32, K_WINDOW_ATTENTION, K_LAYERNORM, K_MERGE_LN, K_SPLIT_LN, K_PATCHIFY,
            K_PERCEIVER_ATTENTION, K_ASSEMBLE, K_UNPATCHIFY, K_COPY2D, K_ABSMAX, K_LINEAR_LN, K_COUNT };
const char* const KIND_NAMES[K_COUNT] = {"linear_bf16", "linear_f32", "window_attention_bf16", "layernorm", "merge_ln",
                                         "split_ln", "patchify", "perceiver_attention", "assemble_tokens", "unpatchify",
                                         "copy2d", "absmax", "linear_layernorm_bf16"};

hipEvent_t take_event(Model& m) {
  if (!m.event_pool.empty()) {
    hipEvent_t e = m.event_pool.back();
    m.event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hip_ok(hipEventCreate(&e), "hipEventCreate");
  return e;
}

// Runs `fn` (one launch), bracketed by an event pair when this kind is being profiled.
template <typename F>
void timed(Model& m, void* stream, int kind, double work, F&& fn) {
  if (m.dry) return;
  const bool on = (m.profile_mask >> kind) & 1u;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (on) {
    e0 = take_event(m);
    e1 = take_event(m);
    hip_ok(hipEventRecord(e0, as_stream(stream)), "hipEventRecord");
  }
  ok(fn());
  if (on) {
    hip_ok(hipEventRecord(e1, as_stream(stream)), "hipEventRecord");
    m.timed.push_back({kind, work, e0, e1});
  }
}

// launches (skipped in a dry run)
struct Launcher {
  Model& m;
  void* stream;
  void linear(const void* A, int64_t lda, const void* Wt, int64_t ldw, const float* bias, void* C, int64_t ldc, int64_t M,
              int N, int K, int dtype, int act = AURORA_ACT_NONE, void* C2 = nullptr, int64_t ldc2 = 0,
              const float* res = nullptr, int64_t ldr = 0, int f32_gemm = -1, const float* guard = nullptr,
              float limit = 0.f) {
    timed(m, stream, dtype == AURORA_BF16 ? K_LINEAR_BF16 : K_LINEAR_F32, 2.0 * (double)M * N * K, [&] {
      return aurora_hip_linear_ex(A, lda, Wt, ldw, bias, C, ldc, C2, ldc2, res, ldr, M, N, K, dtype, act, f32_gemm, guard,
                                  limit, stream);
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

void upload(void* dst, const void* src, size_t bytes) { hip_ok(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), "upload"); }

DevBuf to_device(const std::vector<float>& v) {
  DevBuf b(v.size() * sizeof(float));
  upload(b.p, v.data(), v.size() * sizeof(float));
  return b;
}

int lora_key(const Model& m, int step) {   // lora.py:105-129; -1 = no LoRA
  if (!m.use_lora || step >= m.lora_steps) return -1;
  if (m.lora_mode == 0) return 0;                       // single
  if (m.lora_mode == 1) return step == 0 ? -1 : 0;      // from_second
  return step;                                          // all
}

// A backbone weight in the compute dtype: the fp32 master itself (autocast off), the bf16 entry of a packed file, or a
// bf16 copy of the master made once.  `out_shape0` receives the leading dimension (hidden sizes are read off weights).
const void* compute_weight(Model& m, const std::string& name, void* stream, int64_t* out_shape0 = nullptr) {
  auto h = m.w16.find(name);
  if (h != m.w16.end()) {
    REQUIRE(m.autocast, "'%s' is stored in bf16 only: this packed file serves autocast (bf16 backbone) models", name.c_str());
    if (out_shape0) *out_shape0 = h->second.shape[0];
    return h->second.buf.p;
  }
  const Tensor& t = m.T_(name);
  if (out_shape0) *out_shape0 = t.shape[0];
  if (!m.autocast) return t.f();
  DevBuf b((size_t)t.numel * 2);
  ok(aurora_hip_convert(t.f(), b.p, t.numel, AURORA_F32, stream));
  m.keep.push_back(std::move(b));
  return m.keep.back().p;
}

void build_blocks(Model& m) {
  m.blocks.clear();
  for (int part = 0; part < 2; ++part)
    for (int i = 0; i < m.n_stages; ++i) {
      const int depth = part == 0 ? m.enc_depths[i] : m.dec_depths[i];
      const int stage = part == 0 ? i : m.n_stages - 1 - i;
      for (int j = 0; j < depth; ++j) {
        Block b{};
        b.prefix = std::string(part == 0 ? "backbone.encoder_layers." : "backbone.decoder_layers.") + std::to_string(i) +
                   ".blocks." + std::to_string(j);
        b.dim = m.stage_dim(stage);
        b.stage = stage;
        b.heads = part == 0 ? m.enc_heads[i] : m.dec_heads[i];
        b.shifted = j % 2 == 1;
        REQUIRE(b.dim == b.heads * 64, "the window-attention kernel is built for head_dim 64 (dim %d, %d heads)", b.dim,
                b.heads);
        m.blocks.push_back(b);
      }
    }
}

constexpr float F16_SAFE = 16384.0f;   // activations below this may take the two-term fp16 operand split
// fp32 GEMM mode of the linears whose input is bounded (by construction or by the device-side guard): the two-term fp16
// split, unless the user pinned a mode through AURORA_F32_GEMM
int bounded_mode() {
  static const int mode = getenv("AURORA_F32_GEMM") ? -1 : 2;
  return mode;
}

Resampler pack_resampler(Model& m, const std::string& prefix, int depth, int heads) {
  Resampler r;
  for (int i = 0; i < depth; ++i) {
    const std::string p = prefix + ".layers." + std::to_string(i);
    Resampler::Layer l{};
    l.to_q = m.W(p + ".0.to_q.weight"); l.to_kv = m.W(p + ".0.to_kv.weight"); l.to_out = m.W(p + ".0.to_out.weight");
    l.fc1_w = m.W(p + ".1.net.0.weight"); l.fc1_b = m.W(p + ".1.net.0.bias");
    l.fc2_w = m.W(p + ".1.net.2.weight"); l.fc2_b = m.W(p + ".1.net.2.bias");
    l.ln1_w = m.W(p + ".2.weight"); l.ln1_b = m.W(p + ".2.bias");
    l.ln2_w = m.W(p + ".3.weight"); l.ln2_b = m.W(p + ".3.bias");
    if (m.has(p + ".0.ln_k.weight")) {
      l.ln_k_w = m.W(p + ".0.ln_k.weight"); l.ln_k_b = m.W(p + ".0.ln_k.bias");
      l.ln_q_w = m.W(p + ".0.ln_q.weight"); l.ln_q_b = m.W(p + ".0.ln_q.bias");
    }
    const Tensor& tq = m.T_(p + ".0.to_q.weight");
    const Tensor& tkv = m.T_(p + ".0.to_kv.weight");
    l.inner = (int)tq.shape[0];
    l.head_dim = l.inner / heads;
    l.hidden = (int)m.T_(p + ".1.net.0.weight").shape[0];
    l.dim = (int)m.T_(p + ".0.to_out.weight").shape[0];
    // largest L1 row norm of the value projection: |v| <= v_l1 * max |context| (range guard of the fp16 operand split)
    const int64_t K = tkv.shape[1];
    std::vector<float> host((size_t)l.inner * K);
    hip_ok(hipMemcpy(host.data(), tkv.f() + (int64_t)l.inner * K, host.size() * 4, hipMemcpyDeviceToHost), "download");
    float best = 1e-6f;
    for (int r_ = 0; r_ < l.inner; ++r_) {
      float s = 0.f;
      for (int64_t k = 0; k < K; ++k) s += fabsf(host[(size_t)r_ * K + k]);
      best = std::max(best, s);
    }
    l.v_l1 = best;
    // The two-term fp16 split scales weights by 2^6 and assumes |activation| < 65504: only if every weight of the layer
    // stays below 1000 and what a LayerNorm output can reach (sqrt(D) max|gain| + max|bias|) stays inside the range.
    auto absmax_of = [&](const std::string& name) {
      const Tensor& t = m.T_(name);
      std::vector<float> h((size_t)t.numel);
      hip_ok(hipMemcpy(h.data(), t.f(), h.size() * 4, hipMemcpyDeviceToHost), "download");
      float mx = 0.f;
      for (float v : h) mx = std::max(mx, fabsf(v));
      return mx;
    };
    float w_max = 0.f;
    for (const char* nm : {".0.to_kv.weight", ".0.to_out.weight", ".1.net.0.weight", ".1.net.2.weight"})
      w_max = std::max(w_max, absmax_of(p + nm));
    const float ln_bound = absmax_of(p + ".2.weight") * sqrtf((float)l.dim) + absmax_of(p + ".2.bias");
    l.f16_mode = (w_max < 1000.f && ln_bound < F16_SAFE) ? bounded_mode() : -1;
    if (l.f16_mode == 2) {
      auto presplit = [&](const std::string& name) -> const void* {
        const Tensor& t = m.T_(name);
        const int64_t N = t.shape[0], K = t.shape[1];
        if (N % 256 != 0 || K % 32 != 0 || K < 96) return nullptr;
        r.own.emplace_back((size_t)N * K * 4);
        if (aurora_hip_split_f16(t.f(), K, r.own.back().p, K, N, (int)K, 64.0f, nullptr) != AURORA_OK)
          throw std::runtime_error(aurora_hip_last_error());
        return r.own.back().p;
      };
      l.to_kv_s = presplit(p + ".0.to_kv.weight");
      l.to_out_s = presplit(p + ".0.to_out.weight");
      l.fc1_s = presplit(p + ".1.net.0.weight");
      l.fc2_s = presplit(p + ".1.net.2.weight");
      hip_ok(hipDeviceSynchronize(), "split weights");
    }
    r.layers.push_back(l);
  }
  return r;
}



// What is known on the device about max |context| of a resampler: max|ctx| <= a * (*word) + c.  `pairs`: the context
// buffer holds fp16 pairs iff *word < limit_kv (written so by a guarded two-term producer with that very guard), fp32 otherwise.
struct CtxGuard { const float* word; float a, c, limit_kv; bool pairs; };

// PerceiverResampler (perceiver.py:212-233) for all grid columns at once.  ctx: key j of column (b, l) at row
// b*kv_bstride + j*kv_lstride + l.  First layer: latents (and so q) are shared by every column.  Returns (B*cols*Lq, D).
float* resampler(Model& m, Launcher& L, const Resampler& rs, const float* ctx, int64_t ctx_rows, int ctx_dim, const float* q0,
                 const float* latents0, int B, int64_t cols, int64_t kv_bstride, int64_t kv_lstride, int Lq, int Lk, int heads,
                 float eps, size_t& out_mark, const CtxGuard* cg = nullptr) {
  const int64_t n_rows = (int64_t)B * cols * Lq;
  // The context is as unbounded as the model inputs, so the linears that read it, or averages of its value projection,
  // pick their operand split on the device: from max |ctx|, measured here, or from the bound the caller derived from a
  // word it measured upstream (`cg`).
  const float* ctx_max = cg ? cg->word : m.ctx_max.f();
  const float g_a = cg ? cg->a : 1.0f, g_c = cg ? cg->c : 0.0f;
  const bool ctx_pairs = cg && cg->pairs;
  if (!cg) timed(m, L.stream, K_ABSMAX, 0.0, [&] { return aurora_hip_absmax(ctx, ctx_rows * ctx_dim, m.ctx_max.f(), L.stream); });
  float* lat = nullptr;
  for (size_t i = 0; i < rs.layers.size(); ++i) {
    const auto& ly = rs.layers[i];
    const int inner = ly.inner, Dd = ly.dim;
    const size_t mark0 = m.arena.top;
    // result of this layer first (it outlives the temporaries below; stack order)
    float* y = (float*)m.arena.take((size_t)n_rows * Dd * 4);
    const size_t after_y = m.arena.top;
    float* kv = (float*)m.arena.take((size_t)ctx_rows * 2 * inner * 4);
    // guarded linears with pre-split weights: the two-term launch runs iff the guard holds, the three-term one (fp32
    // weights) iff it does not
    auto guarded = [&](const float* A, int64_t lda, const float* Wf, const void* Ws, float* C_, int64_t ldc, int64_t M_, int N_,
                       int K_, float limit, bool a_pairs = false) {
      if (Ws) {
        L.linear(A, lda, Ws, K_, nullptr, C_, ldc, M_, N_, K_, AURORA_F32, 0, nullptr, 0, nullptr, 0,
                 2 | AURORA_F32_W_SPLIT | (a_pairs ? AURORA_F32_A_SPLIT : 0), ctx_max, limit);
        L.linear(A, lda, Wf, K_, nullptr, C_, ldc, M_, N_, K_, AURORA_F32, 0, nullptr, 0, nullptr, 0, 1, ctx_max, limit);
      } else if (ly.f16_mode == 2) {   // one guarded call: the device word picks the two- or the three-term kernel
        L.linear(A, lda, Wf, K_, nullptr, C_, ldc, M_, N_, K_, AURORA_F32, 0, nullptr, 0, nullptr, 0, 2, ctx_max, limit);
      } else {
        // a pinned mode (AURORA_F32_GEMM) or weights outside the two-term range: NO guard -- a mode-1 launch that carries a
        // guard is the three-term half of a guarded pair and runs only if the guard FAILS (include/aurora_hip.h)
        L.linear(A, lda, Wf, K_, nullptr, C_, ldc, M_, N_, K_, AURORA_F32, 0, nullptr, 0, nullptr, 0, ly.f16_mode, nullptr, 0.f);
      }
    };
    // |ctx| <= g_a * word + g_c < F16_SAFE  <=>  word < (F16_SAFE - g_c) / g_a; a context in pairs comes with its own limit
    REQUIRE(!ctx_pairs || ly.to_kv_s, "resampler: a pair-layout context needs pre-split to_kv weights");
    guarded(ctx, ctx_dim, ly.to_kv, ly.to_kv_s, kv, 2 * inner, ctx_rows, 2 * inner, ctx_dim,
            ctx_pairs ? cg->limit_kv : (F16_SAFE - g_c) / g_a, ctx_pairs);
    if (ly.ln_k_w)   // LayerNorm over the K half, in place (perceiver.py:144-147)
      L.layernorm(kv, 2 * inner, ly.ln_k_w, ly.ln_k_b, nullptr, 0, 0, kv, 2 * inner, nullptr, 0, ctx_rows, inner, 1e-5f,
                  AURORA_F32);
    const float* q = q0;
    int64_t q_stride = 0;
    if (i > 0) {
      float* qb = (float*)m.arena.take((size_t)n_rows * inner * 4);
      L.linear(lat, Dd, ly.to_q, Dd, nullptr, qb, inner, n_rows, inner, Dd, AURORA_F32);
      if (ly.ln_q_w) L.layernorm(qb, inner, ly.ln_q_w, ly.ln_q_b, nullptr, 0, 0, qb, inner, nullptr, 0, n_rows, inner, 1e-5f, AURORA_F32);
      q = qb;
      q_stride = Lq;
    }
    float* att = (float*)m.arena.take((size_t)n_rows * inner * 4);
    // |att| <= max |v| <= (largest L1 row norm of W_v) * max |ctx|: same guard, tighter limit.  With pre-split to_out
    // weights the attention writes fp16 pairs iff that guard holds, and to_out multiplies them without splitting anything.
    const float lim_out = (F16_SAFE / ly.v_l1 - g_c) / g_a;
    const bool att_pairs = ly.to_out_s != nullptr && inner % 32 == 0;
    timed(m, L.stream, K_PERCEIVER_ATTENTION, 0.0, [&] {
      return aurora_hip_perceiver_attention_ex(q, q_stride, kv, att, B, cols, kv_bstride, kv_lstride, Lq, Lk, heads, ly.head_dim,
                                               AURORA_F32, att_pairs ? ctx_max : nullptr, lim_out, L.stream);
    });
    float* o = (float*)m.arena.take((size_t)n_rows * Dd * 4);
    guarded(att, inner, ly.to_out, ly.to_out_s, o, Dd, n_rows, Dd, inner, lim_out, att_pairs);
    // The MLP in the fp16-pair layout end to end: the LayerNorm writes its result already split (and ONLY split), fc1
    // reads that and writes its GELU'd result split, fc2 reads that -- neither GEMM splits anything -- and the LayerNorm
    // behind the MLP takes the split array as its residual.
    const bool pairs = ly.fc1_s && ly.fc2_s && Dd % 32 == 0;
    float* lat1 = (float*)m.arena.take((size_t)n_rows * Dd * 4);   // fp32 values, or their fp16 pairs
    {
      const float* res_ = i == 0 ? latents0 : lat;
      const int64_t mod_ = i == 0 ? Lq : 0;
      if (pairs)
        timed(m, L.stream, K_LAYERNORM, 0.0, [&] {
          return aurora_hip_layernorm_split(o, Dd, ly.ln1_w, ly.ln1_b, res_, Dd, mod_, 0, nullptr, 0, lat1, Dd, n_rows, Dd, eps,
                                            L.stream);
        });
      else L.layernorm(o, Dd, ly.ln1_w, ly.ln1_b, res_, Dd, mod_, lat1, Dd, nullptr, 0, n_rows, Dd, eps, AURORA_F32);
    }
    float* hid = (float*)m.arena.take((size_t)n_rows * ly.hidden * 4);
    // fc1 sees a LayerNorm output (|x| <= sqrt(D) * gain), fc2 its GELU: bounded whatever the inputs are
    if (pairs) {
      const int all = 2 | AURORA_F32_A_SPLIT | AURORA_F32_W_SPLIT;
      L.linear(lat1, Dd, ly.fc1_s, Dd, ly.fc1_b, hid, ly.hidden, n_rows, ly.hidden, Dd, AURORA_F32, AURORA_ACT_GELU, nullptr, 0,
               nullptr, 0, all | AURORA_F32_C_SPLIT);
      L.linear(hid, ly.hidden, ly.fc2_s, ly.hidden, ly.fc2_b, y, Dd, n_rows, Dd, ly.hidden, AURORA_F32, 0, nullptr, 0, nullptr, 0, all);
    } else {
      L.linear(lat1, Dd, ly.fc1_w, Dd, ly.fc1_b, hid, ly.hidden, n_rows, ly.hidden, Dd, AURORA_F32, AURORA_ACT_GELU, nullptr, 0,
               nullptr, 0, ly.f16_mode);
      L.linear(hid, ly.hidden, ly.fc2_w, ly.hidden, ly.fc2_b, y, Dd, n_rows, Dd, ly.hidden, AURORA_F32, 0, nullptr, 0, nullptr, 0,
               ly.f16_mode);
    }
    if (pairs)
      timed(m, L.stream, K_LAYERNORM, 0.0, [&] {
        return aurora_hip_layernorm_split(y, Dd, ly.ln2_w, ly.ln2_b, lat1, Dd, 0, 1, y, Dd, nullptr, 0, n_rows, Dd, eps, L.stream);
      });
    else L.layernorm(y, Dd, ly.ln2_w, ly.ln2_b, lat1, Dd, 0, y, Dd, nullptr, 0, n_rows, Dd, eps, AURORA_F32);
    m.arena.top = after_y;            // temporaries of this layer are dead (a previous layer's result stays below y)
    lat = y;
    if (i == 0) out_mark = mark0;
  }
  return lat;
}

const DevTables& tables_for(Model& m, int stage, bool shifted) {
  auto key = std::make_pair(stage, (int)shifted);
  auto it = m.tables.find(key);
  if (it == m.tables.end()) {
    const WindowTables t = window_tables(m.stage_res[stage], m.window, shifted);
    REQUIRE(t.n_tok <= 144, "windows of more than 144 tokens are not supported");
    DevTables d;
    d.n_windows = t.n_windows;
    d.n_tok = t.n_tok;
    d.tok = DevBuf(t.tok.size() * 4);
    upload(d.tok.p, t.tok.data(), t.tok.size() * 4);
    d.has_grp = !t.grp.empty();
    if (d.has_grp) {
      d.grp = DevBuf(t.grp.size());
      upload(d.grp.p, t.grp.data(), t.grp.size());
    }
    it = m.tables.emplace(key, std::move(d)).first;
  }
  return it->second;
}

// LoRA-merged attention weights of one roll-out phase: W' = W + B A (rank 8, alpha / r = 1), one small GEMM per weight.
const AttnSet& attn_weights(Model& m, int key, void* stream) {
  auto it = m.attn_sets.find(key);
  if (it != m.attn_sets.end()) return it->second;
  AttnSet set;
  for (const Block& blk : m.blocks) {
    for (int which = 0; which < 2; ++which) {
      const std::string name = blk.prefix + (which == 0 ? ".attn.qkv" : ".attn.proj");
      if (key < 0 && m.w16.count(name + ".weight")) {   // packed bf16 file of a model without LoRA
        (which == 0 ? set.qkv : set.proj).push_back(compute_weight(m, name + ".weight", stream));
        continue;
      }
      const Tensor& wt = m.T_(name + ".weight");
      const int64_t out_f = wt.shape[0], in_f = wt.shape[1];
      const float* src = wt.f();
      DevBuf merged;
      if (key >= 0) {
        const std::string lp = blk.prefix + (which == 0 ? ".attn.lora_qkv.loras." : ".attn.lora_proj.loras.") + std::to_string(key);
        const Tensor& a = m.T_(lp + ".lora_A");   // (r, in)
        const Tensor& b = m.T_(lp + ".lora_B");   // (out, r)
        // operands zero-padded to one 32-wide fp32 K-tile: b_p (out, 32), a_t (in, 32) = A^T
        std::vector<float> ha((size_t)a.numel), hb((size_t)b.numel);
        hip_ok(hipMemcpy(ha.data(), a.f(), ha.size() * 4, hipMemcpyDeviceToHost), "download");
        hip_ok(hipMemcpy(hb.data(), b.f(), hb.size() * 4, hipMemcpyDeviceToHost), "download");
        std::vector<float> at((size_t)in_f * 32, 0.f), bp((size_t)out_f * 32, 0.f);
        for (int r_ = 0; r_ < LORA_RANK; ++r_)
          for (int64_t k = 0; k < in_f; ++k) at[(size_t)k * 32 + r_] = ha[(size_t)r_ * in_f + k];
        for (int64_t o = 0; o < out_f; ++o)
          for (int r_ = 0; r_ < LORA_RANK; ++r_) bp[(size_t)o * 32 + r_] = hb[(size_t)o * LORA_RANK + r_];
        DevBuf d_at = to_device(at), d_bp = to_device(bp);
        merged = DevBuf((size_t)out_f * in_f * 4);
        ok(aurora_hip_linear_ex(d_bp.p, 32, d_at.p, 32, nullptr, merged.p, in_f, nullptr, 0, src, in_f, out_f, (int)in_f, 32,
                                AURORA_F32, AURORA_ACT_NONE, -1, nullptr, 0.f, stream));
        hip_ok(hipStreamSynchronize(as_stream(stream)), "sync");   // d_at / d_bp die here
        src = merged.f();
      }
      const void* use = src;
      if (m.autocast) {
        DevBuf h((size_t)out_f * in_f * 2);
        ok(aurora_hip_convert(src, h.p, out_f * in_f, AURORA_F32, stream));
        hip_ok(hipStreamSynchronize(as_stream(stream)), "sync");
        use = h.p;
        set.own.push_back(std::move(h));
      } else if (key >= 0) {
        set.own.push_back(std::move(merged));
      }
      (which == 0 ? set.qkv : set.proj).push_back(use);
    }
  }
  // "all" mode: keep base + the three most recent sets
  while (m.attn_sets.size() > 3) {
    bool erased = false;
    for (auto jt = m.attn_sets.begin(); jt != m.attn_sets.end(); ++jt)
      if (jt->first != -1) { m.attn_sets.erase(jt); erased = true; break; }
    if (!erased) break;
  }
  return m.attn_sets.emplace(key, std::move(set)).first->second;
}

// (D, Kpad) GEMM weight of a LevelPatchEmbed for the model's variable order and T history steps (patchembed.py:100-115):
// per-variable (D, 1, Tmax, P, P) weights cut to T and laid out (v, t, i, j) along K, zero-padded to a multiple of 32.
const float* embed_weight(Model& m, int kind, int T, int& K, int& Kpad) {
  const std::vector<std::string>* names[2] = {nullptr, &m.atmos_vars};
  std::vector<std::string> surf_all = m.surf_vars;
  surf_all.insert(surf_all.end(), m.static_vars.begin(), m.static_vars.end());
  names[0] = &surf_all;
  const std::string prefix = kind == 0 ? "encoder.surf_token_embeds.weights." : "encoder.atmos_token_embeds.weights.";
  const int V = (int)names[kind]->size(), PP = m.P * m.P;
  K = V * T * PP;
  Kpad = round_up(K, 32);
  auto key = std::make_pair(kind, T);
  auto it = m.embed_w.find(key);
  if (it != m.embed_w.end()) return it->second.f();
  std::vector<float> host((size_t)m.D * Kpad, 0.f);
  for (int v = 0; v < V; ++v) {
    const Tensor& t = m.T_(prefix + (*names[kind])[v]);   // (D, 1, Tmax, P, P)
    REQUIRE(t.shape.size() == 5 && t.shape[0] == m.D && t.shape[2] >= T && t.shape[3] == m.P, "bad patch-embed weight shape");
    const int64_t Tmax = t.shape[2];
    std::vector<float> wv((size_t)t.numel);
    hip_ok(hipMemcpy(wv.data(), t.f(), wv.size() * 4, hipMemcpyDeviceToHost), "download");
    for (int d = 0; d < m.D; ++d)
      for (int tt = 0; tt < T; ++tt)
        memcpy(&host[(size_t)d * Kpad + ((size_t)v * T + tt) * PP], &wv[((size_t)d * Tmax + tt) * PP], PP * sizeof(float));
  }
  float l1 = 1e-6f, wmax = 0.f;
  for (int d = 0; d < m.D; ++d) {
    float sum = 0.f;
    for (int k = 0; k < Kpad; ++k) {
      const float a = fabsf(host[(size_t)d * Kpad + k]);
      sum += a;
      wmax = std::max(wmax, a);
    }
    l1 = std::max(l1, sum);
  }
  DevBuf b = to_device(host);
  m.embed_l1[key] = l1;
  // the fp16-pair form for the guarded two-term kernel (the raw, normalised inputs are bounded only by the guard)
  if (bounded_mode() == 2 && wmax < 1000.f && m.D % 256 == 0 && Kpad >= 96) {
    DevBuf sp((size_t)m.D * Kpad * 4);
    if (aurora_hip_split_f16(b.f(), Kpad, sp.p, Kpad, m.D, Kpad, 64.0f, nullptr) != AURORA_OK)
      throw std::runtime_error(aurora_hip_last_error());
    hip_ok(hipDeviceSynchronize(), "split embed weights");
    m.embed_ws.emplace(key, std::move(sp));
  }
  return m.embed_w.emplace(key, std::move(b)).first->second.f();
}


struct StepIO {
  const aurora_hip_step_io* io;
  int B, T, H, W;
};

float* run_step(Model& m, const StepIO& s, void* stream) {
  Launcher L{m, stream};
  Arena& A = m.arena;
  A.top = 0;
  const aurora_hip_step_io& io = *s.io;
  const int B = s.B, T = s.T, P = m.P, D = m.D, Hp = m.Hp, Wp = m.Wp, Cl = m.Cl, C = m.n_levels;
  const int64_t Lp = (int64_t)Hp * Wp;          // patches per level
  const int PP = P * P;
  const int n_surf = (int)m.surf_vars.size(), n_static = (int)m.static_vars.size(), n_atmos = (int)m.atmos_vars.size();
  const float* st = m.stats.f();

  // ================= encoder (encoder.py:198-366) =================
  const size_t enc_mark = A.top;
  float* x_f = (float*)A.take((size_t)B * Cl * Lp * D * 4);                      // residual stream of stage 0 (fp32)
  void* x_b = m.autocast ? A.take((size_t)B * Cl * Lp * D * 2) : nullptr;        // bf16 shadow (GEMM operand)
  const size_t after_x = A.top;
  {
    // ---- surface level: normalise + unfold, patch embedding, MLP, LayerNorm ----
    int K_s, Kpad_s;
    const float* w_s = embed_weight(m, 0, T, K_s, Kpad_s);
    float* A_s = (float*)A.take((size_t)B * Lp * Kpad_s * 4);
    std::vector<aurora_patch_var> descs;
    for (int v = 0; v < n_surf; ++v)
      descs.push_back({io.surf[v], io.surf_strides[0], io.surf_strides[1], 0, io.surf_strides[2], io.surf_strides[3],
                       st + m.surf_stat_off[v], st + m.surf_stat_off[v] + 2, 0, 0.f, 0.f, 0.f});
    for (int v = 0; v < n_static; ++v)
      descs.push_back({io.stat[v], 0, 0, 0, io.static_strides[0], io.static_strides[1], st + m.static_stat_off[v],
                       st + m.static_stat_off[v] + 2, 0, 0.f, 0.f, 0.f});
    for (size_t i = 0; i < descs.size(); i += 32)
      timed(m, stream, K_PATCHIFY, 0.0, [&] {
        return aurora_hip_patchify(descs.data() + i, (int)std::min<size_t>(32, descs.size() - i), A_s, Kpad_s, (int)i * T * PP,
                                   K_s, B, T, 1, Hp, Wp, P, AURORA_F32, stream);
      });
    float* xs0 = (float*)A.take((size_t)B * Lp * D * 4);
    const int hid_s = (int)m.T_("encoder.surf_mlp.net.0.weight").shape[0];
    float* hid = (float*)A.take((size_t)B * Lp * hid_s * 4);
    float* y = (float*)A.take((size_t)B * Lp * D * 4);
    // Guarded like the atmospheric chain: max |normalised input| once, then every linear takes two fp16 terms iff the
    // bound that word implies for ITS activation operand is inside fp16's range -- embedding: the input itself; first
    // MLP linear: |xs0| <= l1_e * w + c; second: |GELU(h)| <= |h| <= l1_0 * (l1_e * w + c) + |b0| -- else three bf16 terms.
    const auto skey = std::make_pair(0, T);
    const void* w_s_s = m.embed_ws.count(skey) ? m.embed_ws.at(skey).p : nullptr;
    if (m.surf_chain && w_s_s) {
      float* word = m.ctx_max.f() + 2;
      timed(m, stream, K_ABSMAX, 0.0, [&] { return aurora_hip_absmax(A_s, (int64_t)B * Lp * Kpad_s, word, stream); });
      const float l1e = m.embed_l1.at(skey);
      const float lim_e = F16_SAFE, lim_0 = (F16_SAFE - m.surf_c) / l1e, lim_2 = ((F16_SAFE - m.surf_b0) / m.surf_l1_0 - m.surf_c) / l1e;
      auto pair = [&](const float* a, int64_t lda, const float* wf, const void* ws, const float* bias, float* c, int64_t ldc, int N_,
                      int K_, int act, const float* res, float limit) {
        L.linear(a, lda, ws, K_, bias, c, ldc, B * Lp, N_, K_, AURORA_F32, act, nullptr, 0, res, 0, 2 | AURORA_F32_W_SPLIT, word, limit);
        L.linear(a, lda, wf, K_, bias, c, ldc, B * Lp, N_, K_, AURORA_F32, act, nullptr, 0, res, 0, 1, word, limit);
      };
      pair(A_s, Kpad_s, w_s, w_s_s, m.W("encoder.surf_token_embeds.bias"), xs0, D, D, Kpad_s, 0, m.W("encoder.surf_level_encoding"), lim_e);
      pair(xs0, D, m.W("encoder.surf_mlp.net.0.weight"), m.surf_w0_s.p, m.W("encoder.surf_mlp.net.0.bias"), hid, hid_s, hid_s, D,
           AURORA_ACT_GELU, nullptr, lim_0);
      pair(hid, hid_s, m.W("encoder.surf_mlp.net.2.weight"), m.surf_w2_s.p, m.W("encoder.surf_mlp.net.2.bias"), y, D, D, hid_s, 0,
           nullptr, lim_2);
    } else {
      L.linear(A_s, Kpad_s, w_s, Kpad_s, m.W("encoder.surf_token_embeds.bias"), xs0, D, B * Lp, D, Kpad_s, AURORA_F32, 0, nullptr,
               0, m.W("encoder.surf_level_encoding"), 0);
      L.linear(xs0, D, m.W("encoder.surf_mlp.net.0.weight"), D, m.W("encoder.surf_mlp.net.0.bias"), hid, hid_s, B * Lp, hid_s, D,
               AURORA_F32, AURORA_ACT_GELU);
      L.linear(hid, hid_s, m.W("encoder.surf_mlp.net.2.weight"), hid_s, m.W("encoder.surf_mlp.net.2.bias"), y, D, B * Lp, D, hid_s,
               AURORA_F32);
    }
    L.layernorm(y, D, m.W("encoder.surf_norm.weight"), m.W("encoder.surf_norm.bias"), xs0, D, 0, y, D, nullptr, 0, B * Lp, D, 1e-5f,
                AURORA_F32);   // xs0 + LN(MLP(xs0)), in place
    const float* xs1 = y;

    // ---- atmospheric levels ----
    int K_a, Kpad_a;
    const float* w_a = embed_weight(m, 1, T, K_a, Kpad_a);
    float* A_a = (float*)A.take((size_t)C * B * Lp * Kpad_a * 4);
    std::vector<aurora_patch_var> adescs;
    for (int v = 0; v < n_atmos; ++v)
      adescs.push_back({io.atmos[v], io.atmos_strides[0], io.atmos_strides[1], io.atmos_strides[2], io.atmos_strides[3],
                        io.atmos_strides[4], st + m.atmos_stat_off[v], st + m.atmos_stat_off[v] + 2 * C, 0, 0.f, 0.f, 0.f});
    for (size_t i = 0; i < adescs.size(); i += 32)
      timed(m, stream, K_PATCHIFY, 0.0, [&] {
        return aurora_hip_patchify(adescs.data() + i, (int)std::min<size_t>(32, adescs.size() - i), A_a, Kpad_a,
                                   (int)i * T * PP, K_a, B, T, C, Hp, Wp, P, AURORA_F32, stream);
      });
    float* xa = (float*)A.take((size_t)C * B * Lp * D * 4);
    const int64_t R = (int64_t)B * Lp;
    // The patch embedding and the level aggregation's to_kv as one guarded chain: max |normalised input| is measured
    // once (a third of the bytes of the embeddings the resampler would otherwise scan), and if it is inside fp16's range
    // -- together with the bound it implies for the embeddings, |x| <= l1 * max|input| + max|bias| -- the embedding runs
    // on two fp16 terms and writes fp16 PAIRS, which to_kv multiplies without splitting anything; otherwise both run on
    // three bf16 terms over fp32 buffers.  One word and one limit decide format and kernels together.
    const auto ekey = std::make_pair(1, T);
    const void* w_a_s = m.embed_ws.count(ekey) ? m.embed_ws.at(ekey).p : nullptr;
    bool chain = w_a_s != nullptr;
    for (const auto& ly : m.enc_rs.layers) chain = chain && ly.f16_mode == 2 && ly.to_kv_s != nullptr;
    CtxGuard cg{};
    if (chain) {
      float* word = m.ctx_max.f() + 1;
      timed(m, stream, K_ABSMAX, 0.0, [&] { return aurora_hip_absmax(A_a, (int64_t)C * R * Kpad_a, word, stream); });
      const float l1 = m.embed_l1.at(ekey), cb = m.enc_bias_max;
      cg = CtxGuard{word, l1, cb, std::min(F16_SAFE, (F16_SAFE - cb) / l1), true};
    }
    for (int c = 0; c < C; ++c) {
      const float* a_c = A_a + (size_t)c * R * Kpad_a;
      const float* b_c = m.enc_bias.f() + (size_t)c * D;
      float* x_c = xa + (size_t)c * R * D;
      if (chain) {
        L.linear(a_c, Kpad_a, w_a_s, Kpad_a, b_c, x_c, D, R, D, Kpad_a, AURORA_F32, 0, nullptr, 0, nullptr, 0,
                 2 | AURORA_F32_W_SPLIT | AURORA_F32_C_SPLIT, cg.word, cg.limit_kv);
        L.linear(a_c, Kpad_a, w_a, Kpad_a, b_c, x_c, D, R, D, Kpad_a, AURORA_F32, 0, nullptr, 0, nullptr, 0, 1, cg.word, cg.limit_kv);
      } else {
        L.linear(a_c, Kpad_a, w_a, Kpad_a, b_c, x_c, D, R, D, Kpad_a, AURORA_F32);
      }
    }

    // ---- level aggregation (Perceiver resampler over the level axis) ----
    size_t rs_mark = 0;
    float* lat = resampler(m, L, m.enc_rs, xa, (int64_t)C * R, D, m.enc_q0.f(), m.W("encoder.atmos_latents"), B, Lp, Lp, R,
                           Cl - 1, C, m.perceiver_heads, m.ln_eps, rs_mark, chain ? &cg : nullptr);

    // ---- assemble tokens + position / scale / time embeddings ----
    float* time_emb = (float*)A.take((size_t)B * D * 4);
    L.linear(m.abs_enc.f(), D, m.W("encoder.absolute_time_embed.weight"), D, m.W("encoder.absolute_time_embed.bias"), time_emb, D,
             B, D, D, AURORA_F32, 0, nullptr, 0, m.lead_emb.f(), 0);
    timed(m, stream, K_ASSEMBLE, 0.0, [&] { return aurora_hip_assemble_tokens(xs1, lat, m.pos_scale.f(), time_emb, x_f, x_b, B, Cl, Lp, D,
                                    m.autocast ? AURORA_BF16 : AURORA_F32, stream); });
  }
  A.top = after_x;   // every encoder temporary is dead
  (void)enc_mark;

  // ================= backbone (swin3d.py:884-936) =================
  const int bb = m.bb();
  const size_t es = m.bbs();
  const bool bf = m.autocast;
  const AttnSet& aw = attn_weights(m, lora_key(m, io.rollout_step), stream);
  const int n = m.n_stages;
  std::vector<float*> skips;
  size_t bi = 0;
  // x_cat (B*L0, 2*D0): decoder output | encoder stage-0 output -- allocated now so that it survives the stack
  const int64_t L0 = (int64_t)Cl * Lp;
  float* x_cat = (float*)A.take((size_t)B * L0 * 2 * D * 4);

  auto run_blocks = [&](int count, float* xf, void* xb, int stage, float* final_out, int64_t final_ld) {
    const Res res = m.stage_res[stage];
    const int64_t Ls = (int64_t)res.c * res.h * res.w, M = (int64_t)B * Ls;
    for (int k = 0; k < count; ++k, ++bi) {
      const Block& blk = m.blocks[bi];
      const int dim = blk.dim;
      const void* a_in = bf ? xb : (const void*)xf;
      const size_t mark = A.top;
      void* qkv = A.take((size_t)M * 3 * dim * es);
      L.linear(a_in, dim, aw.qkv[bi], dim, blk.qkv_b, qkv, 3 * dim, M, 3 * dim, dim, bb);
      const DevTables& tb = tables_for(m, stage, blk.shifted);
      void* ao = A.take((size_t)M * dim * es);
      // algorithmic bytes: q, k, v read + o written once over the (padded) windows
      timed(m, stream, K_WINDOW_ATTENTION, 4.0 * B * tb.n_windows * tb.n_tok * dim * es, [&] {
        return aurora_hip_window_attention(qkv, blk.qkv_b, ao, (const int32_t*)tb.tok.p,
                                           tb.has_grp ? (const uint8_t*)tb.grp.p : nullptr, B, Ls, Ls, dim, blk.heads,
                                           tb.n_windows, tb.n_tok, bb, stream);
      });
      // D = 512 under autocast: the linear, its AdaLN and the residual add are ONE launch (a workgroup owns whole rows)
      // AURORA_FUSE_LN: 0 never, 1 (default) by the fill rule below, 2 always (tests: read per step, not cached)
      const char* fuse_e = getenv("AURORA_FUSE_LN");
      const int fuse_env = fuse_e ? atoi(fuse_e) : 1;
      // (a row-owning tile is 128 rows: only when the launch fills its rounds of one tile per CU -- a latitude band's
      // 270 tiles on 256 CUs would take two rounds for the work of 1.05)
      const int64_t ln_tiles = (M + 127) / 128, cus = device_cus();
      const bool fills = (double)ln_tiles >= 0.85 * (double)(((ln_tiles + cus - 1) / cus) * cus);
      const bool fuse = bf && dim == 512 && (fuse_env == 2 || (fuse_env == 1 && fills));
      auto fused = [&](const void* a, const void* w, const float* bias, int K_, const float* gain, const float* shift, float* xo,
                       int64_t ldo, void* xbo) {
        timed(m, stream, K_LINEAR_LN, 2.0 * (double)M * dim * K_, [&] {
          return aurora_hip_linear_layernorm(a, K_, w, K_, bias, gain, shift, xf, dim, xo, ldo, xbo, dim, M, dim, K_, 1e-5f, stream);
        });
      };
      if (fuse) {
        fused(ao, aw.proj[bi], blk.proj_b, dim, blk.gain1, blk.shift1, xf, dim, xb);
      } else {
        void* y = A.take((size_t)M * dim * es);
        L.linear(ao, dim, aw.proj[bi], dim, blk.proj_b, y, dim, M, dim, dim, bb);
        L.layernorm(y, dim, blk.gain1, blk.shift1, xf, dim, 0, xf, dim, xb, dim, M, dim, 1e-5f, bb);
      }
      A.top = mark;
      void* hid = A.take((size_t)M * blk.hidden * es);
      L.linear(a_in, dim, blk.fc1_w, dim, blk.fc1_b, hid, blk.hidden, M, blk.hidden, dim, bb, AURORA_ACT_GELU);
      const bool last = final_out != nullptr && k == count - 1;
      if (fuse) {
        fused(hid, blk.fc2_w, blk.fc2_b, blk.hidden, blk.gain2, blk.shift2, last ? final_out : xf, last ? final_ld : dim,
              last ? nullptr : xb);
      } else {
        void* y2 = A.take((size_t)M * dim * es);
        L.linear(hid, blk.hidden, blk.fc2_w, blk.hidden, blk.fc2_b, y2, dim, M, dim, blk.hidden, bb);
        L.layernorm(y2, dim, blk.gain2, blk.shift2, xf, dim, 0, last ? final_out : xf, last ? final_ld : dim, last ? nullptr : xb,
                    dim, M, dim, 1e-5f, bb);
      }
      A.top = mark;
    }
  };

  float* xf = x_f;
  void* xb = x_b;
  for (int i = 0; i < n; ++i) {
    run_blocks(m.enc_depths[i], xf, xb, i, nullptr, 0);
    skips.push_back(xf);
    if (i < n - 1) {
      const Res r = m.stage_res[i];
      REQUIRE(r.h > 1 && r.w > 1, "grid (%d, %d, %d) too small to merge", r.c, r.h, r.w);
      const int dim = m.stage_dim(i);
      const int H2 = (r.h + 1) / 2, W2 = (r.w + 1) / 2;
      const int64_t M2 = (int64_t)B * r.c * H2 * W2;
      float* nf = (float*)A.take((size_t)M2 * 2 * dim * 4);
      void* nb = bf ? A.take((size_t)M2 * 2 * dim * 2) : nullptr;
      const size_t mark = A.top;
      void* mg = A.take((size_t)M2 * 4 * dim * es);
      timed(m, stream, K_MERGE_LN, 0.0, [&] { return aurora_hip_merge_ln(xf, m.merges[i].ln_w, m.merges[i].ln_b, mg, B, r.c, r.h, r.w, dim, 1e-5f, bb, stream); });
      if (bf) L.linear(mg, 4 * dim, m.merges[i].w, 4 * dim, nullptr, nb, 2 * dim, M2, 2 * dim, 4 * dim, bb, 0, nf, 2 * dim);
      else L.linear(mg, 4 * dim, m.merges[i].w, 4 * dim, nullptr, nf, 2 * dim, M2, 2 * dim, 4 * dim, bb);
      A.top = mark;
      xf = nf;
      xb = nb;
    }
  }
  for (int i = 0; i < n; ++i) {
    const int idx = n - 1 - i;
    const bool last_layer = i == n - 1;
    run_blocks(m.dec_depths[i], xf, xb, idx, last_layer ? x_cat : nullptr, 2 * D);
    if (last_layer && m.dec_depths[i] == 0 && !m.dry)
      ok(aurora_hip_copy2d(xf, D, x_cat, 2 * D, (int64_t)B * L0, D, AURORA_F32, stream));
    if (i < n - 1) {
      const Res r = m.stage_res[idx];
      const int dim = m.stage_dim(idx);
      const void* a_in = bf ? xb : (const void*)xf;
      const int crop_h = m.merge_pad[idx - 1][0], crop_w = m.merge_pad[idx - 1][1];
      const int Ho = 2 * r.h - crop_h, Wo = 2 * r.w - crop_w;
      const int64_t M1 = (int64_t)B * r.c * r.h * r.w, M2 = (int64_t)B * r.c * Ho * Wo;
      float* nf = (float*)A.take((size_t)M2 * (dim / 2) * 4);
      void* nb = bf ? A.take((size_t)M2 * (dim / 2) * 2) : nullptr;
      const size_t mark = A.top;
      void* y1 = A.take((size_t)M1 * 2 * dim * es);
      L.linear(a_in, dim, m.splits[i].w1, dim, nullptr, y1, 2 * dim, M1, 2 * dim, dim, bb);
      void* sp = A.take((size_t)M2 * (dim / 2) * es);
      timed(m, stream, K_SPLIT_LN, 0.0, [&] { return aurora_hip_split_ln(y1, m.splits[i].ln_w, m.splits[i].ln_b, sp, B, r.c, r.h, r.w, dim / 2, crop_h, crop_w, 1e-5f, bb,
                               stream); });
      // additive skip after the intermediate decoder stages (swin3d.py:930-932)
      const float* res_ = (i > 0 && i < n - 1) ? skips[idx - 1] : nullptr;
      if (bf) L.linear(sp, dim / 2, m.splits[i].w2, dim / 2, nullptr, nb, dim / 2, M2, dim / 2, dim / 2, bb, 0, nf, dim / 2, res_, dim / 2);
      else L.linear(sp, dim / 2, m.splits[i].w2, dim / 2, nullptr, nf, dim / 2, M2, dim / 2, dim / 2, bb, 0, nullptr, 0, res_, dim / 2);
      A.top = mark;
      xf = nf;
      xb = nb;
    }
  }
  timed(m, stream, K_COPY2D, 0.0, [&] { return aurora_hip_copy2d(skips[0], D, x_cat + D, 2 * D, (int64_t)B * L0, D, AURORA_F32, stream); });

  // ================= decoder (decoder.py:168-276) =================
  const int D2 = 2 * D;
  const int H = Hp * P, Wd = Wp * P;
  {
    const size_t mark = A.top;
    // ---- surface heads on latent level 0 ----
    const int n_s = n_surf * PP, ld_s = round_up(n_s, 4);
    float* y_s = (float*)A.take((size_t)B * Lp * ld_s * 4);
    for (int b = 0; b < B; ++b)
      L.linear(x_cat + (size_t)b * Cl * Lp * D2, D2, m.head_surf_w.f(), D2, m.head_surf_b.f(), y_s + (size_t)b * Lp * ld_s, ld_s, Lp,
               n_s, D2, AURORA_F32);
    std::vector<aurora_unpatch_var> ud;
    for (int v = 0; v < n_surf; ++v) {
      aurora_unpatch_var d{};
      d.dst = io.out_surf[v];
      d.loc = st + m.surf_stat_off[v];
      d.scale = st + m.surf_stat_off[v] + 1;
      d.col0 = v * PP;
      d.mod_col0 = d.angle_col0 = d.dens_col0 = -1;
      ud.push_back(d);
    }
    for (size_t i = 0; i < ud.size(); i += 32)
      timed(m, stream, K_UNPATCHIFY, 0.0, [&] {
        return aurora_hip_unpatchify(y_s, ld_s, ud.data() + i, (int)std::min<size_t>(32, ud.size() - i), B, 1, Hp, Wp, P, stream);
      });

    // ---- level de-aggregation ----
    const float* ctx = x_cat + (size_t)Lp * D2;
    float* ctx_copy = nullptr;
    if (B > 1) {   // latent levels 1.. of every batch element, made contiguous
      ctx_copy = (float*)A.take((size_t)B * (Cl - 1) * Lp * D2 * 4);
      for (int b = 0; b < B; ++b)
        timed(m, stream, K_COPY2D, 0.0, [&] {
          return aurora_hip_copy2d(x_cat + ((size_t)b * Cl * Lp + Lp) * D2, D2, ctx_copy + (size_t)b * (Cl - 1) * Lp * D2, D2,
                                   (int64_t)(Cl - 1) * Lp, D2, AURORA_F32, stream);
        });
      ctx = ctx_copy;
    }
    size_t rs_mark = 0;
    float* lat = resampler(m, L, m.dec_rs, ctx, (int64_t)B * (Cl - 1) * Lp, D2, m.dec_q.f(), m.dec_queries.f(), B, Lp,
                           (int64_t)(Cl - 1) * Lp, Lp, C, Cl - 1, m.perceiver_heads, m.ln_eps, rs_mark);
    const int n_a = n_atmos * PP, ld_a = round_up(n_a, 4);
    float* y_a = (float*)A.take((size_t)B * Lp * C * ld_a * 4);
    L.linear(lat, D2, m.head_atmos_w.f(), D2, m.head_atmos_b.f(), y_a, ld_a, (int64_t)B * Lp * C, n_a, D2, AURORA_F32);
    std::vector<aurora_unpatch_var> ad;
    for (int v = 0; v < n_atmos; ++v) {
      aurora_unpatch_var d{};
      d.dst = io.out_atmos[v];
      d.loc = st + m.atmos_stat_off[v];
      d.scale = st + m.atmos_stat_off[v] + C;
      d.col0 = v * PP;
      d.mod_col0 = d.angle_col0 = d.dens_col0 = -1;
      ad.push_back(d);
    }
    for (size_t i = 0; i < ad.size(); i += 32)
      timed(m, stream, K_UNPATCHIFY, 0.0, [&] {
        return aurora_hip_unpatchify(y_a, ld_a, ad.data() + i, (int)std::min<size_t>(32, ad.size() - i), B, C, Hp, Wp, P, stream);
      });
    A.top = mark;
  }
  (void)H; (void)Wd;
  return x_cat;
}

}  // namespace

// ====================================================================================================
// C ABI
// ====================================================================================================
#define GUARDED(...)                                    \
  try {                                                 \
    __VA_ARGS__;                                        \
    return AURORA_OK;                                   \
  } catch (const Fail& f) {                             \
    return f.code;                                      \
  } catch (const std::exception& e) {                   \
    ::aurora::set_error("internal error: %s", e.what()); \
    return AURORA_E_LAUNCH;                             \
  }

extern "C" int aurora_hip_create(const aurora_hip_config* c, aurora_hip_model** out) {
  GUARDED({
    REQUIRE(c != nullptr && out != nullptr, "create: null argument");
    REQUIRE(c->n_stages >= 1 && c->n_stages <= 4, "create: 1..4 backbone stages");
    REQUIRE(c->latent_levels > 1, "At least two latent levels are required.");
    REQUIRE(c->max_history > 0, "At least one history step is required.");
    REQUIRE(c->embed_dim % 32 == 0 && c->patch_size > 0, "create: bad embed_dim / patch_size");
    std::unique_ptr<aurora_hip_model> m(new aurora_hip_model());
    m->D = c->embed_dim; m->P = c->patch_size; m->Cl = c->latent_levels; m->perceiver_heads = c->num_heads;
    m->n_stages = c->n_stages;
    int se = 0, sd = 0;
    for (int i = 0; i < c->n_stages; ++i) {
      m->enc_depths[i] = c->encoder_depths[i]; m->dec_depths[i] = c->decoder_depths[i];
      m->enc_heads[i] = c->encoder_heads[i]; m->dec_heads[i] = c->decoder_heads[i];
      se += c->encoder_depths[i]; sd += c->decoder_depths[i];
    }
    REQUIRE(se == sd, "Encoder and decoder must have the same total depth.");
    for (int a = 0; a < 3; ++a) m->window[a] = c->window[a];
    REQUIRE(m->window[0] * m->window[1] * m->window[2] <= 144, "windows of more than 144 tokens are not supported");
    REQUIRE(m->Cl % m->window[0] == 0, "latent levels must be divisible by the window's level extent");
    m->enc_depth = c->enc_depth; m->dec_depth = c->dec_depth; m->max_history = c->max_history;
    m->ln_eps = c->perceiver_ln_eps; m->timestep_hours = c->timestep_hours;
    m->stabilise = c->stabilise_level_agg != 0; m->use_lora = c->use_lora != 0;
    m->lora_steps = c->lora_steps; m->lora_mode = c->lora_mode; m->autocast = c->autocast != 0;
    REQUIRE(m->lora_mode >= 0 && m->lora_mode <= 2, "create: lora_mode must be 0 (single), 1 (from_second) or 2 (all)");
    for (int i = 0; i < c->n_surf; ++i) m->surf_vars.push_back(c->surf_vars[i]);
    for (int i = 0; i < c->n_static; ++i) m->static_vars.push_back(c->static_vars[i]);
    for (int i = 0; i < c->n_atmos; ++i) m->atmos_vars.push_back(c->atmos_vars[i]);
    REQUIRE(!m->surf_vars.empty() && !m->atmos_vars.empty(), "create: variable lists must not be empty");
    build_blocks(*m);
    m->ctx_max = DevBuf(16);
    *out = m.release();
  })
}

extern "C" void aurora_hip_destroy(aurora_hip_model* m) { delete m; }

extern "C" int aurora_hip_pack_weights(aurora_hip_model* m, const char* name, const void* data, const int64_t* shape, int ndim,
                                       int dtype, int on_device) {
  GUARDED({
    REQUIRE(m && name && data && ndim >= 0 && ndim <= 8, "pack_weights: bad argument");
    REQUIRE(dtype == AURORA_F32, "pack_weights: parameters must be float32 (the engine keeps fp32 masters)");
    Tensor t;
    t.numel = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); t.numel *= shape[i]; }
    t.buf = DevBuf((size_t)t.numel * 4);
    hip_ok(hipMemcpy(t.buf.p, data, (size_t)t.numel * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice),
           "pack_weights copy");
    m->w[name] = std::move(t);
    m->finalized = false;
  })
}

// ---- packed weight files -----------------------------------------------------------------------------
// One self-describing binary that any host can read without pickle / torch:
//   "AURORAHIP1\0" | u32 n_entries | per entry: u32 name_len, name, u32 dtype (0 f32, 1 bf16), u32 ndim, i64 shape[ndim],
//   u64 n_bytes, raw little-endian data
// Saved by a FINALIZED handle: the large backbone matrices (MLP, merge / split, and the attention projections of models
// without LoRA) are written in bf16 when the handle runs the bf16 backbone -- exactly the bits the GEMMs consume --
// everything else as the fp32 master.  1.3 B parameters: 2.6 GB instead of 5 GB.
namespace {
const char PACK_MAGIC[] = "AURORAHIP1";

bool backbone_matrix(const Model& m, const std::string& name) {
  if (name.rfind("backbone.", 0) != 0 || name.size() < 7 || name.compare(name.size() - 7, 7, ".weight") != 0) return false;
  for (const char* tag : {".mlp.fc1.", ".mlp.fc2.", ".downsample.reduction.", ".upsample.lin1.", ".upsample.lin2."})
    if (name.find(tag) != std::string::npos) return true;
  if (!m.use_lora && (name.find(".attn.qkv.") != std::string::npos || name.find(".attn.proj.") != std::string::npos)) return true;
  return false;
}
}  // namespace

extern "C" int aurora_hip_save_packed(aurora_hip_model* mp, const char* path, void* stream) {
  GUARDED({
    REQUIRE(mp && path, "save_packed: null argument");
    Model& m = *mp;
    FILE* f = fopen(path, "wb");
    REQUIRE(f != nullptr, "save_packed: cannot open '%s' for writing", path);
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
    auto put = [&](const void* p, size_t n) { REQUIRE(fwrite(p, 1, n, f) == n, "save_packed: short write"); };
    put(PACK_MAGIC, sizeof(PACK_MAGIC));
    const uint32_t n_entries = (uint32_t)(m.w.size() + m.w16.size());
    put(&n_entries, 4);
    std::vector<char> host;
    auto entry = [&](const std::string& name, const Tensor& t, uint32_t dtype, const void* dev, size_t bytes) {
      const uint32_t len = (uint32_t)name.size(), nd = (uint32_t)t.shape.size();
      put(&len, 4); put(name.data(), len); put(&dtype, 4); put(&nd, 4);
      for (int64_t d : t.shape) put(&d, 8);
      const uint64_t nb = bytes;
      put(&nb, 8);
      host.resize(bytes);
      hip_ok(hipMemcpy(host.data(), dev, bytes, hipMemcpyDeviceToHost), "save_packed download");
      put(host.data(), bytes);
    };
    for (const auto& kv : m.w) {
      if (m.autocast && backbone_matrix(m, kv.first)) {
        DevBuf h((size_t)kv.second.numel * 2);
        ok(aurora_hip_convert(kv.second.f(), h.p, kv.second.numel, AURORA_F32, stream));
        hip_ok(hipStreamSynchronize(as_stream(stream)), "save_packed");
        entry(kv.first, kv.second, AURORA_BF16, h.p, (size_t)kv.second.numel * 2);
      } else {
        entry(kv.first, kv.second, AURORA_F32, kv.second.f(), (size_t)kv.second.numel * 4);
      }
    }
    for (const auto& kv : m.w16) entry(kv.first, kv.second, AURORA_BF16, kv.second.buf.p, (size_t)kv.second.numel * 2);
  })
}

extern "C" int aurora_hip_load_packed(aurora_hip_model* mp, const char* path) {
  GUARDED({
    REQUIRE(mp && path, "load_packed: null argument");
    Model& m = *mp;
    FILE* f = fopen(path, "rb");
    REQUIRE(f != nullptr, "load_packed: cannot open '%s'", path);
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
    auto get = [&](void* p, size_t n) { REQUIRE(fread(p, 1, n, f) == n, "load_packed: truncated file"); };
    char magic[sizeof(PACK_MAGIC)];
    get(magic, sizeof(magic));
    REQUIRE(memcmp(magic, PACK_MAGIC, sizeof(PACK_MAGIC)) == 0, "load_packed: '%s' is not a packed aurora_hip weight file", path);
    uint32_t n_entries = 0;
    get(&n_entries, 4);
    std::vector<char> host;
    for (uint32_t e = 0; e < n_entries; ++e) {
      uint32_t len = 0, dtype = 0, nd = 0;
      get(&len, 4);
      REQUIRE(len < 4096, "load_packed: corrupt entry");
      std::string name(len, '\0');
      get(&name[0], len);
      get(&dtype, 4); get(&nd, 4);
      REQUIRE(nd <= 8 && dtype <= 1, "load_packed: corrupt entry '%s'", name.c_str());
      Tensor t;
      t.numel = 1;
      for (uint32_t i = 0; i < nd; ++i) { int64_t d; get(&d, 8); t.shape.push_back(d); t.numel *= d; }
      uint64_t nb = 0;
      get(&nb, 8);
      REQUIRE(nb == (uint64_t)t.numel * (dtype == AURORA_F32 ? 4 : 2), "load_packed: size mismatch in '%s'", name.c_str());
      host.resize(nb);
      get(host.data(), nb);
      t.buf = DevBuf(nb);
      upload(t.buf.p, host.data(), nb);
      (dtype == AURORA_F32 ? m.w : m.w16)[name] = std::move(t);
    }
    m.finalized = false;
  })
}

extern "C" int aurora_hip_finalize(aurora_hip_model* mp, void* stream) {
  GUARDED({
    REQUIRE(mp != nullptr, "finalize: null model");
    Model& m = *mp;
    m.keep.clear(); m.attn_sets.clear(); m.embed_w.clear(); m.embed_ws.clear(); m.embed_l1.clear(); m.merges.clear(); m.splits.clear();
    Launcher L{m, stream};
    const int D = m.D;
    {   // ---- surface MLP: constants of its guarded two-term chain ----
      auto host_of = [&](const std::string& name) {
        const Tensor& t = m.T_(name);
        std::vector<float> h((size_t)t.numel);
        hip_ok(hipMemcpy(h.data(), t.f(), h.size() * 4, hipMemcpyDeviceToHost), "download");
        return h;
      };
      auto amax = [](const std::vector<float>& h) { float mx = 0.f; for (float v : h) mx = std::max(mx, fabsf(v)); return mx; };
      const std::vector<float> w0 = host_of("encoder.surf_mlp.net.0.weight"), w2 = host_of("encoder.surf_mlp.net.2.weight");
      const Tensor& t0 = m.T_("encoder.surf_mlp.net.0.weight");
      const int64_t N0 = t0.shape[0], K0 = t0.shape[1];
      m.surf_l1_0 = 1e-6f;
      for (int64_t r = 0; r < N0; ++r) {
        float sum = 0.f;
        for (int64_t k = 0; k < K0; ++k) sum += fabsf(w0[(size_t)r * K0 + k]);
        m.surf_l1_0 = std::max(m.surf_l1_0, sum);
      }
      m.surf_b0 = amax(host_of("encoder.surf_mlp.net.0.bias"));
      m.surf_c = amax(host_of("encoder.surf_token_embeds.bias")) + amax(host_of("encoder.surf_level_encoding"));
      auto eligible = [](int64_t N, int64_t K) { return N % 256 == 0 && K % 32 == 0 && K >= 96; };
      m.surf_chain = bounded_mode() == 2 && amax(w0) < 1000.f && amax(w2) < 1000.f &&
                     eligible(N0, K0) && eligible(K0, N0);
      m.surf_w0_s = DevBuf();
      m.surf_w2_s = DevBuf();
      if (m.surf_chain) {
        m.surf_w0_s = DevBuf((size_t)N0 * K0 * 4);
        m.surf_w2_s = DevBuf((size_t)N0 * K0 * 4);
        if (aurora_hip_split_f16(t0.f(), K0, m.surf_w0_s.p, K0, N0, (int)K0, 64.0f, nullptr) != AURORA_OK ||
            aurora_hip_split_f16(m.T_("encoder.surf_mlp.net.2.weight").f(), N0, m.surf_w2_s.p, N0, K0, (int)N0, 64.0f, nullptr) != AURORA_OK)
          throw std::runtime_error(aurora_hip_last_error());
        hip_ok(hipDeviceSynchronize(), "split surface MLP weights");
      }
    }
    // ---- AdaLN modulation of every block: lead time -> time_mlp -> stacked modulation linears (film.py:38-49) ----
    std::vector<float> lead((size_t)D);
    const double hours = (double)(float)m.timestep_hours;
    fourier(LEAD_TIME, &hours, 1, D, lead.data());
    DevBuf d_lead = to_device(lead);
    DevBuf t1((size_t)D * 4), silu_c((size_t)D * 4);
    L.linear(d_lead.p, D, m.W("backbone.time_mlp.0.weight"), D, m.W("backbone.time_mlp.0.bias"), t1.p, D, 1, D, D, AURORA_F32,
             AURORA_ACT_SILU);
    L.linear(t1.p, D, m.W("backbone.time_mlp.2.weight"), D, m.W("backbone.time_mlp.2.bias"), silu_c.p, D, 1, D, D, AURORA_F32,
             AURORA_ACT_SILU);   // SiLU(c): the only way c is ever used
    int64_t rows = 0;
    for (const Block& b : m.blocks) rows += 4 * b.dim;
    DevBuf w_all((size_t)rows * D * 4), b_all((size_t)rows * 4);
    int64_t off = 0;
    for (const Block& b : m.blocks)
      for (const char* nrm : {".norm1", ".norm2"}) {
        const std::string nm = b.prefix + nrm + ".ln_modulation.1";
        hip_ok(hipMemcpy(w_all.f() + off * D, m.W(nm + ".weight"), (size_t)2 * b.dim * D * 4, hipMemcpyDeviceToDevice), "copy");
        hip_ok(hipMemcpy(b_all.f() + off, m.W(nm + ".bias"), (size_t)2 * b.dim * 4, hipMemcpyDeviceToDevice), "copy");
        off += 2 * b.dim;
      }
    m.mod = DevBuf((size_t)rows * 4);
    L.linear(silu_c.p, D, w_all.p, D, b_all.f(), m.mod.p, rows, 1, (int)rows, D, AURORA_F32);
    off = 0;
    for (Block& b : m.blocks) {   // chunk(2): shift first, then scale (film.py:48); scale_bias is 0 in every config
      b.shift1 = m.mod.f() + off; b.gain1 = m.mod.f() + off + b.dim; off += 2 * b.dim;
      b.shift2 = m.mod.f() + off; b.gain2 = m.mod.f() + off + b.dim; off += 2 * b.dim;
      int64_t hidden = 0;
      b.fc1_w = compute_weight(m, b.prefix + ".mlp.fc1.weight", stream, &hidden);
      b.hidden = (int)hidden;
      b.fc2_w = compute_weight(m, b.prefix + ".mlp.fc2.weight", stream);
      b.fc1_b = m.W(b.prefix + ".mlp.fc1.bias"); b.fc2_b = m.W(b.prefix + ".mlp.fc2.bias");
      b.qkv_b = m.W(b.prefix + ".attn.qkv.bias"); b.proj_b = m.W(b.prefix + ".attn.proj.bias");
    }
    for (int i = 0; i + 1 < m.n_stages; ++i) {
      const std::string p = "backbone.encoder_layers." + std::to_string(i) + ".downsample";
      m.merges.push_back({compute_weight(m, p + ".reduction.weight", stream), m.W(p + ".norm.weight"), m.W(p + ".norm.bias")});
      const std::string q = "backbone.decoder_layers." + std::to_string(i) + ".upsample";
      m.splits.push_back({compute_weight(m, q + ".lin1.weight", stream), compute_weight(m, q + ".lin2.weight", stream),
                          m.W(q + ".norm.weight"), m.W(q + ".norm.bias")});
    }
    attn_weights(m, -1, stream);
    // ---- encoder / decoder constants that depend on parameters only ----
    m.lead_emb = DevBuf((size_t)D * 4);
    L.linear(d_lead.p, D, m.W("encoder.lead_time_embed.weight"), D, m.W("encoder.lead_time_embed.bias"), m.lead_emb.p, D, 1, D, D,
             AURORA_F32);
    m.enc_rs = pack_resampler(m, "encoder.level_agg", m.enc_depth, m.perceiver_heads);
    m.dec_rs = pack_resampler(m, "decoder.level_decoder", m.dec_depth, m.perceiver_heads);
    const auto& l0 = m.enc_rs.layers[0];
    const int n_lat = m.Cl - 1;
    m.enc_q0 = DevBuf((size_t)n_lat * l0.inner * 4);
    L.linear(m.W("encoder.atmos_latents"), D, l0.to_q, D, nullptr, m.enc_q0.p, l0.inner, n_lat, l0.inner, D, AURORA_F32);
    if (l0.ln_q_w)
      L.layernorm(m.enc_q0.p, l0.inner, l0.ln_q_w, l0.ln_q_b, nullptr, 0, 0, m.enc_q0.f(), l0.inner, nullptr, 0, n_lat, l0.inner,
                  1e-5f, AURORA_F32);
    // ---- decoder heads, fused over the variables (V fastest inside a patch is handled by unpatchify's col0) ----
    auto fuse_heads = [&](const char* kind, const std::vector<std::string>& names, DevBuf& wd, DevBuf& bd) {
      const int PP = m.P * m.P, D2 = 2 * D;
      wd = DevBuf((size_t)names.size() * PP * D2 * 4);
      bd = DevBuf((size_t)names.size() * PP * 4);
      for (size_t v = 0; v < names.size(); ++v) {
        const std::string p = std::string("decoder.") + kind + "_heads." + names[v];
        const Tensor& wt = m.T_(p + ".weight");
        REQUIRE(wt.shape.size() == 2 && wt.shape[0] == PP && wt.shape[1] == D2, "bad head weight shape for %s", p.c_str());
        hip_ok(hipMemcpy(wd.f() + v * PP * D2, wt.f(), (size_t)PP * D2 * 4, hipMemcpyDeviceToDevice), "copy");
        hip_ok(hipMemcpy(bd.f() + v * PP, m.W(p + ".bias"), (size_t)PP * 4, hipMemcpyDeviceToDevice), "copy");
      }
    };
    fuse_heads("surf", m.surf_vars, m.head_surf_w, m.head_surf_b);
    fuse_heads("atmos", m.atmos_vars, m.head_atmos_w, m.head_atmos_b);
    hip_ok(hipStreamSynchronize(as_stream(stream)), "finalize sync");   // temporaries above die here
    m.finalized = true;
  })
}

extern "C" int aurora_hip_pos_scale_encoding(const double* lat, const double* lon, int n_lat, int n_lon, int patch_size,
                                             int embed_dim, float* pos_out, float* scale_out) {
  GUARDED({
    REQUIRE(lat && lon && pos_out && scale_out, "pos_scale_encoding: null argument");
    REQUIRE(patch_size > 0 && n_lon % patch_size == 0 && n_lat >= patch_size && embed_dim % 4 == 0,
            "pos_scale_encoding: bad grid %d x %d for patch size %d / embed_dim %d", n_lat, n_lon, patch_size, embed_dim);
    pos_scale_tables(lat, lon, n_lat / patch_size, n_lon / patch_size, patch_size, embed_dim, pos_out, scale_out);
  })
}

extern "C" int aurora_hip_precompute(aurora_hip_model* mp, const aurora_hip_grid* g, void* stream) {
  GUARDED({
    REQUIRE(mp && g, "precompute: null argument");
    Model& m = *mp;
    REQUIRE(m.finalized, "precompute: call aurora_hip_finalize after packing the weights");
    Launcher L{m, stream};
    const int P = m.P, D = m.D;
    REQUIRE(g->n_lon % P == 0, "Width of the data must be a multiple of the patch size.");
    REQUIRE(g->n_lat % P == 0 || g->n_lat % P == 1, "There can at most be one latitude too many.");
    const int H = g->n_lat - g->n_lat % P, W = g->n_lon;
    m.n_lat = H; m.n_lon = W; m.Hp = H / P; m.Wp = W / P;
    const int64_t Lp = (int64_t)m.Hp * m.Wp;
    // ---- stage resolutions (swin3d.py:868-882) ----
    m.stage_res.clear(); m.merge_pad.clear(); m.tables.clear();
    m.stage_res.push_back({m.Cl, m.Hp, m.Wp});
    for (int s = 1; s < m.n_stages; ++s) {
      const Res r = m.stage_res.back();
      m.merge_pad.push_back({r.h % 2, r.w % 2});
      m.stage_res.push_back({r.c, (r.h + r.h % 2) / 2, (r.w + r.w % 2) / 2});
    }
    m.merge_pad.push_back({0, 0});
    // ---- position / scale encodings of the patch grid (posencoding.py:61-192) ----
    std::vector<float> pos((size_t)Lp * D), scale((size_t)Lp * D);
    if (g->pos_encoding && g->scale_encoding) {
      memcpy(pos.data(), g->pos_encoding, pos.size() * 4);
      memcpy(scale.data(), g->scale_encoding, scale.size() * 4);
    } else {
      REQUIRE(g->lat && g->lon, "precompute: latitudes / longitudes (or the encodings themselves) are required");
      pos_scale_tables(g->lat, g->lon, m.Hp, m.Wp, P, D, pos.data(), scale.data());
    }
    {
      DevBuf d_pos = to_device(pos), d_scale = to_device(scale), pe((size_t)Lp * D * 4);
      m.pos_scale = DevBuf((size_t)Lp * D * 4);
      L.linear(d_pos.p, D, m.W("encoder.pos_embed.weight"), D, m.W("encoder.pos_embed.bias"), pe.p, D, Lp, D, D, AURORA_F32);
      L.linear(d_scale.p, D, m.W("encoder.scale_embed.weight"), D, m.W("encoder.scale_embed.bias"), m.pos_scale.p, D, Lp, D, D,
               AURORA_F32, 0, nullptr, 0, pe.f(), D);
      hip_ok(hipStreamSynchronize(as_stream(stream)), "precompute sync");
    }
    // ---- pressure levels: per-level patch-embedding bias, decoder queries (encoder.py:318-330, decoder.py:176-200) ----
    const int C = g->n_levels;
    REQUIRE(C >= 1 && g->levels, "precompute: pressure levels are required");
    m.n_levels = C;
    std::vector<double> lv(C);
    for (int c = 0; c < C; ++c) lv[c] = g->levels_float32 ? (double)(float)g->levels[c] : g->levels[c];
    {
      std::vector<float> enc((size_t)C * D), dec((size_t)C * 2 * D);
      fourier(LEVELS, lv.data(), C, D, enc.data());
      fourier(LEVELS, lv.data(), C, 2 * D, dec.data());
      DevBuf d_enc = to_device(enc), d_dec = to_device(dec);
      m.enc_bias = DevBuf((size_t)C * D * 4);
      L.linear(d_enc.p, D, m.W("encoder.atmos_levels_embed.weight"), D, m.W("encoder.atmos_levels_embed.bias"), m.enc_bias.p, D, C,
               D, D, AURORA_F32, 0, nullptr, 0, m.W("encoder.atmos_token_embeds.bias"), 0);
      m.dec_queries = DevBuf((size_t)C * 2 * D * 4);
      L.linear(d_dec.p, 2 * D, m.W("decoder.atmos_levels_embed.weight"), 2 * D, m.W("decoder.atmos_levels_embed.bias"),
               m.dec_queries.p, 2 * D, C, 2 * D, 2 * D, AURORA_F32);
      const auto& d0 = m.dec_rs.layers[0];
      m.dec_q = DevBuf((size_t)C * d0.inner * 4);
      L.linear(m.dec_queries.p, 2 * D, d0.to_q, 2 * D, nullptr, m.dec_q.p, d0.inner, C, d0.inner, 2 * D, AURORA_F32);
      if (d0.ln_q_w)
        L.layernorm(m.dec_q.p, d0.inner, d0.ln_q_w, d0.ln_q_b, nullptr, 0, 0, m.dec_q.f(), d0.inner, nullptr, 0, C, d0.inner, 1e-5f,
                    AURORA_F32);
      hip_ok(hipStreamSynchronize(as_stream(stream)), "precompute sync");
      std::vector<float> eb((size_t)C * D);
      hip_ok(hipMemcpy(eb.data(), m.enc_bias.p, eb.size() * 4, hipMemcpyDeviceToHost), "download");
      m.enc_bias_max = 0.f;
      for (float v : eb) m.enc_bias_max = std::max(m.enc_bias_max, fabsf(v));
    }
    // ---- normalisation statistics: loc, scale, 1/scale (computed in fp64) per variable (and level) ----
    const int ns = (int)m.surf_vars.size(), nst = (int)m.static_vars.size(), na = (int)m.atmos_vars.size();
    REQUIRE(g->surf_loc && g->surf_scale && g->atmos_loc && g->atmos_scale && (nst == 0 || (g->static_loc && g->static_scale)),
            "precompute: normalisation statistics are required");
    std::vector<float> hs;
    m.surf_stat_off.clear(); m.static_stat_off.clear(); m.atmos_stat_off.clear();
    auto push1 = [&](std::vector<size_t>& offs, double loc, double sc) {
      offs.push_back(hs.size());
      hs.push_back((float)loc); hs.push_back((float)sc); hs.push_back((float)(1.0 / sc)); hs.push_back(0.f);
    };
    for (int v = 0; v < ns; ++v) push1(m.surf_stat_off, g->surf_loc[v], g->surf_scale[v]);
    for (int v = 0; v < nst; ++v) push1(m.static_stat_off, g->static_loc[v], g->static_scale[v]);
    for (int v = 0; v < na; ++v) {
      m.atmos_stat_off.push_back(hs.size());
      for (int c = 0; c < C; ++c) hs.push_back((float)g->atmos_loc[v * C + c]);
      for (int c = 0; c < C; ++c) hs.push_back((float)g->atmos_scale[v * C + c]);
      for (int c = 0; c < C; ++c) hs.push_back((float)(1.0 / g->atmos_scale[v * C + c]));
      while (hs.size() % 4) hs.push_back(0.f);
    }
    m.stats = to_device(hs);
    m.have_grid = true;
  })
}

extern "C" int aurora_hip_set_time(aurora_hip_model* mp, const double* time_hours, int B, void* stream) {
  GUARDED({
    REQUIRE(mp && time_hours && B >= 1, "set_time: bad argument");
    Model& m = *mp;
    std::vector<double> t(B);
    // the reference converts the timestamps to a float32 tensor before expanding (encoder.py:359-362)
    for (int b = 0; b < B; ++b) t[b] = (double)(float)time_hours[b];
    const size_t bytes = (size_t)B * m.D * 4;
    if (m.abs_B < B) {
      hip_ok(hipDeviceSynchronize(), "set_time");
      m.abs_enc = DevBuf(bytes);
      m.abs_B = B;
    }
    auto& slot = m.pinned[m.pinned_next++ & 3];
    if (slot.done) hip_ok(hipEventSynchronize(slot.done), "set_time");   // the copy that used this slot four uploads ago
    else hip_ok(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming), "set_time");
    if (slot.bytes < bytes) {
      if (slot.host) (void)hipHostFree(slot.host);
      hip_ok(hipHostMalloc((void**)&slot.host, bytes, hipHostMallocDefault), "set_time");
      slot.bytes = bytes;
    }
    fourier(ABS_TIME, t.data(), B, m.D, slot.host);
    hip_ok(hipMemcpyAsync(m.abs_enc.p, slot.host, bytes, hipMemcpyHostToDevice, as_stream(stream)), "set_time");
    hip_ok(hipEventRecord(slot.done, as_stream(stream)), "set_time");
  })
}

extern "C" int aurora_hip_step(aurora_hip_model* mp, const aurora_hip_step_io* io, void* stream) {
  GUARDED({
    REQUIRE(mp && io, "step: null argument");
    Model& m = *mp;
    REQUIRE(m.finalized && m.have_grid, "step: finalize the weights and precompute the grid first");
    REQUIRE(io->B >= 1 && io->T >= 1, "step: empty batch");
    REQUIRE(io->T <= m.max_history, "%d > %d.", io->T, m.max_history);
    REQUIRE(m.abs_B >= io->B, "step: call aurora_hip_set_time for this batch first");
    REQUIRE(io->surf && io->atmos && io->out_surf && io->out_atmos && (m.static_vars.empty() || io->stat), "step: null field list");
    StepIO s{io, io->B, io->T, m.n_lat, m.n_lon};
    // LoRA sets are merged outside the dry run (they allocate and launch)
    attn_weights(m, lora_key(m, io->rollout_step), stream);
    m.dry = true;
    m.arena.peak = 0;
    try { run_step(m, s, stream); } catch (...) { m.dry = false; throw; }
    m.dry = false;
    if (m.arena.peak > m.arena.cap) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(as_stream(stream), &cap);
      REQUIRE(cap == hipStreamCaptureStatusNone, "step: the workspace must grow; run one step outside graph capture first");
      hip_ok(hipDeviceSynchronize(), "sync before growing the workspace");
      if (m.arena.base) (void)hipFree(m.arena.base);
      m.arena.base = nullptr;
      m.arena.cap = 0;
      void* p = nullptr;
      hip_ok(hipMalloc(&p, m.arena.peak), "workspace allocation");
      m.arena.base = (char*)p;
      m.arena.cap = m.arena.peak;
    }
    run_step(m, s, stream);
  })
}

extern "C" int aurora_hip_profile_begin(aurora_hip_model* m, uint32_t kind_mask) {
  GUARDED({
    REQUIRE(m != nullptr, "profile_begin: null model");
    for (auto& t : m->timed) { m->event_pool.push_back(t.e0); m->event_pool.push_back(t.e1); }
    m->timed.clear();
    m->profile_mask = kind_mask;
  })
}

extern "C" int aurora_hip_profile_end(aurora_hip_model* m, aurora_hip_profile_entry* out, int capacity, int* n_out) {
  GUARDED({
    REQUIRE(m && out && n_out && capacity >= K_COUNT, "profile_end: need room for %d entries", (int)K_COUNT);
    m->profile_mask = 0;
    hip_ok(hipDeviceSynchronize(), "profile_end");
    for (int k = 0; k < K_COUNT; ++k) out[k] = aurora_hip_profile_entry{KIND_NAMES[k], 0, 0.0, 0.0};
    for (auto& t : m->timed) {
      float ms = 0.f;
      hip_ok(hipEventElapsedTime(&ms, t.e0, t.e1), "hipEventElapsedTime");
      out[t.kind].launches += 1;
      out[t.kind].ms += ms;
      out[t.kind].work += t.work;
      m->event_pool.push_back(t.e0);
      m->event_pool.push_back(t.e1);
    }
    m->timed.clear();
    *n_out = K_COUNT;
  })
}

extern "C" int64_t aurora_hip_workspace_bytes(const aurora_hip_model* m) { return m ? (int64_t)m->arena.cap : 0; }
