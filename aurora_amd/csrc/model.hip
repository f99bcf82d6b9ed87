// The model handle of libaurora_hip.so: creation, weights, tables.
//
// aurora_hip_create / _pack_weights / _finalize / _precompute / _set_time / _step / _destroy (include/aurora_hip.h) own
// everything the reference's `Aurora.forward` (aurora/model/aurora.py:265-392) needs besides the input fields: the
// configuration, the weights (fp32 masters + bf16 backbone copies, LoRA merged per roll-out phase), the tables that
// depend on parameters / grid / levels only (AdaLN modulation, Fourier position / scale / level encodings, window
// token tables, a latitude band's halo plans), the workspace, and the SEQUENCE of kernel launches of a step (step.hip) on
// a caller-supplied stream.  No torch, no Python: any host language that can call C can run Aurora on an MI355X through
// these functions; aurora_amd's own Python `Engine` is a thin binding of them.
//
// Scope: every public model class -- Aurora, AuroraPretrained, AuroraSmallPretrained, Aurora12hPretrained, AuroraHighRes
// (any patch size / depths / history / LoRA mode, stabilised level aggregation, batch > 1), AuroraAirPollution and
// AuroraWave (level-conditioned embeddings / heads, dynamic and static inputs, feature combiners, difference prediction,
// second decoder Perceiver, NaN / density / angle channels) -- on one device or as one latitude band of a forecast
// sharded over several (aurora_hip_set_band).
//
// Host code only; every launch goes through the operator ABI of this same library.
#include <stdarg.h>

#include <algorithm>
#include <exception>

#include "model.h"

namespace aurora {

namespace {

constexpr double PI = 3.14159265358979323846;
constexpr int LORA_RANK = 8;
const char* const DYNAMIC_NAMES[6] = {"tod_cos", "tod_sin", "dow_cos", "dow_sin", "doy_cos", "doy_sin"};   // encoder.py:246

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

}  // namespace

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

int bounded_mode() {
  static const int mode = getenv("AURORA_F32_GEMM") ? -1 : 2;
  return mode;
}

std::string level_to_str(double level) {
  const double value = round(level * 1000.0) / 1000.0;
  char buf[64];
  if (value == (double)(long long)value) snprintf(buf, sizeof buf, "%lld", (long long)value);
  else {
    snprintf(buf, sizeof buf, "%.3f", value);
    std::string t(buf);
    while (!t.empty() && t.back() == '0') t.pop_back();   // Python's str(float): shortest form of a 3-decimal value
    for (char& ch : t) if (ch == '.') ch = '_';
    return t;
  }
  return buf;
}

int lora_key(const Model& m, int step) {   // lora.py:105-129; -1 = no LoRA
  if (!m.use_lora || step >= m.lora_steps) return -1;
  if (m.lora_mode == 0) return 0;                       // single
  if (m.lora_mode == 1) return step == 0 ? -1 : 0;      // from_second
  return step;                                          // all
}

namespace {

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
    l.ctx_dim = (int)tkv.shape[1];
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



// Scores without a key projection (first layer of a Perceiver: perceiver.py:141-152 with the queries of perceiver.py:224-226 /
// decoder.py:225-231, which are model constants).  q_l . (W_k x) = (W_k^T q_l) . x, so `to_kv` becomes
//   [ W_v  |  one row W_k,h^T q_l,h / sqrt(head_dim) per (query l, head h)  |  zero rows up to a multiple of 256 ]
// -- Lq * heads rows instead of heads * head_dim: 48 instead of 512 in the encoder's level aggregation, 208 instead of 1,024 in
// the decoder's de-aggregation -- and a context row leaves that linear with its values and its SCALED SCORES against every
// query (embed.hip: perceiver_attention_scores_kernel; perceiver_out.hip: perceiver_probs_kernel<.., true>).  The rows are
// summed in double on the host (64 terms each) and rounded once.  Not with a LayerNorm on the keys (`ln_k_q`), and only where
// the pre-split form exists iff to_kv's does (a context in the fp16-pair layout needs pre-split weights, step.hip).
void score_weights(Model& m, Resampler& r, const float* q0, int Lq, int heads) {
  r.vs_w = DevBuf();
  r.vs_ws = DevBuf();
  r.n_s = r.n_vs = r.vs_lq = 0;
  if (!m.score_weights || r.layers.empty() || q0 == nullptr) return;
  const auto& l = r.layers[0];
  if (l.ln_k_w != nullptr || l.head_dim * heads != l.inner || l.f16_mode < 0) return;
  const int inner = l.inner, hd = l.head_dim, K = l.ctx_dim, n_s = Lq * heads;
  const int n_vs = round_up(inner + n_s, 256);
  if (n_vs >= 2 * inner) return;   // nothing saved
  std::vector<float> wkv((size_t)2 * inner * K), q((size_t)Lq * inner), vs((size_t)n_vs * K, 0.f);
  hip_ok(hipMemcpy(wkv.data(), l.to_kv, wkv.size() * 4, hipMemcpyDeviceToHost), "download");
  hip_ok(hipMemcpy(q.data(), q0, q.size() * 4, hipMemcpyDeviceToHost), "download");
  std::copy(wkv.begin() + (size_t)inner * K, wkv.end(), vs.begin());   // the value half: rows inner .. 2 inner of to_kv
  const double scale = 1.0 / std::sqrt((double)hd);
  std::vector<double> acc((size_t)K);
  float s_max = 0.f;
  for (int lq = 0; lq < Lq; ++lq)
    for (int h = 0; h < heads; ++h) {
      std::fill(acc.begin(), acc.end(), 0.0);
      for (int d = 0; d < hd; ++d) {
        const double qd = q[(size_t)lq * inner + h * hd + d];
        const float* wr = wkv.data() + (size_t)(h * hd + d) * K;
        for (int c = 0; c < K; ++c) acc[c] += qd * wr[c];
      }
      float* dst = vs.data() + (size_t)(inner + lq * heads + h) * K;
      for (int c = 0; c < K; ++c) {
        dst[c] = (float)(acc[c] * scale);
        s_max = std::max(s_max, fabsf(dst[c]));
      }
    }
  if (!(s_max < 1000.f)) return;   // (the two-term split scales weights by 2^6: the same bound as pack_resampler's)
  DevBuf w(vs.size() * 4), ws;
  hip_ok(hipMemcpy(w.p, vs.data(), vs.size() * 4, hipMemcpyHostToDevice), "upload");
  if (l.to_kv_s != nullptr && l.f16_mode == 2 && K % 32 == 0 && K >= 96) {
    ws = DevBuf(vs.size() * 4);
    if (aurora_hip_split_f16(w.f(), K, ws.p, K, n_vs, K, 64.0f, nullptr) != AURORA_OK) throw std::runtime_error(aurora_hip_last_error());
    hip_ok(hipDeviceSynchronize(), "split weights");
  }
  if ((l.to_kv_s != nullptr) != (ws.p != nullptr)) return;
  r.vs_w = std::move(w);
  r.vs_ws = std::move(ws);
  r.n_s = n_s;
  r.n_vs = n_vs;
  r.vs_lq = Lq;
}

// What is known on the device about max |context| of a resampler: max|ctx| <= a * (*word) + c.  `pairs`: the context
// buffer holds fp16 pairs iff *word < limit_kv (written so by a guarded two-term producer with that very guard), fp32 otherwise.
}  // namespace

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

// The attention plan of one block flavour of this rank's band, on the device.
const DevPlan& plan_for(Model& m, int stage, bool shifted) {
  auto key = std::make_pair(stage, (int)shifted);
  auto it = m.plans.find(key);
  if (it == m.plans.end()) {
    BandPlan p;
    if (!band_plan(m.stage_res[stage], m.window, shifted, m.band.rank, m.rows[stage], p)) throw Fail{AURORA_E_ARG};
    REQUIRE(p.n_tok <= 144, "windows of more than 144 tokens are not supported");
    DevPlan d;
    d.n_windows = p.n_windows; d.n_tok = p.n_tok; d.n_own = p.n_own; d.n_halo = p.n_halo; d.n_interior = p.n_interior;
    d.tok = DevBuf(p.tok.size() * 4);
    upload(d.tok.p, p.tok.data(), p.tok.size() * 4);
    d.has_grp = !p.grp.empty();
    if (d.has_grp) {
      d.grp = DevBuf(p.grp.size());
      upload(d.grp.p, p.grp.data(), p.grp.size());
    }
    std::vector<int32_t> both;
    for (int side = 0; side < 2; ++side) {
      d.recv_off[side] = p.recv_off[side]; d.recv_cnt[side] = p.recv_cnt[side];
      d.send_cnt[side] = (int)p.send_idx[side].size();
      both.insert(both.end(), p.send_idx[side].begin(), p.send_idx[side].end());
    }
    REQUIRE(d.recv_cnt[0] == 0 || d.recv_cnt[1] == 0 || d.recv_off[1] == d.recv_off[0] + d.recv_cnt[0],
            "band plan: the halo rows of the two neighbours are not adjacent");
    if (!both.empty()) {
      d.send_idx = DevBuf(both.size() * 4);
      upload(d.send_idx.p, both.data(), both.size() * 4);
    }
    it = m.plans.emplace(key, std::move(d)).first;
  }
  return it->second;
}

// (groups, D, Kpad) GEMM weight of a LevelPatchEmbed for the channels that are present and T history steps
// (patchembed.py:100-115): per-variable (D, 1, Tmax, P, P) weights cut to T and laid out (v, t, i, j) along K, zero-padded
// to a multiple of 32.  Level-conditioned models (levelcond.py:36-69) hold one such weight per pressure level.
const EmbedPack& embed_pack(Model& m, int kind, int T, const std::vector<char>& present) {
  const std::vector<Channel>& chans = kind == 0 ? m.surf_channels : m.atmos_channels;
  int64_t mask = 0, mask_hi = 0;
  REQUIRE(chans.size() <= 126, "more than 126 input channels");
  for (size_t i = 0; i < chans.size(); ++i)
    if (present[i]) (i < 63 ? mask : mask_hi) |= (int64_t)1 << (i % 63);
  const std::array<int64_t, 3> key{(int64_t)kind * 1024 + T, mask, mask_hi};
  auto it = m.embed_packs.find(key);
  if (it != m.embed_packs.end()) return it->second;
  EmbedPack pk;
  for (size_t i = 0; i < chans.size(); ++i)
    if (present[i]) pk.channels.push_back((int)i);
  REQUIRE(!pk.channels.empty(), "no %s variable given", kind == 0 ? "surface-level" : "atmospheric");
  const bool per_level = kind == 1 && !m.level_condition.empty();
  pk.groups = per_level ? m.n_levels : 1;
  const int V = (int)pk.channels.size(), PP = m.P * m.P;
  pk.K = V * T * PP;
  pk.Kpad = round_up(pk.K, 32);
  std::vector<float> host((size_t)pk.groups * m.D * pk.Kpad, 0.f);
  for (int g = 0; g < pk.groups; ++g) {
    const std::string prefix = kind == 0 ? "encoder.surf_token_embeds.weights."
                               : per_level ? "encoder.atmos_token_embeds.layers." + level_to_str(m.levels[g]) + ".weights."
                                           : "encoder.atmos_token_embeds.weights.";
    for (int v = 0; v < V; ++v) {
      const Tensor& t = m.T_(prefix + chans[pk.channels[v]].name);   // (D, 1, Tmax, P, P)
      REQUIRE(t.shape.size() == 5 && t.shape[0] == m.D && t.shape[2] >= T && t.shape[3] == m.P, "bad patch-embed weight shape");
      const int64_t Tmax = t.shape[2];
      const std::vector<float> wv = to_host(t);
      for (int d = 0; d < m.D; ++d)
        for (int tt = 0; tt < T; ++tt)
          memcpy(&host[((size_t)g * m.D + d) * pk.Kpad + ((size_t)v * T + tt) * PP], &wv[((size_t)d * Tmax + tt) * PP], PP * sizeof(float));
    }
  }
  float l1 = 1e-6f, wmax = 0.f;
  for (size_t r = 0; r < (size_t)pk.groups * m.D; ++r) {
    float sum = 0.f;
    for (int k = 0; k < pk.Kpad; ++k) {
      const float a = fabsf(host[r * pk.Kpad + k]);
      sum += a;
      wmax = std::max(wmax, a);
    }
    l1 = std::max(l1, sum);
  }
  pk.l1 = l1;
  pk.w = to_device(host);
  // the fp16-pair form for the guarded two-term kernel (the raw, normalised inputs are bounded only by the guard)
  if (bounded_mode() == 2 && wmax < 1000.f && m.D % 256 == 0 && pk.Kpad >= 96) {
    pk.ws = DevBuf(host.size() * 4);
    if (aurora_hip_split_f16(pk.w.f(), pk.Kpad, pk.ws.p, pk.Kpad, (int64_t)pk.groups * m.D, pk.Kpad, 64.0f, nullptr) != AURORA_OK)
      throw std::runtime_error(aurora_hip_last_error());
    hip_ok(hipDeviceSynchronize(), "split embed weights");
  }
  return m.embed_packs.emplace(key, std::move(pk)).first->second;
}

namespace {

bool contains(const std::vector<std::string>& v, const std::string& s) { return std::find(v.begin(), v.end(), s) != v.end(); }

// Input channels of the two patch embeddings and the decoder's heads / outputs, from the variant keywords
// (encoder.py:226-303; aurora.py:733-742, 892-932; decoder.py:214-263).
void build_channels(Model& m) {
  m.surf_channels.clear(); m.atmos_channels.clear(); m.surf_heads.clear(); m.surf_out.clear(); m.atmos_heads.clear();
  if (m.variant == 2) {
    // ocean wave: the caller supplies the variables themselves; the model sees value (NaN -> 0) + density channels and
    // sin / cos of the directions -- the variables that stay, then the appended channels (aurora.py:892-912)
    std::vector<Channel> kept, appended;
    for (size_t i = 0; i < m.surf_inputs.size(); ++i) {
      const std::string& k = m.surf_inputs[i];
      const bool dens = contains(m.density_vars, k), ang = contains(m.angle_vars, k);
      if (!ang) kept.push_back({k, SRC_SURF, (int)i, dens ? 4 : 0});
      if (dens) appended.push_back({k + "_density", SRC_SURF, (int)i, 3});
      if (ang) {
        appended.push_back({k + "_sin", SRC_SURF, (int)i, 5});
        appended.push_back({k + "_cos", SRC_SURF, (int)i, 6});
      }
    }
    for (const auto& c : kept) { m.surf_channels.push_back(c); m.surf_out.push_back(c.name); }
    for (const auto& c : appended) m.surf_channels.push_back(c);
    for (const auto& c : m.surf_channels) m.surf_heads.push_back(c.name);
    for (const auto& a : m.angle_vars)   // predicted directions follow the variables that stayed (aurora.py:914-932)
      if (contains(m.surf_heads, a + "_sin") && !contains(m.surf_out, a)) m.surf_out.push_back(a);
    for (const auto& c : m.surf_channels)
      REQUIRE(contains(m.surf_vars, c.name), "ocean wave: channel '%s' is not among the model's surf_vars", c.name.c_str());
  } else {
    for (size_t i = 0; i < m.surf_inputs.size(); ++i) {
      const std::string& k = m.surf_inputs[i];
      const int tr = contains(m.pos_surf, k) ? (m.variant == 1 ? 2 : 1) : 0;   // clamp, or clamp + log feature combiner
      m.surf_channels.push_back({k, SRC_SURF, (int)i, tr});
      m.surf_heads.push_back(k);
      m.surf_out.push_back(k);
    }
    for (const auto& k : m.surf_inputs)
      if (contains(m.mod_heads, k)) m.surf_heads.push_back(k + "_mod");
  }
  for (size_t i = 0; i < m.static_vars.size(); ++i) m.surf_channels.push_back({m.static_vars[i], SRC_STATIC, (int)i, 0});
  if (m.dynamic_vars)
    for (int i = 0; i < 6; ++i) m.surf_channels.push_back({DYNAMIC_NAMES[i], SRC_DYN, i, 0});

  for (size_t i = 0; i < m.atmos_vars.size(); ++i) {
    const std::string& k = m.atmos_vars[i];
    const int tr = contains(m.pos_atmos, k) ? (m.variant == 1 ? 2 : 1) : 0;
    m.atmos_channels.push_back({k, SRC_ATMOS, (int)i, tr});
    m.atmos_heads.push_back(k);
  }
  for (const auto& k : m.atmos_vars)
    if (contains(m.mod_heads, k)) m.atmos_heads.push_back(k + "_mod");
  if (m.atmos_static_vars) {
    // static (and dynamic) variables at every level; prefixed when the dynamic ones are there (encoder.py:248-269)
    const std::string pre = m.dynamic_vars ? "static_" : "";
    for (size_t i = 0; i < m.static_vars.size(); ++i) m.atmos_channels.push_back({pre + m.static_vars[i], SRC_STATIC, (int)i, 0});
    if (m.dynamic_vars)
      for (int i = 0; i < 6; ++i) m.atmos_channels.push_back({pre + DYNAMIC_NAMES[i], SRC_DYN, i, 0});
  }
  if (m.index_bug) {
    // the slot of `static_z` is fed with `z`'s data (encoder.py:293-303, compat.py:156-159)
    int iz = -1, isz = -1;
    for (size_t i = 0; i < m.atmos_channels.size(); ++i) {
      if (m.atmos_channels[i].name == "z") iz = (int)i;
      if (m.atmos_channels[i].name == "static_z") isz = (int)i;
    }
    if (iz >= 0) {
      REQUIRE(isz >= 0, "'static_z' is not in list");
      Channel c = m.atmos_channels[iz];
      c.name = "static_z";
      m.atmos_channels[isz] = c;
    }
  }
}

// Fused decoder heads of a group of variables: [groups][n * P * P][2D] weights, V fastest inside a patch is handled by
// unpatchify's col0.  groups > 1: one head per pressure level (levelcond.py:36-69).
void build_heads(Model& m, HeadGroup& hg, const char* kind, const std::vector<std::string>& names, bool per_level) {
  const int PP = m.P * m.P, D2 = 2 * m.D;
  hg.names = names;
  hg.groups = per_level ? m.n_levels : 1;
  if (names.empty()) return;
  const size_t ldb = (size_t)round_up((int)names.size() * PP, 4);   // per-group bias rows padded: every group stays 16-byte aligned
  hg.w = DevBuf((size_t)hg.groups * names.size() * PP * D2 * 4);
  hg.b = DevBuf((size_t)hg.groups * ldb * 4);
  hip_ok(hipMemset(hg.b.p, 0, hg.b.bytes), "memset");
  for (int g = 0; g < hg.groups; ++g)
    for (size_t v = 0; v < names.size(); ++v) {
      std::string p = std::string("decoder.") + kind + "_heads." + names[v];
      if (per_level) p += ".layers." + level_to_str(m.levels[g]);
      const Tensor& wt = m.T_(p + ".weight");
      REQUIRE(wt.shape.size() == 2 && wt.shape[0] == PP && wt.shape[1] == D2, "bad head weight shape for %s", p.c_str());
      hip_ok(hipMemcpy(hg.w.f() + ((size_t)g * names.size() + v) * PP * D2, wt.f(), (size_t)PP * D2 * 4, hipMemcpyDeviceToDevice), "copy");
      hip_ok(hipMemcpy(hg.b.f() + (size_t)g * ldb + v * PP, m.W(p + ".bias"), (size_t)PP * 4, hipMemcpyDeviceToDevice), "copy");
    }
  // ---- the two-term form: N padded to whole 128-column tiles of the fp16-pair GEMM (zero rows cost MFMAs, not bytes of A) ----
  hg.n_pad = 0;
  hg.ws = DevBuf(); hg.bs = DevBuf();
  if (std::string(kind) != "atmos" || bounded_mode() != 2 || D2 % 32 != 0 || D2 < 96) return;
  const int n = (int)names.size() * PP, n_pad = round_up(n, 128);   // the tile width of the 256 x 128 two-term kernel
  std::vector<float> hw((size_t)hg.groups * n * D2);
  hip_ok(hipMemcpy(hw.data(), hg.w.p, hw.size() * 4, hipMemcpyDeviceToHost), "download");
  float wmax = 0.f;
  for (float v : hw) wmax = std::max(wmax, fabsf(v));
  if (!(wmax < 1000.f)) return;
  DevBuf padded((size_t)hg.groups * n_pad * D2 * 4);
  hip_ok(hipMemset(padded.p, 0, padded.bytes), "memset");
  hg.bs = DevBuf((size_t)hg.groups * n_pad * 4);
  hip_ok(hipMemset(hg.bs.p, 0, hg.bs.bytes), "memset");
  for (int g = 0; g < hg.groups; ++g) {
    hip_ok(hipMemcpy(padded.f() + (size_t)g * n_pad * D2, hg.w.f() + (size_t)g * n * D2, (size_t)n * D2 * 4, hipMemcpyDeviceToDevice), "copy");
    hip_ok(hipMemcpy(hg.bs.f() + (size_t)g * n_pad, hg.b.f() + (size_t)g * ldb, (size_t)n * 4, hipMemcpyDeviceToDevice), "copy");
  }
  hg.ws = DevBuf(padded.bytes);
  if (aurora_hip_split_f16(padded.f(), D2, hg.ws.p, D2, (int64_t)hg.groups * n_pad, D2, 64.0f, nullptr) != AURORA_OK)
    throw std::runtime_error(aurora_hip_last_error());
  hip_ok(hipDeviceSynchronize(), "split head weights");
  hg.n_pad = n_pad;
}

}  // namespace

}  // namespace aurora

using namespace aurora;

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
    auto names = [](const char* const* p, int n, std::vector<std::string>& dst) {
      for (int i = 0; i < n; ++i) dst.push_back(p[i]);
    };
    names(c->surf_vars, c->n_surf, m->surf_vars);
    names(c->static_vars, c->n_static, m->static_vars);
    names(c->atmos_vars, c->n_atmos, m->atmos_vars);
    REQUIRE(!m->surf_vars.empty() && !m->atmos_vars.empty(), "create: variable lists must not be empty");
    // ---- variant keywords ----
    m->variant = c->variant;
    REQUIRE(m->variant >= 0 && m->variant <= 2, "create: variant must be 0 (base), 1 (air pollution) or 2 (ocean wave)");
    for (int i = 0; i < c->n_level_condition; ++i) m->level_condition.push_back(c->level_condition[i]);
    m->dynamic_vars = c->dynamic_vars != 0; m->atmos_static_vars = c->atmos_static_vars != 0;
    m->clamp_first = c->clamp_at_first_step != 0; m->index_bug = c->simulate_indexing_bug != 0;
    names(c->separate_perceiver, c->n_separate_perceiver, m->sep_perceiver);
    names(c->modulation_heads, c->n_modulation_heads, m->mod_heads);
    names(c->positive_surf_vars, c->n_positive_surf, m->pos_surf);
    names(c->positive_atmos_vars, c->n_positive_atmos, m->pos_atmos);
    names(c->surf_inputs, c->n_surf_inputs, m->surf_inputs);
    names(c->density_channel_surf_vars, c->n_density, m->density_vars);
    names(c->angle_surf_vars, c->n_angle, m->angle_vars);
    if (m->surf_inputs.empty()) m->surf_inputs = m->surf_vars;
    REQUIRE(m->variant == 2 || m->surf_inputs == m->surf_vars, "create: surf_inputs are for the ocean-wave variant");
    if (c->difference_history)
      for (int i = 0; i < c->n_modulation_heads; ++i)
        if (c->difference_history[i] >= 0) m->diff_index[m->mod_heads[i]] = c->difference_history[i];
    REQUIRE(m->static_vars.size() <= 60 && m->surf_inputs.size() <= 60 && m->atmos_vars.size() <= 60, "create: too many variables");
    build_blocks(*m);
    build_channels(*m);
    m->ctx_max = DevBuf(16);
    // how this handle runs its steps: fields of the configuration (0 = the library's default), never the environment
    const auto& tu = c->tuning;
    REQUIRE(tu.fuse_ln >= 0 && tu.fuse_ln <= 3, "create: tuning.fuse_ln = %d (0 default, 1 never, 2 by the fill rule, 3 always)", tu.fuse_ln);
    auto sw = [](int32_t v, bool dflt, const char* what) {
      REQUIRE(v >= 0 && v <= 2, "create: tuning.%s = %d (0 default, 1 off, 2 on)", what, v);
      return v == 0 ? dflt : v == 2;
    };
    m->fuse_ln = tu.fuse_ln == 0 ? 1 : tu.fuse_ln - 1;
    m->split_attention = sw(tu.band_split_attention, false, "band_split_attention");
    m->qkv_planes = sw(tu.qkv_planes, true, "qkv_planes");
    m->split_k = sw(tu.split_k, true, "split_k");
    m->reassoc_out = sw(tu.perceiver_reassoc, true, "perceiver_reassoc");
    m->score_weights = sw(tu.score_weights, true, "score_weights");
    m->tickets = DevBuf(SPLIT_TICKETS * sizeof(int32_t));
    hip_ok(hipMemset(m->tickets.p, 0, SPLIT_TICKETS * sizeof(int32_t)), "hipMemset");
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
    m.keep.clear(); m.attn_sets.clear(); m.embed_packs.clear(); m.merges.clear(); m.splits.clear();
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
    hip_ok(hipStreamSynchronize(as_stream(stream)), "precompute sync");
    score_weights(m, m.enc_rs, m.enc_q0.f(), n_lat, m.perceiver_heads);
    // ---- second decoder Perceiver for the variables of `separate_perceiver` (decoder.py:232-248) ----
    std::vector<std::string> sep = m.sep_perceiver;
    if (!m.mod_heads.empty())
      for (const auto& v : m.sep_perceiver) sep.push_back(v + "_mod");
    m.has_alt = !sep.empty();
    if (m.has_alt) m.dec_rs_alt = pack_resampler(m, "decoder.level_decoder_alternate", m.dec_depth, m.perceiver_heads);
    // ---- decoder heads, fused over the variables of a group (level-conditioned atmospheric heads: per level set) ----
    build_heads(m, m.head_surf, "surf", m.surf_heads, false);
    if (m.level_condition.empty()) {
      std::vector<std::string> main_names, alt_names;
      for (const auto& n : m.atmos_heads) (contains(sep, n) ? alt_names : main_names).push_back(n);
      build_heads(m, m.head_main, "atmos", main_names, false);
      build_heads(m, m.head_alt, "atmos", alt_names, false);
    }
    // ---- air pollution: Linear(2, 1) feature combiners of the positive variables (aurora.py:733-742) ----
    for (int kind = 0; kind < 2; ++kind)
      for (Channel& ch : kind == 0 ? m.surf_channels : m.atmos_channels)
        if (ch.transform == 2) {
          const std::string p = std::string(kind == 0 ? "surf" : "atmos") + "_feature_combiner." + ch.name;
          const std::vector<float> wv = to_host(m.T_(p + ".weight")), bv = to_host(m.T_(p + ".bias"));
          REQUIRE(wv.size() == 2 && bv.size() == 1, "bad feature combiner shape for %s", p.c_str());
          ch.tw0 = wv[0]; ch.tw1 = wv[1]; ch.tb = bv[0];
        }
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

extern "C" int aurora_hip_set_band(aurora_hip_model* mp, const aurora_hip_band* band) {
  GUARDED({
    REQUIRE(mp != nullptr, "set_band: null model");
    Model& m = *mp;
    if (band == nullptr || band->world <= 1) {
      m.band = aurora_hip_band{0, 1, nullptr, nullptr, nullptr};
    } else {
      REQUIRE(band->rank >= 0 && band->rank < band->world, "set_band: rank %d of %d", band->rank, band->world);
      REQUIRE(band->post && band->wait, "set_band: the halo transport callbacks are required");
      m.band = *band;
    }
    m.have_grid = false;   // the grid tables are per band: precompute again
    m.plans.clear(); m.rows.clear();
    m.stage_send = m.stage_recv = nullptr;
    m.staging_bytes = m.staging_need = 0;
  })
}

extern "C" int aurora_hip_band_rows(const aurora_hip_model* m, int32_t* row0, int32_t* row1) {
  GUARDED({
    REQUIRE(m && row0 && row1 && m->have_grid, "band_rows: precompute the grid first");
    const int h0 = m->sharded() ? m->rows[0][m->band.rank][0] : 0;
    *row0 = h0 * m->P;
    *row1 = (h0 + m->Hp) * m->P;
  })
}

extern "C" int64_t aurora_hip_band_staging_bytes(const aurora_hip_model* m) { return m ? m->staging_need : 0; }

extern "C" int aurora_hip_set_band_staging(aurora_hip_model* m, void* send, void* recv, int64_t staging_bytes) {
  GUARDED({
    REQUIRE(m != nullptr, "set_band_staging: null model");
    REQUIRE(staging_bytes >= m->staging_need, "set_band_staging: %lld bytes per buffer, %lld needed", (long long)staging_bytes,
            (long long)m->staging_need);
    REQUIRE(m->staging_need == 0 || (send && recv && (uintptr_t)send % 16 == 0 && (uintptr_t)recv % 16 == 0),
            "set_band_staging: two 16-byte aligned device buffers are required");
    m->stage_send = send;
    m->stage_recv = recv;
    m->staging_bytes = staging_bytes;
  })
}

extern "C" int aurora_hip_output_vars(const aurora_hip_model* m, const char** names, int capacity) {
  if (!m) return 0;
  if (names)
    for (int i = 0; i < capacity && i < (int)m->surf_out.size(); ++i) names[i] = m->surf_out[i].c_str();
  return (int)m->surf_out.size();
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
    m.full_Hp = H / P; m.Wp = W / P; m.n_lon = W;
    // ---- stage resolutions of the whole grid (swin3d.py:868-882) ----
    m.stage_res = stage_resolutions(Res{m.Cl, m.full_Hp, m.Wp}, m.n_stages);
    m.merge_pad.clear(); m.tables.clear(); m.plans.clear(); m.embed_packs.clear();
    for (int s = 0; s + 1 < m.n_stages; ++s) m.merge_pad.push_back({m.stage_res[s].h % 2, m.stage_res[s].w % 2});
    m.merge_pad.push_back({0, 0});
    // ---- this rank's rows: everything (un-sharded) or a latitude band ----
    int h0 = 0;
    m.Hp = m.full_Hp;
    m.rows.clear();
    if (m.sharded()) {
      if (!band_rows(m.stage_res, m.window, m.band.world, m.rows)) throw Fail{AURORA_E_ARG};
      h0 = m.rows[0][m.band.rank][0];
      m.Hp = m.rows[0][m.band.rank][1] - h0;
    }
    m.n_lat = m.Hp * P;
    const int64_t Lp_full = (int64_t)m.full_Hp * m.Wp, Lp = (int64_t)m.Hp * m.Wp;
    // ---- position / scale encodings of the patch grid (posencoding.py:61-192) ----
    std::vector<float> pos((size_t)Lp_full * D), scale((size_t)Lp_full * D);
    if (g->pos_encoding && g->scale_encoding) {
      memcpy(pos.data(), g->pos_encoding, pos.size() * 4);
      memcpy(scale.data(), g->scale_encoding, scale.size() * 4);
    } else {
      REQUIRE(g->lat && g->lon, "precompute: latitudes / longitudes (or the encodings themselves) are required");
      pos_scale_tables(g->lat, g->lon, m.full_Hp, m.Wp, P, D, pos.data(), scale.data());
    }
    {
      DevBuf d_pos((size_t)Lp * D * 4), d_scale((size_t)Lp * D * 4), pe((size_t)Lp * D * 4);
      upload(d_pos.p, pos.data() + (size_t)h0 * m.Wp * D, (size_t)Lp * D * 4);       // the band's patch rows
      upload(d_scale.p, scale.data() + (size_t)h0 * m.Wp * D, (size_t)Lp * D * 4);
      m.pos_scale = DevBuf((size_t)Lp * D * 4);
      L.linear(d_pos.p, D, m.W("encoder.pos_embed.weight"), D, m.W("encoder.pos_embed.bias"), pe.p, D, Lp, D, D, AURORA_F32);
      L.linear(d_scale.p, D, m.W("encoder.scale_embed.weight"), D, m.W("encoder.scale_embed.bias"), m.pos_scale.p, D, Lp, D, D,
               AURORA_F32, 0, nullptr, 0, pe.f(), D);
      hip_ok(hipStreamSynchronize(as_stream(stream)), "precompute sync");
    }
    // ---- pressure levels: per-level patch-embedding bias, decoder queries (encoder.py:318-330, decoder.py:176-200) ----
    const int C = g->n_levels;
    REQUIRE(C >= 1 && C <= 32 && g->levels, "precompute: 1..32 pressure levels are required");
    m.n_levels = C;
    m.levels.assign(C, 0.0);
    for (int c = 0; c < C; ++c) m.levels[c] = g->levels_float32 ? (double)(float)g->levels[c] : g->levels[c];
    {
      std::vector<float> enc((size_t)C * D), dec((size_t)C * 2 * D);
      fourier(LEVELS, m.levels.data(), C, D, enc.data());
      fourier(LEVELS, m.levels.data(), C, 2 * D, dec.data());
      DevBuf d_enc = to_device(enc), d_dec = to_device(dec);
      m.enc_bias = DevBuf((size_t)C * D * 4);
      if (m.level_condition.empty()) {
        L.linear(d_enc.p, D, m.W("encoder.atmos_levels_embed.weight"), D, m.W("encoder.atmos_levels_embed.bias"), m.enc_bias.p, D, C,
                 D, D, AURORA_F32, 0, nullptr, 0, m.W("encoder.atmos_token_embeds.bias"), 0);
      } else {   // every level has its own patch embedding, bias included (levelcond.py:36-69)
        DevBuf pb((size_t)C * D * 4);
        for (int c = 0; c < C; ++c)
          hip_ok(hipMemcpy(pb.f() + (size_t)c * D, m.W("encoder.atmos_token_embeds.layers." + level_to_str(m.levels[c]) + ".bias"),
                           (size_t)D * 4, hipMemcpyDeviceToDevice), "copy");
        L.linear(d_enc.p, D, m.W("encoder.atmos_levels_embed.weight"), D, m.W("encoder.atmos_levels_embed.bias"), m.enc_bias.p, D, C,
                 D, D, AURORA_F32, 0, nullptr, 0, pb.f(), D);
        hip_ok(hipStreamSynchronize(as_stream(stream)), "precompute sync");
      }
      m.dec_queries = DevBuf((size_t)C * 2 * D * 4);
      L.linear(d_dec.p, 2 * D, m.W("decoder.atmos_levels_embed.weight"), 2 * D, m.W("decoder.atmos_levels_embed.bias"),
               m.dec_queries.p, 2 * D, C, 2 * D, 2 * D, AURORA_F32);
      auto first_q = [&](const Resampler& rs, DevBuf& q) {
        const auto& d0 = rs.layers[0];
        q = DevBuf((size_t)C * d0.inner * 4);
        L.linear(m.dec_queries.p, 2 * D, d0.to_q, 2 * D, nullptr, q.p, d0.inner, C, d0.inner, 2 * D, AURORA_F32);
        if (d0.ln_q_w)
          L.layernorm(q.p, d0.inner, d0.ln_q_w, d0.ln_q_b, nullptr, 0, 0, q.f(), d0.inner, nullptr, 0, C, d0.inner, 1e-5f, AURORA_F32);
      };
      first_q(m.dec_rs, m.dec_q);
      if (m.has_alt) first_q(m.dec_rs_alt, m.dec_q_alt);
      hip_ok(hipStreamSynchronize(as_stream(stream)), "precompute sync");
      score_weights(m, m.dec_rs, m.dec_q.f(), C, m.perceiver_heads);
      if (m.has_alt) score_weights(m, m.dec_rs_alt, m.dec_q_alt.f(), C, m.perceiver_heads);
      // What a decoder Perceiver can put out, whatever the inputs: every layer returns LN2(.) + LN1(.) + its residual, the
      // first residual being the level queries -- |LN(x) g + b| <= sqrt(D) max|g| + max|b|.  Decides whether the output may
      // leave in the fp16-pair layout for the output heads' two-term GEMM (step.hip).
      {
        std::vector<float> q((size_t)C * 2 * D);
        hip_ok(hipMemcpy(q.data(), m.dec_queries.p, q.size() * 4, hipMemcpyDeviceToHost), "download");
        float qmax = 0.f;
        for (float v : q) qmax = std::max(qmax, fabsf(v));
        auto amax = [&](const float* dev, size_t n) {
          std::vector<float> h(n);
          hip_ok(hipMemcpy(h.data(), dev, n * 4, hipMemcpyDeviceToHost), "download");
          float mx = 0.f;
          for (float v : h) mx = std::max(mx, fabsf(v));
          return mx;
        };
        auto bound_of = [&](const Resampler& rs) {
          float b = qmax;
          for (const auto& ly : rs.layers) {
            const float rt = sqrtf((float)ly.dim);
            b += amax(ly.ln1_w, ly.dim) * rt + amax(ly.ln1_b, ly.dim) + amax(ly.ln2_w, ly.dim) * rt + amax(ly.ln2_b, ly.dim);
          }
          return b;
        };
        m.dec_out_bound = bound_of(m.dec_rs);
        m.dec_out_bound_alt = m.has_alt ? bound_of(m.dec_rs_alt) : 0.f;
      }
      std::vector<float> eb((size_t)C * D);
      hip_ok(hipMemcpy(eb.data(), m.enc_bias.p, eb.size() * 4, hipMemcpyDeviceToHost), "download");
      m.enc_bias_max = 0.f;
      for (float v : eb) m.enc_bias_max = std::max(m.enc_bias_max, fabsf(v));
    }
    if (!m.level_condition.empty()) {   // level-conditioned heads depend on the level set
      std::vector<std::string> sep = m.sep_perceiver;
      if (!m.mod_heads.empty())
        for (const auto& v : m.sep_perceiver) sep.push_back(v + "_mod");
      std::vector<std::string> main_names, alt_names;
      for (const auto& n : m.atmos_heads) (contains(sep, n) ? alt_names : main_names).push_back(n);
      build_heads(m, m.head_main, "atmos", main_names, true);
      build_heads(m, m.head_alt, "atmos", alt_names, true);
    }
    // ---- normalisation statistics: loc, scale, 1/scale (computed in fp64) per variable (and level) ----
    const int ns = (int)m.surf_inputs.size(), nst = (int)m.static_vars.size(), na = (int)m.atmos_vars.size();
    REQUIRE(g->surf_loc && g->surf_scale && g->atmos_loc && g->atmos_scale && (nst == 0 || (g->static_loc && g->static_scale)),
            "precompute: normalisation statistics are required");
    std::vector<float> hs;
    m.surf_stat_off.clear(); m.static_stat_off.clear(); m.atmos_stat_off.clear(); m.static_lvl_stat_off.clear(); m.static_loc.clear();
    auto push1 = [&](std::vector<size_t>& offs, double loc, double sc) {
      offs.push_back(hs.size());
      hs.push_back((float)loc); hs.push_back((float)sc); hs.push_back((float)(1.0 / sc)); hs.push_back(0.f);
    };
    auto pushC = [&](std::vector<size_t>& offs, const double* loc, const double* sc, int stride) {   // C x loc | scale | 1/scale
      offs.push_back(hs.size());
      for (int c = 0; c < C; ++c) hs.push_back((float)loc[c * stride]);
      for (int c = 0; c < C; ++c) hs.push_back((float)sc[c * stride]);
      for (int c = 0; c < C; ++c) hs.push_back((float)(1.0 / sc[c * stride]));
      while (hs.size() % 4) hs.push_back(0.f);
    };
    for (int v = 0; v < ns; ++v) push1(m.surf_stat_off, g->surf_loc[v], g->surf_scale[v]);
    for (int v = 0; v < nst; ++v) {
      push1(m.static_stat_off, g->static_loc[v], g->static_scale[v]);
      m.static_loc.push_back(g->static_loc[v]);
    }
    for (int v = 0; v < na; ++v) pushC(m.atmos_stat_off, g->atmos_loc + (size_t)v * C, g->atmos_scale + (size_t)v * C, 1);
    // static variables fed at every level keep their surface statistics; dynamic planes are not normalised
    for (int v = 0; v < nst; ++v) pushC(m.static_lvl_stat_off, g->static_loc + v, g->static_scale + v, 0);
    {
      const double zero = 0.0, one = 1.0;
      std::vector<size_t> tmp;
      pushC(tmp, &zero, &one, 0);
      m.one_stat_off = tmp[0];
    }
    m.stats = to_device(hs);
    // ---- a band's halo plans, and the staging each side of an exchange needs ----
    m.staging_need = 0;
    if (m.sharded())
      for (int s = 0; s < m.n_stages; ++s)
        for (int sh = 0; sh < 2; ++sh) {
          const DevPlan& pl = plan_for(m, s, sh != 0);
          const int64_t row_bytes = (int64_t)m.stage_dim(s) * (int64_t)m.bbs();   // the block's input rows travel (step.hip)
          m.staging_need = std::max(m.staging_need, (int64_t)std::max(pl.send_cnt[0] + pl.send_cnt[1], pl.recv_cnt[0] + pl.recv_cnt[1]) * row_bytes);
        }
    m.have_grid = true;
    m.generation += 1;
  })
}

namespace {
// Civil date of a day count since 1970-01-01 (proleptic Gregorian; H. Hinnant's days_from_civil inverse).
void civil_from_days(int64_t z, int& y, int& mth, int& d) {
  z += 719468;
  const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
  const unsigned doe = (unsigned)(z - era * 146097);
  const unsigned yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const unsigned doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const unsigned mp = (5 * doy + 2) / 153;
  d = (int)(doy - (153 * mp + 2) / 5 + 1);
  mth = (int)(mp < 10 ? mp + 3 : mp - 9);
  y = (int)(yoe + era * 400 + (mth <= 2));
}
}  // namespace

extern "C" int aurora_hip_set_time_ex(aurora_hip_model* mp, const double* time_hours, const int32_t* calendar, int B, void* stream) {
  GUARDED({
    REQUIRE(mp && time_hours && B >= 1, "set_time: bad argument");
    Model& m = *mp;
    std::vector<double> t(B);
    // the reference converts the timestamps to a float32 tensor before expanding (encoder.py:359-362)
    for (int b = 0; b < B; ++b) t[b] = (double)(float)time_hours[b];
    if (m.abs_B < B) {
      hip_ok(hipDeviceSynchronize(), "set_time");
      m.abs_enc = DevBuf((size_t)B * m.D * 4);
      m.dyn_planes = DevBuf((size_t)6 * B * 4);
      m.abs_B = B;
      m.generation += 1;
    }
    const size_t n_abs = (size_t)B * m.D, n_dyn = (size_t)6 * m.abs_B, bytes = (n_abs + n_dyn) * 4;
    auto& slot = m.pinned[m.pinned_next++ & 3];
    if (slot.done) hip_ok(hipEventSynchronize(slot.done), "set_time");   // the copy that used this slot four uploads ago
    else hip_ok(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming), "set_time");
    if (slot.bytes < bytes) {
      if (slot.host) (void)hipHostFree(slot.host);
      hip_ok(hipHostMalloc((void**)&slot.host, bytes, hipHostMallocDefault), "set_time");
      slot.bytes = bytes;
    }
    fourier(ABS_TIME, t.data(), B, m.D, slot.host);
    // time of day / day of week / "day of year" planes of the dynamic variables (encoder.py:226-246: the last really is the
    // day of the MONTH over 365.25), plane i of batch element b at [i][b]
    float* dyn = slot.host + n_abs;
    for (size_t i = 0; i < n_dyn; ++i) dyn[i] = 0.f;
    for (int b = 0; b < B; ++b) {
      int hour, weekday, day;
      if (calendar) { hour = calendar[3 * b]; weekday = calendar[3 * b + 1]; day = calendar[3 * b + 2]; }
      else {
        const double hrs = time_hours[b];
        const int64_t days = (int64_t)floor(hrs / 24.0);
        hour = (int)floor(hrs - 24.0 * (double)days);
        weekday = (int)(((days % 7) + 7 + 3) % 7);   // 1970-01-01 was a Thursday; Monday = 0
        int y, mo;
        civil_from_days(days, y, mo, day);
      }
      const double vals[6] = {cos(2 * PI * hour / 24), sin(2 * PI * hour / 24), cos(2 * PI * weekday / 7), sin(2 * PI * weekday / 7),
                              cos(2 * PI * day / 365.25), sin(2 * PI * day / 365.25)};
      for (int i = 0; i < 6; ++i) dyn[(size_t)i * m.abs_B + b] = (float)vals[i];
    }
    hip_ok(hipMemcpyAsync(m.abs_enc.p, slot.host, n_abs * 4, hipMemcpyHostToDevice, as_stream(stream)), "set_time");
    hip_ok(hipMemcpyAsync(m.dyn_planes.p, dyn, n_dyn * 4, hipMemcpyHostToDevice, as_stream(stream)), "set_time");
    hip_ok(hipEventRecord(slot.done, as_stream(stream)), "set_time");
  })
}

extern "C" int aurora_hip_set_time(aurora_hip_model* mp, const double* time_hours, int B, void* stream) {
  return aurora_hip_set_time_ex(mp, time_hours, nullptr, B, stream);
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
    if (m.sharded()) {
      REQUIRE(io->B == 1, "latitude-band sharding runs one forecast (batch size 1) across the ranks");
      REQUIRE(m.staging_need == 0 || m.staging_bytes >= m.staging_need, "step: hand over the band's staging buffers first "
              "(aurora_hip_set_band_staging, %lld bytes each)", (long long)m.staging_need);
    }
    StepIO s{io, io->B, io->T};
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
      m.generation += 1;
    }
    // split-K tickets must be zero when a launch starts (gemm.hip).  Every launch leaves them zero, but a launch that was
    // aborted on the DEVICE (a fault, a reset) returns normally to the host and may have left counts behind: a sharded step --
    // the only kind that splits along K -- clears them first, whatever happened before (one 16 KiB memset node; a captured
    // graph replays it too).
    if (m.sharded() && m.split_k)
      hip_ok(hipMemsetAsync(m.tickets.p, 0, SPLIT_TICKETS * sizeof(int32_t), as_stream(stream)), "zeroing the split-K tickets");
    run_step(m, s, stream);
  })
}

namespace {
const char* const KIND_NAMES[K_COUNT] = {"linear_bf16", "linear_f32", "window_attention_bf16", "layernorm", "merge_ln",
                                         "split_ln", "patchify", "perceiver_attention", "assemble_tokens", "unpatchify",
                                         "copy2d", "absmax", "linear_layernorm_bf16", "gather_rows", "perceiver_out"};
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

extern "C" int aurora_hip_profile_end_list(aurora_hip_model* m, aurora_hip_profile_entry* out, int capacity, int* n_out) {
  GUARDED({
    REQUIRE(m && n_out && (out || capacity == 0), "profile_end_list: bad arguments");
    m->profile_mask = 0;
    hip_ok(hipDeviceSynchronize(), "profile_end_list");
    *n_out = (int)m->timed.size();
    if (capacity < *n_out) return AURORA_OK;   // (query: the launches stay recorded)
    int i = 0;
    for (auto& t : m->timed) {
      float ms = 0.f;
      hip_ok(hipEventElapsedTime(&ms, t.e0, t.e1), "hipEventElapsedTime");
      out[i++] = aurora_hip_profile_entry{KIND_NAMES[t.kind], 1, ms, t.work};
      m->event_pool.push_back(t.e0);
      m->event_pool.push_back(t.e1);
    }
    m->timed.clear();
  })
}

extern "C" int64_t aurora_hip_generation(const aurora_hip_model* m) { return m ? m->generation : 0; }

extern "C" int aurora_hip_abi_sizes(int32_t* out, int capacity) {
  const int32_t sizes[] = {(int32_t)sizeof(aurora_hip_config), (int32_t)sizeof(aurora_hip_grid), (int32_t)sizeof(aurora_hip_step_io),
                           (int32_t)sizeof(aurora_hip_band), (int32_t)sizeof(aurora_hip_halo_msg), (int32_t)sizeof(aurora_hip_plan_info),
                           (int32_t)sizeof(aurora_patch_var), (int32_t)sizeof(aurora_unpatch_var), (int32_t)sizeof(aurora_hip_profile_entry)};
  const int n = (int)(sizeof(sizes) / sizeof(sizes[0]));
  for (int i = 0; i < n && i < capacity; ++i) out[i] = sizes[i];
  return n;
}

extern "C" int64_t aurora_hip_workspace_bytes(const aurora_hip_model* m) { return m ? (int64_t)m->arena.cap : 0; }

extern "C" int aurora_hip_guard_words(const aurora_hip_model* m, float out[4], void* stream) {
  GUARDED({
    REQUIRE(m && out && m->ctx_max.p, "guard_words: bad arguments");
    static_assert(AURORA_F16_SAFE_RANGE == F16_SAFE, "the header's constant is the step's");
    hip_ok(hipMemcpyAsync(out, m->ctx_max.p, 4 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream), "guard_words");
    hip_ok(hipStreamSynchronize((hipStream_t)stream), "guard_words");
  })
}
