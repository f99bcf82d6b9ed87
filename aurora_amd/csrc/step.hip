// The launch sequence of ONE FORECAST STEP (aurora_hip_step): encoder (encoder.py:198-366), 3D Swin U-net
// (swin3d.py:884-936, 440-509), decoder (decoder.py:168-276), for every model class -- the ERA5 family and the
// air-pollution / ocean-wave variants (aurora.py:726-796, 854-932; levelcond.py:36-69) -- on one device or on one
// latitude band of a sharded forecast (halo exchange through the host's transport callbacks, include/aurora_hip.h).
// Host code only; every launch goes through the operator ABI of this same library.
#include <algorithm>

#include "model.h"

namespace aurora {

namespace {

struct CtxGuard { const float* word; float a, c, limit_kv; bool pairs; };

// PerceiverResampler (perceiver.py:212-233) for all grid columns at once.  ctx: key j of column (b, l) at row
// b*kv_bstride + j*kv_lstride + l.  First layer: latents (and so q) are shared by every column.  Returns (B*cols*Lq, D).
float* resampler(Model& m, Launcher& L, const Resampler& rs, const float* ctx, int64_t ctx_rows, int ctx_dim, const float* q0,
                 const float* latents0, int B, int64_t cols, int64_t kv_bstride, int64_t kv_lstride, int Lq, int Lk, int heads,
                 float eps, size_t& out_mark, const CtxGuard* cg = nullptr, bool out_pairs = false, int own_word = 0,
                 bool scan_ctx = true) {
  const int64_t n_rows = (int64_t)B * cols * Lq;
  // The context is as unbounded as the model inputs, so the linears that read it, or averages of its value projection,
  // pick their operand split on the device: from max |ctx|, measured here, or from the bound the caller derived from a
  // word it measured upstream (`cg`).
  const float* ctx_max = cg ? cg->word : m.ctx_max.f() + own_word;
  const float g_a = cg ? cg->a : 1.0f, g_c = cg ? cg->c : 0.0f;
  const bool ctx_pairs = cg && cg->pairs;
  // (the words were cleared at the start of the step; of the two decoder Perceivers of a `separate_perceiver` model only the
  // first scans their common context: `scan_ctx`)
  if (!cg && scan_ctx)
    timed(m, L.stream, K_ABSMAX, 0.0, [&] { return aurora_hip_absmax_fold(ctx, ctx_rows * ctx_dim, m.ctx_max.f() + own_word, L.stream); });
  float* lat = nullptr;
  for (size_t i = 0; i < rs.layers.size(); ++i) {
    const auto& ly = rs.layers[i];
    const int inner = ly.inner, Dd = ly.dim;
    const size_t mark0 = m.arena.top;
    // Workspace of a layer: three regions instead of one buffer per intermediate (the decoder's intermediates are 3.5 GB each
    // at 0.25 degree) --
    //   Y  the layer's result (it outlives the rest; stack order); until fc2 writes it, it holds the attention output
    //   L  the MLP's input / residual (LayerNorm 1 output)
    //   S  scratch: k | v (and q), then to_out's result, then the MLP's hidden layer, each dead before the next is written;
    //      the MLP runs in row chunks so that a chunk's hidden layer fits
    const size_t unit = (size_t)n_rows * Dd * 4;
    float* y = (float*)m.arena.take(unit);
    const size_t after_y = m.arena.top;
    float* lat1 = (float*)m.arena.take(unit);   // fp32 values, or their fp16 pairs
    // First layer, queries known at pack time: the context rows leave `to_kv` as [v | scores with every query] -- no keys
    // (model.hip:score_weights); else k | v.
    const bool scores = i == 0 && rs.n_vs > 0 && rs.vs_lq == Lq && ly.ln_k_w == nullptr;
    const int kv_ld = scores ? rs.n_vs : 2 * inner;
    const size_t kv_bytes = (size_t)ctx_rows * kv_ld * 4, q_bytes = i > 0 ? (size_t)n_rows * inner * 4 : 0;
    const size_t att_bytes = (size_t)n_rows * inner * 4;
    const bool att_in_y = att_bytes <= unit;   // (inner <= dim in every published model; else behind everything it coexists with)
    const size_t kvq_bytes = ((kv_bytes + 255) & ~size_t(255)) + ((q_bytes + 255) & ~size_t(255));
    const size_t att_off = (std::max(unit, kvq_bytes) + 255) & ~size_t(255);
    const size_t hid_row = (size_t)ly.hidden * 4;
    const size_t hid_min = (size_t)std::min<int64_t>(n_rows, 256) * hid_row;   // at least one row tile of the hidden layer
    // The decoder's de-aggregation (first layer: queries shared by all columns, three keys per column) runs RE-ASSOCIATED
    // (perceiver_out.hip): to_out of the three value rows per column and head, then the Lq x 3 convex combinations per head
    // in registers -- the attention output and its Lq-row `to_out` GEMM do not exist.  Its inputs: the softmax weights P
    // (behind to_out's result in the scratch region) and the value rows as fp16 pairs (in the result region, until fc2
    // writes there).  Two fp16 terms: decided on the device by the guard of the linear it replaces.
    const int64_t n_cols = (int64_t)B * cols;
    const bool att_pairs = ly.to_out_s != nullptr && inner % 32 == 0;
    const bool reassoc = m.reassoc_out && i == 0 && att_pairs && att_in_y && ly.f16_mode == 2 &&
                         (size_t)n_cols * Lk * inner * 4 <= unit &&
                         aurora_hip_perceiver_out_supported(Lq, Lk, heads, ly.head_dim, Dd) != 0;
    const size_t p_off = (unit + 255) & ~size_t(255), p_bytes = reassoc ? (size_t)n_cols * heads * 64 * 4 : 0;
    const size_t s_bytes = std::max(std::max(att_in_y ? std::max(unit, kvq_bytes) : att_off + att_bytes, hid_min),
                                    reassoc ? p_off + p_bytes : (size_t)0);
    char* S = (char*)m.arena.take(s_bytes);
    float* kv = (float*)S;
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
    guarded(ctx, ctx_dim, scores ? rs.vs_w.f() : ly.to_kv, scores ? rs.vs_ws.p : ly.to_kv_s, kv, kv_ld, ctx_rows, kv_ld, ctx_dim,
            ctx_pairs ? cg->limit_kv : (F16_SAFE - g_c) / g_a, ctx_pairs);
    if (ly.ln_k_w)   // LayerNorm over the K half, in place (perceiver.py:144-147)
      L.layernorm(kv, 2 * inner, ly.ln_k_w, ly.ln_k_b, nullptr, 0, 0, kv, 2 * inner, nullptr, 0, ctx_rows, inner, 1e-5f,
                  AURORA_F32);
    const float* q = q0;
    int64_t q_stride = 0;
    if (i > 0) {
      float* qb = (float*)(S + ((kv_bytes + 255) & ~size_t(255)));
      L.linear(lat, Dd, ly.to_q, Dd, nullptr, qb, inner, n_rows, inner, Dd, AURORA_F32);
      if (ly.ln_q_w) L.layernorm(qb, inner, ly.ln_q_w, ly.ln_q_b, nullptr, 0, 0, qb, inner, nullptr, 0, n_rows, inner, 1e-5f, AURORA_F32);
      q = qb;
      q_stride = Lq;
    }
    float* att = att_in_y ? y : (float*)(S + att_off);
    // |att| <= max |v| <= (largest L1 row norm of W_v) * max |ctx|: same guard, tighter limit.  With pre-split to_out
    // weights the attention writes fp16 pairs iff that guard holds, and to_out multiplies them without splitting anything.
    const float lim_out = (F16_SAFE / ly.v_l1 - g_c) / g_a;
    float* o = (float*)S;   // (k | v and q are dead)
    if (reassoc) {
      float* P = (float*)(S + p_off);
      void* Vp = y;
      timed(m, L.stream, K_PERCEIVER_ATTENTION, 0.0, [&] {
        if (scores)
          return aurora_hip_perceiver_probs_scores(kv, kv_ld, inner, P, Vp, B, cols, kv_bstride, kv_lstride, Lq, Lk, heads, ly.head_dim,
                                                   ctx_max, lim_out, L.stream);
        return aurora_hip_perceiver_probs(q, kv, P, Vp, B, cols, kv_bstride, kv_lstride, Lq, Lk, heads, ly.head_dim, ctx_max, lim_out,
                                          L.stream);
      });
      // (work: the three value rows of a column through to_out, and the Lq x Lk combinations per head)
      timed(m, L.stream, K_PERCEIVER_OUT, 2.0 * (double)n_cols * Lk * Dd * inner + 2.0 * (double)n_rows * Dd * heads * Lk, [&] {
        return aurora_hip_perceiver_out(Vp, ly.to_out_s, inner, P, nullptr, o, Dd, n_cols, Lq, Lk, heads, ly.head_dim, Dd, ctx_max,
                                        lim_out, L.stream);
      });
      // Values outside fp16's range (the same word decides, on the device): the plain pair -- attention output in fp32, to_out
      // on three bf16 terms -- runs instead; inside the range both launches retire at once.
      timed(m, L.stream, K_PERCEIVER_ATTENTION, 0.0, [&] {
        if (scores)
          return aurora_hip_perceiver_attention_scores(kv, kv_ld, inner, att, B, cols, kv_bstride, kv_lstride, Lq, Lk, heads,
                                                       ly.head_dim, nullptr, 0.f, ctx_max, lim_out, L.stream);
        return aurora_hip_perceiver_attention_unless(q, q_stride, kv, att, B, cols, kv_bstride, kv_lstride, Lq, Lk, heads,
                                                     ly.head_dim, AURORA_F32, nullptr, 0.f, ctx_max, lim_out, L.stream);
      });
      L.linear(att, inner, ly.to_out, inner, nullptr, o, Dd, n_rows, Dd, inner, AURORA_F32, 0, nullptr, 0, nullptr, 0, 1, ctx_max, lim_out);
    } else {
      timed(m, L.stream, K_PERCEIVER_ATTENTION, 0.0, [&] {
        if (scores)
          return aurora_hip_perceiver_attention_scores(kv, kv_ld, inner, att, B, cols, kv_bstride, kv_lstride, Lq, Lk, heads,
                                                       ly.head_dim, att_pairs ? ctx_max : nullptr, lim_out, nullptr, 0.f, L.stream);
        return aurora_hip_perceiver_attention_ex(q, q_stride, kv, att, B, cols, kv_bstride, kv_lstride, Lq, Lk, heads, ly.head_dim,
                                                 AURORA_F32, att_pairs ? ctx_max : nullptr, lim_out, L.stream);
      });
      guarded(att, inner, ly.to_out, ly.to_out_s, o, Dd, n_rows, Dd, inner, lim_out, att_pairs);
    }
    // The MLP in the fp16-pair layout end to end: the LayerNorm writes its result already split (and ONLY split), fc1
    // reads that and writes its GELU'd result split, fc2 reads that -- neither GEMM splits anything -- and the LayerNorm
    // behind the MLP takes the split array as its residual.
    const bool pairs = ly.fc1_s && ly.fc2_s && Dd % 32 == 0;
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
    // fc1 sees a LayerNorm output (|x| <= sqrt(D) * gain), fc2 its GELU: bounded whatever the inputs are.  Row chunks of the
    // pair fc1 -> fc2, so that the hidden layer of a chunk fits the scratch region (to_out's result is dead by now).
    float* hid = (float*)S;
    int64_t chunk_rows = std::min<int64_t>(n_rows, (int64_t)(s_bytes / hid_row));
    if (chunk_rows < n_rows) chunk_rows = chunk_rows / 256 * 256;   // whole row tiles per chunk (>= 256 rows fit: hid_min)
    REQUIRE(chunk_rows >= 1 && (size_t)chunk_rows * hid_row <= s_bytes, "resampler: scratch region too small for the MLP");
    for (int64_t r0 = 0; r0 < n_rows; r0 += chunk_rows) {
      const int64_t nr = std::min(chunk_rows, n_rows - r0);
      const float* a_ = lat1 + (size_t)r0 * Dd;
      float* y_ = y + (size_t)r0 * Dd;
      if (pairs) {
        const int all = 2 | AURORA_F32_A_SPLIT | AURORA_F32_W_SPLIT;
        L.linear(a_, Dd, ly.fc1_s, Dd, ly.fc1_b, hid, ly.hidden, nr, ly.hidden, Dd, AURORA_F32, AURORA_ACT_GELU, nullptr, 0, nullptr, 0,
                 all | AURORA_F32_C_SPLIT);
        L.linear(hid, ly.hidden, ly.fc2_s, ly.hidden, ly.fc2_b, y_, Dd, nr, Dd, ly.hidden, AURORA_F32, 0, nullptr, 0, nullptr, 0, all);
      } else {
        L.linear(a_, Dd, ly.fc1_w, Dd, ly.fc1_b, hid, ly.hidden, nr, ly.hidden, Dd, AURORA_F32, AURORA_ACT_GELU, nullptr, 0, nullptr, 0,
                 ly.f16_mode);
        L.linear(hid, ly.hidden, ly.fc2_w, ly.hidden, ly.fc2_b, y_, Dd, nr, Dd, ly.hidden, AURORA_F32, 0, nullptr, 0, nullptr, 0,
                 ly.f16_mode);
      }
    }
    // (`out_pairs`: the LAST layer's result leaves in the fp16-pair layout, in place -- a row is in registers before any of
    // it is written --, for a consumer that multiplies it without splitting anything: the decoder's output heads)
    const bool y_pairs = out_pairs && pairs && i + 1 == rs.layers.size();
    if (pairs)
      timed(m, L.stream, K_LAYERNORM, 0.0, [&] {
        return aurora_hip_layernorm_split(y, Dd, ly.ln2_w, ly.ln2_b, lat1, Dd, 0, 1, y_pairs ? nullptr : y, Dd, y_pairs ? y : nullptr, Dd,
                                          n_rows, Dd, eps, L.stream);
      });
    else L.layernorm(y, Dd, ly.ln2_w, ly.ln2_b, lat1, Dd, 0, y, Dd, nullptr, 0, n_rows, Dd, eps, AURORA_F32);
    m.arena.top = after_y;            // temporaries of this layer are dead (a previous layer's result stays below y)
    lat = y;
    if (i == 0) out_mark = mark0;
  }
  return lat;
}


inline int index_of(const std::vector<std::string>& v, const std::string& s) {
  const auto it = std::find(v.begin(), v.end(), s);
  return it == v.end() ? -1 : (int)(it - v.begin());
}

// patchify descriptor of one input channel (embed.hip): where its pixels come from, its normalisation, its transform
aurora_patch_var channel_desc(const Model& m, const aurora_hip_step_io& io, const Channel& ch, bool atmos_level, int C) {
  const float* st = m.stats.f();
  aurora_patch_var d{};
  d.transform = ch.transform; d.tw0 = ch.tw0; d.tw1 = ch.tw1; d.tb = ch.tb;
  switch (ch.kind) {
    case SRC_SURF:
      d.src = io.surf[ch.src];
      d.stride_b = io.surf_strides[0]; d.stride_t = io.surf_strides[1]; d.stride_h = io.surf_strides[2]; d.stride_w = io.surf_strides[3];
      d.loc = st + m.surf_stat_off[ch.src]; d.inv_scale = d.loc + 2;
      break;
    case SRC_STATIC:   // broadcast over batch / history (/ level)
      d.src = io.stat[ch.src];
      d.stride_h = io.static_strides[0]; d.stride_w = io.static_strides[1];
      if (atmos_level) { d.loc = st + m.static_lvl_stat_off[ch.src]; d.inv_scale = d.loc + 2 * C; }
      else { d.loc = st + m.static_stat_off[ch.src]; d.inv_scale = d.loc + 2; }
      break;
    case SRC_DYN:      // one constant plane per batch element (encoder.py:226-246), identity normalisation
      d.src = m.dyn_planes.f() + (size_t)ch.src * m.abs_B;
      d.stride_b = 1;
      d.loc = st + m.one_stat_off; d.inv_scale = d.loc + 2 * C;
      break;
    case SRC_ATMOS:
      d.src = io.atmos[ch.src];
      d.stride_b = io.atmos_strides[0]; d.stride_t = io.atmos_strides[1]; d.stride_c = io.atmos_strides[2];
      d.stride_h = io.atmos_strides[3]; d.stride_w = io.atmos_strides[4];
      d.loc = st + m.atmos_stat_off[ch.src]; d.inv_scale = d.loc + 2 * C;
      break;
  }
  return d;
}

bool channel_present(const aurora_hip_step_io& io, const Channel& ch) {
  switch (ch.kind) {
    case SRC_SURF: return io.surf[ch.src] != nullptr;
    case SRC_STATIC: return io.stat != nullptr && io.stat[ch.src] != nullptr;
    case SRC_ATMOS: return io.atmos[ch.src] != nullptr;
    default: return true;
  }
}

}  // namespace

void run_step(Model& m, const StepIO& s, void* stream) {
  Launcher L{m, stream};
  Arena& A = m.arena;
  A.top = 0;
  const aurora_hip_step_io& io = *s.io;
  const int B = s.B, T = s.T, P = m.P, D = m.D, Hp = m.Hp, Wp = m.Wp, Cl = m.Cl, C = m.n_levels;
  const int64_t Lp = (int64_t)Hp * Wp;          // patches per level (of this rank's rows)
  const int PP = P * P;
  const float* st = m.stats.f();
  const bool sharded = m.sharded();
  const int rank = m.band.rank, world = m.band.world;
  const int new_step = io.rollout_step + 1;
  const bool clamp_now = m.clamp_first ? new_step >= 1 : new_step > 1;   // aurora.py:368-388

  // the four guard words of the step's operand-split decisions (0: encoder context when it is not part of the guarded
  // chain, 1: atmospheric / 2: surface patch-embedding inputs, 3: decoder context), cleared by ONE launch; their producers
  // fold the maxima in (patchify) or scan (absmax_fold)
  if (!m.dry) ok(aurora_hip_zero_words(m.ctx_max.f(), 4, stream));
  // ================= encoder (encoder.py:198-366) =================
  float* x_f = (float*)A.take((size_t)B * Cl * Lp * D * 4);                      // residual stream of stage 0 (fp32)
  void* x_b = m.autocast ? A.take((size_t)B * Cl * Lp * D * 2) : nullptr;        // bf16 shadow (GEMM operand)
  const size_t after_x = A.top;
  {
    // ---- surface level: normalise + unfold, patch embedding, MLP, LayerNorm ----
    std::vector<char> present(m.surf_channels.size());
    for (size_t i = 0; i < present.size(); ++i) present[i] = channel_present(io, m.surf_channels[i]);
    const EmbedPack& ps = embed_pack(m, 0, T, present);
    const int K_s = ps.K, Kpad_s = ps.Kpad;
    const float* w_s = ps.w.f();
    float* A_s = (float*)A.take((size_t)B * Lp * Kpad_s * 4);
    std::vector<aurora_patch_var> descs;
    for (int ci : ps.channels) descs.push_back(channel_desc(m, io, m.surf_channels[ci], false, C));
    const bool surf_guarded = m.surf_chain && ps.ws.p != nullptr;
    for (size_t i = 0; i < descs.size(); i += 32)
      timed(m, stream, K_PATCHIFY, 0.0, [&] {
        return aurora_hip_patchify_absmax(descs.data() + i, (int)std::min<size_t>(32, descs.size() - i), A_s, Kpad_s,
                                          (int)i * T * PP, K_s, B, T, 1, Hp, Wp, P, AURORA_F32,
                                          surf_guarded ? m.ctx_max.f() + 2 : nullptr, stream);
      });
    float* xs0 = (float*)A.take((size_t)B * Lp * D * 4);
    const int hid_s = (int)m.T_("encoder.surf_mlp.net.0.weight").shape[0];
    float* hid = (float*)A.take((size_t)B * Lp * hid_s * 4);
    float* y = (float*)A.take((size_t)B * Lp * D * 4);
    // Guarded like the atmospheric chain: max |normalised input| once, then every linear takes two fp16 terms iff the
    // bound that word implies for ITS activation operand is inside fp16's range -- embedding: the input itself; first
    // MLP linear: |xs0| <= l1_e * w + c; second: |GELU(h)| <= |h| <= l1_0 * (l1_e * w + c) + |b0| -- else three bf16 terms.
    const void* w_s_s = ps.ws.p;
    if (surf_guarded) {
      float* word = m.ctx_max.f() + 2;   // max |normalised surface input|, folded in by patchify above
      const float l1e = ps.l1;
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
    std::vector<char> apresent(m.atmos_channels.size());
    for (size_t i = 0; i < apresent.size(); ++i) apresent[i] = channel_present(io, m.atmos_channels[i]);
    const EmbedPack& pa = embed_pack(m, 1, T, apresent);
    const int K_a = pa.K, Kpad_a = pa.Kpad;
    float* A_a = (float*)A.take((size_t)C * B * Lp * Kpad_a * 4);
    std::vector<aurora_patch_var> adescs;
    for (int ci : pa.channels) adescs.push_back(channel_desc(m, io, m.atmos_channels[ci], true, C));
    const void* w_a_s = pa.ws.p;
    bool chain = w_a_s != nullptr;
    for (const auto& ly : m.enc_rs.layers) chain = chain && ly.f16_mode == 2 && ly.to_kv_s != nullptr;
    for (size_t i = 0; i < adescs.size(); i += 32)
      timed(m, stream, K_PATCHIFY, 0.0, [&] {
        return aurora_hip_patchify_absmax(adescs.data() + i, (int)std::min<size_t>(32, adescs.size() - i), A_a, Kpad_a,
                                          (int)i * T * PP, K_a, B, T, C, Hp, Wp, P, AURORA_F32,
                                          chain ? m.ctx_max.f() + 1 : nullptr, stream);
      });
    float* xa = (float*)A.take((size_t)C * B * Lp * D * 4);
    const int64_t R = (int64_t)B * Lp;
    // The patch embedding and the level aggregation's to_kv as one guarded chain: max |normalised input| is measured
    // once (a third of the bytes of the embeddings the resampler would otherwise scan), and if it is inside fp16's range
    // -- together with the bound it implies for the embeddings, |x| <= l1 * max|input| + max|bias| -- the embedding runs
    // on two fp16 terms and writes fp16 PAIRS, which to_kv multiplies without splitting anything; otherwise both run on
    // three bf16 terms over fp32 buffers.  One word and one limit decide format and kernels together.
    // All C levels are ONE strided-batch launch: level c reads rows [c R, (c + 1) R) of the unfolded input, its own bias
    // (level embedding + patch bias) and -- level-conditioned models (levelcond.py:36-69) -- its own weight.
    CtxGuard cg{};
    const int64_t sw = pa.groups > 1 ? (int64_t)D * Kpad_a : 0;
    if (chain) {
      float* word = m.ctx_max.f() + 1;   // max |normalised atmospheric input|, folded in by patchify above
      const float l1 = pa.l1, cb = m.enc_bias_max;
      cg = CtxGuard{word, l1, cb, std::min(F16_SAFE, (F16_SAFE - cb) / l1), true};
      L.linear(A_a, Kpad_a, w_a_s, Kpad_a, m.enc_bias.f(), xa, D, R, D, Kpad_a, AURORA_F32, 0, nullptr, 0, nullptr, 0,
               2 | AURORA_F32_W_SPLIT | AURORA_F32_C_SPLIT, cg.word, cg.limit_kv, C, R * Kpad_a, sw, D, R * D);
      L.linear(A_a, Kpad_a, pa.w.f(), Kpad_a, m.enc_bias.f(), xa, D, R, D, Kpad_a, AURORA_F32, 0, nullptr, 0, nullptr, 0, 1, cg.word,
               cg.limit_kv, C, R * Kpad_a, sw, D, R * D);
    } else {
      L.linear(A_a, Kpad_a, pa.w.f(), Kpad_a, m.enc_bias.f(), xa, D, R, D, Kpad_a, AURORA_F32, 0, nullptr, 0, nullptr, 0, -1, nullptr,
               0.f, C, R * Kpad_a, sw, D, R * D);
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

  // ================= backbone (swin3d.py:884-936) =================
  const int bb = m.bb();
  const size_t es = m.bbs();
  const bool bf = m.autocast;
  const AttnSet& aw = attn_weights(m, lora_key(m, io.rollout_step), stream);
  const int n = m.n_stages;
  std::vector<float*> skips;
  size_t bi = 0;
  // token grid of this rank at a stage: the whole grid, or its band of latitude rows
  auto local_res = [&](int stage) {
    Res r = m.stage_res[stage];
    if (sharded) r.h = m.rows[stage][rank][1] - m.rows[stage][rank][0];
    return r;
  };
  // x_cat (B*L0, 2*D0): decoder output | encoder stage-0 output -- allocated now so that it survives the stack
  const int64_t L0 = (int64_t)Cl * Lp;
  float* x_cat = (float*)A.take((size_t)B * L0 * 2 * D * 4);

  auto run_blocks = [&](int count, float* xf, void* xb, int stage, float* final_out, int64_t final_ld) {
    const Res res = local_res(stage);
    const int64_t Ls = (int64_t)res.c * res.h * res.w, M = (int64_t)B * Ls;
    for (int k = 0; k < count; ++k, ++bi) {
      const Block& blk = m.blocks[bi];
      const int dim = blk.dim;
      const void* a_in = bf ? xb : (const void*)xf;
      const size_t mark = A.top;
      void* ao = nullptr;
      // bf16 blocks: q | k | v in head planes (head h: [rows][q | k | v = 192]) -- what the attention gathers per (token,
      // head) is then 384 contiguous bytes, and a window's runs of consecutive tokens are contiguous in DRAM
      // (m.qkv_planes, AURORA_QKV_PLANES=0 at creation: rows of 3 dim).  Same bytes, same arithmetic; the planes of a
      // band have own + halo rows.
      const bool planes = bf && m.qkv_planes;
      int64_t plane_stride = 0;   // elements; set where qkv is allocated
      auto attend = [&](const void* qkv, const int32_t* tok, const uint8_t* grp, int n_windows, int n_tok, int64_t Lq, int64_t Lo) {
        // algorithmic bytes: q, k, v read + o written once over the (padded) windows
        timed(m, stream, K_WINDOW_ATTENTION, 4.0 * B * n_windows * n_tok * dim * es, [&] {
          return aurora_hip_window_attention_planes(qkv, plane_stride, blk.qkv_b, ao, tok, grp, B, Lq, Lo, dim, blk.heads, n_windows,
                                                    n_tok, bb, stream);
        });
      };
      if (!sharded) {
        void* qkv = A.take((size_t)M * 3 * dim * es);
        if (planes) {
          plane_stride = M * 192;
          L.linear_planes(a_in, dim, aw.qkv[bi], dim, blk.qkv_b, qkv, plane_stride, blk.heads, M, 3 * dim, dim);
        } else {
          L.linear(a_in, dim, aw.qkv[bi], dim, blk.qkv_b, qkv, 3 * dim, M, 3 * dim, dim, bb);
        }
        const DevTables& tb = tables_for(m, stage, blk.shifted);
        ao = A.take((size_t)M * dim * es);
        attend(qkv, (const int32_t*)tb.tok.p, tb.has_grp ? (const uint8_t*)tb.grp.p : nullptr, tb.n_windows, tb.n_tok, Ls, Ls);
      } else {
        // A band: the attention table indexes [own rows | halo rows]; outputs are written for owned tokens only.
        const DevPlan& pl = plan_for(m, stage, blk.shifted);
        REQUIRE(pl.n_own == Ls, "band plan of stage %d holds %d rows, the step %lld", stage, pl.n_own, (long long)Ls);
        const int64_t Lq = Ls + pl.n_halo;
        char* qkv = (char*)A.take((size_t)Lq * 3 * dim * es);
        REQUIRE(!planes || B == 1, "a latitude band runs one batch element");
        if (planes) plane_stride = Lq * 192;
        ao = A.take((size_t)M * dim * es);
        const int32_t* tok = (const int32_t*)pl.tok.p;
        const uint8_t* grp = pl.has_grp ? (const uint8_t*)pl.grp.p : nullptr;
        const bool exchange = pl.n_halo > 0 || pl.send_cnt[0] > 0 || pl.send_cnt[1] > 0;
        if (exchange) {
          // What travels is the INPUT of the block, not k | v: the halo rows' activations (dim wide: half the bytes of
          // k | v, a third of q | k | v) leave before this rank's own qkv GEMM is even launched, so the transfer has that
          // GEMM and the interior windows to hide under; the receiver projects the halo rows to k | v itself (a small GEMM
          // straight into the halo region of `qkv`: no placement copy).  A halo row is only ever a key / value -- its own
          // rank computes its queries.
          const int64_t row_bytes = (int64_t)dim * es;
          const int n_send = pl.send_cnt[0] + pl.send_cnt[1], n_recv = pl.recv_cnt[0] + pl.recv_cnt[1];
          REQUIRE(m.dry || (std::max(n_send, n_recv) * row_bytes <= m.staging_bytes && m.stage_send && m.stage_recv),
                  "band staging buffers are missing or too small");
          if (n_send > 0)   // one launch packs the rows for both neighbours: the previous rank's first, the next rank's behind
            timed(m, stream, K_GATHER, 0.0, [&] {
              return aurora_hip_gather_rows(a_in, row_bytes, (const int32_t*)pl.send_idx.p, m.stage_send, row_bytes, n_send, row_bytes,
                                            stream);
            });
          aurora_hip_halo_msg sends[2], recvs[2];
          int ns = 0, nr = 0;
          for (int side = 0; side < 2; ++side) {
            const int peer = side == 0 ? rank - 1 : rank + 1;
            if (pl.send_cnt[side] > 0)
              sends[ns++] = aurora_hip_halo_msg{peer, 0, (side == 0 ? 0 : pl.send_cnt[0]) * row_bytes, pl.send_cnt[side] * row_bytes};
            if (pl.recv_cnt[side] > 0)
              recvs[nr++] = aurora_hip_halo_msg{peer, 0, (side == 0 ? 0 : pl.recv_cnt[0]) * row_bytes, pl.recv_cnt[side] * row_bytes};
          }
          if (!m.dry) {
            const int rc = m.band.post(m.band.user, sends, ns, recvs, nr, stream);
            REQUIRE(rc == 0, "the host's halo `post` callback failed (%d)", rc);
          }
        }
        if (planes) L.linear_planes(a_in, dim, aw.qkv[bi], dim, blk.qkv_b, qkv, plane_stride, blk.heads, M, 3 * dim, dim);
        else L.linear(a_in, dim, aw.qkv[bi], dim, blk.qkv_b, qkv, 3 * dim, M, 3 * dim, dim, bb);
        if (exchange) {
          // The halo rows were posted ahead of the qkv GEMM above, so the transfer has that whole GEMM to hide under.  By
          // default ALL windows then run in one launch behind the halo projection: a band's interior / boundary launches
          // are latency-bound (~15 us each for a few hundred workgroups), two of them cost a rank 0.4 ms per step.
          // `split_attention` (AURORA_BAND_SPLIT_ATTENTION=1 at creation) keeps the interior windows as a launch of their
          // own in front of `wait`, for transports that need those extra microseconds of cover.
          if (m.split_attention && pl.n_interior > 0) attend(qkv, tok, grp, pl.n_interior, pl.n_tok, Lq, Ls);
          if (!m.dry) {
            const int rc = m.band.wait(m.band.user, stream);
            REQUIRE(rc == 0, "the host's halo `wait` callback failed (%d)", rc);
          }
          // k | v of the received rows: rows [dim, 3 dim) of the qkv weight, written into columns [dim, 3 dim) of the halo rows
          const char* w_kv = (const char*)aw.qkv[bi] + (size_t)dim * dim * es;
          const int n_recv = pl.recv_cnt[0] + pl.recv_cnt[1];
          const int first = pl.recv_cnt[0] > 0 ? pl.recv_off[0] : pl.recv_off[1];   // the two neighbours' halo rows are adjacent
          if (n_recv > 0 && planes)   // k | v of rows Ls + first ... of every head's plane (64 elements into the row: behind q)
            L.linear_planes(m.stage_recv, dim, w_kv, dim, blk.qkv_b + dim, qkv + ((size_t)(Ls + first) * 192 + 64) * es, plane_stride,
                            blk.heads, n_recv, 2 * dim, dim);
          else if (n_recv > 0)
            L.linear(m.stage_recv, dim, w_kv, dim, blk.qkv_b + dim, qkv + ((size_t)(Ls + first) * 3 * dim + dim) * es, 3 * dim, n_recv,
                     2 * dim, dim, bb);
          const int w0 = (m.split_attention && pl.n_interior > 0) ? pl.n_interior : 0;
          if (pl.n_windows > w0)
            attend(qkv, tok + (size_t)w0 * pl.n_tok, grp ? grp + (size_t)w0 * pl.n_tok : nullptr, pl.n_windows - w0, pl.n_tok, Lq, Ls);
        } else {
          attend(qkv, tok, grp, pl.n_windows, pl.n_tok, Lq, Ls);
        }
      }
      // D = 512 under autocast: the linear, its AdaLN and the residual add are ONE launch (a workgroup owns whole rows)
      // m.fuse_ln (AURORA_FUSE_LN when the handle was created): 0 never, 1 (default) by the fill rule below, 2 always
      const int fuse_env = m.fuse_ln;
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
      const Res g = m.stage_res[i], r = local_res(i);
      REQUIRE(g.h > 1 && g.w > 1, "grid (%d, %d, %d) too small to merge", g.c, g.h, g.w);
      const int dim = m.stage_dim(i);
      const int H2 = (r.h + 1) / 2, W2 = (r.w + 1) / 2;
      REQUIRE(H2 == local_res(i + 1).h, "band rows of stages %d / %d do not nest", i, i + 1);
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
      const Res r = local_res(idx);
      const int dim = m.stage_dim(idx);
      const void* a_in = bf ? xb : (const void*)xf;
      // the odd bottom row of the finer stage belongs to the last band only
      const int crop_h = (sharded && rank != world - 1) ? 0 : m.merge_pad[idx - 1][0], crop_w = m.merge_pad[idx - 1][1];
      const int Ho = 2 * r.h - crop_h, Wo = 2 * r.w - crop_w;
      REQUIRE(Ho == local_res(idx - 1).h && Wo == local_res(idx - 1).w, "band rows of stages %d / %d do not nest", idx - 1, idx);
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
  {
    const size_t mark = A.top;
    // ---- surface heads on latent level 0 ----
    const HeadGroup& hs = m.head_surf;
    const int n_s = (int)hs.names.size() * PP, ld_s = round_up(n_s, 4);
    float* y_s = (float*)A.take((size_t)B * Lp * ld_s * 4);
    for (int b = 0; b < B; ++b)
      L.linear(x_cat + (size_t)b * Cl * Lp * D2, D2, hs.w.f(), D2, hs.b.f(), y_s + (size_t)b * Lp * ld_s, ld_s, Lp, n_s, D2, AURORA_F32);
    // difference prediction (aurora.py:761-779): y + (1 + y_mod) * normalised previous state of the same variable
    auto diff_fields = [&](aurora_unpatch_var& d, const std::string& name, const std::vector<std::string>& heads, bool atmos, int src,
                           const float* stat_loc, int n_lvl) {
      d.mod_col0 = -1;
      const auto it = m.diff_index.find(name);
      const int mod = index_of(heads, name + "_mod");
      if (m.variant != 1 || it == m.diff_index.end() || mod < 0) return;
      d.mod_col0 = mod * PP;
      const int idx = it->second;
      if (atmos) {
        d.prev = io.atmos[src] + (int64_t)idx * io.atmos_strides[1];
        d.prev_sb = io.atmos_strides[0]; d.prev_sc = io.atmos_strides[2]; d.prev_sh = io.atmos_strides[3];
        REQUIRE(io.atmos_strides[4] == 1, "difference prediction needs unit longitude stride");
      } else {
        d.prev = io.surf[src] + (int64_t)idx * io.surf_strides[1];
        d.prev_sb = io.surf_strides[0]; d.prev_sc = 0; d.prev_sh = io.surf_strides[2];
        REQUIRE(io.surf_strides[3] == 1, "difference prediction needs unit longitude stride");
      }
      REQUIRE(idx < T, "difference prediction of '%s' refers to history index %d, %d states given", name.c_str(), idx, T);
      d.inv_scale = stat_loc + 2 * n_lvl;
    };
    std::vector<aurora_unpatch_var> ud;
    for (size_t v = 0; v < m.surf_out.size(); ++v) {
      const std::string& name = m.surf_out[v];
      const int src = index_of(m.surf_inputs, name);
      REQUIRE(src >= 0, "surface output '%s' is not a surface input", name.c_str());
      if (io.out_surf[v] == nullptr || io.surf[src] == nullptr) continue;
      aurora_unpatch_var d{};
      d.dst = io.out_surf[v];
      d.loc = st + m.surf_stat_off[src];
      d.scale = d.loc + 1;
      d.clamp_min0 = clamp_now && index_of(m.pos_surf, name) >= 0;
      d.angle_col0 = d.dens_col0 = -1;
      const int plain = index_of(hs.names, name);
      if (m.variant == 2 && plain < 0) {   // a direction: atan2 of its sin / cos heads (aurora.py:914-932)
        d.col0 = index_of(hs.names, name + "_sin") * PP;
        d.angle_col0 = index_of(hs.names, name + "_cos") * PP;
      } else {
        REQUIRE(plain >= 0, "no decoder head for '%s'", name.c_str());
        d.col0 = plain * PP;
      }
      diff_fields(d, name, hs.names, false, src, d.loc, 1);
      if (m.variant == 2) {
        const int dens = index_of(hs.names, name + "_density");
        if (dens >= 0) {   // keep the value only over water and where the density head says "present"
          const int wmb = index_of(m.static_vars, "wmb");
          REQUIRE(wmb >= 0 && io.stat && io.stat[wmb], "the ocean-wave variant needs the static variable 'wmb'");
          REQUIRE(io.static_strides[1] == 1, "the water-body mask needs unit longitude stride");
          d.dens_col0 = dens * PP;
          d.mask = io.stat[wmb];
          d.mask_sh = io.static_strides[0];
          d.mask_thresh = (float)m.static_loc[wmb];   // normalised value > 0
        }
      }
      ud.push_back(d);
    }
    for (size_t i = 0; i < ud.size(); i += 32)
      timed(m, stream, K_UNPATCHIFY, 0.0, [&] {
        return aurora_hip_unpatchify(y_s, ld_s, ud.data() + i, (int)std::min<size_t>(32, ud.size() - i), B, 1, Hp, Wp, P, stream);
      });

    // ---- level de-aggregation ----
    const float* ctx = x_cat + (size_t)Lp * D2;
    if (B > 1) {   // latent levels 1.. of every batch element, made contiguous
      float* ctx_copy = (float*)A.take((size_t)B * (Cl - 1) * Lp * D2 * 4);
      for (int b = 0; b < B; ++b)
        timed(m, stream, K_COPY2D, 0.0, [&] {
          return aurora_hip_copy2d(x_cat + ((size_t)b * Cl * Lp + Lp) * D2, D2, ctx_copy + (size_t)b * (Cl - 1) * Lp * D2, D2,
                                   (int64_t)(Cl - 1) * Lp, D2, AURORA_F32, stream);
        });
      ctx = ctx_copy;
    }
    // The main Perceiver decodes every variable except those named in `separate_perceiver`, which get their own
    // (decoder.py:232-248); each group: resampler -> heads (one strided-batch launch over the levels when every level has
    // its own head, levelcond.py:36-69) -> unpatchify with the post-decoder hooks fused.
    struct Group { const HeadGroup* h; const Resampler* rs; const float* q; };
    const Group groups[2] = {{&m.head_main, &m.dec_rs, m.dec_q.f()}, {&m.head_alt, &m.dec_rs_alt, m.dec_q_alt.f()}};
    for (int gi = 0; gi < (m.has_alt ? 2 : 1); ++gi) {
      const HeadGroup& hg = *groups[gi].h;
      if (hg.names.empty()) continue;
      const size_t gmark = A.top;
      // The output heads have few columns (80 at patch size 4, 500 at 10): on the native-fp32 128 x 128 kernel they ran at
      // 77 TFLOP/s.  When the Perceiver's output is bounded inside fp16's range by its LayerNorm parameters alone (it is: a few
      // hundred), its last LayerNorm writes fp16 pairs and the heads -- rows zero-padded to the 256-column tile, weights
      // pre-split -- run on the VALU-free two-term kernel instead.
      const Resampler& rs = *groups[gi].rs;
      const auto& last_ly = rs.layers.back();
      const bool last_pairs = last_ly.fc1_s && last_ly.fc2_s && last_ly.dim % 32 == 0;
      const bool two_term = hg.n_pad > 0 && last_pairs && (gi == 0 ? m.dec_out_bound : m.dec_out_bound_alt) < F16_SAFE;
      size_t rs_mark = 0;
      float* lat = resampler(m, L, rs, ctx, (int64_t)B * (Cl - 1) * Lp, D2, groups[gi].q, m.dec_queries.f(), B, Lp,
                             (int64_t)(Cl - 1) * Lp, Lp, C, Cl - 1, m.perceiver_heads, m.ln_eps, rs_mark, nullptr, two_term, 3,
                             /*scan_ctx=*/gi == 0 || m.head_main.names.empty());
      const int n_a = two_term ? hg.n_pad : (int)hg.names.size() * PP, ld_a = round_up(n_a, 4);
      const float* hw = two_term ? (const float*)hg.ws.p : hg.w.f();
      const float* hb = two_term ? hg.bs.f() : hg.b.f();
      const int mode = two_term ? (2 | AURORA_F32_A_SPLIT | AURORA_F32_W_SPLIT) : -1;
      float* y_a = (float*)A.take((size_t)B * Lp * C * ld_a * 4);
      if (hg.groups > 1)   // level c: rows (b L + l) C + c of `lat` -> the same rows of y_a, with that level's head
        L.linear(lat, (int64_t)C * D2, hw, D2, hb, y_a, (int64_t)C * ld_a, (int64_t)B * Lp, n_a, D2, AURORA_F32, 0, nullptr, 0,
                 nullptr, 0, mode, nullptr, 0.f, C, D2, (int64_t)n_a * D2, ld_a, ld_a);
      else
        L.linear(lat, D2, hw, D2, hb, y_a, ld_a, (int64_t)B * Lp * C, n_a, D2, AURORA_F32, 0, nullptr, 0, nullptr, 0, mode);
      std::vector<aurora_unpatch_var> ad;
      for (size_t hi = 0; hi < hg.names.size(); ++hi) {
        const std::string& name = hg.names[hi];
        const int v = index_of(m.atmos_vars, name);
        if (v < 0) continue;               // a `<v>_mod` head: consumed by its base variable
        if (io.out_atmos[v] == nullptr || io.atmos[v] == nullptr) continue;
        aurora_unpatch_var d{};
        d.dst = io.out_atmos[v];
        d.loc = st + m.atmos_stat_off[v];
        d.scale = d.loc + C;
        d.clamp_min0 = clamp_now && index_of(m.pos_atmos, name) >= 0;
        d.col0 = (int)hi * PP;
        d.angle_col0 = d.dens_col0 = -1;
        diff_fields(d, name, hg.names, true, v, d.loc, C);
        if (m.variant == 1 && m.use_lora && name == "so2")   // aurora.py:781-794
          for (int c = 0; c < C; ++c)
            if (m.levels[c] >= 850) d.clamp_max1_levels |= 1u << c;
        ad.push_back(d);
      }
      for (size_t i = 0; i < ad.size(); i += 32)
        timed(m, stream, K_UNPATCHIFY, 0.0, [&] {
          return aurora_hip_unpatchify(y_a, ld_a, ad.data() + i, (int)std::min<size_t>(32, ad.size() - i), B, C, Hp, Wp, P, stream);
        });
      A.top = gmark;
    }
    A.top = mark;
  }
}

}  // namespace aurora
