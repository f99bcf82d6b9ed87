// Host-side geometry: window tables, the latitude-band partition and its halo plans (see band.h).
//
// Strong scaling of ONE forecast over R devices: every rank owns a contiguous band of latitude
// rows at every backbone stage.  Window attention needs, for every window that contains an owned token, the k | v rows of
// the window's other tokens; because the attention kernel gathers through a token table, a band needs no special kernel:
// its table indexes a local buffer [own rows | halo rows].  Foreign tokens that no owned query can see are not exchanged
// at all: in a shifted block the -100 mask (swin3d.py:333-358) only lets tokens of the same group attend to each other,
// so a foreign position whose group contains no owned position of that window becomes an absent one (-1, its group label
// kept).  That is what makes the cyclic wrap of the latitude roll free: the window row holding the last three and --
// rolled -- the first three latitude rows puts them in different groups, so the first and the last rank never talk.
#include "band.h"

#include <algorithm>
#include <numeric>

#include "common.h"

namespace aurora {

WindowTables window_tables(Res res, const int window[3], bool shifted) {
  const int dims[3] = {res.c, res.h, res.w};
  int ws[3], ss[3], pad[3], front[3], nwin[3];
  bool any_shift = false;
  for (int a = 0; a < 3; ++a) {
    ws[a] = window[a];
    ss[a] = shifted ? window[a] / 2 : 0;
    if (dims[a] <= window[a]) { ws[a] = dims[a]; ss[a] = 0; }   // util.py:53-71
    pad[a] = (ws[a] - dims[a] % ws[a]) % ws[a];
    front[a] = pad[a] / 2;                                       // two-sided padding, front = pad // 2
    nwin[a] = (dims[a] + pad[a]) / ws[a];
    any_shift |= ss[a] != 0;
  }
  auto label = [&](int a, int x) {   // swin3d.py:333-342
    if (ss[a] == 0) return 2;
    return x < dims[a] - ws[a] ? 0 : x < dims[a] - ss[a] ? 1 : 2;
  };
  WindowTables t;
  t.n_windows = nwin[0] * nwin[1] * nwin[2];
  t.n_tok = ws[0] * ws[1] * ws[2];
  t.tok.resize((size_t)t.n_windows * t.n_tok);
  if (any_shift) t.grp.resize(t.tok.size());
  size_t at = 0;
  for (int c1 = 0; c1 < nwin[0]; ++c1)
    for (int h1 = 0; h1 < nwin[1]; ++h1)
      for (int w1 = 0; w1 < nwin[2]; ++w1)
        for (int wc = 0; wc < ws[0]; ++wc)
          for (int wh = 0; wh < ws[1]; ++wh)
            for (int ww = 0; ww < ws[2]; ++ww, ++at) {
              const int r[3] = {c1 * ws[0] + wc - front[0], h1 * ws[1] + wh - front[1], w1 * ws[2] + ww - front[2]};
              const bool valid = r[0] >= 0 && r[0] < dims[0] && r[1] >= 0 && r[1] < dims[1] && r[2] >= 0 && r[2] < dims[2];
              int g = 0, o[3];
              for (int a = 0; a < 3; ++a) {
                o[a] = ((r[a] + ss[a]) % dims[a] + dims[a]) % dims[a];   // torch.roll(x, -s): rolled[i] = x[(i + s) % n]
                const int cl = r[a] < 0 ? 0 : r[a] >= dims[a] ? dims[a] - 1 : r[a];
                int la = label(a, cl);
                if (a == 2 && la == 1) la = 2;   // longitude wraps: W slices 1 and 2 communicate
                g = g * 3 + la;
              }
              t.tok[at] = valid ? (o[0] * res.h + o[1]) * res.w + o[2] : -1;
              if (any_shift) t.grp[at] = (uint8_t)(valid ? g : 27);
            }
  return t;
}

std::vector<Res> stage_resolutions(Res res0, int n_stages) {
  std::vector<Res> out{res0};
  for (int s = 1; s < n_stages; ++s) {
    const Res r = out.back();
    out.push_back({r.c, (r.h + r.h % 2) / 2, (r.w + r.w % 2) / 2});
  }
  return out;
}

namespace {

// Owned rows of every rank at every stage from boundaries on the coarsest stage (doubled per finer stage, clipped to the
// stage's rows); false if a rank ends up without rows somewhere.
bool rows_from_bounds(const std::vector<Res>& all_res, const std::vector<int>& bounds,
                      std::vector<std::vector<std::array<int, 2>>>& rows, int* bad_rank, int* bad_stage) {
  const int n = (int)all_res.size(), world = (int)bounds.size() - 1;
  rows.assign(n, {});
  for (int s = 0; s < n; ++s) {
    const int mult = 1 << (n - 1 - s), Hs = all_res[s].h;
    for (int r = 0; r < world; ++r) {
      const int h0 = std::min(bounds[r] * mult, Hs), h1 = r < world - 1 ? std::min(bounds[r + 1] * mult, Hs) : Hs;
      if (h1 <= h0) {
        if (bad_rank) *bad_rank = r;
        if (bad_stage) *bad_stage = s;
        return false;
      }
      rows[s].push_back({h0, h1});
    }
  }
  return true;
}

// Can every rank get what its windows need from its two neighbours?  Tokens that attend to each other -- a window's
// positions of one mask group (band_plan below drops the rest) -- must lie on at most two adjacent ranks, in both block
// flavours at every stage.  Only the latitude structure matters (a grid one window wide), and because bands are contiguous
// and ordered, a set of rows spans the ranks owner[lowest row] .. owner[highest row]: every (window, group) reduces to ONE
// row interval, and those depend on the grid alone.  They are collected once per search (GroupSpans); a candidate
// partition then costs a few hundred look-ups instead of rebuilding every stage's window tables (which made an infeasible
// 12-rank request take minutes to fail).
struct GroupSpans { std::vector<std::vector<std::array<int, 2>>> per_stage; };   // [stage] -> distinct (lowest, highest) rows

GroupSpans group_spans(const std::vector<Res>& all_res, const int window[3]) {
  GroupSpans out;
  for (size_t s = 0; s < all_res.size(); ++s) {
    const Res res{all_res[s].c, all_res[s].h, std::min(all_res[s].w, window[2])};
    std::vector<std::array<int, 2>> spans;
    for (int shifted = 0; shifted < 2; ++shifted) {
      const WindowTables t = window_tables(res, window, shifted != 0);
      for (int w = 0; w < t.n_windows; ++w) {
        int lo[28], hi[28];
        for (int g = 0; g < 28; ++g) { lo[g] = 1 << 30; hi[g] = -1; }
        for (int i = 0; i < t.n_tok; ++i) {
          const int32_t tk = t.tok[(size_t)w * t.n_tok + i];
          if (tk < 0) continue;
          const int g = t.grp.empty() ? 0 : t.grp[(size_t)w * t.n_tok + i], h = (tk / res.w) % res.h;
          lo[g] = std::min(lo[g], h);
          hi[g] = std::max(hi[g], h);
        }
        for (int g = 0; g < 28; ++g)
          if (hi[g] > lo[g]) spans.push_back({lo[g], hi[g]});
      }
    }
    std::sort(spans.begin(), spans.end());
    spans.erase(std::unique(spans.begin(), spans.end()), spans.end());
    out.per_stage.push_back(std::move(spans));
  }
  return out;
}

bool neighbours_suffice(const GroupSpans& gs, const std::vector<Res>& all_res,
                        const std::vector<std::vector<std::array<int, 2>>>& rows) {
  std::vector<int> owner;
  for (size_t s = 0; s < all_res.size(); ++s) {
    owner.assign(all_res[s].h, -1);
    for (size_t r = 0; r < rows[s].size(); ++r)
      for (int h = rows[s][r][0]; h < rows[s][r][1]; ++h) owner[h] = (int)r;
    for (const auto& sp : gs.per_stage[s])
      if (owner[sp[1]] - owner[sp[0]] > 1) return false;
  }
  return true;
}

}  // namespace

bool band_rows(const std::vector<Res>& all_res, const int window[3], int world,
               std::vector<std::vector<std::array<int, 2>>>& rows) {
  const int n = (int)all_res.size();
  const int Hc = all_res.back().h;
  if (Hc < world) {
    set_error("cannot split %d latitude rows of the coarsest stage over %d ranks", Hc, world);
    return false;
  }
  // First choice: boundaries on window rows of the finer stages (then un-shifted blocks need no halo there).
  int unit = n > 1 ? window[1] / std::gcd(window[1], 2) : window[1];
  if (unit < 1 || (Hc + unit - 1) / unit < world) unit = 1;
  const int n_units = (Hc + unit - 1) / unit;
  const int base = n_units / world, extra = n_units % world;
  std::vector<int> bounds{0};
  for (int r = 0; r < world; ++r) bounds.push_back(bounds.back() + (base + (r < extra ? 1 : 0)) * unit);
  for (int& b : bounds) b = std::min(b, Hc);
  bounds.back() = Hc;
  int bad_rank = 0, bad_stage = 0;
  const bool has_rows = rows_from_bounds(all_res, bounds, rows, &bad_rank, &bad_stage);
  // ... kept unless it is badly balanced: its largest band more than 1/12 above the smallest possible largest band
  // (38 rows over 4 ranks: 12 + 9 + 9 + 8 against 10 + 10 + 9 + 9 -- the slowest rank bounds the step)
  const int m_opt = (Hc + world - 1) / world;
  int m_unit = 0;
  for (int r = 0; r < world; ++r) m_unit = std::max(m_unit, bounds[r + 1] - bounds[r]);
  const GroupSpans gs = group_spans(all_res, window);
  if (has_rows && (m_unit - m_opt) * 12 <= m_opt && neighbours_suffice(gs, all_res, rows)) return true;
  // Otherwise -- and for thin bands (many ranks for the grid: the 0.4-degree grid on 8), where a window of the aligned split
  // may reach past a whole band -- search the partitions with the smallest largest band, thick bands first, for one whose
  // windows stay within neighbouring ranks.
  // (BAND_SEARCH_BUDGET candidates, a fraction of a second; engine/partition.py searches the same candidates in the same
  // order with the same budget, so the two agree on success, on the split found and on giving up)
  std::vector<int> sizes(world, 0);
  long budget = BAND_SEARCH_BUDGET;
  for (int m = (Hc + world - 1) / world; m <= Hc; ++m) {
    // iterative depth-first search over band sizes 1 .. m that sum to Hc
    int r = 0, remaining = Hc;
    sizes[0] = std::min(m, remaining - (world - 1)) + 1;   // "one above the first candidate"
    while (r >= 0 && budget >= 0) {
      const int left = world - 1 - r;                      // bands still to size after this one
      int sz = sizes[r] - 1;                               // next candidate for band r
      const int least = std::max(1, remaining - left * m); // smaller would leave too much for the rest
      if (sz < least) {                                    // exhausted: back up
        if (--r >= 0) remaining += sizes[r];
        continue;
      }
      sizes[r] = sz;
      if (left == 0) {                                     // (then sz == remaining by the bounds above, or no fit)
        if (sz == remaining && --budget >= 0) {
          std::vector<int> b{0};
          for (int q = 0; q < world; ++q) b.push_back(b.back() + sizes[q]);
          if (rows_from_bounds(all_res, b, rows, nullptr, nullptr) && neighbours_suffice(gs, all_res, rows)) return true;
        }
        sizes[r] = least;                                  // no other size fits the last band
        continue;
      }
      remaining -= sz;
      ++r;
      sizes[r] = std::min(m, remaining - (world - 1 - r)) + 1;
    }
    if (budget < 0) break;
  }
  if (!has_rows)
    set_error("latitude-band partition: rank %d of %d ends up without rows at stage %d (%d rows)", bad_rank, world, bad_stage,
              all_res[bad_stage].h);
  else if (budget < 0)
    set_error("latitude-band partition: none of the first %ld splits of %d coarsest-stage rows over %d ranks (most balanced first) "
              "keeps every window within two neighbouring ranks; search stopped (bands too thin: use fewer ranks)",
              (long)BAND_SEARCH_BUDGET, Hc, world);
  else
    set_error("latitude-band partition: no split of %d coarsest-stage rows over %d ranks keeps every window within two "
              "neighbouring ranks (bands too thin)", Hc, world);
  return false;
}

bool band_plan(Res res, const int window[3], bool shifted, int rank, const std::vector<std::array<int, 2>>& rows,
               BandPlan& out) {
  const int H = res.h, W = res.w, world = (int)rows.size();
  const WindowTables t = window_tables(res, window, shifted);
  const int N = t.n_tok, nW = t.n_windows;
  const bool has_grp = !t.grp.empty();
  std::vector<int> owner_of_row(H, -1);
  for (int r = 0; r < world; ++r)
    for (int h = rows[r][0]; h < rows[r][1]; ++h) owner_of_row[h] = r;
  auto owner_of = [&](int32_t tok) { return tok < 0 ? -1 : owner_of_row[(tok / W) % H]; };
  auto local_index = [&](int32_t tok, int q) {   // (c * rows + h - h0) * W + w inside rank q's band
    const int h0 = rows[q][0], h1 = rows[q][1];
    const int c = tok / (H * W), rem = tok % (H * W), h = rem / W, w = rem % W;
    return (int32_t)((c * (h1 - h0) + (h - h0)) * W + w);
  };
  // What rank q lists as foreign (sorted by owning rank, then by token id), and -- for q == rank -- its filtered tables.
  struct Mine { std::vector<int> windows; std::vector<int32_t> tok; std::vector<int> own; };
  auto scan = [&](int q, std::vector<int32_t>& foreign, std::vector<int>& f_owner, Mine* mine) {
    std::vector<int32_t> seen;
    for (int w = 0; w < nW; ++w) {
      const int32_t* tw = &t.tok[(size_t)w * N];
      const uint8_t* gw = has_grp ? &t.grp[(size_t)w * N] : nullptr;
      bool touches = false;
      uint32_t bits = 0;
      for (int i = 0; i < N; ++i)
        if (owner_of(tw[i]) == q) {
          touches = true;
          if (gw) bits |= 1u << gw[i];
        }
      if (!touches) continue;
      if (mine) mine->windows.push_back(w);
      for (int i = 0; i < N; ++i) {
        int32_t tk = tw[i];
        int ow = owner_of(tk);
        // a foreign position whose mask group holds no owned position of the same window is never seen: dropped
        if (ow != q && ow >= 0 && gw && !((bits >> gw[i]) & 1u)) { tk = -1; ow = -1; }
        if (ow != q && ow >= 0) seen.push_back(tk);
        if (mine) { mine->tok.push_back(tk); mine->own.push_back(ow); }
      }
    }
    std::sort(seen.begin(), seen.end());
    seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
    std::stable_sort(seen.begin(), seen.end(), [&](int32_t a, int32_t b) { return owner_of(a) < owner_of(b); });
    foreign = seen;
    f_owner.resize(foreign.size());
    for (size_t i = 0; i < foreign.size(); ++i) f_owner[i] = owner_of(foreign[i]);
  };

  Mine mine;
  std::vector<int32_t> foreign;
  std::vector<int> f_owner;
  scan(rank, foreign, f_owner, &mine);
  out = BandPlan{};
  out.n_tok = N;
  out.n_windows = (int)mine.windows.size();
  out.n_own = res.c * (rows[rank][1] - rows[rank][0]) * W;
  out.n_halo = (int)foreign.size();
  for (size_t i = 0; i < foreign.size(); ++i) {
    const int side = f_owner[i] == rank - 1 ? 0 : f_owner[i] == rank + 1 ? 1 : -1;
    if (side < 0) {
      set_error("latitude-band partition: rank %d needs halo rows of rank %d, not a neighbour (bands too thin for %d ranks)",
                rank, f_owner[i], world);
      return false;
    }
    if (out.recv_cnt[side] == 0) out.recv_off[side] = (int)i;
    out.recv_cnt[side] += 1;
  }
  // local token table: owned -> index inside the band, foreign -> n_own + position in the halo list
  std::vector<int32_t> loc(mine.tok.size());
  std::vector<char> boundary(out.n_windows, 0);
  for (size_t i = 0; i < mine.tok.size(); ++i) {
    const int32_t tk = mine.tok[i];
    if (tk < 0 || mine.own[i] < 0) { loc[i] = -1; continue; }
    if (mine.own[i] == rank) { loc[i] = local_index(tk, rank); continue; }
    const auto it = std::lower_bound(foreign.begin() + out.recv_off[mine.own[i] == rank - 1 ? 0 : 1],
                                     foreign.begin() + out.recv_off[mine.own[i] == rank - 1 ? 0 : 1] +
                                         out.recv_cnt[mine.own[i] == rank - 1 ? 0 : 1], tk);
    loc[i] = out.n_own + (int32_t)(it - foreign.begin());
    boundary[i / N] = 1;
  }
  // interior windows first (stable), boundary windows behind them
  std::vector<int> order;
  for (int w = 0; w < out.n_windows; ++w) if (!boundary[w]) order.push_back(w);
  out.n_interior = (int)order.size();
  for (int w = 0; w < out.n_windows; ++w) if (boundary[w]) order.push_back(w);
  out.tok.resize(loc.size());
  if (has_grp) out.grp.resize(loc.size());
  for (int k = 0; k < out.n_windows; ++k) {
    const int w = order[k];
    std::copy(loc.begin() + (size_t)w * N, loc.begin() + (size_t)(w + 1) * N, out.tok.begin() + (size_t)k * N);
    if (has_grp)
      std::copy(t.grp.begin() + (size_t)mine.windows[w] * N, t.grp.begin() + (size_t)(mine.windows[w] + 1) * N,
                out.grp.begin() + (size_t)k * N);
  }
  // what I send = what my neighbours list as foreign and I own, in THEIR order
  for (int side = 0; side < 2; ++side) {
    const int q = side == 0 ? rank - 1 : rank + 1;
    if (q < 0 || q >= world) continue;
    std::vector<int32_t> fq;
    std::vector<int> oq;
    scan(q, fq, oq, nullptr);
    for (size_t i = 0; i < fq.size(); ++i)
      if (oq[i] == rank) out.send_idx[side].push_back(local_index(fq[i], rank));
  }
  return true;
}

}  // namespace aurora

using namespace aurora;

extern "C" int aurora_hip_band_partition(int n_stages, const int32_t res0[3], const int32_t window[3], int world, int rank,
                                         int stage, int32_t* h0, int32_t* h1) {
  AURORA_CHECK_ARG(res0 && window && h0 && h1 && n_stages >= 1 && n_stages <= 8 && world >= 1 && rank >= 0 && rank < world &&
                       stage >= 0 && stage < n_stages, "band_partition: bad argument");
  const int win[3] = {window[0], window[1], window[2]};
  std::vector<std::vector<std::array<int, 2>>> rows;
  if (!band_rows(stage_resolutions(Res{res0[0], res0[1], res0[2]}, n_stages), win, world, rows)) return AURORA_E_ARG;
  *h0 = rows[stage][rank][0];
  *h1 = rows[stage][rank][1];
  return AURORA_OK;
}

extern "C" int aurora_hip_band_plan(const int32_t res[3], const int32_t window[3], int shifted, int world, int rank,
                                    const int32_t* rows, aurora_hip_plan_info* info, int32_t* tok, uint8_t* grp,
                                    int32_t* send_idx_prev, int32_t* send_idx_next) {
  AURORA_CHECK_ARG(res && window && rows && info && world >= 1 && rank >= 0 && rank < world, "band_plan: bad argument");
  const int win[3] = {window[0], window[1], window[2]};
  std::vector<std::array<int, 2>> r(world);
  for (int q = 0; q < world; ++q) r[q] = {rows[2 * q], rows[2 * q + 1]};
  BandPlan p;
  if (!band_plan(Res{res[0], res[1], res[2]}, win, shifted != 0, rank, r, p)) return AURORA_E_ARG;
  *info = aurora_hip_plan_info{p.n_windows, p.n_tok, p.n_own, p.n_halo, p.n_interior,
                               {p.recv_off[0], p.recv_off[1]}, {p.recv_cnt[0], p.recv_cnt[1]},
                               {(int32_t)p.send_idx[0].size(), (int32_t)p.send_idx[1].size()}, p.grp.empty() ? 0 : 1};
  if (tok) std::copy(p.tok.begin(), p.tok.end(), tok);
  if (grp && !p.grp.empty()) std::copy(p.grp.begin(), p.grp.end(), grp);
  if (send_idx_prev) std::copy(p.send_idx[0].begin(), p.send_idx[0].end(), send_idx_prev);
  if (send_idx_next) std::copy(p.send_idx[1].begin(), p.send_idx[1].end(), send_idx_next);
  return AURORA_OK;
}
