// Host-side geometry of the 3D Swin backbone and of its latitude-band partition: window token / group tables, owned
// rows per rank and stage, and the halo plans of window attention.  Pure host code (no device work); shared by the model
// handle (model.hip / step.hip) and exposed through aurora_hip_band_partition / aurora_hip_band_plan.
#pragma once

#include <stdint.h>

#include <array>
#include <vector>

namespace aurora {

struct Res { int c, h, w; };

// Window token / group tables of one block flavour (the closed form of the reference's roll -> pad -> partition chain
// and mask, swin3d.py:177-360, 471-505; Python twin: aurora_amd/engine/geometry.py, tests/test_geometry.py).
struct WindowTables {
  std::vector<int32_t> tok;   // [n_windows][n_tok] token index (c * H + h) * W + w, or -1 for a zero-padded position
  std::vector<uint8_t> grp;   // communication-group labels; empty when the block is not shifted (or fits one window)
  int n_windows = 0, n_tok = 0;
};
WindowTables window_tables(Res res, const int window[3], bool shifted);

// Token grids of the U-net stages (swin3d.py:868-882): the level axis is never merged.
std::vector<Res> stage_resolutions(Res res0, int n_stages);

// Owned latitude rows [h0, h1) of every rank at every stage: boundaries are chosen on the coarsest stage and doubled
// per finer stage, so 2 x 2 merges / splits never cross a rank (Python twin: aurora_amd/engine/partition.py:band_rows).
// Returns false (with the error set) if the coarsest stage has fewer row units than ranks.
bool band_rows(const std::vector<Res>& all_res, const int window[3], int world,
               std::vector<std::vector<std::array<int, 2>>>& rows);

// What one rank needs to run window attention of one block flavour on its band (partition.py:block_plans).
// candidates the partition search of band_rows tries before it gives up (engine/partition.py: the same number)
constexpr long BAND_SEARCH_BUDGET = 200000;

struct BandPlan {
  std::vector<int32_t> tok;      // [n_windows][n_tok] index into [own rows | halo rows]; -1 = padding / unseen
  std::vector<uint8_t> grp;      // [n_windows][n_tok] (empty when not shifted)
  int n_windows = 0, n_tok = 0, n_own = 0, n_halo = 0;
  int n_interior = 0;            // the first n_interior windows touch no halo row (they run while the halos travel)
  int recv_off[2] = {0, 0}, recv_cnt[2] = {0, 0};   // [0] from rank - 1, [1] from rank + 1: slice of the halo rows
  std::vector<int32_t> send_idx[2];                 // owned rows (local indices) to send, in the receiver's halo order
};
// `rows`: owned rows of every rank at this stage.  Returns false (error set) if a halo row belongs to a rank that is not
// an immediate neighbour (bands thinner than a window row's shift).
bool band_plan(Res res, const int window[3], bool shifted, int rank, const std::vector<std::array<int, 2>>& rows,
               BandPlan& out);

}  // namespace aurora
