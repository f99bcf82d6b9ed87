// REFERENCE COPY of a round-5 experiment -- NOT built into libaurora_hip.so.  It was part of the library at commits dada72d and
// ad67596 (aurora_amd/csrc/gemm_w4.hip, dispatched from gemm.hip's linear_impl behind AURORA_GEMM_W4_MIN_K / AURORA_GEMM_W4_STAGES;
// build.py compiled it WITHOUT -amdgpu-mfma-vgpr-form=1; gemm.hip carried an `#ifndef AURORA_GEMM_W4_TU` guard around its host
// side for the #include below): check those commits out to run it (the register-staged kernel below is one step further: inline-asm loads, forms (7) and (8) of the log: the last kernel of this file runs on 32x32x16 MFMAs.  Every variant passed tests/test_gpu_ops.py -k "linear_bf16 or
// planes"; none beat the eight-wave ping-pong kernel by more than 3 % (K >= 4096 only) -- profiles/r05_ab_gemm_w4.log, DESIGN.md 10.
//
// Four-wave form of the 256 x 256 bf16 GEMM tile: ONE wave per SIMD with the whole 512-register file, wave tile 128 x 128.
//
// Why (round 5): the ping-pong kernel (gemm.hip: eight waves, wave tile 128 x 64) was measured against the vendor library
// for the first time (profiles/r05_vendor_gemm_yardstick.log) -- it wins on the K = 512 shapes and LOSES 9-17 % on every
// K >= 1024 shape.  rocprofv3 names the vendor's kernel: MT256x256x64, MIWT8_8 (wave tile 128 x 128), 256 threads,
// 16x16 MFMAs.  The arithmetic of the difference is LDS traffic: per 32-wide K-step a 128 x 64 wave tile reads
// (128 + 64) x 64 B of fragments for 32 MFMAs, a 128 x 128 one (128 + 128) x 64 B for 64 -- 96 KiB against 64 KiB per
// CU and K-step, next to the 32 KiB the LDS-DMA writes.  The ping-pong schedule needs its load phase (~600 cycles of LDS
// reads) as long as its matrix phase (543); here the reads of stage s+1 sit in the shadow of the 64 MFMAs of stage s.
//
// Structure (per workgroup = per CU): the same 4-stage LDS ring, stage image, swizzles, tile order and epilogues as
// gemm.hip's 256 x 256 kernels (this file includes that one for them) -- so results are bit-identical: every output
// accumulates K in the same 32-wide steps.  Waves 2 (m) x 2 (n).  Registers: 256 accumulators (AGPRs), two fragment
// sets of 16 x 4 registers (stage s being multiplied, stage s+1 arriving).  Iteration s:
//     eight groups of { 2 fragment reads of stage s+1, 1 LDS-DMA piece of stage s+4 (into the slot of stage s),
//                       8 MFMAs of stage s }, pinned by sched_barrier so that the reads and the DMA issue between MFMAs;
//     s_waitcnt vmcnt(16)   this wave's pieces of stage s+2 have landed (s+3, s+4 stay in flight);
//     s_waitcnt lgkmcnt(0)  its reads of stage s+1 are done;
//     ONE s_barrier         RAW: stage s+2 is complete for everyone;  WAR: nobody reads stage s+1's slot any more.
// One barrier per 64 MFMAs (the ping-pong form: two per 32).  The ring is refilled unconditionally (stages past the
// end re-load the last one into a dead slot), so every iteration is the same code and the counted waits never change.
#define AURORA_GEMM_W4_TU 1
#include "gemm.hip"

namespace aurora {

namespace {

constexpr int W4_THREADS = 256;

// c += a . b^T on the matrix pipe with the accumulator tile IN PLACE in AGPRs.  As a builtin, hipcc allocates destination
// and source accumulator separately at one wave per SIMD and shuffles tiles between AGPRs and VGPRs around every MFMA of
// the loop (168 v_accvgpr moves per two K-steps); the tied "+a" operand leaves nothing to allocate.  The statements are
// volatile: they issue in source order, which is the schedule.
__device__ __forceinline__ void mma_acc(f32x4& c, u32x4 a, u32x4 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// ABLATE (timing probes with WRONG results, never dispatched by the library's own rule): 1 no barrier in the loop, 2 no
// fragment reads, 3 no LDS-DMA refill, 4 none of the three (MFMAs only)
template <int NST, int ABLATE = 0>   // stages of the LDS ring: 4 (128 KiB, stages s+2 .. s+4 in flight) or 5 (160 KiB, s+2 .. s+5)
__global__ __launch_bounds__(W4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void linear_kernel_256w4(const LinearArgs p_in) {
  const LinearArgs p = batch_problem(p_in);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn2 = wave & 1;
  const uint32_t nb = (uint32_t)p.n_blocks;
  uint32_t tile_m, tile_n;
  tile_of_block(blockIdx.x, nb, nb / (uint32_t)p.tiles_n, (uint32_t)p.tiles_n, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * BM2;
  const int n0 = (int)tile_n * BN2;
  const int nt = p.k_tiles;

  const char* src_x[4];
  const char* src_w[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int id = r * W4_THREADS + tid;
    const int row = id >> 2, c = id & 3;
    int64_t gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    int gn = n0 + row;
    gn = gn < p.N ? gn : p.N - 1;
    src_x[r] = p.A + gm * p.lda_b + ((c ^ swz2_x(row)) << 4);
    src_w[r] = p.W + (int64_t)gn * p.ldw_b + ((c ^ swz2_w(row)) << 4);
  }
  // piece j (0..7) of stage kt into ring slot `slot`: four activation pieces, then four weight pieces.  Past the end the last
  // stage is loaded again (into a slot nobody reads): every iteration is the same code, the counted waits never change.
  auto stage_piece_at = [&](int kt, int slot, int j) {
    const int kc = kt < nt ? kt : nt - 1;
    const int64_t koff = (int64_t)kc * ROW2;
    char* base = smem + slot * STAGE2;
    const int r = j & 3;
    if (j < 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_x[r] + koff),
                                       (lds_ptr_t)(base + (r * W4_THREADS + wave * 64) * 16), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_w[r] + koff),
                                       (lds_ptr_t)(base + OPER2 + (r * W4_THREADS + wave * 64) * 16), 16, 0, 0);
  };

  // fragment addresses: the swizzle of a row does not depend on the fragment index, so the eight fragments of an operand
  // are one base plus immediates
  const int i16 = lane & 15, g = lane >> 4;
  int off_x0, off_w0;
  {
    const int row = wm * 128 + i16;
    off_x0 = row * ROW2 + ((g ^ swz2_x(row)) << 4);
    const int roww = wn2 * 128 + 16 * (i16 >> 2) + (i16 & 3);
    off_w0 = OPER2 + roww * ROW2 + ((g ^ swz2_w(roww)) << 4);
  }
  auto read_x = [&](const char* buf, int f) { return *reinterpret_cast<const u32x4*>(buf + off_x0 + f * 16 * ROW2); };
  // weight fragment j = 4 h + fn: half h (64 columns) of the wave's 128, n-fragment fn (rows 4 fn + .. interleaved)
  auto read_w = [&](const char* buf, int j) {
    return *reinterpret_cast<const u32x4*>(buf + off_w0 + (j >> 2) * 64 * ROW2 + (j & 3) * 4 * ROW2);
  };

  f32x4 acc[2][4][8];   // [half][fn][fm]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[h][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ax[8], aw[8], bx[8], bw[8];
  // One K-step: the 64 MFMAs of the fragments in (cx, cw), the 16 reads of the next stage (ring slot `rs`) into (nx, nw) and
  // the 8 LDS-DMA pieces of stage `kt` into slot `ws` -- at most one of them between two MFMAs, each in the shadow of the 16
  // cycles the matrix pipe needs for the MFMA before it (pinned by the sched_barriers: hipcc knows no latency of an asm
  // statement and would put all loads of a group in front of its MFMAs).
  auto step = [&](int rs, int kt, int ws, u32x4 (&cx)[8], u32x4 (&cw)[8], u32x4 (&nx)[8], u32x4 (&nw)[8]) {
    const char* nbuf = smem + rs * STAGE2;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      mma_acc(acc[0][0][q], cw[0], cx[q]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ABLATE != 2 && ABLATE != 4) nw[q] = read_w(nbuf, q);
      __builtin_amdgcn_sched_barrier(0);
      mma_acc(acc[0][1][q], cw[1], cx[q]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ABLATE != 2 && ABLATE != 4) nx[q] = read_x(nbuf, q);
      __builtin_amdgcn_sched_barrier(0);
      mma_acc(acc[0][2][q], cw[2], cx[q]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ABLATE != 3 && ABLATE != 4) stage_piece_at(kt, ws, q);
      __builtin_amdgcn_sched_barrier(0);
      mma_acc(acc[0][3][q], cw[3], cx[q]);
      mma_acc(acc[1][0][q], cw[4], cx[q]);
      mma_acc(acc[1][1][q], cw[5], cx[q]);
      mma_acc(acc[1][2][q], cw[6], cx[q]);
      mma_acc(acc[1][3][q], cw[7], cx[q]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // own pieces of the stage after next have landed (NST - 2 younger stages stay in flight), the fragment reads are done --
    // as ONE s_waitcnt the compiler can see (simm16: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14): behind an
    // opaque asm wait hipcc's counter model still holds the reads pending and puts an lgkmcnt(0) in front of the next MFMA
    if constexpr (ABLATE == 3 || ABLATE == 4) __builtin_amdgcn_s_waitcnt(0x0070);
    else if constexpr (NST == 4) __builtin_amdgcn_s_waitcnt(0x4070);   // vmcnt(16) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x4078);                      // vmcnt(24) lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ABLATE != 1 && ABLATE != 4) __builtin_amdgcn_s_barrier();   // RAW: that stage is complete for everyone.  WAR: nobody reads slot `rs` any more.
    asm volatile("" ::: "memory");
  };
#pragma unroll
  for (int kt = 0; kt < NST; ++kt)
#pragma unroll
    for (int j = 0; j < 8; ++j) stage_piece_at(kt, kt, j);
  if constexpr (NST == 4) __builtin_amdgcn_s_waitcnt(0x4F78);   // vmcnt(24): own pieces of stage 0
  else __builtin_amdgcn_s_waitcnt(0x8F70);                      // vmcnt(32)
  __builtin_amdgcn_s_barrier();                                 // stage 0 is complete
  asm volatile("" ::: "memory");
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    aw[f] = read_w(smem, f);
    ax[f] = read_x(smem, f);
  }
  if constexpr (NST == 4) __builtin_amdgcn_s_waitcnt(0x4070);   // own pieces of stage 1; the fragments of stage 0
  else __builtin_amdgcn_s_waitcnt(0x4078);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();   // stage 1 is complete; nobody reads slot 0 any more
  asm volatile("" ::: "memory");
  if constexpr (ABLATE == 2 || ABLATE == 4) {   // (no reads: both fragment sets hold stage 0 for good)
#pragma unroll
    for (int q = 0; q < 8; ++q) { bx[q] = ax[q]; bw[q] = aw[q]; }
  }
  // iteration s: registers hold stage s (slot r, free), reads take stage s+1 (slot r+1), the DMA puts stage s+NST into slot r
  int s = 0, r = 0;
  auto next = [](int v) { return v + 1 == NST ? 0 : v + 1; };
  for (; s + 1 < nt; s += 2) {
    const int r1 = next(r);
    step(r1, s + NST, r, ax, aw, bx, bw);
    r = next(r1);
    step(r, s + 1 + NST, r1, bx, bw, ax, aw);
  }
  if (s < nt) step(next(r), s + NST, r, ax, aw, bx, bw);
  __builtin_amdgcn_s_waitcnt(0x0070);   // the re-loads past the end have landed: the ring is dead
  __builtin_amdgcn_s_barrier();
  // (the MFMAs are opaque to hipcc's hazard recogniser: the last results must have left the matrix pipe before the
  // epilogue's v_accvgpr_read -- 16 passes = 64 cycles at most)
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  // the epilogues of the eight-wave kernels, once per 64-column half: half h of wave (wm, wn2) is their wave (wm, 2 wn2 + h)
  // (written out, not a loop over h: hipcc declines to unroll a loop around the two inlined epilogues, and a run-time index
  // into the accumulators would put all 256 of them on the stack)
  const bool whole_rows = p.C2 == nullptr && p.res == nullptr && p.vec_store;   // (uniform)
  if (whole_rows) {
    epilogue_256_bf16_coalesced<1>(p, acc[0], m0, n0, wm, 2 * wn2, wm * 4 + 2 * wn2, lane, smem);
    epilogue_256_bf16_coalesced<1>(p, acc[1], m0, n0, wm, 2 * wn2 + 1, wm * 4 + 2 * wn2 + 1, lane, smem);
  } else {
    epilogue_256<bf16_t>(p, acc[0], m0, n0, wm, 2 * wn2, i16, g);
    epilogue_256<bf16_t>(p, acc[1], m0, n0, wm, 2 * wn2 + 1, i16, g);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same tile with the operands staged THROUGH REGISTERS (buffer_load -> VGPR -> ds_write) in WHOLE 128-BYTE LINES.
// Why: (1) the ablation of the kernel above (profiles/r05_ab_gemm_w4.log) -- without its barrier +1 %, without its fragment
// reads +5 %, without its refill +25 % (1,713 against 1,370 TFLOP/s at 8192^3; MFMAs alone 1,812): what the tile pays for is
// getting its operands, not multiplying them.  (2) Every kernel of gemm.hip asks for an operand row in 64-byte pieces, one
// K-stage (32 bf16) at a time: a 128-byte cache line is requested from L2 twice, half a microsecond apart (the CU's 16 KiB
// L1 has seen 32 KiB in between).  The vendor's kernel stages K = 64 (MT256x256x64): whole lines, half the requests.  LDS-DMA
// cannot do that here without halving the ring (a lane's 16 bytes land at lane * 16: whole-line rows make 64 KiB stages),
// registers can: eight lanes load the eight pieces of a row's line and write them into the images of TWO K-stages.
//   even step s: 64 MFMAs of stage s; fragment reads of stage s+1; ds_write of stages s+2 and s+3 (loaded two steps ago);
//                buffer_load of stages s+4 and s+5 (16 pieces) into the other register set; lgkmcnt(0); barrier
//   odd step:    64 MFMAs of stage s+1; fragment reads of stage s+2.  No barrier: what it reads was complete at the last one,
//                and nothing is written.
// One barrier per 128 MFMAs; LDS: the four 32 KiB stage images as before (read s+1, write s+2 and s+3, s dead); registers:
// 2 x 64 staged + 96 fragment registers (the activation fragment of group q is re-read in place behind its MFMAs).
// Rows past M read as zero through the buffer descriptor's range check (their results are never stored).  K % 64 == 0.
template <int DUMMY>
__global__ __launch_bounds__(W4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void linear_kernel_256w4v(const LinearArgs p_in) {
  const LinearArgs p = batch_problem(p_in);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn2 = wave & 1;
  const uint32_t nb = (uint32_t)p.n_blocks;
  uint32_t tile_m, tile_n;
  tile_of_block(blockIdx.x, nb, nb / (uint32_t)p.tiles_n, (uint32_t)p.tiles_n, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * BM2;
  const int n0 = (int)tile_n * BN2;
  const int nt = p.k_tiles;          // 32-wide K-stages (even)
  const int nu = nt >> 1;            // 64-wide units of loading

  // piece r (0..7) of an operand: row r * 32 + (tid >> 3), 16-byte piece c8 = tid & 7 of the row's 128-byte line
  const int c8 = tid & 7, row0 = tid >> 3;
  const int64_t rows_x = p.M - m0 < BM2 ? p.M - m0 : BM2;
  const int64_t bytes_x = rows_x * p.lda_b, bytes_w = (int64_t)BN2 * p.ldw_b;
  // buffer descriptors by hand (base, stride 0, bytes, gfx9 raw-buffer flags), uniform: the loads below are inline asm
  auto descriptor = [](const char* base, int64_t bytes) {
    const uint64_t b = (uint64_t)base;
    return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b),
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu)),
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(bytes < 0x7fffffff ? bytes : 0x7fffffff)), 0x00020000u};
  };
  const u32x4 rs_x = descriptor(p.A + m0 * p.lda_b, bytes_x), rs_w = descriptor(p.W + (int64_t)n0 * p.ldw_b, bytes_w);
  const int vo_x0 = (int)(row0 * p.lda_b) + c8 * 16, vo_w0 = (int)(row0 * p.ldw_b) + c8 * 16;
  const int ld32_x = (int)(32 * p.lda_b), ld32_w = (int)(32 * p.ldw_b);   // (uniform)
  // piece j of unit u: j < 8 activations (r = j), else weights.  The row part goes into the per-lane offset (one add), so
  // that the descriptor's range check sees it; the K part is the scalar offset.
  // INLINE ASM on purpose: hipcc puts an s_waitcnt vmcnt(n) in front of EVERY ds_write that stores a loaded register (16 per
  // step, ~2 issue cycles per MFMA: profiles/r05_pmc_w4_forms.txt -- 19.6 cycles per MFMA where the vendor's wave needs
  // 16.4).  It cannot see these loads; the step waits ONCE, by hand, in front of its first write.
  auto load_piece = [&](int u, int j) -> u32x4 {
    const int uc = u < nu ? u : nu - 1;   // past the end: the last unit again (written to stage images nobody reads)
    const int koff = uc * 128;
    const int r = j & 7;
    u32x4 v;
    if (j < 8) {
      const int vo = vo_x0 + r * ld32_x;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(v) : "v"(vo), "s"(rs_x), "s"(koff) : "memory");
    } else {
      const int vo = vo_w0 + r * ld32_w;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(v) : "v"(vo), "s"(rs_w), "s"(koff) : "memory");
    }
    return v;
  };
  // where a piece goes: stage image (c8 >> 2) of the unit's two, row, position (c8 & 3) ^ swizzle -- the image the LDS-DMA
  // kernels write (gemm.hip).  The activation swizzle does not depend on r; the weight one does through r & 1.
  const int half_off = (c8 >> 2) * STAGE2;
  char* const wr_x = smem + half_off + row0 * ROW2 + (((c8 & 3) ^ swz2_x(row0)) << 4);
  char* const wr_w0 = smem + half_off + OPER2 + row0 * ROW2 + (((c8 & 3) ^ swz2_w(row0)) << 4);
  char* const wr_w1 = smem + half_off + OPER2 + (row0 + 32) * ROW2 + (((c8 & 3) ^ swz2_w(row0 + 32)) << 4);
  auto write_piece = [&](int unit_slot, int j, u32x4 v) {   // unit_slot: 0 / 1 = stage images {0, 1} / {2, 3}
    const int r = j & 7;
    char* dst = j < 8 ? wr_x + r * 32 * ROW2 : ((r & 1) ? wr_w1 + (r >> 1) * 64 * ROW2 : wr_w0 + (r >> 1) * 64 * ROW2);
    *reinterpret_cast<u32x4*>(dst + unit_slot * 2 * STAGE2) = v;
  };

  const int i16 = lane & 15, g = lane >> 4;
  int off_x0, off_w0;
  {
    const int row = wm * 128 + i16;
    off_x0 = row * ROW2 + ((g ^ swz2_x(row)) << 4);
    const int roww = wn2 * 128 + 16 * (i16 >> 2) + (i16 & 3);
    off_w0 = OPER2 + roww * ROW2 + ((g ^ swz2_w(roww)) << 4);
  }
  auto read_x = [&](const char* buf, int f) { return *reinterpret_cast<const u32x4*>(buf + off_x0 + f * 16 * ROW2); };
  auto read_w = [&](const char* buf, int j) {
    return *reinterpret_cast<const u32x4*>(buf + off_w0 + (j >> 2) * 64 * ROW2 + (j & 3) * 4 * ROW2);
  };

  f32x4 acc[2][4][8];   // [half][fn][fm]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[h][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 xf[8], aw[8], bw[8];   // fragments: one activation set (re-read in place), two weight sets
  u32x4 ga[16], gb[16];        // two units on their way from memory to LDS
  // even step s (unit s / 2): writes `gw` = unit s/2 + 1 into the stage images of s+2, s+3; loads unit s/2 + 2 into `gl`
  auto even_step = [&](int s, u32x4 (&cw)[8], u32x4 (&nw)[8], u32x4 (&gl)[16], u32x4 (&gw)[16]) {
    const char* nbuf = smem + ((s + 1) & 3) * STAGE2;
    const int wslot = ((s + 2) >> 1) & 1, u = (s >> 1) + 2;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      mma_acc(acc[0][0][q], cw[0], xf[q]);
      __builtin_amdgcn_sched_barrier(0);
      nw[q] = read_w(nbuf, q);
      __builtin_amdgcn_sched_barrier(0);
      mma_acc(acc[0][1][q], cw[1], xf[q]);
      __builtin_amdgcn_sched_barrier(0);
      if (q == 0) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the unit loaded two steps ago (nothing younger is in flight)
      write_piece(wslot, 2 * q, gw[2 * q]);
      __builtin_amdgcn_sched_barrier(0);
      mma_acc(acc[0][2][q], cw[2], xf[q]);
      __builtin_amdgcn_sched_barrier(0);
      write_piece(wslot, 2 * q + 1, gw[2 * q + 1]);
      __builtin_amdgcn_sched_barrier(0);
      mma_acc(acc[0][3][q], cw[3], xf[q]);
      __builtin_amdgcn_sched_barrier(0);
      gl[2 * q] = load_piece(u, 2 * q);
      __builtin_amdgcn_sched_barrier(0);
      mma_acc(acc[1][0][q], cw[4], xf[q]);
      __builtin_amdgcn_sched_barrier(0);
      gl[2 * q + 1] = load_piece(u, 2 * q + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_acc(acc[1][1][q], cw[5], xf[q]);
      mma_acc(acc[1][2][q], cw[6], xf[q]);
      mma_acc(acc[1][3][q], cw[7], xf[q]);
      __builtin_amdgcn_sched_barrier(0);
      xf[q] = read_x(nbuf, q);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the fragment reads and this wave's ds_writes are done
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();   // RAW: stages s+2, s+3 are complete.  WAR: the images of s, s+1 may be written from s+2 on.
    asm volatile("" ::: "memory");
  };
  auto odd_step = [&](int s, u32x4 (&cw)[8], u32x4 (&nw)[8]) {
    const char* nbuf = smem + ((s + 1) & 3) * STAGE2;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      mma_acc(acc[0][0][q], cw[0], xf[q]);
      __builtin_amdgcn_sched_barrier(0);
      nw[q] = read_w(nbuf, q);
      __builtin_amdgcn_sched_barrier(0);
      mma_acc(acc[0][1][q], cw[1], xf[q]);
      mma_acc(acc[0][2][q], cw[2], xf[q]);
      mma_acc(acc[0][3][q], cw[3], xf[q]);
      mma_acc(acc[1][0][q], cw[4], xf[q]);
      mma_acc(acc[1][1][q], cw[5], xf[q]);
      mma_acc(acc[1][2][q], cw[6], xf[q]);
      mma_acc(acc[1][3][q], cw[7], xf[q]);
      __builtin_amdgcn_sched_barrier(0);
      xf[q] = read_x(nbuf, q);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // prologue: units 0 and 1 into the register sets, unit 0 into stage images 0, 1
#pragma unroll
  for (int j = 0; j < 16; ++j) gb[j] = load_piece(0, j);
#pragma unroll
  for (int j = 0; j < 16; ++j) ga[j] = load_piece(1, j);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // (both units: the counter is in order)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 16; ++j) write_piece(0, j, gb[j]);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();   // stages 0 and 1 are complete in LDS
  asm volatile("" ::: "memory");
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    aw[f] = read_w(smem, f);
    xf[f] = read_x(smem, f);
  }
  int s = 0;
  for (; s + 4 <= nt; s += 4) {
    even_step(s, aw, bw, gb, ga);       // writes unit s/2+1 (in ga), loads unit s/2+2 into gb
    odd_step(s + 1, bw, aw);
    even_step(s + 2, aw, bw, ga, gb);
    odd_step(s + 3, bw, aw);
  }
  if (s < nt) {
    even_step(s, aw, bw, gb, ga);
    odd_step(s + 1, bw, aw);
  }
  __builtin_amdgcn_s_waitcnt(0x0070);   // loads past the end have landed (their registers are dead), the last reads are done
  __builtin_amdgcn_s_barrier();
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  const bool whole_rows = p.C2 == nullptr && p.res == nullptr && p.vec_store;   // (uniform)
  if (whole_rows) {
    epilogue_256_bf16_coalesced<1>(p, acc[0], m0, n0, wm, 2 * wn2, wm * 4 + 2 * wn2, lane, smem);
    epilogue_256_bf16_coalesced<1>(p, acc[1], m0, n0, wm, 2 * wn2 + 1, wm * 4 + 2 * wn2 + 1, lane, smem);
  } else {
    epilogue_256<bf16_t>(p, acc[0], m0, n0, wm, 2 * wn2, i16, g);
    epilogue_256<bf16_t>(p, acc[1], m0, n0, wm, 2 * wn2 + 1, i16, g);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The register-staged four-wave tile on v_mfma_f32_32x32x16_bf16.
// Why: the counters of the forms above (profiles/r05_pmc_w4_forms.txt) -- a lone wave per SIMD spends 19.6 cycles per 16-cycle
// MFMA where the vendor's spends 16.4: every LDS / VMEM instruction between two MFMAs costs issue cycles that a 16-cycle
// matrix instruction does not cover.  A 32x32x16 MFMA occupies the pipe for 32 cycles: the same 48 memory instructions per
// K-step fall between 32 MFMAs instead of 64, 1.5 per gap where the guide counts five as free.
// Fragments (gfx950): A / B operand lane l = row / column l & 31, k = 8 (l >> 5) .. + 7; D register 4 a + b of lane l = row
// 8 a + 4 (l >> 5) + b, column l & 31.  The MFMA's A operand is the WEIGHT tile again (D = C^T).  Weight row of operand row i
// in 32-column block t of a 64-column half: 16 (2 t + ((i >> 2) & 1)) + 4 (i >> 3) + (i & 3) -- then register a, component b of
// lane (c, h2) holds feature 16 (2 t + h2) + 4 a + b of token c: what lane (i16 = c & 15, g = 2 t + h2) of the eight-wave
// kernels holds in acc[fn = a][fm = 2 tm + (c >> 4)].  Two swaps per register pair (v_permlane32_swap, then
// v_permlane16_swap, on the blocks t = 0 / 1) put it there, and the eight-wave epilogues run unchanged.
// LDS image of a stage: rows of 64 bytes as before, piece c of row r at position c ^ f((r >> 2) & 7), f(x) = (x & 3) ^ 3 (x >> 2):
// the 16 lanes of a ds_read_b128 cycle (rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31} of a fragment, one piece) then hit 16
// distinct 16-byte slots -- for the activation rows and for the interleaved weight rows alike.
// NOT bit-identical to the 16x16x32 kernels (K is summed in 16-wide steps and in the 32x32 instruction's internal order).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2_sw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mma32_acc(f32x16& c, u32x4 a, u32x4 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ int swz8(int row) {
  const int x = (row >> 2) & 7;
  return (x & 3) ^ ((x >> 2) * 3);
}

template <int DUMMY>
__global__ __launch_bounds__(W4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void linear_kernel_256w4m(const LinearArgs p_in) {
  const LinearArgs p = batch_problem(p_in);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn2 = wave & 1;
  const uint32_t nb = (uint32_t)p.n_blocks;
  uint32_t tile_m, tile_n;
  tile_of_block(blockIdx.x, nb, nb / (uint32_t)p.tiles_n, (uint32_t)p.tiles_n, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * BM2;
  const int n0 = (int)tile_n * BN2;
  const int nt = p.k_tiles;          // 32-wide K-stages (even)
  const int nu = nt >> 1;            // 64-wide units of loading

  // ---- staging: as linear_kernel_256w4v (whole 128-byte lines through registers), with the image's swizzle swz8 ----
  const int c8 = tid & 7, row0 = tid >> 3;
  const int64_t rows_x = p.M - m0 < BM2 ? p.M - m0 : BM2;
  const int64_t bytes_x = rows_x * p.lda_b, bytes_w = (int64_t)BN2 * p.ldw_b;
  auto descriptor = [](const char* base, int64_t bytes) {
    const uint64_t b = (uint64_t)base;
    return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b),
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu)),
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(bytes < 0x7fffffff ? bytes : 0x7fffffff)), 0x00020000u};
  };
  const u32x4 rs_x = descriptor(p.A + m0 * p.lda_b, bytes_x), rs_w = descriptor(p.W + (int64_t)n0 * p.ldw_b, bytes_w);
  const int vo_x0 = (int)(row0 * p.lda_b) + c8 * 16, vo_w0 = (int)(row0 * p.ldw_b) + c8 * 16;
  const int ld32_x = (int)(32 * p.lda_b), ld32_w = (int)(32 * p.ldw_b);   // (uniform)
  auto load_piece = [&](int u, int j) -> u32x4 {
    const int uc = u < nu ? u : nu - 1;
    const int koff = uc * 128;
    const int r = j & 7;
    u32x4 v;
    if (j < 8) {
      const int vo = vo_x0 + r * ld32_x;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(v) : "v"(vo), "s"(rs_x), "s"(koff) : "memory");
    } else {
      const int vo = vo_w0 + r * ld32_w;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(v) : "v"(vo), "s"(rs_w), "s"(koff) : "memory");
    }
    return v;
  };
  // rows r * 32 + row0: (row >> 2) & 7 = (row0 >> 2) & 7 whatever r, for both operands -- one write address each
  char* const wr_x = smem + (c8 >> 2) * STAGE2 + row0 * ROW2 + (((c8 & 3) ^ swz8(row0)) << 4);
  char* const wr_w = wr_x + OPER2;
  auto write_piece = [&](int unit_slot, int j, u32x4 v) {   // unit_slot: 0 / 1 = stage images {0, 1} / {2, 3}
    char* dst = (j < 8 ? wr_x : wr_w) + (j & 7) * 32 * ROW2;
    *reinterpret_cast<u32x4*>(dst + unit_slot * 2 * STAGE2) = v;
  };

  // ---- fragments ----
  const int c32 = lane & 31, h2 = lane >> 5;
  int off_x0, off_w0[2];
  {
    const int row = wm * 128 + c32;                       // + 32 tm
    off_x0 = row * ROW2;
    // weight operand row i = c32 of block t (t = 0, 1 within a 64-column half; + 64 per half)
    const int wrow = wn2 * 128 + 16 * ((c32 >> 2) & 1) + 4 * (c32 >> 3) + (c32 & 3);   // + 32 t + 64 h
    off_w0[0] = OPER2 + wrow * ROW2;
    off_w0[1] = wrow;   // (row number, for the swizzle)
  }
  const int sx = swz8(wm * 128 + c32);                   // (+ 32 tm does not change (row >> 2) & 7)
  const int sw = swz8(off_w0[1]);                        // (+ 32 t + 64 h neither)
  // piece 2 kh + h2 of the row, kh = 0 / 1: position (2 kh + h2) ^ swizzle
  const int px0 = ((h2 ^ sx) << 4), px1 = (((2 + h2) ^ sx) << 4);
  const int pw0 = ((h2 ^ sw) << 4), pw1 = (((2 + h2) ^ sw) << 4);
  auto read_x = [&](const char* buf, int tm, int kh) {
    return *reinterpret_cast<const u32x4*>(buf + off_x0 + tm * 32 * ROW2 + (kh ? px1 : px0));
  };
  auto read_w = [&](const char* buf, int tn, int kh) {   // tn = 0 .. 3: block t = tn & 1 of half tn >> 1
    return *reinterpret_cast<const u32x4*>(buf + off_w0[0] + tn * 32 * ROW2 + (kh ? pw1 : pw0));
  };

  f32x16 acc[4][4];   // [tm][tn]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  u32x4 xf[4][2], aw[4][2], bw[4][2];   // [tm / tn][kh]: one activation set (re-read in place), two weight sets
  u32x4 ga[16], gb[16];
  // even step: 4 groups (tm) of 8 MFMAs; per group 2 weight reads, 4 writes, 4 loads, 2 in-place activation reads
  auto even_step = [&](int s, u32x4 (&cw)[4][2], u32x4 (&nw)[4][2], u32x4 (&gl)[16], u32x4 (&gw)[16]) {
    const char* nbuf = smem + ((s + 1) & 3) * STAGE2;
    const int wslot = ((s + 2) >> 1) & 1, u = (s >> 1) + 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      mma32_acc(acc[q][0], cw[0][0], xf[q][0]);
      __builtin_amdgcn_sched_barrier(0);
      nw[q][0] = read_w(nbuf, q, 0);
      if (q == 0) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the unit loaded two steps ago
      write_piece(wslot, 4 * q, gw[4 * q]);
      __builtin_amdgcn_sched_barrier(0);
      mma32_acc(acc[q][1], cw[1][0], xf[q][0]);
      __builtin_amdgcn_sched_barrier(0);
      nw[q][1] = read_w(nbuf, q, 1);
      write_piece(wslot, 4 * q + 1, gw[4 * q + 1]);
      __builtin_amdgcn_sched_barrier(0);
      mma32_acc(acc[q][2], cw[2][0], xf[q][0]);
      __builtin_amdgcn_sched_barrier(0);
      write_piece(wslot, 4 * q + 2, gw[4 * q + 2]);
      write_piece(wslot, 4 * q + 3, gw[4 * q + 3]);
      __builtin_amdgcn_sched_barrier(0);
      mma32_acc(acc[q][3], cw[3][0], xf[q][0]);
      __builtin_amdgcn_sched_barrier(0);
      gl[4 * q] = load_piece(u, 4 * q);
      xf[q][0] = read_x(nbuf, q, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma32_acc(acc[q][0], cw[0][1], xf[q][1]);
      __builtin_amdgcn_sched_barrier(0);
      gl[4 * q + 1] = load_piece(u, 4 * q + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma32_acc(acc[q][1], cw[1][1], xf[q][1]);
      __builtin_amdgcn_sched_barrier(0);
      gl[4 * q + 2] = load_piece(u, 4 * q + 2);
      __builtin_amdgcn_sched_barrier(0);
      mma32_acc(acc[q][2], cw[2][1], xf[q][1]);
      __builtin_amdgcn_sched_barrier(0);
      gl[4 * q + 3] = load_piece(u, 4 * q + 3);
      __builtin_amdgcn_sched_barrier(0);
      mma32_acc(acc[q][3], cw[3][1], xf[q][1]);
      __builtin_amdgcn_sched_barrier(0);
      xf[q][1] = read_x(nbuf, q, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the fragment reads and this wave's ds_writes are done
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto odd_step = [&](int s, u32x4 (&cw)[4][2], u32x4 (&nw)[4][2]) {
    const char* nbuf = smem + ((s + 1) & 3) * STAGE2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      mma32_acc(acc[q][0], cw[0][0], xf[q][0]);
      __builtin_amdgcn_sched_barrier(0);
      nw[q][0] = read_w(nbuf, q, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma32_acc(acc[q][1], cw[1][0], xf[q][0]);
      __builtin_amdgcn_sched_barrier(0);
      nw[q][1] = read_w(nbuf, q, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma32_acc(acc[q][2], cw[2][0], xf[q][0]);
      mma32_acc(acc[q][3], cw[3][0], xf[q][0]);
      __builtin_amdgcn_sched_barrier(0);
      xf[q][0] = read_x(nbuf, q, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma32_acc(acc[q][0], cw[0][1], xf[q][1]);
      mma32_acc(acc[q][1], cw[1][1], xf[q][1]);
      mma32_acc(acc[q][2], cw[2][1], xf[q][1]);
      mma32_acc(acc[q][3], cw[3][1], xf[q][1]);
      __builtin_amdgcn_sched_barrier(0);
      xf[q][1] = read_x(nbuf, q, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // prologue: units 0 and 1 into the register sets, unit 0 into stage images 0, 1
#pragma unroll
  for (int j = 0; j < 16; ++j) gb[j] = load_piece(0, j);
#pragma unroll
  for (int j = 0; j < 16; ++j) ga[j] = load_piece(1, j);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 16; ++j) write_piece(0, j, gb[j]);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();   // stages 0 and 1 are complete in LDS
  asm volatile("" ::: "memory");
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      aw[f][kh] = read_w(smem, f, kh);
      xf[f][kh] = read_x(smem, f, kh);
    }
  int s = 0;
  for (; s + 4 <= nt; s += 4) {
    even_step(s, aw, bw, gb, ga);
    odd_step(s + 1, bw, aw);
    even_step(s + 2, aw, bw, ga, gb);
    odd_step(s + 3, bw, aw);
  }
  if (s < nt) {
    even_step(s, aw, bw, gb, ga);
    odd_step(s + 1, bw, aw);
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_s_barrier();
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  // ---- to the eight-wave kernels' accumulator layout, one 64-column half at a time, and their epilogues ----
  const int i16 = lane & 15, g = lane >> 4;
  const bool whole_rows = p.C2 == nullptr && p.res == nullptr && p.vec_store;   // (uniform)
  auto half = [&](f32x16 (&t0)[4], f32x16 (&t1)[4], int hh) {   // t0[tm], t1[tm]: blocks t = 0 / 1 of half hh
    f32x4 old[4][8];   // [fn][fm]
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          // X = [X0 X1 X2 X3] (16-lane rows), Y likewise:  32-swap -> [X0 X1 Y0 Y1], [X2 X3 Y2 Y3];  16-swap -> [X0 X2 Y0 Y2], [X1 X3 Y1 Y3]
          const u32x2_sw r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t0[tm][4 * a + b]), __float_as_uint(t1[tm][4 * a + b]), false, false);
          const u32x2_sw q = __builtin_amdgcn_permlane16_swap(r.x, r.y, false, false);
          old[a][2 * tm][b] = __uint_as_float(q.x);
          old[a][2 * tm + 1][b] = __uint_as_float(q.y);
        }
    const int wn = 2 * wn2 + hh;
    if (whole_rows) epilogue_256_bf16_coalesced<1>(p, old, m0, n0, wm, wn, wm * 4 + wn, lane, smem);
    else epilogue_256<bf16_t>(p, old, m0, n0, wm, wn, i16, g);
  };
  {
    f32x16 t0[4] = {acc[0][0], acc[1][0], acc[2][0], acc[3][0]}, t1[4] = {acc[0][1], acc[1][1], acc[2][1], acc[3][1]};
    half(t0, t1, 0);
  }
  {
    f32x16 t0[4] = {acc[0][2], acc[1][2], acc[2][2], acc[3][2]}, t1[4] = {acc[0][3], acc[1][3], acc[2][3], acc[3][3]};
    half(t0, t1, 1);
  }
}

}  // namespace

}  // namespace aurora

// The launcher gemm.hip's dispatch calls (C linkage: LinearArgs lives in each translation unit's anonymous namespace;
// both see the same definition, gemm.hip's).
extern "C" __attribute__((visibility("hidden"))) int aurora_w4_launch(const void* linear_args, unsigned n_blocks, unsigned batch,
                                                                     int stages, void* stream) {
  using namespace aurora;
  static bool attr_done_dev[64] = {false};
  bool& attr_done = attr_done_dev[current_device() & 63];
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)linear_kernel_256w4<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256w4<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256w4v<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256w4m<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256w4<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256w4<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256w4<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE2);
    (void)hipFuncSetAttribute((const void*)linear_kernel_256w4<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE2);
    attr_done = true;
  }
  const LinearArgs& p = *static_cast<const LinearArgs*>(linear_args);
  const dim3 gr(n_blocks, batch), bl(W4_THREADS);
  switch (stages) {
    case 5: hipLaunchKernelGGL(linear_kernel_256w4<5>, gr, bl, 5 * STAGE2, as_stream(stream), p); break;
    case 6:   // whole-line register staging: K in units of 64
      if (p.k_tiles % 2 == 0) hipLaunchKernelGGL(linear_kernel_256w4v<0>, gr, bl, 4 * STAGE2, as_stream(stream), p);
      else hipLaunchKernelGGL(linear_kernel_256w4<4>, gr, bl, 4 * STAGE2, as_stream(stream), p);
      break;
    case 7:   // 32x32x16 MFMAs
      if (p.k_tiles % 2 == 0) hipLaunchKernelGGL(linear_kernel_256w4m<0>, gr, bl, 4 * STAGE2, as_stream(stream), p);
      else hipLaunchKernelGGL(linear_kernel_256w4<4>, gr, bl, 4 * STAGE2, as_stream(stream), p);
      break;
    case 14: hipLaunchKernelGGL((linear_kernel_256w4<4, 1>), gr, bl, 4 * STAGE2, as_stream(stream), p); break;
    case 24: hipLaunchKernelGGL((linear_kernel_256w4<4, 2>), gr, bl, 4 * STAGE2, as_stream(stream), p); break;
    case 34: hipLaunchKernelGGL((linear_kernel_256w4<4, 3>), gr, bl, 4 * STAGE2, as_stream(stream), p); break;
    case 44: hipLaunchKernelGGL((linear_kernel_256w4<4, 4>), gr, bl, 4 * STAGE2, as_stream(stream), p); break;
    default: hipLaunchKernelGGL(linear_kernel_256w4<4>, gr, bl, 4 * STAGE2, as_stream(stream), p);
  }
  return 0;
}
