// Per-CU vector-memory throughput probe (L2-resident working set), gfx950.
//   mode 0: global_load_lds_dwordx4 (LDS-DMA)         mode 1: global_load_dwordx4 -> VGPR
//   mode 2: global_load_dwordx4 -> VGPR -> ds_write_b128   mode 3: global_store_dwordx4 (coalesced rows)
// Build: hipcc --offload-arch=gfx950 -O3 -o l1_probe tools/probes/l1_probe.hip ; run: ./l1_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE, int NTHR>
__global__ __launch_bounds__(NTHR) void probe(const char* src, char* dst, int iters, int wg_bytes, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (size_t)blockIdx.x * wg_bytes;
  char* obase = dst + (size_t)blockIdx.x * wg_bytes;
  const int chunk = NTHR * 16 * 4;           // bytes per iteration per WG (4 pieces per thread)
  u32x4 accv = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const int off = (it * chunk) % wg_bytes;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const char* p = base + off + (r * NTHR + tid) * 16;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (lds_ptr_t)(smem + ((it & 3) * 4 + r) * NTHR * 16 + wave * 1024), 16, 0, 0);
      } else if (MODE == 1 || MODE == 2) {
        u32x4 v = *reinterpret_cast<const u32x4*>(p);
        if (MODE == 2) *reinterpret_cast<u32x4*>(smem + ((it & 3) * 4 + r) * NTHR * 16 + tid * 16) = v;
        else accv ^= v;
      } else {
        *reinterpret_cast<u32x4*>(obase + off + (r * NTHR + tid) * 16) = u32x4{(uint32_t)it, 1, 2, 3};
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE == 0 || MODE == 2) accv.x ^= *reinterpret_cast<uint32_t*>(smem + tid * 4);
  if (accv.x == 0x12345678u) sink[0] = accv.y ^ accv.z ^ accv.w;
}

template <int MODE, int NTHR>
void run(const char* name, const char* src, char* dst, uint32_t* sink, int wgs, int wg_bytes, size_t lds) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)probe<MODE, NTHR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, NTHR><<<wgs, NTHR, lds>>>(src, dst, 50, wg_bytes, sink);
  hipEventRecord(e0);
  probe<MODE, NTHR><<<wgs, NTHR, lds>>>(src, dst, iters, wg_bytes, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * iters * NTHR * 64;
  printf("%-46s wgs=%4d thr=%d  %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU (at 2.4 GHz, 256 CUs)\n", name, wgs, NTHR, ms,
         bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

// GEMM-operand pattern: every wave instruction fetches ROWS rows x (1024 / ROWS) bytes, row stride `ld` bytes, from
// a per-WG panel of 256 rows that is walked along K; 4 instructions per wave and step = a 256-row x (1024/ROWS)-byte
// stage.  Panels are shared by `share` consecutive WGs (L2 reuse like the n-tiles of one m-tile).
template <int ROWS, int PAIR = 0>
__global__ __launch_bounds__(512) void panel_dma(const char* src, int iters, int64_t ld, int kbytes, int share, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  constexpr int RB = 1024 / ROWS;                 // bytes per row per instruction
  const char* panel = src + (size_t)(blockIdx.x / share) * 256 * ld;
  for (int it = 0; it < iters; ++it) {
    const int koff = PAIR ? (((it >> 1) * 2 * RB + (it & 1) * RB) % kbytes) : ((it * RB) % kbytes);
    // PAIR == 2: both halves of the lines inside ONE step (the 4 instructions alternate halves)
#pragma unroll
    for (int r = 0; r < 256 / ROWS / 8; ++r) {     // instructions per wave and step
      const int row = (r * 8 + wave) * ROWS + lane / (RB / 16);
      const char* p = panel + (size_t)row * ld + koff + (lane % (RB / 16)) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (lds_ptr_t)(smem + ((it & 3) * 16 + r) * 8192 + wave * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (*reinterpret_cast<uint32_t*>(smem + tid * 4) == 0x12345678u) sink[0] = 1;
}

template <int ROWS, int PAIR = 0>
void run_panel(const char* src, uint32_t* sink, int64_t ld, int kbytes, int share) {
  const int iters = 4000, wgs = 256;
  hipFuncSetAttribute((const void*)panel_dma<ROWS, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  panel_dma<ROWS, PAIR><<<wgs, 512, 128 * 1024>>>(src, 50, ld, kbytes, share, sink);
  hipEventRecord(e0);
  panel_dma<ROWS, PAIR><<<wgs, 512, 128 * 1024>>>(src, iters, ld, kbytes, share, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * iters * 256 * (1024 / ROWS);
  printf("panel DMA%s: %2d rows x %3d B per instr, ld=%5lld, K-bytes=%5d, %d WGs/panel: %7.3f ms %6.2f TB/s %6.1f B/clk/CU\n", PAIR ? " (halves back-to-back)" : "", ROWS,
         1024 / ROWS, (long long)ld, kbytes, share, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

// GEMM-epilogue pattern: each WG writes 256-row x 512-byte tiles (row stride `ld` bytes) to fresh memory.
template <int PAT>
__global__ __launch_bounds__(512) void tile_store(char* dst, int tiles_per_wg, int tiles_n, int64_t ld) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 2, wn = wave & 3, rr = lane >> 3, cc = lane & 7;
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int64_t tile = (int64_t)blockIdx.x * tiles_per_wg + t;
    const int64_t tm = tile / tiles_n, tn = tile % tiles_n;
    if (PAT == 0) {          // 8 rows x 128 B per instruction, adjacent lanes on adjacent pieces
      char* base = dst + (tm * 256 + wm * 128) * ld + tn * 512 + wn * 128 + cc * 16;
#pragma unroll
      for (int it = 0; it < 16; ++it)
        *reinterpret_cast<u32x4*>(base + (it * 8 + rr) * ld) = u32x4{(uint32_t)t, 1, 2, 3};
    } else if (PAT == 1) {   // 16 rows x 64 B per instruction, lane = row + 16 * piece (MFMA C^T ownership)
      char* base = dst + (tm * 256 + (lane & 15)) * ld + tn * 512 + wave * 64 + (lane >> 4) * 16;
#pragma unroll
      for (int it = 0; it < 16; ++it)
        *reinterpret_cast<u32x4*>(base + (it * 16) * ld) = u32x4{(uint32_t)t, 1, 2, 3};
    } else {                 // 2 rows x 512 B per instruction
      char* base = dst + (tm * 256 + wave * 32 + (lane >> 5)) * ld + tn * 512 + (lane & 31) * 16;
#pragma unroll
      for (int it = 0; it < 16; ++it)
        *reinterpret_cast<u32x4*>(base + (it * 2) * ld) = u32x4{(uint32_t)t, 1, 2, 3};
    }
  }
}

template <int PAT>
void run_tiles(char* big, int wgs, int tiles_per_wg) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int tiles_n = 8; const int64_t ld = 4096;
  tile_store<PAT><<<wgs, 512>>>(big, 2, tiles_n, ld);
  hipEventRecord(e0);
  tile_store<PAT><<<wgs, 512>>>(big, tiles_per_wg, tiles_n, ld);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * tiles_per_wg * 131072;
  printf("tile stores, pattern %d: %4d WGs x %3d tiles  %8.3f ms  %6.2f TB/s  %6.1f B/clk per ACTIVE CU  (%.2f us per tile)\n",
         PAT, wgs, tiles_per_wg, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / wgs / 2.4e9, ms * 1e3 / tiles_per_wg);
}

int main() {
  {
    char* big; hipMalloc(&big, (size_t)8 << 30);
    run_tiles<0>(big, 256, 64); run_tiles<1>(big, 256, 64); run_tiles<2>(big, 256, 64);
    run_tiles<0>(big, 64, 64);  run_tiles<1>(big, 64, 64);  run_tiles<2>(big, 64, 64);
    run_tiles<0>(big, 1, 64);   run_tiles<1>(big, 1, 64);   run_tiles<2>(big, 1, 64);
    for (int share : {1, 8}) {
      run_panel<16>(big, (uint32_t*)big + (1 << 28), 1024, 1024, share);
      run_panel<16, 1>(big, (uint32_t*)big + (1 << 28), 1024, 1024, share);
      run_panel<16, 1>(big, (uint32_t*)big + (1 << 28), 8192, 8192, share);
      run_panel<8>(big, (uint32_t*)big + (1 << 28), 1024, 1024, share);
      run_panel<16>(big, (uint32_t*)big + (1 << 28), 8192, 8192, share);
      run_panel<8>(big, (uint32_t*)big + (1 << 28), 8192, 8192, share);
      run_panel<4>(big, (uint32_t*)big + (1 << 28), 8192, 8192, share);
    }
    hipFree(big);
  }
  const int wg_bytes = 64 * 1024;
  const int max_wgs = 1024;
  char *src, *dst; uint32_t* sink;
  hipMalloc(&src, (size_t)max_wgs * wg_bytes); hipMalloc(&dst, (size_t)max_wgs * wg_bytes); hipMalloc(&sink, 64);
  hipMemset(src, 1, (size_t)max_wgs * wg_bytes);
  run<0, 512>("LDS-DMA dwordx4, 1 WG/CU (512 thr)", src, dst, sink, 256, wg_bytes, 128 * 1024);
  run<0, 256>("LDS-DMA dwordx4, 2 WG/CU (256 thr)", src, dst, sink, 512, wg_bytes, 64 * 1024);
  run<0, 256>("LDS-DMA dwordx4, 4 WG/CU (256 thr)", src, dst, sink, 1024, wg_bytes, 32 * 1024);
  run<1, 512>("global_load_dwordx4 -> VGPR, 1 WG/CU", src, dst, sink, 256, wg_bytes, 1024);
  run<1, 256>("global_load_dwordx4 -> VGPR, 4 WG/CU (256)", src, dst, sink, 1024, wg_bytes, 1024);
  run<2, 512>("global_load_dwordx4 -> ds_write_b128, 1 WG/CU", src, dst, sink, 256, wg_bytes, 128 * 1024);
  run<3, 512>("global_store_dwordx4 coalesced, 1 WG/CU", src, dst, sink, 256, wg_bytes, 1024);
  run<3, 256>("global_store_dwordx4 coalesced, 4 WG/CU (256)", src, dst, sink, 1024, wg_bytes, 1024);
  return 0;
}
