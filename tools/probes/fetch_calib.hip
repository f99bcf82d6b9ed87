// FETCH_SIZE calibration for the ring GEMM's staging pattern (gfx950, run under rocprofv3 --pmc FETCH_SIZE).
//
// The guide (MI355X_MICROARCH.md, HBM section) says FETCH_SIZE reports HALF the bytes of a wide coalesced stream
// (128-B requests tallied at 64 B) and that other access shapes are uncalibrated.  The GEMM stages its operands with
// `global_load_lds_dwordx4` instructions that cover 16 rows x 64 B (four lanes per row piece, rows one matrix row
// apart).  This probe streams a 2 GiB row-major matrix (8x the 256 MiB Infinity Cache) EXACTLY ONCE with
//   mode 0: that pattern (16 rows x 64 B per wave instruction, walking along the rows in 64-byte steps)
//   mode 1: 8 rows x 128 B per wave instruction
//   mode 2: 1 KiB contiguous per wave instruction (the "wide coalesced stream" of the guide)
// so that FETCH_SIZE x 1024 / 2^31 is the factor to apply to each pattern.  Kernel names carry the mode.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/fetch_calib tools/probes/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE>
__global__ __launch_bounds__(256) void fetch_calib(const char* src, long row_bytes, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 4096];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // rows per workgroup and bytes along the row per step, by mode
  constexpr int ROWS = MODE == 0 ? 64 : MODE == 1 ? 32 : 4;
  constexpr int STEP = MODE == 0 ? 64 : MODE == 1 ? 128 : 1024;
  const int row = MODE == 0 ? tid >> 2 : MODE == 1 ? tid >> 3 : tid >> 6;
  const int piece = MODE == 0 ? tid & 3 : MODE == 1 ? tid & 7 : tid & 63;
  const char* p = src + ((long)blockIdx.x * ROWS + row) * row_bytes + piece * 16;
  const int steps = (int)(row_bytes / STEP);
  for (int k = 0; k < steps; ++k) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (long)k * STEP),
                                     (lds_ptr_t)(smem + (k & 3) * 4096 + wave * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (*reinterpret_cast<unsigned*>(smem + tid * 4) == 0x12345678u) sink[0] = 1;
}

template <int MODE>
void run(const char* src, long rows, long row_bytes, unsigned* sink) {
  constexpr int ROWS = MODE == 0 ? 64 : MODE == 1 ? 32 : 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  fetch_calib<MODE><<<(unsigned)(rows / ROWS), 256>>>(src, row_bytes, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mode %d: %ld bytes read once in %.3f ms = %.2f TB/s\n", MODE, rows * row_bytes, ms, rows * row_bytes / ms / 1e9);
}

int main() {
  const long rows = 1L << 20, row_bytes = 2048;   // 2 GiB
  char* src; unsigned* sink;
  hipMalloc(&src, rows * row_bytes); hipMalloc(&sink, 4);
  hipMemset(src, 1, rows * row_bytes);
  hipDeviceSynchronize();
  run<0>(src, rows, row_bytes, sink);
  run<1>(src, rows, row_bytes, sink);
  run<2>(src, rows, row_bytes, sink);
  return 0;
}
