// Correctness + time of the bf16 linears of libaurora_hip.so, without Python (a fresh GPU box spends 1-2 minutes importing
// torch; this starts in a second).  Every shape is multiplied through the C ABI and compared, on sampled rows (every
// `stride`-th row plus the last 300 -- all 16 fragment rows and every 256-row tile are hit), with an fp32 dot product of the
// same bf16 operands computed by a plain kernel here; then timed with HIP events.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/gemm_check.hip -o tools/probes/gemm_check -Laurora_amd/_lib -laurora_hip \
//         -Wl,-rpath,'$ORIGIN/../../aurora_amd/_lib'
//   tools/probes/gemm_check [set ...]        sets: step band8 band4 band2 small, or shape=M,N,K[,act]
//   CHECK_NO_WS=1 no scratch is lent (no split-K); CHECK_SPLIT=s forces s K-slices; CHECK_REPS=n timed launches
// (Round 4's A/B of the persistent / register-only-epilogue / n-group variants ran this binary under the development
// switches those variants had -- commits 4e8846e .. 8f2b2a1; the logs are profiles/r04_ab_gemm_*.)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/aurora_hip.h"

#define HIP_OK(x)                                                                    \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                       \
    }                                                                                \
  } while (0)

__device__ __host__ inline float bf2f(uint16_t v) {
  union { uint32_t u; float f; } c;
  c.u = (uint32_t)v << 16;
  return c.f;
}
__device__ inline uint16_t f2bf(float f) {   // round to nearest even
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__global__ void fill_kernel(uint16_t* p, int64_t n, uint32_t seed, float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = f2bf(((float)(x & 0xffffff) / 8388608.0f - 1.0f) * scale);
  }
}
__global__ void fill_f32_kernel(float* p, int64_t n, uint32_t seed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x = (uint32_t)i * 2654435761u ^ seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
  p[i] = (float)(x & 0xffff) / 65536.0f - 0.5f;
}
// one thread per (sampled row, column): err[0] = max |c - ref| / (2^-7 |ref| + 2e-3), i.e. <= ~1 when c is ref to one bf16 ulp
__global__ void check_kernel(const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* C, int64_t M, int N, int K,
                             int act, const int64_t* rows, int n_rows, float* err) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (int64_t)n_rows * N) return;
  const int64_t m = rows[id / N];
  const int n = (int)(id % N);
  const uint16_t* a = A + m * K;
  const uint16_t* w = W + (int64_t)n * K;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int k = 0; k < K; k += 4) {
    s0 = fmaf(bf2f(a[k]), bf2f(w[k]), s0);
    s1 = fmaf(bf2f(a[k + 1]), bf2f(w[k + 1]), s1);
    s2 = fmaf(bf2f(a[k + 2]), bf2f(w[k + 2]), s2);
    s3 = fmaf(bf2f(a[k + 3]), bf2f(w[k + 3]), s3);
  }
  float ref = (s0 + s1) + (s2 + s3) + (bias ? bias[n] : 0.f);
  if (act == AURORA_ACT_GELU) ref = 0.5f * ref * (1.0f + erff(ref * 0.70710678f));
  const float c = bf2f(C[m * N + n]);
  const float e = fabsf(c - ref) / (fabsf(ref) * 0.0078125f + 2e-3f);
  if (e > 0.f) atomicMax(reinterpret_cast<unsigned int*>(err), __float_as_uint(e));
}

struct Shape { const char* name; int64_t M; int N, K, act, weight; };

static std::vector<Shape> shapes_of(const std::string& set) {
  std::vector<Shape> v;
  if (set.rfind("shape=", 0) == 0) {   // shape=M,N,K,act : one ad-hoc problem (tools/pmc_gemm_fetch.sh profiles one per run)
    static char nm[64];
    long long M = 0;
    int N = 0, K = 0, act = 0;
    if (sscanf(set.c_str(), "shape=%lld,%d,%d,%d", &M, &N, &K, &act) >= 3) {
      snprintf(nm, sizeof nm, "%lldx%dx%d", M, N, K);
      v.push_back({nm, M, N, K, act, 0});
    }
    return v;
  }
  if (set == "step") {   // the un-sharded 0.25-degree step (weight = launches per step)
    v = {{"s0.qkv", 259200, 1536, 512, 0, 12}, {"s0.proj", 259200, 512, 512, 0, 12}, {"s0.fc1", 259200, 2048, 512, 1, 12},
         {"s0.fc2", 259200, 512, 2048, 0, 12}, {"s1.qkv", 64800, 3072, 1024, 0, 20}, {"s1.proj", 64800, 1024, 1024, 0, 20},
         {"s1.fc1", 64800, 4096, 1024, 1, 20}, {"s1.fc2", 64800, 1024, 4096, 0, 20}, {"s2.qkv", 16200, 6144, 2048, 0, 16},
         {"s2.proj", 16200, 2048, 2048, 0, 16}, {"s2.fc1", 16200, 8192, 2048, 1, 16}, {"s2.fc2", 16200, 2048, 8192, 0, 16}};
  } else if (set == "band8" || set == "band4" || set == "band2") {   // per-rank shapes of a latitude band (largest rank)
    const int r = set == "band8" ? 8 : set == "band4" ? 4 : 2;
    const int64_t rows0 = r == 8 ? 24 : r == 4 ? 48 : 96;   // stage-0 rows of the largest band (15 units of 12: 2, 4, 8)
    const int64_t m0 = rows0 * 360 * 4, m1 = m0 / 4, m2 = m1 / 4;
    static char names[12][16];
    const char* base[12] = {"s0.qkv", "s0.proj", "s0.fc1", "s0.fc2", "s1.qkv", "s1.proj", "s1.fc1", "s1.fc2", "s2.qkv", "s2.proj", "s2.fc1", "s2.fc2"};
    const int64_t Ms[3] = {m0, m1, m2};
    const int D[3] = {512, 1024, 2048}, cnt[3] = {12, 20, 16};
    for (int s = 0; s < 3; ++s) {
      const int d = D[s];
      const int NK[4][3] = {{3 * d, d, 0}, {d, d, 0}, {4 * d, d, 1}, {d, 4 * d, 0}};
      for (int j = 0; j < 4; ++j) {
        snprintf(names[4 * s + j], 16, "r%d.%s", r, base[4 * s + j]);
        v.push_back({names[4 * s + j], Ms[s], NK[j][0], NK[j][1], NK[j][2], cnt[s]});
      }
    }
  } else if (set == "small") {   // ragged / tiny shapes: every code path of the tile and split logic
    v = {{"t.1", 1024, 256, 512, 0, 0},   {"t.2", 1300, 512, 1024, 1, 0},  {"t.3", 2160, 2048, 2048, 0, 0},
         {"t.4", 2160, 2048, 8192, 0, 0}, {"t.5", 70000, 512, 512, 0, 0},  {"t.6", 66000, 768, 256, 1, 0},
         {"t.7", 65537, 256, 1024, 0, 0}, {"t.8", 4320, 1024, 4096, 0, 0}, {"t.9", 131072, 256, 128, 0, 0}};
  }
  return v;
}

int main(int argc, char** argv) {
  std::vector<std::string> sets;
  for (int i = 1; i < argc; ++i) sets.push_back(argv[i]);
  if (sets.empty()) sets = {"small", "band8", "step"};
  const int forced_split = getenv("CHECK_SPLIT") ? atoi(getenv("CHECK_SPLIT")) : 0;   // 0: library's choice
  const bool use_ws = !getenv("CHECK_NO_WS");
  const int reps = getenv("CHECK_REPS") ? atoi(getenv("CHECK_REPS")) : 20;
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  const int64_t ws_bytes = 160 << 20;
  void* ws;
  int32_t* tickets;
  float* err;
  HIP_OK(hipMalloc(&ws, ws_bytes));
  HIP_OK(hipMalloc(&tickets, 4096 * 4));
  HIP_OK(hipMemset(tickets, 0, 4096 * 4));
  HIP_OK(hipMalloc(&err, 4));
  printf("# CHECK_SPLIT=%d ws=%d\n", forced_split, (int)use_ws);
  int bad = 0;
  for (const auto& set : sets) {
    double tot_us = 0, tot_fl = 0;
    for (const Shape& sh : shapes_of(set)) {
      uint16_t *A, *W, *C;
      float* bias;
      HIP_OK(hipMalloc(&A, sh.M * sh.K * 2));
      HIP_OK(hipMalloc(&W, (int64_t)sh.N * sh.K * 2));
      HIP_OK(hipMalloc(&C, sh.M * sh.N * 2));
      HIP_OK(hipMalloc(&bias, sh.N * 4));
      fill_kernel<<<4096, 256, 0, st>>>(A, sh.M * sh.K, 0x1234567u, 1.0f);
      fill_kernel<<<4096, 256, 0, st>>>(W, (int64_t)sh.N * sh.K, 0x7654321u, 1.7f / sqrtf((float)sh.K));
      fill_f32_kernel<<<(sh.N + 255) / 256, 256, 0, st>>>(bias, sh.N, 99u);
      HIP_OK(hipMemsetAsync(C, 0xff, sh.M * sh.N * 2, st));   // NaN pattern: an unwritten element fails the check
      auto run = [&]() {
        int rc;
        if (use_ws)
          rc = aurora_hip_linear_ws(A, sh.K, W, sh.K, bias, C, sh.N, nullptr, 0, nullptr, 0, sh.M, sh.N, sh.K, AURORA_BF16, sh.act,
                                    ws, ws_bytes, tickets, 4096, forced_split, st);
        else
          rc = aurora_hip_linear(A, sh.K, W, sh.K, bias, C, sh.N, nullptr, 0, nullptr, 0, sh.M, sh.N, sh.K, AURORA_BF16, sh.act, st);
        if (rc != 0) {
          fprintf(stderr, "%s: %s\n", sh.name, aurora_hip_last_error());
          exit(3);
        }
      };
      run();
      // sampled rows
      std::vector<int64_t> rows;
      const int64_t stride = sh.M > 8192 ? 97 : 1;
      for (int64_t m = 0; m < sh.M; m += stride) rows.push_back(m);
      if (stride > 1)
        for (int64_t m = sh.M > 300 ? sh.M - 300 : 0; m < sh.M; ++m) rows.push_back(m);
      int64_t* d_rows;
      HIP_OK(hipMalloc(&d_rows, rows.size() * 8));
      HIP_OK(hipMemcpyAsync(d_rows, rows.data(), rows.size() * 8, hipMemcpyHostToDevice, st));
      HIP_OK(hipMemsetAsync(err, 0, 4, st));
      const int64_t items = (int64_t)rows.size() * sh.N;
      check_kernel<<<(unsigned)((items + 255) / 256), 256, 0, st>>>(A, W, bias, C, sh.M, sh.N, sh.K, sh.act, d_rows, (int)rows.size(), err);
      float h_err = -1.f;
      HIP_OK(hipMemcpyAsync(&h_err, err, 4, hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      // second call on the SAME buffers (tickets must have been left zero), then timing
      run();
      run();
      HIP_OK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) run();
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipStreamSynchronize(st));
      float ms = 0;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      // and check once more after the timing loop (persistent / split state survived repeated launches)
      HIP_OK(hipMemsetAsync(err, 0, 4, st));
      check_kernel<<<(unsigned)((items + 255) / 256), 256, 0, st>>>(A, W, bias, C, sh.M, sh.N, sh.K, sh.act, d_rows, (int)rows.size(), err);
      float h_err2 = -1.f;
      HIP_OK(hipMemcpyAsync(&h_err2, err, 4, hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      const double us = ms * 1e3 / reps, fl = 2.0 * sh.M * sh.N * sh.K;
      const bool ok = h_err >= 0.f && h_err < 1.6f && h_err2 >= 0.f && h_err2 < 1.6f && h_err == h_err && h_err2 == h_err2;
      bad += !ok;
      printf("%-10s M=%7lld N=%5d K=%5d act=%d  %9.1f us  %7.1f TF/s  err %.3f %.3f %s  ws=%lld\n", sh.name, (long long)sh.M, sh.N, sh.K,
             sh.act, us, fl / us / 1e6, h_err, h_err2, ok ? "ok" : "FAIL", (long long)aurora_hip_linear_workspace(sh.M, sh.N, sh.K, AURORA_BF16));
      fflush(stdout);
      tot_us += us * sh.weight;
      tot_fl += fl * sh.weight;
      HIP_OK(hipFree(A)); HIP_OK(hipFree(W)); HIP_OK(hipFree(C)); HIP_OK(hipFree(bias)); HIP_OK(hipFree(d_rows));
    }
    if (tot_us > 0) printf("%s weighted: %.2f ms per step, %.0f TF/s\n", set.c_str(), tot_us / 1e3, tot_fl / tot_us / 1e6);
  }
  printf(bad ? "FAILED: %d shapes\n" : "all shapes ok\n", bad);
  return bad ? 1 : 0;
}
