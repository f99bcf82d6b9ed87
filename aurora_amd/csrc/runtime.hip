// Error plumbing and small utility entry points of libaurora_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace aurora {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return AURORA_E_LAUNCH;
  }
  return AURORA_OK;
}

int current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev;
}

int device_cus() {
  static int cus[64] = {0};
  const int dev = current_device() & 63;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

namespace {

template <typename S, typename D>
__global__ void convert_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n8) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    float v[8];
    load8(src + i * 8, v);
    store8(dst + i * 8, v);
  }
}
template <typename S, typename D>
__global__ void convert_tail_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t from, int64_t n) {
  const int64_t i = from + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) elem<D>::store(dst + i, elem<S>::load(src + i));
}

template <typename T>
__global__ void copy2d_kernel(const T* __restrict__ src, int64_t lds_, T* __restrict__ dst, int64_t ldd,
                              int64_t rows, int64_t cols16) {
  // one 16-byte piece per thread, grid-stride over rows * cols16 pieces
  const int64_t total = rows * cols16, stride = (int64_t)gridDim.x * blockDim.x;
  constexpr int E = 16 / sizeof(T);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / cols16, c = i - r * cols16;
    *reinterpret_cast<u32x4*>(dst + r * ldd + c * E) = *reinterpret_cast<const u32x4*>(src + r * lds_ + c * E);
  }
}

__global__ void gather_rows_kernel(const char* __restrict__ src, int64_t src_pitch, const int32_t* __restrict__ idx,
                                   char* __restrict__ dst, int64_t dst_pitch, int64_t n_rows, int pieces) {
  // one 16-byte piece per thread: dst row r = src row idx[r]
  const int64_t total = n_rows * pieces, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / pieces, c = i - r * pieces;
    *reinterpret_cast<u32x4*>(dst + r * dst_pitch + c * 16) =
        *reinterpret_cast<const u32x4*>(src + (int64_t)idx[r] * src_pitch + c * 16);
  }
}

}  // namespace
}  // namespace aurora

using namespace aurora;

extern "C" const char* aurora_hip_last_error(void) { return g_err; }
extern "C" int aurora_hip_version(void) { return 1; }

namespace aurora {
namespace {
// max |x| over a contiguous fp32 array: 16-byte loads, wave + block reduction, at most one atomic per block (non-negative
// floats order like their bit patterns).  NaN compares false everywhere and is ignored; +-inf wins.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n4, int64_t n, float* out) {
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
  // One atomic per BLOCK, and only if it can still raise the word: 8192 same-address atomics (one per wave of 2048 blocks)
  // serialise at ~12 ns each -- a fixed ~100 us per launch, more than the scan of a latitude band's inputs takes.
  __shared__ float part[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    const unsigned int bits = __float_as_uint(m);
    if (bits > __hip_atomic_load(reinterpret_cast<unsigned int*>(out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(reinterpret_cast<unsigned int*>(out), bits);
  }
}
}  // namespace
}  // namespace aurora

namespace aurora {
namespace {
__global__ void zero_words_kernel(float* w, int n) {
  if ((int)threadIdx.x < n) w[threadIdx.x] = 0.f;
}
}  // namespace
}  // namespace aurora

extern "C" int aurora_hip_zero_words(float* words, int n, void* stream) {
  AURORA_CHECK_ARG(words != nullptr && n > 0 && n <= 64, "zero_words: 1 <= n <= 64");
  hipLaunchKernelGGL(aurora::zero_words_kernel, dim3(1), dim3(64), 0, as_stream(stream), words, n);
  return check_launch("zero_words");
}

extern "C" int aurora_hip_absmax(const float* x, int64_t n, float* out, void* stream) {
  AURORA_CHECK_ARG(out != nullptr, "absmax: bad arguments");
  const int rc = aurora_hip_zero_words(out, 1, stream);
  return rc != AURORA_OK ? rc : aurora_hip_absmax_fold(x, n, out, stream);
}

extern "C" int aurora_hip_absmax_fold(const float* x, int64_t n, float* out, void* stream) {
  AURORA_CHECK_ARG(x != nullptr && out != nullptr && n > 0 && (uintptr_t)x % 16 == 0, "absmax: bad arguments");
  const int64_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 + 1 : 2048);
  hipLaunchKernelGGL(aurora::absmax_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, n4, n, out);
  return check_launch("absmax");
}

extern "C" int aurora_hip_convert(const void* src, void* dst, int64_t n, int src_dtype, void* stream) {
  AURORA_CHECK_ARG(src_dtype == AURORA_F32 || src_dtype == AURORA_BF16, "convert: bad dtype");
  if (n <= 0) return AURORA_OK;
  const bool al = ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
  const int64_t n8 = al ? n / 8 : 0;
  const int64_t tail = n - n8 * 8;
  const int blocks = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
  if (src_dtype == AURORA_F32) {
    if (n8) hipLaunchKernelGGL((convert_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, as_stream(stream),
                               (const float*)src, (bf16_t*)dst, n8);
    if (tail) hipLaunchKernelGGL((convert_tail_kernel<float, bf16_t>), dim3((unsigned)((tail + 255) / 256)), dim3(256), 0,
                                 as_stream(stream), (const float*)src, (bf16_t*)dst, n8 * 8, n);
  } else {
    if (n8) hipLaunchKernelGGL((convert_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, as_stream(stream),
                               (const bf16_t*)src, (float*)dst, n8);
    if (tail) hipLaunchKernelGGL((convert_tail_kernel<bf16_t, float>), dim3((unsigned)((tail + 255) / 256)), dim3(256), 0,
                                 as_stream(stream), (const bf16_t*)src, (float*)dst, n8 * 8, n);
  }
  return check_launch("convert");
}

extern "C" int aurora_hip_copy2d(const void* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows,
                                 int64_t cols, int dtype, void* stream) {
  AURORA_CHECK_ARG(dtype == AURORA_F32 || dtype == AURORA_BF16, "copy2d: bad dtype");
  const int es = dtype == AURORA_F32 ? 4 : 2;
  AURORA_CHECK_ARG((cols * es) % 16 == 0 && (lds_ * es) % 16 == 0 && (ldd * es) % 16 == 0 &&
                       (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0,
                   "copy2d: rows must be 16-byte multiples and aligned");
  if (rows <= 0 || cols <= 0) return AURORA_OK;
  const int64_t cols16 = cols * es / 16, total = rows * cols16;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype == AURORA_F32)
    hipLaunchKernelGGL(copy2d_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)src, lds_,
                       (float*)dst, ldd, rows, cols16);
  else
    hipLaunchKernelGGL(copy2d_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), (const bf16_t*)src, lds_,
                       (bf16_t*)dst, ldd, rows, cols16);
  return check_launch("copy2d");
}

extern "C" int aurora_hip_gather_rows(const void* src, int64_t src_pitch_bytes, const int32_t* idx, void* dst,
                                      int64_t dst_pitch_bytes, int64_t n_rows, int64_t row_bytes, void* stream) {
  AURORA_CHECK_ARG(row_bytes % 16 == 0 && src_pitch_bytes % 16 == 0 && dst_pitch_bytes % 16 == 0 &&
                       (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0,
                   "gather_rows: rows must be 16-byte multiples and aligned");
  if (n_rows <= 0 || row_bytes <= 0) return AURORA_OK;
  const int64_t total = n_rows * (row_bytes / 16);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (const char*)src,
                     src_pitch_bytes, idx, (char*)dst, dst_pitch_bytes, n_rows, (int)(row_bytes / 16));
  return check_launch("gather_rows");
}
