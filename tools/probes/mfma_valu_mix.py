"""The C phase of perceiver_out_kernel in isolation: 24 MFMAs (8 accumulators x 3 terms) and the 70 VALU instructions of a row
fragment's combine (14 x [64-bit DPP broadcast + 4 packed FMAs]) in different interleavings, one wave per SIMD.

    python tools/probes/mfma_valu_mix.py /tmp/mvm.hip && hipcc --offload-arch=gfx950 -O3 /tmp/mvm.hip -o /tmp/mvm && /tmp/mvm
"""
import sys


def mfma(i):
    acc = 100 + 4 * (i % 8)
    return f"v_mfma_f32_16x16x32_f16 v[{acc}:{acc + 3}], v[8:11], v[12:15], v[{acc}:{acc + 3}]"


def valu(q, wreg=22):
    e, sub = divmod(q, 5)
    if sub == 0:
        return f"v_mov_b64_dpp v[{wreg}:{wreg + 1}], v[20:21] row_newbcast:{e} row_mask:0xf bank_mask:0xf"
    acc = 140 + 2 * ((e % 7) * 4 + sub - 1)
    u = 40 + 4 * (sub - 1)
    return f"v_pk_fma_f32 v[{acc}:{acc + 1}], v[{u}:{u + 1}], v[{wreg}:{wreg + 1}], v[{acc}:{acc + 1}] op_sel_hi:[0,1,1]"


def scalar(q):
    """instruction q of the 32-bit form: 26 x [v_mov_b32_dpp + 4 v_fmac_f32]"""
    e, sub = divmod(q, 5)
    if sub == 0:
        return f"v_mov_b32_dpp v22, v20 row_newbcast:{e % 16} row_mask:0xf bank_mask:0xf"
    acc = 140 + ((e % 13) * 4 + sub - 1)
    return f"v_fmac_f32 v{acc}, v22, v{40 + 4 * (sub - 1)}"


def pattern(pat):
    s = []
    if pat == 7:
        s = [scalar(q) for q in range(130)]
    if pat == 8:   # 32-bit form, singly
        for i in range(24):
            s.append(mfma(i))
            s += [scalar(q) for q in range(i * 130 // 24, (i + 1) * 130 // 24)]
    if pat == 9:   # 32-bit form, by whole steps
        for i in range(24):
            s.append(mfma(i))
            for e in range(i * 26 // 24, (i + 1) * 26 // 24):
                s += [scalar(q) for q in range(5 * e, 5 * e + 5)]
    if pat == 10:   # packed form, all MFMAs first, then all VALU
        s = [mfma(i) for i in range(24)] + [valu(q) for q in range(70)]
    if pat == 11:   # packed form: 8 MFMAs, 23 VALU, three times
        for b in range(3):
            s += [mfma(i) for i in range(8 * b, 8 * b + 8)] + [valu(q) for q in range(b * 70 // 3, (b + 1) * 70 // 3)]
    if pat == 0:
        s = [mfma(i) for i in range(24)]
    if pat == 1:
        s = [valu(q) for q in range(70)]
    if pat == 2:   # singly
        for i in range(24):
            s.append(mfma(i))
            s += [valu(q) for q in range(i * 70 // 24, (i + 1) * 70 // 24)]
    if pat == 3:   # by whole steps
        for i in range(24):
            s.append(mfma(i))
            for e in range(i * 14 // 24, (i + 1) * 14 // 24):
                s += [valu(q) for q in range(5 * e, 5 * e + 5)]
    if pat == 4:   # singly, the broadcast a step ahead (two weight pairs in turn)
        s.append(valu(0, 22))
        for i in range(24):
            s.append(mfma(i))
            for q in range(i * 70 // 24, (i + 1) * 70 // 24):
                e, sub = divmod(q, 5)
                if sub == 0:
                    if e + 1 < 14:
                        s.append(valu(5 * (e + 1), 22 + 2 * ((e + 1) & 1)))
                else:
                    s.append(valu(q, 22 + 2 * (e & 1)))
    if pat == 5:   # two VALU instructions behind every MFMA (the free ones), the rest at the end
        q = 0
        for i in range(24):
            s.append(mfma(i))
            s += [valu(q), valu(q + 1)]
            q += 2
        s += [valu(x) for x in range(q, 70)]
    if pat == 6:   # MFMA, nop-free: 3 VALU behind every MFMA but the broadcasts all up front into 14 pairs?  (not register-feasible; bound only)
        for i in range(24):
            s.append(mfma(i))
            s += [valu(q) for q in range(i * 70 // 24, (i + 1) * 70 // 24) if q % 5]
    return "".join(f'  "{x}\\n\\t"\n' for x in s)


NAMES = ["24 MFMAs", "70 VALU", "singly interleaved", "by whole steps", "singly, broadcast a step ahead", "2 VALU per MFMA, rest behind",
         "singly, without the broadcasts", "130 VALU of the 32-bit form", "32-bit form singly", "32-bit form by whole steps",
         "packed: MFMAs, then VALU", "packed: 3 x (8 MFMAs, 23 VALU)"]
out = ['#include <hip/hip_runtime.h>\n#include <stdio.h>\n#include <stdint.h>\n']
for p in range(len(NAMES)):
    out.append(f"#define PAT{p} \\\n" + pattern(p).replace("\n", " \\\n") + '  ""\n')
out.append('''template <int PAT>
__global__ __launch_bounds__(256) void k(uint64_t* t) {
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  for (int it = 0; it < 64; ++it) {
''' + "".join(f"    if constexpr (PAT == {p}) asm volatile(PAT{p} ::: \"memory\");\n" for p in range(len(NAMES))) + '''  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  if (threadIdx.x == 0 && blockIdx.x == 0) t[PAT] = t1 - t0;
}
// two waves per SIMD: waves 0-3 run one pattern, waves 4-7 another -- does the matrix pipe overlap with the partner's VALU?
template <int PA, int PB>
__global__ __launch_bounds__(512) void k2(uint64_t* t) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  if (wave < 4) {
    for (int it = 0; it < 64; ++it) {
      if constexpr (PA == 0) asm volatile(PAT0 ::: "memory");
      if constexpr (PA == 1) asm volatile(PAT1 ::: "memory");
      if constexpr (PA == 7) asm volatile(PAT7 ::: "memory");
    }
  } else {
    for (int it = 0; it < 64; ++it) {
      if constexpr (PB == 0) asm volatile(PAT0 ::: "memory");
      if constexpr (PB == 1) asm volatile(PAT1 ::: "memory");
      if constexpr (PB == 7) asm volatile(PAT7 ::: "memory");
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  if ((threadIdx.x & 255) == 0 && blockIdx.x == 0) t[wave >> 2] = t1 - t0;
}
template <int PA, int PB>
void run2(uint64_t* t, const char* what) {
  hipLaunchKernelGGL((k2<PA, PB>), dim3(256), dim3(512), 0, 0, t);
  uint64_t h[2]; (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
  printf("%-60s waves 0-3: %7.1f   waves 4-7: %7.1f ticks per phase\\n", what, h[0] / 64.0, h[1] / 64.0);
}
int main() {
  uint64_t* t; (void)hipMalloc(&t, 128);
  run2<0, 0>(t, "two waves per SIMD: MFMAs | MFMAs");
  run2<0, 1>(t, "two waves per SIMD: MFMAs | packed VALU");
  run2<0, 7>(t, "two waves per SIMD: MFMAs | 32-bit VALU");
  run2<1, 1>(t, "two waves per SIMD: packed VALU | packed VALU");
  run2<7, 7>(t, "two waves per SIMD: 32-bit VALU | 32-bit VALU");
''' + "".join(f"  hipLaunchKernelGGL(k<{p}>, dim3(256), dim3(256), 0, 0, t);\n" for p in range(len(NAMES))) + '''  uint64_t h[16]; (void)hipMemcpy(h, t, 128, hipMemcpyDeviceToHost);
  const char* n[] = {''' + ", ".join(f'"{n}"' for n in NAMES) + '''};
  for (int i = 0; i < ''' + str(len(NAMES)) + '''; ++i) printf("%-36s %7.1f ticks per C phase\\n", n[i], h[i] / 64.0);
}
''')
open(sys.argv[1], "w").write("".join(out))
