// Probe the semantics of ds_read_b64_tr_b16 on gfx950: LDS holds u16 element e at index e.
// Every lane reads 8 bytes from its own address `base[lane]`; we print what each lane received.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(const int* lane_addr, unsigned short* out) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  unsigned addr = (unsigned)(size_t)(&lds[0]) + lane_addr[threadIdx.x] * 2;  // byte address in LDS
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  u2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
  out[threadIdx.x * 4 + 0] = r.x & 0xffff; out[threadIdx.x * 4 + 1] = r.x >> 16;
  out[threadIdx.x * 4 + 2] = r.y & 0xffff; out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
  int h_addr[64]; unsigned short h_out[256];
  int *d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int variant = 0; variant < 3; ++variant) {
    for (int l = 0; l < 64; ++l) {
      if (variant == 0) h_addr[l] = l * 4;                                   // lane-linear 8-byte pieces
      if (variant == 1) h_addr[l] = ((l >> 4) * 4 + ((l & 15) >> 2)) * 64 + (l & 3) * 4;  // group g: rows 4g..4g+3 of a [16][64] image, 16-col block 0
      if (variant == 2) h_addr[l] = (l & 15) * 64 + (l >> 4) * 4;            // lane i -> row i, 4 elems at col 4g
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("variant %d\n", variant);
    for (int l = 0; l < 64; ++l)
      printf("  lane %2d addr_elem %4d -> %4d %4d %4d %4d\n", l, h_addr[l], h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
  }
  return 0;
}
