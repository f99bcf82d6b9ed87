/* One-rank loop-back of the RCCL halo transport (examples/c_host/rccl_transport.c): the rank is its own neighbour, so
 * ncclSend / ncclRecv inside one ncclGroup move a message from the `send` staging buffer to the `recv` one on the side
 * stream, ordered by the transport's two events against a launch stream that is still busy when the host posts.  Checks
 * the bytes.  What a one-GPU box can show of the transport; exit code 0 = ok, 77 = RCCL could not initialise.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I examples/c_host -I /opt/rocm/include rccl_loopback.c rccl_transport.c \
 *       -L /opt/rocm/lib -lamdhip64 -lrccl -o rccl_loopback && ./rccl_loopback /tmp/nccl_id
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rccl_transport.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const char* id_file = argc > 1 ? argv[1] : "/tmp/aurora_rccl_loopback.id";
  const int64_t n = 1 << 20;
  CHECK_HIP(hipSetDevice(0));
  rccl_transport t;
  if (rccl_transport_init(&t, 0, 1, id_file, 0, 30.0) != 0) {
    fprintf(stderr, "RCCL-UNAVAILABLE %s\n", t.error);
    return 77;
  }
  if (rccl_transport_allocate(&t, 2 * n) != 0) { fprintf(stderr, "allocate: %s\n", t.error); return 1; }
  hipStream_t launch;
  CHECK_HIP(hipStreamCreate(&launch));
  unsigned char* host = (unsigned char*)malloc((size_t)n);
  unsigned char* back = (unsigned char*)malloc((size_t)n);
  for (int64_t i = 0; i < n; ++i) host[i] = (unsigned char)((i * 7 + 3) % 251);
  /* keep the launch stream busy in front of the fill: 64 device-to-device copies of 256 MiB */
  char* big;
  CHECK_HIP(hipMalloc((void**)&big, (size_t)512 << 20));
  for (int r = 0; r < 64; ++r) CHECK_HIP(hipMemcpyAsync(big + ((size_t)256 << 20), big, (size_t)256 << 20, hipMemcpyDeviceToDevice, launch));
  CHECK_HIP(hipMemcpyAsync(t.send, host, (size_t)n, hipMemcpyHostToDevice, launch));
  CHECK_HIP(hipMemsetAsync(t.recv, 0xEE, (size_t)n, launch));
  aurora_hip_halo_msg m;
  memset(&m, 0, sizeof m);
  m.peer = 0; m.offset = 0; m.bytes = n;
  if (rccl_transport_post(&t, &m, 1, &m, 1, launch) != 0) { fprintf(stderr, "post: %s\n", t.error); return 1; }
  if (rccl_transport_wait(&t, launch) != 0) { fprintf(stderr, "wait: %s\n", t.error); return 1; }
  CHECK_HIP(hipMemcpyAsync(back, t.recv, (size_t)n, hipMemcpyDeviceToHost, launch));   /* behind the receive, by `wait` */
  CHECK_HIP(hipMemsetAsync(t.send, 0, (size_t)n, launch));                             /* after the send was ordered */
  CHECK_HIP(hipStreamSynchronize(launch));
  int64_t bad = 0;
  for (int64_t i = 0; i < n; ++i) bad += back[i] != host[i];
  printf("%s exchanges=%lld bytes_sent=%lld wrong=%lld\n", bad ? "RCCL-LOOPBACK-WRONG" : "RCCL-LOOPBACK-OK", (long long)t.exchanges,
         (long long)t.bytes_sent, (long long)bad);
  rccl_transport_destroy(&t);
  free(host); free(back);
  return bad ? 1 : 0;
}
