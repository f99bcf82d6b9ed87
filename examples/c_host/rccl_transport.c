/* See rccl_transport.h.  C99 + the HIP runtime's C API + RCCL's C API. */
#include "rccl_transport.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define T_FAIL(t, ...)                                   \
  do {                                                   \
    snprintf((t)->error, sizeof (t)->error, __VA_ARGS__); \
    return -1;                                           \
  } while (0)
#define T_HIP(t, call)                                                         \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) T_FAIL(t, "%s: %s", #call, hipGetErrorString(e_));   \
  } while (0)
#define T_NCCL(t, call)                                                        \
  do {                                                                         \
    ncclResult_t r_ = (call);                                                  \
    if (r_ != ncclSuccess) T_FAIL(t, "%s: %s", #call, ncclGetErrorString(r_)); \
  } while (0)

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* what the bootstrap file holds */
typedef struct { unsigned long long magic, nonce; ncclUniqueId id; } id_record;
#define ID_MAGIC 0x4155524f52414944ull   /* "AURORAID" */

int rccl_transport_init(rccl_transport* t, int rank, int world, const char* id_file, unsigned long long run_nonce,
                        double timeout_s) {
  memset(t, 0, sizeof *t);
  t->rank = rank, t->world = world;
  if (world < 1 || rank < 0 || rank >= world) T_FAIL(t, "bad rank %d of %d", rank, world);   /* (world 1: rccl_loopback.c) */
  if (!id_file || strlen(id_file) + 5 > sizeof t->id_file) T_FAIL(t, "bootstrap path missing or longer than %zu bytes", sizeof t->id_file - 5);
  id_record rec;
  memset(&rec, 0, sizeof rec);
  if (rank == 0) {
    rec.magic = ID_MAGIC, rec.nonce = run_nonce;
    T_NCCL(t, ncclGetUniqueId(&rec.id));
    char tmp[sizeof t->id_file + 8];
    snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
    FILE* f = fopen(tmp, "wb");
    if (!f) T_FAIL(t, "cannot write %.200s", tmp);
    const size_t n = fwrite(&rec, sizeof rec, 1, f);
    if (fclose(f) != 0 || n != 1) T_FAIL(t, "cannot write %.200s", tmp);
    if (rename(tmp, id_file) != 0) T_FAIL(t, "cannot rename %.200s", tmp);   /* (replaces a stale file atomically) */
    snprintf(t->id_file, sizeof t->id_file, "%s", id_file);                  /* removed again by rccl_transport_destroy */
  } else {
    /* A file left behind by a run that crashed must not be taken for this run's.  With a launcher-provided nonce: only a
     * record that carries it counts (whatever its age, whatever the clocks say).  Without one: only a file written within
     * the last `timeout_s` seconds (rank 0 of THIS run renames its file into place within that window or this rank gives up
     * anyway). */
    const double t0 = now_s();
    for (;;) {
      struct stat st;
      const int fresh = run_nonce != 0 || (stat(id_file, &st) == 0 && difftime(time(NULL), st.st_mtime) <= timeout_s + 2.0);
      FILE* f = fresh ? fopen(id_file, "rb") : NULL;
      if (f) {
        const size_t n = fread(&rec, sizeof rec, 1, f);
        fclose(f);
        if (n == 1 && rec.magic == ID_MAGIC && rec.nonce == run_nonce) break;
      }
      if (now_s() - t0 > timeout_s)
        T_FAIL(t, "no ncclUniqueId of run %llu in %.160s after %.0f s", run_nonce, id_file, timeout_s);
      struct timespec nap = {0, 20 * 1000 * 1000};
      nanosleep(&nap, NULL);
    }
  }
  const ncclUniqueId id = rec.id;
  T_NCCL(t, ncclCommInitRank(&t->comm, world, id, rank));
  T_HIP(t, hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking));
  T_HIP(t, hipEventCreateWithFlags(&t->ready, hipEventDisableTiming));
  T_HIP(t, hipEventCreateWithFlags(&t->done, hipEventDisableTiming));
  return 0;
}

int rccl_transport_allocate(rccl_transport* t, int64_t staging_bytes) {
  if (staging_bytes < 16) staging_bytes = 16;
  T_HIP(t, hipMalloc((void**)&t->send, (size_t)staging_bytes));
  T_HIP(t, hipMalloc((void**)&t->recv, (size_t)staging_bytes));
  t->staging_bytes = staging_bytes;
  return 0;
}

/* A band talks to rank - 1 and rank + 1 only; a one-rank communicator may talk to itself (rccl_loopback.c: what a one-GPU
 * box can exercise of this file). */
static int peer_ok(const rccl_transport* t, int peer) {
  if (peer < 0 || peer >= t->world) return 0;
  return t->world == 1 ? peer == t->rank : abs(peer - t->rank) == 1;
}

int rccl_transport_post(void* user, const aurora_hip_halo_msg* sends, int32_t n_sends, const aurora_hip_halo_msg* recvs,
                        int32_t n_recvs, void* stream) {
  rccl_transport* t = (rccl_transport*)user;
  for (int i = 0; i < n_sends; ++i)
    if (sends[i].offset < 0 || sends[i].offset + sends[i].bytes > t->staging_bytes || !peer_ok(t, sends[i].peer))
      T_FAIL(t, "bad send message %d (peer %d, %lld + %lld bytes)", i, sends[i].peer, (long long)sends[i].offset, (long long)sends[i].bytes);
  for (int i = 0; i < n_recvs; ++i)
    if (recvs[i].offset < 0 || recvs[i].offset + recvs[i].bytes > t->staging_bytes || !peer_ok(t, recvs[i].peer))
      T_FAIL(t, "bad receive message %d (peer %d, %lld + %lld bytes)", i, recvs[i].peer, (long long)recvs[i].offset, (long long)recvs[i].bytes);
  /* one event pair: correct only while post and wait strictly alternate (the step does: it waits before the next post) */
  if (t->posted) T_FAIL(t, "post: the previous exchange has not been waited for (one exchange is in flight at a time)");
  T_HIP(t, hipEventRecord(t->ready, (hipStream_t)stream));
  T_HIP(t, hipStreamWaitEvent(t->side, t->ready, 0));
  T_NCCL(t, ncclGroupStart());
  for (int i = 0; i < n_sends; ++i) {
    T_NCCL(t, ncclSend(t->send + sends[i].offset, (size_t)sends[i].bytes, ncclUint8, sends[i].peer, t->comm, t->side));
    t->bytes_sent += sends[i].bytes;
  }
  for (int i = 0; i < n_recvs; ++i)
    T_NCCL(t, ncclRecv(t->recv + recvs[i].offset, (size_t)recvs[i].bytes, ncclUint8, recvs[i].peer, t->comm, t->side));
  T_NCCL(t, ncclGroupEnd());
  T_HIP(t, hipEventRecord(t->done, t->side));
  t->exchanges += 1;
  t->posted = 1;
  return 0;
}

int rccl_transport_wait(void* user, void* stream) {
  rccl_transport* t = (rccl_transport*)user;
  if (!t->posted) T_FAIL(t, "wait: no exchange was posted");
  T_HIP(t, hipStreamWaitEvent((hipStream_t)stream, t->done, 0));
  t->posted = 0;
  return 0;
}

int rccl_transport_selftest(rccl_transport* t, int64_t bytes, hipStream_t stream) {
  if (bytes > t->staging_bytes / 2) bytes = t->staging_bytes / 2;
  bytes &= ~(int64_t)15;
  if (bytes <= 0) return 0;
  /* send buffer: [to previous | to next], byte i of the message for peer p = (rank * 31 + p * 7 + i) mod 251 */
  unsigned char* host = (unsigned char*)malloc((size_t)(2 * bytes));
  aurora_hip_halo_msg sends[2], recvs[2];
  int ns = 0;
  for (int side = 0; side < 2; ++side) {
    const int peer = side == 0 ? t->rank - 1 : t->rank + 1;
    if (peer < 0 || peer >= t->world) continue;
    for (int64_t i = 0; i < bytes; ++i) host[ns * bytes + i] = (unsigned char)((t->rank * 31 + peer * 7 + i) % 251);
    sends[ns].peer = recvs[ns].peer = peer;
    sends[ns].reserved = recvs[ns].reserved = 0;
    sends[ns].offset = recvs[ns].offset = ns * bytes;
    sends[ns].bytes = recvs[ns].bytes = bytes;
    ++ns;
  }
  T_HIP(t, hipMemcpyAsync(t->send, host, (size_t)(ns * bytes), hipMemcpyHostToDevice, stream));
  T_HIP(t, hipMemsetAsync(t->recv, 0xee, (size_t)(ns * bytes), stream));
  if (rccl_transport_post(t, sends, ns, recvs, ns, stream) != 0 || rccl_transport_wait(t, stream) != 0) {
    free(host);
    return -1;
  }
  T_HIP(t, hipMemcpyAsync(host, t->recv, (size_t)(ns * bytes), hipMemcpyDeviceToHost, stream));
  T_HIP(t, hipStreamSynchronize(stream));
  for (int m = 0; m < ns; ++m)
    for (int64_t i = 0; i < bytes; ++i)
      if (host[m * bytes + i] != (unsigned char)((recvs[m].peer * 31 + t->rank * 7 + i) % 251)) {
        const int got = host[m * bytes + i];
        free(host);
        T_FAIL(t, "self-test: byte %lld of the message from rank %d is %d", (long long)i, recvs[m].peer, got);
      }
  free(host);
  return 0;
}

void rccl_transport_destroy(rccl_transport* t) {
  if (t->send) (void)hipFree(t->send);
  if (t->recv) (void)hipFree(t->recv);
  if (t->ready) (void)hipEventDestroy(t->ready);
  if (t->done) (void)hipEventDestroy(t->done);
  if (t->side) (void)hipStreamDestroy(t->side);
  if (t->comm) (void)ncclCommDestroy(t->comm);
  if (t->id_file[0]) (void)unlink(t->id_file);   /* rank 0: the next run must not find this run's id */
  memset(t, 0, sizeof *t);
}
