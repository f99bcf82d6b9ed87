/* The halo transport of a latitude band over RCCL, in plain C: the two callbacks aurora_hip_band asks for
 * (include/aurora_hip.h: `post` starts the point-to-point messages of one exchange asynchronously to the launch stream,
 * `wait` makes the launch stream -- not the host -- wait for them), SURVEY.md section 8e: ncclSend / ncclRecv pairs in one
 * ncclGroup per exchange, rank +- 1 only.  aurora_amd/engine/native.py:_Transport does the same through
 * torch.distributed.batch_isend_irecv; with this file a band runs with no Python in the process.
 *
 * Ordering (the same three edges as the torch transport):
 *   post:  the side stream waits for an event recorded on the launch stream (the gather kernel that filled `send`, and
 *          every earlier reader of `recv`, are done before a byte moves); the group's sends / receives run on the side
 *          stream, so the rank's own qkv GEMM on the launch stream overlaps the transfer;
 *   wait:  the launch stream waits for the event recorded behind the group on the side stream.
 * One exchange is in flight at a time (the step calls wait before the next post), so one event pair suffices -- and ONLY
 * then: post and wait must strictly alternate, which both callbacks check (`posted`) and refuse otherwise.
 */
#ifndef AURORA_RCCL_TRANSPORT_H
#define AURORA_RCCL_TRANSPORT_H

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>

#include "aurora_hip.h"

typedef struct rccl_transport {
  ncclComm_t comm;
  int rank, world;
  hipStream_t side;          /* the stream the messages run on */
  hipEvent_t ready, done;    /* launch stream -> side stream, side stream -> launch stream */
  char* send;                /* staging buffers handed to aurora_hip_set_band_staging (device memory) */
  char* recv;
  int64_t staging_bytes;
  int64_t exchanges, bytes_sent;   /* counters for reports */
  int posted;                /* an exchange is in flight: posted, not yet waited for */
  char id_file[1024];        /* rank 0: the bootstrap file it wrote, removed again by rccl_transport_destroy */
  char error[256];           /* last failure, for the host's message (the callbacks only return -1) */
} rccl_transport;

/* Bootstrap without MPI: rank 0 creates the ncclUniqueId and writes {magic, run_nonce, id} to `id_file` (atomically: temp
 * file + rename), the other ranks poll for a file that carries THEIR run_nonce -- the launcher hands every rank of one run
 * the same non-zero number (a time stamp, a job id), so a file left behind by a crashed run, however recent, is never taken
 * for this run's.  run_nonce 0: no launcher-provided number; a file then counts if it is at most timeout_s seconds old (a
 * relaunch within that window can still meet the old file: pass a nonce, or a path that is new per run).  Rank 0 removes
 * its file in rccl_transport_destroy.  Then ncclCommInitRank.  The caller has selected its device.  `id_file` must fit
 * rccl_transport.id_file. */
int rccl_transport_init(rccl_transport* t, int rank, int world, const char* id_file, unsigned long long run_nonce,
                        double timeout_s);
/* Allocates the two staging buffers (call after aurora_hip_precompute: aurora_hip_band_staging_bytes is known then). */
int rccl_transport_allocate(rccl_transport* t, int64_t staging_bytes);
/* Every rank sends a rank-stamped pattern of `bytes` to its neighbours and checks what it received: run once before the
 * first step (a mis-wired fabric or rank order shows up here, not as a wrong forecast).  0 on success. */
int rccl_transport_selftest(rccl_transport* t, int64_t bytes, hipStream_t stream);
void rccl_transport_destroy(rccl_transport* t);

/* The callbacks; `user` is the rccl_transport. */
int rccl_transport_post(void* user, const aurora_hip_halo_msg* sends, int32_t n_sends, const aurora_hip_halo_msg* recvs,
                        int32_t n_recvs, void* stream);
int rccl_transport_wait(void* user, void* stream);

#endif
