/*
 * tcmi_rccl.h — the RCCL side of tcmi_split_step (include/tcmi.h), in its own small library (libtcmi_rccl.so): libtcmi.so itself
 * links no collective library, the one exchange of the path is a hook of the caller's.  What it replaces: the reference piles ONE
 * BAM file up in a single pass (TrueConsense/indexing.py:96-100); when the file is shared by several GPUs (BASELINE configs[4])
 * every rank tallies its range of BGZF blocks and the int32 count matrices are summed to rank 0 — ncclReduce over xGMI, queued on
 * the context's stream behind the tally, so no host wait lies between the two.
 *
 *   unsigned char id[TCMI_RCCL_ID_BYTES];
 *   if (rank == 0) tcmi_rccl_unique_id(id);            // ... and hand the bytes to the other ranks (a file, a socket, MPI_Bcast)
 *   void *comm; tcmi_rccl_comm_init(world, rank, id, &comm);           // the device is the calling thread's current one
 *   struct tcmi_rccl_user u = { comm, 0 };
 *   tcmi_split_step(ctx, file, first_block, n_blocks, L, ld, d_counts, mincov, 1, tcmi_rccl_reduce, &u, rank, world, &rs, &plain, &alt, &flags);
 *   tcmi_rccl_comm_destroy(comm);
 *
 * Plain C: pointers, sizes, int status codes (0 = ok, else the ncclResult_t; tcmi_rccl_last_error() has RCCL's words for it).
 */
#ifndef TCMI_RCCL_H
#define TCMI_RCCL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCMI_RCCL_ID_BYTES 128                        /* sizeof(ncclUniqueId) */

struct tcmi_rccl_user { void *comm; int root; };      /* `user` of tcmi_rccl_reduce: an ncclComm_t and the rank that receives the sum */

int  tcmi_rccl_unique_id(void *id_out);                                          /* ncclGetUniqueId                                 */
int  tcmi_rccl_comm_init(int world, int rank, const void *id, void **comm_out);  /* ncclCommInitRank on the current device           */
int  tcmi_rccl_comm_destroy(void *comm);                                         /* ncclCommDestroy                                  */
int  tcmi_rccl_comm_info(void *comm, int *world, int *rank, int *device);        /* ncclCommCount / ncclCommUserRank / ncclCommCuDevice */
/* a tcmi_reduce_fn: ncclReduce(d_counts, d_counts, n_int32, ncclInt32, ncclSum, user->root, user->comm, stream) */
int  tcmi_rccl_reduce(void *user, void *d_counts, int64_t n_int32, void *stream);
const char *tcmi_rccl_last_error(void);                                          /* of the calling thread                            */

#ifdef __cplusplus
}
#endif
#endif
