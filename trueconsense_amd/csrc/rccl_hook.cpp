// rccl_hook.cpp — libtcmi_rccl.so (include/tcmi_rccl.h): the reduce hook of tcmi_split_step for a caller that owns an RCCL
// communicator, and the three calls that make one.  The count matrix of ONE BAM file shared by several GPUs (BASELINE configs[4];
// the reference piles the file up in one pass, indexing.py:96-100) is summed to rank `root` with ncclReduce on the context's stream.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <rccl/rccl.h>
#include "../../include/tcmi_rccl.h"

static_assert(sizeof(ncclUniqueId) == TCMI_RCCL_ID_BYTES, "TCMI_RCCL_ID_BYTES must be sizeof(ncclUniqueId)");

static thread_local char t_err[256] = "";

static int note(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return 0;
    snprintf(t_err, sizeof t_err, "%s: %s", what, ncclGetErrorString(r));
    return (int)r;
}

extern "C" {

int tcmi_rccl_unique_id(void *id_out)
{
    if (!id_out) return note(ncclInvalidArgument, "tcmi_rccl_unique_id");
    ncclUniqueId id;
    const int rc = note(ncclGetUniqueId(&id), "ncclGetUniqueId");
    if (!rc) memcpy(id_out, &id, sizeof id);
    return rc;
}

int tcmi_rccl_comm_init(int world, int rank, const void *id_bytes, void **comm_out)
{
    if (!id_bytes || !comm_out || world < 1 || rank < 0 || rank >= world) return note(ncclInvalidArgument, "tcmi_rccl_comm_init");
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    ncclComm_t comm = nullptr;
    const int rc = note(ncclCommInitRank(&comm, world, id, rank), "ncclCommInitRank");
    *comm_out = rc ? nullptr : (void *)comm;
    return rc;
}

int tcmi_rccl_comm_destroy(void *comm)
{
    return comm ? note(ncclCommDestroy((ncclComm_t)comm), "ncclCommDestroy") : 0;
}

int tcmi_rccl_comm_info(void *comm, int *world, int *rank, int *device)
{
    if (!comm) return note(ncclInvalidArgument, "tcmi_rccl_comm_info");
    int rc = 0;
    if (world && !rc) rc = note(ncclCommCount((ncclComm_t)comm, world), "ncclCommCount");
    if (rank && !rc) rc = note(ncclCommUserRank((ncclComm_t)comm, rank), "ncclCommUserRank");
    if (device && !rc) rc = note(ncclCommCuDevice((ncclComm_t)comm, device), "ncclCommCuDevice");
    return rc;
}

int tcmi_rccl_reduce(void *user, void *d_counts, int64_t n_int32, void *stream)
{
    const struct tcmi_rccl_user *u = (const struct tcmi_rccl_user *)user;
    if (!u || !u->comm || !d_counts || n_int32 < 0) return note(ncclInvalidArgument, "tcmi_rccl_reduce");
    return note(ncclReduce(d_counts, d_counts, (size_t)n_int32, ncclInt32, ncclSum, u->root, (ncclComm_t)u->comm, (hipStream_t)stream), "ncclReduce");
}

const char *tcmi_rccl_last_error(void) { return t_err; }

} // extern "C"
