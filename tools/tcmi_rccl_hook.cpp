/* tcmi_rccl_hook.cpp — the reduce hook of tcmi_split_step (include/tcmi.h) for a C caller that owns an RCCL communicator:
 * the count matrix of ONE BAM file shared by several GPUs (BASELINE configs[4]; the reference piles the file up in one pass,
 * indexing.py:96-100) is summed to rank `root` with ncclReduce on the context's stream.  libtcmi itself links no collective library.
 *
 *   hipcc -shared -fPIC -o libtcmi_rccl.so tools/tcmi_rccl_hook.cpp -L/opt/rocm/lib -lrccl        (rccl.h wants the HIP headers: C++)
 *
 *   struct tcmi_rccl_user u = { comm, 0 };            // ncclComm_t of this rank, root rank
 *   tcmi_split_step(ctx, file, first_block, n_blocks, L, ld, d_counts, mincov, 1, tcmi_rccl_reduce, &u, rank == 0, &rs, &plain, &alt, &flags);
 */
#include <stdint.h>
#include <rccl/rccl.h>

struct tcmi_rccl_user { ncclComm_t comm; int root; };

extern "C" int tcmi_rccl_reduce(void *user, void *d_counts, int64_t n_int32, void *stream)
{
    const struct tcmi_rccl_user *u = (const struct tcmi_rccl_user *)user;
    return ncclReduce(d_counts, d_counts, (size_t)n_int32, ncclInt32, ncclSum, u->root, u->comm, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
}
