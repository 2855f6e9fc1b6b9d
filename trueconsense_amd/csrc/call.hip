// call.hip — position-local part of stage B on gfx950: counters -> call record.
//
// One lane per reference position, branch-flattened.  Replaces, per position:
//   Sequences.GetNucleotide / GetDistribution   Sequences.py:119-165  (rank by (count, letter))
//   Ambig.IsAmbiguous + helpers                 Ambig.py:18-228       (fp64 percentages, IUPAC)
//   Events.MinorityDel                          Events.py:85-106      ((X/cov)*100 >= 15)
//   candidate test of Events.ListInserts        Events.py:29-36       ((I/cov)*100 > 55)
//   the case rule                               Sequences.py:229-235  (lower-case iff count < mincov)
// The sequential walk (deletion runs, ORF logic, insert splice) stays on the host
// (consensus_walk.cpp) and reads only these records.
//
// fp64 exactness: percentages are (c/cov)*100 in IEEE double exactly as CPython computes
// them; the library is built with -ffp-contract=off so no fma is formed (SURVEY §7 hard parts).
#include "call_device.h"
#include "tcmi_internal.h"

namespace {

constexpr int BLOCK = 256;

__global__ __launch_bounds__(BLOCK) void call_kernel(int32_t *__restrict__ counts, int64_t L, int64_t ld,
                                                      int32_t mincov, int include_ambig, int clean, uint8_t *__restrict__ plain,
                                                      uint8_t *__restrict__ alt, uint8_t *__restrict__ flags,
                                                      int32_t *__restrict__ events, int32_t *__restrict__ event_counts)
{
    __shared__ int wave_cnt[BLOCK / 64];
    const int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = p < L;
    unsigned f = 0;
    if (valid) {
        const int64_t cov = counts[(int64_t)TCMI_COV * ld + p];
        const int64_t nA = counts[(int64_t)TCMI_A * ld + p], nT = counts[(int64_t)TCMI_T * ld + p];
        const int64_t nC = counts[(int64_t)TCMI_C * ld + p], nG = counts[(int64_t)TCMI_G * ld + p];
        const int64_t nX = counts[(int64_t)TCMI_X * ld + p], nI = counts[(int64_t)TCMI_I * ld + p];
        if (clean) {    // leave the matrix zeroed for the next tally into this workspace (saves a memset launch)
#pragma unroll
            for (int c = 0; c < TCMI_NCOL; ++c) counts[(int64_t)c * ld + p] = 0;
        }
        const tcmi_calldev::Record rec = tcmi_calldev::call_position(cov, nA, nT, nC, nG, nX, nI, mincov, include_ambig);
        f = rec.flags;
        plain[p] = rec.plain;
        alt[p] = rec.alt;
        flags[p] = rec.flags;
    }
    if (events) {   // ordered compaction of event positions inside the block
        const bool ev = valid && (f & TCMI_EVENT_MASK);
        const unsigned long long m = __ballot(ev);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int base = 0, total = 0;
        for (int w = 0; w < BLOCK / 64; ++w) {
            if (w < wave) base += wave_cnt[w];
            total += wave_cnt[w];
        }
        if (ev) events[(int64_t)blockIdx.x * BLOCK + base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)p;
        if (threadIdx.x == 0) event_counts[blockIdx.x] = total;
    }
}

} // namespace

int tcmi_launch_call(tcmi_ctx *ctx, int32_t *d_counts, int64_t L, int64_t ld, int32_t mincov, int include_ambig, int clean,
                     uint8_t *d_plain, uint8_t *d_alt, uint8_t *d_flags, int32_t *d_events, int32_t *d_event_counts)
{
    const int64_t grid = (L + BLOCK - 1) / BLOCK;
    tcmi_prof_begin(ctx, TCMI_K_CALL);
    (void)hipGetLastError();                               // drop any stale error of this thread
    hipLaunchKernelGGL(call_kernel, dim3((unsigned)grid), dim3(BLOCK), 0, ctx->stream, d_counts, L, ld, mincov,
                       include_ambig, clean, d_plain, d_alt, d_flags, d_events, d_event_counts);
    tcmi_prof_end(ctx, TCMI_K_CALL);
    TCMI_HIP(ctx, hipGetLastError());
    return TCMI_OK;
}
