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
#include "tcmi_internal.h"

namespace {

constexpr int BLOCK = 256;

__device__ inline void cswap(int64_t &a, int64_t &b)
{   // descending
    const int64_t hi = a > b ? a : b, lo = a > b ? b : a;
    a = hi; b = lo;
}

__device__ inline char iupac_of(unsigned mask)
{   // bit0 A, bit1 C, bit2 G, bit3 T -> "?ACMGRSV" "TWYHKDBN" packed little-endian (Ambig.py:1-15 tables)
    const uint64_t tab = (mask & 8u) ? 0x4E42444B48595754ull : 0x565352474D43413Full;
    return (char)((tab >> (8u * (mask & 7u))) & 0xFFu);
}

__device__ inline char letter_of(unsigned rank)
{   // "ACGTX" packed little-endian
    return (char)((0x5854474341ull >> (8u * rank)) & 0xFFu);
}

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
        // key = count*8 + letter rank (A<C<G<T<X): Python's sort of (count, letter) tuples
        int64_t k0 = nA * 8 + 0, k1 = nC * 8 + 1, k2 = nG * 8 + 2, k3 = nT * 8 + 3, k4 = nX * 8 + 4;
        // 9-comparator sorting network for 5 keys, descending
        cswap(k0, k1); cswap(k3, k4); cswap(k2, k4); cswap(k2, k3); cswap(k0, k3);
        cswap(k0, k2); cswap(k1, k4); cswap(k1, k3); cswap(k1, k2);
        const unsigned r1 = (unsigned)(k0 & 7), r2 = (unsigned)(k1 & 7), r3 = (unsigned)(k2 & 7);
        const int64_t c1 = k0 >> 3, c2 = k1 >> 3, c3 = k2 >> 3, c4 = k3 >> 3;

        if (cov < mincov) f |= TCMI_F_LOWCOV;
        if (r1 == 4) f |= TCMI_F_PRIMX;
        if (cov > mincov) f |= TCMI_F_COVGT;
        const double dcov = (double)cov;
        if (cov == 0) f |= TCMI_F_COVZERO;
        else if (((double)nX / dcov) * 100.0 >= 15.0) f |= TCMI_F_MINDEL;
        if (cov >= mincov && cov != 0 && nI != 0 && ((double)nI / dcov) * 100.0 > 55.0) f |= TCMI_F_INSCAND;

        char amb = 0;
        if (cov != 0 && r1 != 4 && r2 != 4) {
            const double p1 = ((double)c1 / dcov) * 100.0, p2 = ((double)c2 / dcov) * 100.0;
            const double p3 = ((double)c3 / dcov) * 100.0, p4 = ((double)c4 / dcov) * 100.0;
            if (fabs(p1 - p2) <= 10.0) {
                const unsigned m2 = (1u << r1) | (1u << r2);
                if (fabs(p1 - p3) <= 10.0 && fabs(p2 - p3) <= 10.0) {
                    if (fabs(p1 - p4) <= 10.0 && fabs(p2 - p4) <= 10.0 && fabs(p3 - p4) <= 10.0) amb = 'N';
                    else if (r3 == 4) amb = 'N';
                    else amb = iupac_of(m2 | (1u << r3));
                } else amb = iupac_of(m2);
            }
        }
        if (amb) f |= TCMI_F_AMBIG;
        char ch1 = letter_of(r1), ch2 = letter_of(r2);
        if (c1 < mincov) ch1 = (char)(ch1 | 0x20);
        if (c2 < mincov) ch2 = (char)(ch2 | 0x20);
        plain[p] = (uint8_t)((f & TCMI_F_LOWCOV) ? 'N' : ((include_ambig && amb) ? amb : ch1));
        alt[p] = (uint8_t)ch2;
        flags[p] = (uint8_t)f;
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
