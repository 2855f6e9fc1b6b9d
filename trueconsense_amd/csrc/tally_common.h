// tally_common.h — shared by the aligned-read tally kernel (tally_planes.hip) and the device packer
// (pack_device.hip): launch arguments, the tail blocks that count event words, the block scan and the
// ride-along call.
#pragma once
#include "call_device.h"
#include "tcmi_internal.h"

constexpr int FB = TCMI_F_BLOCK;                // lanes per workgroup
constexpr int MAXPOS = TCMI_F_MAXW * 8;         // positions in the largest window
constexpr int NLD = TCMI_F_SEQCAP / (4 * FB);   // 16-byte loads per lane that cover the largest stage
static_assert(NLD >= 1 && NLD <= 6, "prefetch registers are written out for up to 6 loads per lane");

struct FastArgs {
    const uint32_t *lenoff;         // packed read headers
    const uint32_t *seq;            // {lo, hi} plane pairs
    const tcmi_fast_chunk *chunks;
    const uint32_t *events;
    const uint32_t *covrun;         // coverage runs (tcmi_fast_chunk::run0 / n_runs)
    int32_t *counts;
    int64_t ld;
    int64_t n_events;
    int32_t n_chunks;
    int32_t L;
    int32_t mincov, include_ambig;  // of the ride-along call
    uint8_t *plain, *alt, *flags;
    // ride-along call: the first n_call2 blocks call a matrix that an EARLIER launch finished
    int32_t *counts2;
    int64_t ld2;
    int32_t L2, n_call2;
    int32_t pair_ok;                // the matrix allows 64-bit adds over two adjacent positions (8-byte aligned, even ld)
    // A read set whose totals are still on the device (the one-sync file path: the launch is queued behind the packer, nobody has
    // read its counts back): {chunks, event words} as the packer left them; n_chunks / n_events above are then the CAPACITIES the
    // grid was sized from — blocks beyond the real counts leave at once — and n_tail blocks stride over the event words.
    const uint32_t *dev_counts;
    int32_t n_tail;                 // tail blocks of the launch (>= 1 when there can be events)
    int32_t n_chunk_blocks;         // blocks that take chunks: block b the chunks b, b + n_chunk_blocks, ..
};

constexpr int TILE = FB;            // positions per block of the ride-along call: one lane each

// Ride-along call: block t calls tile t of ANOTHER matrix, complete since an earlier launch on the stream (plain
// loads), with this launch's mincov / ambiguity switch, and leaves it zeroed.  The pipeline attaches the call of
// step k to the tally launch of step k + 1 (another workspace): one launch per step, and the call's PCIe stores
// overlap with the tally's streaming instead of sitting between two launches.
static __device__ inline void call_other_tile(const FastArgs &a, int t)
{
    const int64_t p = (int64_t)t * TILE + threadIdx.x;
    if (p >= a.L2) return;
    int32_t v[TCMI_NCOL];
#pragma unroll
    for (int c = 0; c < TCMI_NCOL; ++c) {
        int32_t *q = &a.counts2[(int64_t)c * a.ld2 + p];
        v[c] = *q;
        *q = 0;
    }
    const tcmi_calldev::Record rec = tcmi_calldev::call_position(v[TCMI_COV], v[TCMI_A], v[TCMI_T], v[TCMI_C], v[TCMI_G],
                                                                 v[TCMI_X], v[TCMI_I], a.mincov, a.include_ambig);
    a.plain[p] = rec.plain;
    a.alt[p] = rec.alt;
    a.flags[p] = rec.flags;
}

// inclusive block scan of one int over the workgroup
static __device__ inline int block_scan_incl(int v, int *wave_tot /* LDS [FB / 64] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    __syncthreads();                            // wave_tot may still be read from a previous scan
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < FB / 64; ++w)
        if (w < wave) base += wave_tot[w];
    return v + base;
}

// Tail blocks (behind the ride-along blocks, in front of the chunk blocks): the tokens that are no plain A/C/G/T bases, as event words
// (position | kind).  Equal words are counted inside the wave (ballot match), one atomic per distinct word
// and wave: at an indel site thousands of reads carry the same event.  A covered position without an
// A/C/G/T base was counted into column A by subtraction in the chunk blocks and is taken out here.
static __device__ inline void tally_tail_block(const FastArgs &a, int tail /* which of the n_tail tail blocks */, int64_t n_events)
{
    const int tid = threadIdx.x, lane = tid & 63;
    for (int64_t i0 = (int64_t)tail * FB; i0 < n_events; i0 += (int64_t)a.n_tail * FB) {
    const int64_t i = i0 + tid;
    const bool valid = i < n_events;
    const uint32_t key = valid ? a.events[i] : 0u;
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k = (uint32_t)__shfl((int)key, leader, 64);
        const unsigned long long same = __ballot(valid && key == k);
        if (lane == leader) {
            const int n = __popcll(same);
            const int p = (int)(k & (TCMI_F_EVPOS - 1u));
            if (p < a.L) {
                if (k & TCMI_F_EV_OTHER) atomicSub(&a.counts[(int64_t)TCMI_A * a.ld + p], n);
                if (k & TCMI_F_EV_X) atomicAdd(&a.counts[(int64_t)TCMI_X * a.ld + p], n);
                if (k & TCMI_F_EV_I) atomicAdd(&a.counts[(int64_t)TCMI_I * a.ld + p], n);
            }
        }
        todo &= ~same;
    }
    }
}
