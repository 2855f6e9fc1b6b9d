// tally_fast_common.h — shared by the two aligned-read tally kernels (tally_fast.hip: one-hot nibbles,
// tally_planes.hip: 2-bit codes as two bit planes): launch arguments, the tail blocks that count event
// words, the block scan, and the optional fused call (see tally_fast.hip, "fused call").
#pragma once
#include "call_device.h"
#include "tcmi_internal.h"

constexpr int FB = TCMI_F_BLOCK;                // lanes per workgroup
constexpr int MAXPOS = TCMI_F_MAXW * 8;         // positions in the largest window
constexpr int NLD = TCMI_F_SEQCAP / (4 * FB);   // 16-byte loads per lane that cover the largest stage
static_assert(NLD >= 1 && NLD <= 6, "prefetch registers are written out for up to 6 loads per lane");

struct FastArgs {
    const int32_t *pos;
    const uint32_t *lenoff;
    const uint32_t *seq;
    const tcmi_fast_chunk *chunks;
    const uint32_t *events;
    const uint32_t *covrun;         // format 2: coverage runs (tcmi_fast_chunk::run0 / n_runs)
    int32_t *counts;
    int64_t ld;
    int64_t n_events;
    int32_t n_chunks;
    int32_t L;
    // fused call (FUSED launches only)
    const int32_t *tile_need;       // [n_tiles] workgroups that add into tile t
    int32_t *tile_done;             // [>= all tiles below L] sign-offs so far; zero between launches
    const int32_t *ev_tile_off;     // [n_tail + 1] tiles touched by tail block b: ev_tile[ev_tile_off[b] .. ev_tile_off[b+1])
    const int32_t *ev_tile;
    const int32_t *orphans;         // [n_orphans] tiles below n_tiles nobody adds into
    int32_t n_tail, n_tiles, n_orphans;
    int32_t mincov, include_ambig;
    uint8_t *plain, *alt, *flags;
    // ride-along call (unfused launches): the first n_call2 blocks call a matrix that an EARLIER launch finished
    int32_t *counts2;
    int64_t ld2;
    int32_t L2, n_call2;
    int32_t pair_ok;                // the matrix allows 64-bit adds over two adjacent positions (8-byte aligned, even ld)
    int32_t other_col;              // column the chunk blocks count class-less covered positions into (by subtraction)
};

void tcmi_dispatch_tally_planes(const FastArgs &a, unsigned grid, hipStream_t stream, bool fused);   // tally_planes.hip

constexpr int TILE = FB;            // positions per tile of the fused call: one lane each

// Memory ordering of the fused call.  Everything the workgroups tell each other goes through AGENT-scope atomics
// (the adds into the matrix, the sign-off counter, the caller's sc1 loads of the finished counters): on gfx942/950
// those are performed at the device's coherence point, never held dirty in one XCD's L2 (two XCDs adding into one
// counter is what the tally relies on anyway).  So a lane only has to WAIT until its adds have
// been performed (s_waitcnt vmcnt(0)) before the workgroup signs off — no `__threadfence()`: its agent-scope
// release / acquire would write back and invalidate a whole L2 per wave (measured: 1.2 ms instead of 80 us).
static __device__ inline void wait_until_adds_are_performed()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // compiler: nothing moves below
    __builtin_amdgcn_s_waitcnt(0);                              // vmcnt(0) expcnt(0) lgkmcnt(0)
}

// The calling workgroup of tile t: every workgroup that adds into the tile has signed off, so the counters are
// final.  Agent-scope atomic loads: they must not hit in a stale L1 / another XCD's L2 line.
static __device__ inline void call_tile(const FastArgs &a, int t)
{
    const int64_t p = (int64_t)t * TILE + threadIdx.x;
    if (p < a.L) {
        int32_t v[TCMI_NCOL];
#pragma unroll
        for (int c = 0; c < TCMI_NCOL; ++c) {
            int32_t *q = &a.counts[(int64_t)c * a.ld + p];
            v[c] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int c = 0; c < TCMI_NCOL; ++c)                     // zero for the next launch (visible after the kernel boundary)
            __hip_atomic_store(&a.counts[(int64_t)c * a.ld + p], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const tcmi_calldev::Record rec = tcmi_calldev::call_position(v[TCMI_COV], v[TCMI_A], v[TCMI_T], v[TCMI_C], v[TCMI_G],
                                                                     v[TCMI_X], v[TCMI_I], a.mincov, a.include_ambig);
        a.plain[p] = rec.plain;
        a.alt[p] = rec.alt;
        a.flags[p] = rec.flags;
    }
    if (threadIdx.x == 0) __hip_atomic_store(&a.tile_done[t], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

// Sign off `nt` tiles (tile index for slot k from `tile_of(k)`), then call those this workgroup completed.
// All lanes of the workgroup must arrive; every lane has issued its adds into the matrix before.
template <class TileOf>
static __device__ inline void sign_off_and_call(const FastArgs &a, int nt, TileOf tile_of, int *s_last /* LDS [FB] */)
{
    wait_until_adds_are_performed();
    __syncthreads();
    for (int k0 = 0; k0 < nt; k0 += FB) {
        const int k = k0 + (int)threadIdx.x;
        const int n = min(FB, nt - k0);
        if (k < nt) {
            const int t = tile_of(k);
            const int old = __hip_atomic_fetch_add(&a.tile_done[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last[threadIdx.x] = old == a.tile_need[t] - 1 ? t : -1;
        }
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const int t = s_last[j];
            if (t >= 0) call_tile(a, t);
        }
        __syncthreads();
    }
}


// inclusive block scan of one int over the workgroup
static __device__ inline int block_scan_incl(int v, int *wave_tot /* LDS [4] */)
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


// Tail blocks (block index >= n_chunks): the tokens that are no plain A/C/G/T bases, as event words
// (position | kind).  Equal words are counted inside the wave (ballot match), one atomic per distinct word
// and wave: at an indel site thousands of reads carry the same event.  A covered position without an
// A/C/G/T base was counted into column `other_col` by subtraction in the chunk blocks and is taken out here.
template <bool FUSED>
static __device__ inline void tally_tail_block(const FastArgs &a, int bid /* block index behind the ride-along blocks */, int *s_last /* LDS [FB] */)
{
    const int tid = threadIdx.x, lane = tid & 63;
    {
        const int64_t i = (int64_t)(bid - a.n_chunks) * FB + tid;
        const bool valid = i < a.n_events;
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
                    if (k & TCMI_F_EV_OTHER) atomicSub(&a.counts[(int64_t)a.other_col * a.ld + p], n);
                    if (k & TCMI_F_EV_X) atomicAdd(&a.counts[(int64_t)TCMI_X * a.ld + p], n);
                    if (k & TCMI_F_EV_I) atomicAdd(&a.counts[(int64_t)TCMI_I * a.ld + p], n);
                }
            }
            todo &= ~same;
        }
        if constexpr (FUSED) {
            const int b = bid - a.n_chunks;
            if (b < a.n_tail) {
                const int o = a.ev_tile_off[b];
                sign_off_and_call(a, a.ev_tile_off[b + 1] - o, [&](int k) { return a.ev_tile[o + k]; },
                                  s_last);
            } else {                            // a tile nobody adds into: call it straight away
                const int k = b - a.n_tail;
                call_tile(a, k < a.n_orphans ? a.orphans[k] : a.n_tiles + (k - a.n_orphans));
            }
        }
    }
}
