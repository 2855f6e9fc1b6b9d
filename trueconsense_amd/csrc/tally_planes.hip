// tally_planes.hip — stage A for ALIGNED reads on gfx950, bases as 2-bit codes in two bit planes.
// Counterpart of the per-token loop of indexing.py:102-132 for the tokens that are plain bases
// (SURVEY §8-P2).
//
// Data (tcmi_internal.h): per read ONE packed header word and its aligned bases as codes
// A=0 C=1 G=2 T=3 (anything else 0, listed as an OTHER event), 32 bases per pair of 32-bit words
// {lo plane, hi plane}, with one zero pair between reads: 52 bytes for a 150-bp read.  The layout is
// produced on the device by pack_device.hip (default) or on the host by readset.cpp.
//
// One workgroup per chunk (<= 8 stages of <= 510 reads), lane (g, s) owns 32 positions g of the
// window and depth slice s of the reads.  Per read of the slice: one 64-bit LDS header, ONE
// ds_read2_b64 (two pairs), two v_alignbit funnel shifts bring the read's planes onto the lane's
// 32 positions (a lane that straddles an end of the read sees the zero pair there; a lane wholly
// outside the read is told so by its clamped pair index and takes zeros); lo, hi and lo&hi
// (= C|T, G|T, T) are then COUNTED BIT-SLICED: carry-save adders (sum and carry: one v_bitop3_b32
// each) fold eight reads into the ones / twos / fours planes and an eights carry that ripples through
// the upper planes (8 planes: <= 255 reads per lane and chunk).  ~19 VALU instructions per read and
// 32 positions.  At the end of the chunk the planes are spread into byte counters once, the slices
// are summed through LDS, and per position
//     C = n(lo) - n(lo&hi),  G = n(hi) - n(lo&hi),  T = n(lo&hi),  A = coverage - C - G - T
// (covered positions without an A/C/G/T base land in A and are taken out by the tail blocks).
// Coverage comes from the packer's per-chunk list of runs of reads with equal (position, length):
// a (+n, -n) pair per run in an LDS difference array, prefix-summed at the end.
//
// HBM-streaming integer work: no MFMA (BASELINE.json north_star).
#include <algorithm>

#include "tally_common.h"

namespace {

#ifndef TCMI_P_BODY8
#define TCMI_P_BODY8 1   // 0: four-read bodies only (fewer live registers, more carry ripples)
#endif
constexpr int NPL = TCMI_P_NPL;                 // counter planes per vector
constexpr int NVEC = 3;                         // lo, hi, lo & hi
constexpr int NREG = NVEC * 8;                  // byte-counter registers per lane after the spread
constexpr int CPL = (MAXPOS + FB - 1) / FB;     // coverage entries per lane in the final prefix sum
constexpr int HSLOTS = 576;                     // header slots (TCMI_P_SUB + the dummy); the buffer later holds the window counters
static_assert(TCMI_P_SUB <= 2 * FB && TCMI_P_SUB < HSLOTS && HSLOTS * 8 >= NVEC * MAXPOS * 2, "s_hdr doubles as the 16-bit window counters");
static_assert(NREG * FB <= TCMI_F_SEQCAP, "slice partials must fit the stage buffer");
static_assert(4 * (FB / 2) <= HSLOTS - 3, "a lone four-read body of the widest slice layout must stay inside the header array");
static_assert(NLD * FB * 4 <= TCMI_F_SEQCAP, "the unconditional stage stores must fit the stage buffer");

// carry-save adder on bit vectors: sum and carry of three inputs (one v_bitop3_b32 each on gfx950;
// truth table: bit i of the immediate = f(a = i >> 2 & 1, b = i >> 1 & 1, c = i & 1))
#define TCMI_XOR3(a_, b_, c_) __builtin_amdgcn_bitop3_b32((a_), (b_), (c_), 0x96)
#define TCMI_MAJ(a_, b_, c_) __builtin_amdgcn_bitop3_b32((a_), (b_), (c_), 0xE8)

struct Planes {                                 // one bit-sliced counter per bit position: value = sum p[k] << k
    uint32_t p[NPL];
};

// fold eight bit vectors into the counter: seven carry-save adders, then the eights carry ripples upward
__device__ inline void add8(Planes &c, const uint32_t (&x)[8])
{
    const uint32_t s1 = TCMI_XOR3(c.p[0], x[0], x[1]), c1 = TCMI_MAJ(c.p[0], x[0], x[1]);
    const uint32_t s2 = TCMI_XOR3(s1, x[2], x[3]), c2 = TCMI_MAJ(s1, x[2], x[3]);
    const uint32_t s3 = TCMI_XOR3(s2, x[4], x[5]), c3 = TCMI_MAJ(s2, x[4], x[5]);
    c.p[0] = TCMI_XOR3(s3, x[6], x[7]);
    const uint32_t c4 = TCMI_MAJ(s3, x[6], x[7]);
    const uint32_t t1 = TCMI_XOR3(c.p[1], c1, c2), d1 = TCMI_MAJ(c.p[1], c1, c2);
    c.p[1] = TCMI_XOR3(t1, c3, c4);
    const uint32_t d2 = TCMI_MAJ(t1, c3, c4);
    uint32_t e = TCMI_MAJ(c.p[2], d1, d2);
    c.p[2] = TCMI_XOR3(c.p[2], d1, d2);
#pragma unroll
    for (int k = 3; k < NPL - 1; ++k) {
        const uint32_t t = c.p[k] & e;
        c.p[k] ^= e;
        e = t;
    }
    c.p[NPL - 1] ^= e;
}

// fold four bit vectors into the counter (the remainder of a stage)
__device__ inline void add4(Planes &c, const uint32_t (&x)[4])
{
    const uint32_t s1 = TCMI_XOR3(c.p[0], x[0], x[1]), c1 = TCMI_MAJ(c.p[0], x[0], x[1]);
    c.p[0] = TCMI_XOR3(s1, x[2], x[3]);
    const uint32_t c2 = TCMI_MAJ(s1, x[2], x[3]);
    uint32_t e = TCMI_MAJ(c.p[1], c1, c2);
    c.p[1] = TCMI_XOR3(c.p[1], c1, c2);
#pragma unroll
    for (int k = 2; k < NPL - 1; ++k) {
        const uint32_t t = c.p[k] & e;
        c.p[k] ^= e;
        e = t;
    }
    c.p[NPL - 1] ^= e;
}

// byte i of the result = count at bit position J + 8 i, from the planes [0, NP) (the others are known to be zero):
// one shift and one v_and_or_b32 per plane
template <int J, int NP>
__device__ inline uint32_t spread(const Planes &c)
{
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const uint32_t m = 0x01010101u << k;
        const uint32_t v = J >= k ? (c.p[k] >> (J - k)) : (c.p[k] << (k - J));
        r = __builtin_amdgcn_bitop3_b32(v, m, r, 0xEA);     // (v & m) | r
    }
    return r;
}

// the three bit-sliced counters of a lane -> 24 registers of four byte counters, stored [register][lane]
template <int NP>
__device__ inline void spread_all(const Planes (&cnt)[NVEC], uint32_t *s_part, int tid)
{
#pragma unroll
    for (int v = 0; v < NVEC; ++v) {
        s_part[(v * 8 + 0) * FB + tid] = spread<0, NP>(cnt[v]);
        s_part[(v * 8 + 1) * FB + tid] = spread<1, NP>(cnt[v]);
        s_part[(v * 8 + 2) * FB + tid] = spread<2, NP>(cnt[v]);
        s_part[(v * 8 + 3) * FB + tid] = spread<3, NP>(cnt[v]);
        s_part[(v * 8 + 4) * FB + tid] = spread<4, NP>(cnt[v]);
        s_part[(v * 8 + 5) * FB + tid] = spread<5, NP>(cnt[v]);
        s_part[(v * 8 + 6) * FB + tid] = spread<6, NP>(cnt[v]);
        s_part[(v * 8 + 7) * FB + tid] = spread<7, NP>(cnt[v]);
    }
}

__global__ __launch_bounds__(FB, TCMI_P_WAVES) void tally_planes_kernel(FastArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_seq[TCMI_F_SEQCAP];   // staged planes; later the slice partials
    __shared__ __attribute__((aligned(8))) uint2 s_hdr[HSLOTS];              // {pos - P0 | pairs << 16, byte offset in s_seq}
    __shared__ int32_t s_cov[MAXPOS + 8];                                     // coverage difference array
    __shared__ int s_scan[FB / 64];

    const int tid = threadIdx.x;
    if ((int)blockIdx.x < a.n_call2) {           // ride-along call of an earlier step's matrix (first in the grid: done early)
        call_other_tile(a, (int)blockIdx.x);
        return;
    }
    // then the tail blocks (event words) — in FRONT of the chunk blocks: behind them they started only when chunk blocks had left
    // (the chunk blocks of a 1M-read BAM fill every slot of the chip) and ran on their own at the launch's end
    const int b1 = (int)blockIdx.x - a.n_call2;
    if (b1 < a.n_tail) {
        tally_tail_block(a, b1, a.dev_counts ? (int64_t)min(a.dev_counts[1], (uint32_t)a.n_events) : a.n_events);
        return;
    }
    // A chunk block takes the chunks b, b + n_chunk_blocks, ..: when the counts are still on the device (the one-sync file path) the
    // grid is sized from the resident slots, not from the packer's CAPACITY — 5 700 blocks for the ~1 100 chunks of a 1M-read BAM, each
    // of the idle ones a trip to memory for the count while it held a slot (LDS and registers) that a chunk block was waiting for.
    const int n_real = a.dev_counts ? (int)min(a.dev_counts[0], (uint32_t)a.n_chunks) : a.n_chunks;
    for (int bid = b1 - a.n_tail; bid < n_real; bid += a.n_chunk_blocks) {
    const tcmi_fast_chunk *chp = a.chunks + bid;
    const int64_t read0 = chp->read0, word0 = chp->word0;
    const int n_reads = chp->n_reads, P0 = chp->P0, Wn = chp->Wn, sub_reads = chp->sub_reads;
    const int npos = Wn * 8;
    const int Gn = max(2, (npos + 31) >> 5);    // lane groups of 32 positions (at least two: S <= 128 keeps a body's four
                                                // header slots of a lane inside the header array)
    const int S = FB / Gn;                      // depth slices
    const int s = tid / Gn, gi = tid - s * Gn;
    const int base32p = gi * 32 + 32;           // first owned position relative to P0, + 32
    const int n_stage = (n_reads + sub_reads - 1) / sub_reads;

    for (int i = tid; i <= npos; i += FB) s_cov[i] = 0;
    // the last three header slots: a dummy read far to the right (no pairs: every lane is outside it) for the lanes
    // beyond the last depth slice and for the unused slots of a short stage, and 16 bytes of zeros that a lane outside
    // a read loads instead of the read's pairs
    if (tid < 3) s_hdr[HSLOTS - 3 + tid] = make_uint2(tid == 0 ? 0x7FFFu : 0u, 0u);
    const int zero_off = (int)(reinterpret_cast<const char *>(&s_hdr[HSLOTS - 2]) - reinterpret_cast<const char *>(s_seq));
    const int hb_first = s < S ? s * 8 : (HSLOTS - 3) * 8;       // byte offset of the lane's first header of a stage
    const int hb_step = s < S ? S * 8 : 0;
    __syncthreads();                            // before any wave adds coverage runs into it

    Planes cnt[NVEC];
#pragma unroll
    for (int v = 0; v < NVEC; ++v) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) cnt[v].p[k] = 0;
    }

    // ---- prefetch registers: the next stage's headers (two slots per lane) and planes --------------
    uint32_t h_lo0 = 0, h_lo1 = 0;              // packed headers: position - P0 | len << 10 | pair offset in the stage << 20
    uint4 pre0 = {}, pre1 = {}, pre2 = {}, pre3 = {}, pre4 = {}, pre5 = {};
    int st_begin = 0, st_end = chp->stage_end[0];   // word range of the stage (from word0)
    int st_end_next = chp->stage_end[1];            // fetched one stage ahead (a scalar load: its round trip hides under a stage)
    // Uniform base pointers + 32-bit lane offsets: the loads take the scalar-base form (no 64-bit address math
    // per lane).  Every lane loads (indices clamped into the stage): no exec-masked branch, so the loads stay in
    // flight across the inner loop.  (A macro, not a lambda: a closure kept the registers in scratch memory.)
    const uint32_t *lenoff_base = a.lenoff + read0;
    const uint32_t *seq_base = a.seq + word0;
#define TCMI_ISSUE_STAGE(stage_, begin_, end_)                                                        \
    do {                                                                                              \
        const uint32_t r0_ = (uint32_t)min((stage_) * sub_reads + tid, n_reads - 1);                  \
        const uint32_t r1_ = (uint32_t)min((stage_) * sub_reads + FB + tid, n_reads - 1);             \
        h_lo0 = lenoff_base[r0_];                                                                     \
        h_lo1 = lenoff_base[r1_];                                                                     \
        const int mis_ = (begin_) & 3; /* keep the 16-byte loads aligned (word0 is a multiple of 4) */ \
        const uint4 *src_ = reinterpret_cast<const uint4 *>(seq_base + ((begin_) - mis_));            \
        const uint32_t last_ = (uint32_t)(((end_) - (begin_) + mis_ + 3) / 4 - 1);                    \
        pre0 = src_[min((uint32_t)(0 * FB + tid), last_)];                                            \
        pre1 = src_[min((uint32_t)(1 * FB + tid), last_)];                                            \
        pre2 = src_[min((uint32_t)(2 * FB + tid), last_)];                                            \
        pre3 = src_[min((uint32_t)(3 * FB + tid), last_)];                                            \
        pre4 = src_[min((uint32_t)(4 * FB + tid), last_)];                                            \
        pre5 = src_[min((uint32_t)(5 * FB + tid), last_)];                                            \
    } while (0)
    static_assert(NLD == 6, "six 16-byte loads per lane cover a stage");
    TCMI_ISSUE_STAGE(0, st_begin, st_end);
    // coverage: the packer lists the chunk's reads as runs of equal (position, length) — a few dozen words for a few
    // thousand reads of a sorted BAM; each becomes a (+n, -n) pair in the difference array (prefix-summed at the end)
    {
        const uint32_t *runs = a.covrun + chp->run0;
        const int n_runs = chp->n_runs;
        for (int i = tid; i < n_runs; i += FB) {
            const uint32_t w = runs[i];
            const int rel = (int)(w & 1023u), len = (int)((w >> 10) & 1023u), n = (int)(w >> 20);
            atomicAdd(&s_cov[rel], n);
            atomicAdd(&s_cov[rel + len], -n);
        }
    }

    for (int stage = 0; stage < n_stage; ++stage) {
        const int ns = min(sub_reads, n_reads - stage * sub_reads);
        const int mis = st_begin & 3;
        // ---- A: headers, coverage runs and planes of this stage -> LDS ------------------------------
        const bool valid0 = tid < ns, valid1 = tid + FB < ns;
        // header slots up to the end of the stage's last inner-loop body: real reads, then dummies
        const int Rs = (ns + S - 1) / S;
        const int k_end = Rs <= 4 ? 4 : Rs <= 8 ? 8 : (Rs + 3) & ~3;   // bodies: 8, 8, ..., then 4 (mirrors the loop below)
        const int pad_end = k_end * S;
        {
            uint2 h0 = make_uint2(0x7FFFu, 0u), h1 = h0;
            if (valid0) {
                const int rel0 = (int)(h_lo0 & 1023u), len0 = (int)((h_lo0 >> 10) & 1023u);
                const int off = (int)(h_lo0 >> 20) * 2 + mis;          // word index of the read in s_seq (even)
                h0 = make_uint2((uint32_t)rel0 | ((uint32_t)(len0 + 31) >> 5) << 16, (uint32_t)(off - 2) * 4u);
            }
            if (valid1) {
                const int rel1 = (int)(h_lo1 & 1023u), len1 = (int)((h_lo1 >> 10) & 1023u);
                const int off = (int)(h_lo1 >> 20) * 2 + mis;
                h1 = make_uint2((uint32_t)rel1 | ((uint32_t)(len1 + 31) >> 5) << 16, (uint32_t)(off - 2) * 4u);
            }
            if (tid < pad_end) s_hdr[tid] = h0;
            if (tid + FB < pad_end) s_hdr[tid + FB] = h1;
        }
        {   // all six stores, whatever the stage's length: the loads were clamped into the stage, the buffer holds
            // 6 * 256 * 16 bytes, and nothing reads past the stage's last zero pair
            uint4 *dst = reinterpret_cast<uint4 *>(s_seq);
            {
                dst[0 * FB + tid] = pre0;
                dst[1 * FB + tid] = pre1;
                dst[2 * FB + tid] = pre2;
                dst[3 * FB + tid] = pre3;
                dst[4 * FB + tid] = pre4;
                dst[5 * FB + tid] = pre5;
            }
        }
        // ---- B: issue the next stage's loads at once — in front of the barrier, so that this workgroup has loads in
        //      flight while it waits there (the LDS stores above have read their registers); they complete while C runs
        if (stage + 1 < n_stage) {
            st_begin = st_end - 2;                               // the zero pair behind the last read comes along
            st_end = st_end_next;
            st_end_next = chp->stage_end[min(stage + 2, TCMI_F_MAXSTAGE - 1)];
            TCMI_ISSUE_STAGE(stage + 1, st_begin, st_end);
        }
        __syncthreads();
        // ---- C: this lane's slice of the staged reads: r = s, s + S, s + 2S, ...  Branch-free bodies of eight
        //      reads, then at most one body of four (a stage holds S * 4 * m reads); the slots past the stage's reads
        //      hold dummy headers.
        const int Rc = Rs;
        int hb = hb_first;                                       // byte offset of the lane's next header
#define TCMI_FETCH4(lo_, hi_, both_, at_)                                                                         \
    do {                                                                                                          \
        uint2 h_[4];                                                                                              \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                           \
            h_[u] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(s_hdr) + hb);                 \
            hb += hb_step;                                                                                        \
        }                                                                                                         \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                           \
            /* d = first owned position relative to the read start; t = pair holding it, + 1 */                    \
            const int dp_ = base32p - (int)(h_[u].x & 0xFFFFu);  /* d + 32 */                                       \
            const int t_ = dp_ >> 5;                                                                              \
            /* one zero pair lies on either side of a read: pairs t - 1 and t are loaded for 0 <= t <= pairs; a   \
               lane further out (clamped index) is outside the read altogether and loads the 16 zero bytes */       \
            const int tc_ = max(0, min(t_, (int)(h_[u].x >> 16)));                                                \
            const int at_b_ = tc_ == t_ ? (int)h_[u].y + tc_ * 8 : zero_off;                                      \
            const uint2 *wp_ = reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(s_seq) + at_b_);    \
            const uint2 w0_ = wp_[0], w1_ = wp_[1];             /* {lo, hi} of pairs t - 1 and t */                 \
            lo_[(at_) + u] = __builtin_amdgcn_alignbit(w1_.x, w0_.x, (uint32_t)dp_);   /* bits [4:0] = d mod 32 */  \
            hi_[(at_) + u] = __builtin_amdgcn_alignbit(w1_.y, w0_.y, (uint32_t)dp_);                               \
            both_[(at_) + u] = lo_[(at_) + u] & hi_[(at_) + u];                                                   \
        }                                                                                                         \
    } while (0)
        int k = 0;
        for (; TCMI_P_BODY8 && Rc - k > 4; k += 8) {
            uint32_t lo[8], hi[8], both[8];
            TCMI_FETCH4(lo, hi, both, 0);
            TCMI_FETCH4(lo, hi, both, 4);
            add8(cnt[0], lo);
            add8(cnt[1], hi);
            add8(cnt[2], both);
        }
        for (; k < Rc; k += 4) {
            uint32_t lo[4], hi[4], both[4];
            TCMI_FETCH4(lo, hi, both, 0);
            add4(cnt[0], lo);
            add4(cnt[1], hi);
            add4(cnt[2], both);
        }
#undef TCMI_FETCH4
        __syncthreads();                                        // every lane is done with this stage's LDS
    }
    // ---- planes -> byte counters -> LDS, layout [register j][lane] (conflict-free both ways) ---------
    uint32_t *s_part = s_seq;
    uint16_t (*s_fin)[MAXPOS] = reinterpret_cast<uint16_t (*)[MAXPOS]>(s_hdr);   // window counters of lo, hi, lo&hi
    {
        const int per_lane = n_stage * ((sub_reads + S - 1) / S);                  // bound on the reads per lane (dummies count nothing)
        // planes that can be non-zero: one uniform branch, then straight-line code (few reads per lane leave the top
        // planes empty)
        if (per_lane < 32) spread_all<5>(cnt, s_part, tid);
        else if (per_lane < 64) spread_all<6>(cnt, s_part, tid);
        else if (per_lane < 128) spread_all<(NPL < 7 ? NPL : 7)>(cnt, s_part, tid);
        else spread_all<NPL>(cnt, s_part, tid);
    }
    __syncthreads();
    // ---- sum the slices; register j of group g holds 4 positions (j%8 + 8 i) of one vector ------------
    for (int item = tid; item < Gn * NREG; item += FB) {
        const int j = item / Gn, g = item - j * Gn;
        uint32_t e = 0, o = 0;                                  // bytes 0,2 and bytes 1,3 as 16-bit sums
        const uint32_t *row = s_part + j * FB + g;
        for (int t = 0; t < S; t += 4) {                        // four independent LDS loads in flight
            const uint32_t v0 = row[t * Gn];
            const uint32_t v1 = t + 1 < S ? row[(t + 1) * Gn] : 0u;
            const uint32_t v2 = t + 2 < S ? row[(t + 2) * Gn] : 0u;
            const uint32_t v3 = t + 3 < S ? row[(t + 3) * Gn] : 0u;
            e += (v0 & 0x00FF00FFu) + (v1 & 0x00FF00FFu) + (v2 & 0x00FF00FFu) + (v3 & 0x00FF00FFu);
            o += ((v0 >> 8) & 0x00FF00FFu) + ((v1 >> 8) & 0x00FF00FFu) + ((v2 >> 8) & 0x00FF00FFu) + ((v3 >> 8) & 0x00FF00FFu);
        }
        const int v = j >> 3;
        const int p = g * 32 + (j & 7);                         // byte i of the register <-> position p + 8 i
        uint16_t *f = &s_fin[v][p];
        if (p < npos) f[0] = (uint16_t)(e & 0xFFFFu);
        if (p + 8 < npos) f[8] = (uint16_t)(o & 0xFFFFu);
        if (p + 16 < npos) f[16] = (uint16_t)(e >> 16);
        if (p + 24 < npos) f[24] = (uint16_t)(o >> 16);
    }
    // ---- coverage: inclusive prefix sum of the difference array, CPL entries per lane ----------------
    {
        const int i0 = tid * CPL;
        int d[CPL], sum = 0;
#pragma unroll
        for (int k = 0; k < CPL; ++k) { d[k] = i0 + k < npos ? s_cov[i0 + k] : 0; sum += d[k]; }
        int run = block_scan_incl(sum, s_scan) - sum;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            run += d[k];
            if (i0 + k < npos) s_cov[i0 + k] = run;
        }
    }
    __syncthreads();
    // ---- global atomics: coverage, C, G, T of TWO adjacent positions per 64-bit add (the columns never go
    //      negative and never carry out of 32 bits), A one position at a time (the tail blocks subtract from it, so
    //      it may be transiently negative and a carry would spill into the neighbour) ---------------------------
#ifdef TCMI_TALLY_NO_ATOMICS                        // (diagnostic build: the kernel without its adds to the matrix — how much of its time they are)
    if (a.L < 0) {
#else
    if (a.pair_ok) {
#endif
        for (int p = 2 * tid; p < npos; p += 2 * FB) {
            const int gp = P0 + p;                                  // even: P0 is a multiple of 8
            if (gp >= a.L) continue;
            const bool two = gp + 1 < a.L;                          // (npos is a multiple of 8: p + 1 is inside the window)
            const int cv0 = s_cov[p], cv1 = two ? s_cov[p + 1] : 0;
            if ((cv0 | cv1) == 0) continue;
            const int nT0 = s_fin[2][p], nC0 = s_fin[0][p] - nT0, nG0 = s_fin[1][p] - nT0;
            const int nT1 = two ? s_fin[2][p + 1] : 0, nC1 = two ? s_fin[0][p + 1] - nT1 : 0, nG1 = two ? s_fin[1][p + 1] - nT1 : 0;
            const int nA0 = cv0 - nC0 - nG0 - nT0, nA1 = cv1 - nC1 - nG1 - nT1;   // include the class-less positions (tail blocks)
            auto add2 = [&](int col, int v0, int v1) {
                if (v0 | v1)
                    atomicAdd(reinterpret_cast<unsigned long long *>(&a.counts[(int64_t)col * a.ld + gp]),
                              (unsigned long long)(uint32_t)v0 | ((unsigned long long)(uint32_t)v1 << 32));
            };
            add2(TCMI_COV, cv0, cv1);
            add2(TCMI_C, nC0, nC1);
            add2(TCMI_G, nG0, nG1);
            add2(TCMI_T, nT0, nT1);
            if (nA0) atomicAdd(&a.counts[(int64_t)TCMI_A * a.ld + gp], nA0);
            if (nA1) atomicAdd(&a.counts[(int64_t)TCMI_A * a.ld + gp + 1], nA1);
        }
#ifdef TCMI_TALLY_NO_ATOMICS
    } else if (a.L < 0) {
#else
    } else {
#endif
        for (int p = tid; p < npos; p += FB) {
            const int gp = P0 + p;
            if (gp >= a.L) continue;
            const int cv = s_cov[p];
            if (cv == 0) continue;
            const int nT = s_fin[2][p], nC = s_fin[0][p] - nT, nG = s_fin[1][p] - nT;
            const int nA = cv - nC - nG - nT;                       // includes the class-less positions, taken out by the tail blocks
            atomicAdd(&a.counts[(int64_t)TCMI_COV * a.ld + gp], cv);
            if (nA) atomicAdd(&a.counts[(int64_t)TCMI_A * a.ld + gp], nA);
            if (nC) atomicAdd(&a.counts[(int64_t)TCMI_C * a.ld + gp], nC);
            if (nG) atomicAdd(&a.counts[(int64_t)TCMI_G * a.ld + gp], nG);
            if (nT) atomicAdd(&a.counts[(int64_t)TCMI_T * a.ld + gp], nT);
        }
    }
    __syncthreads();                            // (the next chunk's set-up writes what the adds above read)
    }
}

#undef TCMI_ISSUE_STAGE
#undef TCMI_XOR3
#undef TCMI_MAJ

} // namespace



// Launch over the aligned set of a read set: [ride-along call blocks][tail blocks: event words][chunk blocks].
int tcmi_launch_tally_fast(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld, int32_t *d_counts)
{
    FastArgs a = {};
    a.lenoff = rs->d_flenoff; a.seq = rs->d_fseq; a.chunks = rs->d_fchunk; a.events = rs->d_fevent; a.covrun = rs->d_fcovrun;
    a.counts = d_counts; a.ld = ld; a.n_events = rs->f_events; a.L = (int32_t)L;
    a.pair_ok = (ld % 2 == 0) && (reinterpret_cast<uintptr_t>(d_counts) % 8 == 0);
    a.n_tail = (int32_t)((rs->f_events + FB - 1) / FB);
    int64_t n_chunk_blocks = rs->f_chunks;
    if (rs->d_dev_counts) {                                     // totals still on the device: f_chunks / f_events are the capacities
        a.dev_counts = rs->d_dev_counts;
        a.n_tail = (int32_t)std::min<int64_t>(a.n_tail, 128);
        // (one round of the chip's slots and a half: pk_pack cuts a file into about one chunk per slot; a file with more takes turns)
        n_chunk_blocks = std::min<int64_t>(n_chunk_blocks, (int64_t)ctx->n_cu * ctx->wg_per_cu * 3 / 2);
    }
    a.n_chunk_blocks = (int32_t)std::max<int64_t>(1, n_chunk_blocks);
    int64_t grid = (rs->f_chunks ? n_chunk_blocks : 0) + a.n_tail;
    if (ctx->ride && !ctx->ride->taken) {                     // carry another workspace's call in this launch
        tcmi_ride *r = ctx->ride;
        a.counts2 = r->counts; a.ld2 = r->ld; a.L2 = (int32_t)r->L; a.n_call2 = (int32_t)((r->L + TILE - 1) / TILE);
        a.mincov = r->mincov; a.include_ambig = r->amb; a.plain = r->plain; a.alt = r->alt; a.flags = r->flags;
        grid += a.n_call2;
        r->taken = true;
    }
    if (grid > INT32_MAX) return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "too many chunks");
    if (grid == 0) return TCMI_OK;
    a.n_chunks = (int32_t)rs->f_chunks;
    tcmi_prof_begin(ctx, TCMI_K_TALLY);
    (void)hipGetLastError();                                   // drop any stale error of this thread
    hipLaunchKernelGGL(tally_planes_kernel, dim3((unsigned)grid), dim3(FB), 0, ctx->stream, a);
    tcmi_prof_end(ctx, TCMI_K_TALLY);
    TCMI_HIP(ctx, hipGetLastError());
    return TCMI_OK;
}
