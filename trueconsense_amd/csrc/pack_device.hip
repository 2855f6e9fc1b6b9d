// pack_device.hip — DEVICE: BAM-native reads -> the bit-plane layout tally_planes.hip consumes.
//
// What the reference does per pileup token in Python (indexing.py:100-139: pileup membership, the token of every
// covered position, parse_query_sequences' classification) is decided here per READ by HIP kernels, straight from
// the arrays a BAM holds (SAM spec §4.2: pos, flag, l_seq, CIGAR words, 4-bit SEQ) — either as the flat arrays of
// struct tcmi_reads copied to the device as they are, or from the inflated BAM byte stream itself (bam_device.hip):
//
//   pk_classify   one lane per read: does it pile up, reference span, CIGAR shape ([H][S]M[S][H] reads
//                 are taken as they are, anything else is projected onto the reference), words it will occupy;
//                 per-workgroup sums for the scan
//   pk_scan       exclusive scan of the per-workgroup sums (one workgroup)
//   pk_scatter    compacted index of every kept read + its word offset (block scan + the scanned sums)
//   pk_pack       one workgroup per run of consecutive kept reads: cuts it into chunks (window <= 768 positions,
//                 <= 255 reads per lane, <= 8 stages that fill the tally kernel's LDS stage buffer), writes the chunk
//                 records, ONE packed header word per read, the coverage runs (reads of equal position and length),
//                 and per read the bases as {lo, hi} bit planes: 4-bit codes are classified eight at a time with
//                 SWAR bit tricks (one-hot test, C|T and G|T planes, 3-step bit squeeze), CIGAR ops are walked by the
//                 read's lane (M/=/X copy bit fields, D -> X events, insertions -> I events on the base before,
//                 every covered position without an A/C/G/T base -> OTHER event)
//
// HBM-streaming byte / bit work: no MFMA.  Input 91 B + 20 B of offsets per 150-bp read, output 52 B.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tally_common.h"
#include "bgzf_device.h"

namespace {

constexpr int PB = 256;                          // lanes per workgroup of every kernel here
constexpr int PK_CMAX = 1024;                    // most reads one pk_pack workgroup takes
constexpr uint32_t NIB = 0x11111111u;

using PackSrc = tcmi_pack_src;

struct ReadView {
    int32_t tid, pos, l_seq;
    uint32_t flag, n_cigar;
    const uint8_t *cigar;       // n_cigar little-endian words, not necessarily aligned
    const uint8_t *seq;         // ceil(l_seq / 2) bytes
    bool bad;                   // inconsistent offsets / lengths
    bool broken;                // ... of a BAM record (any record, mapped or not: the file is not a BAM file then)
};

__device__ inline uint32_t ld_u32(const uint8_t *p)
{
    uint32_t w;                                 // (the record fields of a BAM stream sit at any byte offset: one unaligned dword load)
    __builtin_memcpy(&w, p, 4);
    return w;
}

// a record of the inflated BAM stream, `rec` at its block_size field (any byte address)
__device__ inline ReadView view_rec(const uint8_t *rec)
{
    ReadView v;
    v.bad = false;
    v.broken = false;
    const uint8_t *r = rec + 4;                             // behind block_size
    // the fixed fields in two loads at the record's own (any) byte address — unaligned access mode; twelve aligned dword loads
    // and funnel shifts kept the kernel waiting on the address unit: every lane's record lies in a cache line of its own
    uint32_t h[6];                                         // block_size, refID, pos, l_read_name|mapq|bin, n_cigar_op|flag, l_seq
    __builtin_memcpy(h, r - 4, 16);
    __builtin_memcpy(h + 4, r + 12, 8);
    v.tid = (int32_t)h[1];
    v.pos = (int32_t)h[2];
    const uint32_t w2 = h[3], w3 = h[4];
    const uint32_t l_name = w2 & 0xFFu;
    v.n_cigar = w3 & 0xFFFFu;
    v.flag = w3 >> 16;
    v.l_seq = (int32_t)h[5];
    // The record walk only checked block_size itself: the variable-length fields must fit into it (what bam_reader.cpp's
    // "alignment record fields overrun block_size" refuses) — a forged l_seq or n_cigar_op would otherwise send the kernels
    // that follow the CIGAR and the bases far behind the record, or behind the stream.
    const uint32_t block_size = h[0];
    const uint64_t need = 32ull + l_name + 4ull * v.n_cigar + ((uint64_t)(uint32_t)v.l_seq + 1) / 2 + (uint64_t)(uint32_t)v.l_seq;
    v.bad = v.l_seq < 0 || l_name == 0 || need > block_size;
    v.broken = v.bad;
    if (v.bad) v.n_cigar = 0;
    v.cigar = r + 32 + l_name;
    v.seq = v.cigar + 4 * (size_t)v.n_cigar;
    return v;
}

__device__ inline ReadView view(const PackSrc &s, int64_t i)
{
    ReadView v;
    v.bad = false;
    v.broken = false;
    if (s.mode == 0) {
        v.tid = s.tid ? s.tid[i] : 0;
        v.pos = s.pos[i];
        v.l_seq = s.l_qseq[i];
        v.flag = s.flag[i];
        const uint64_t c0 = s.cigar_off[i], c1 = s.cigar_off[i + 1], q0 = s.seq_off[i], q1 = s.seq_off[i + 1];
        v.bad = c1 < c0 || c1 - c0 > 65535u || q1 < q0 || v.l_seq < 0 || (int64_t)(q1 - q0) < ((int64_t)v.l_seq + 1) / 2;
        v.n_cigar = v.bad ? 0u : (uint32_t)(c1 - c0);
        v.cigar = reinterpret_cast<const uint8_t *>(s.cigar + c0);
        v.seq = s.seq + q0;
    } else v = view_rec(s.stream + s.rec_off[i]);
    return v;
}

__device__ inline uint32_t nib_at(const uint8_t *seq, int32_t q)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(seq + (q >> 1));
    const uint32_t w = *reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3) >> ((a & 3) * 8);
    return (q & 1) ? (w & 15u) : ((w >> 4) & 15u);
}
__device__ inline uint32_t byte_at(const uint8_t *p)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    return (*reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3) >> ((a & 3) * 8)) & 0xFFu;
}

// does the record carry a CG:B aux field (the real CIGAR of a read with more than 65 535 operations, SAM spec §4.2.2)?
__device__ inline bool has_cg_tag(const uint8_t *aux, const uint8_t *end)
{
    for (int guard = 0; guard < 4096 && aux + 3 <= end; ++guard) {
        const uint32_t t0 = byte_at(aux), t1 = byte_at(aux + 1), ty = byte_at(aux + 2);
        if (t0 == 'C' && t1 == 'G' && ty == 'B') return true;
        aux += 3;
        if (ty == 'A' || ty == 'c' || ty == 'C') aux += 1;
        else if (ty == 's' || ty == 'S') aux += 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') aux += 4;
        else if (ty == 'Z' || ty == 'H') { while (aux < end && byte_at(aux)) ++aux; ++aux; }
        else if (ty == 'B' && aux + 5 <= end) {
            const uint32_t st = byte_at(aux), cnt = ld_u32(aux + 1);
            const uint32_t sz = (st == 'c' || st == 'C') ? 1u : (st == 's' || st == 'S') ? 2u : 4u;
            if (cnt > (1u << 28)) return false;
            aux += 5 + (size_t)cnt * sz;
        } else return false;
    }
    return false;
}

__device__ inline bool consumes_ref(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
__device__ inline bool is_match(uint32_t op) { return op == 0 || op == 7 || op == 8; }

// per-read word of pk_classify: len (10 bits, <= TCMI_D_MAXLEN) | projected << 10 | kept << 11 | y0 << 12
constexpr uint32_t INFO_PROJ = 1u << 10, INFO_KEPT = 1u << 11;
// flags raised for the host
enum { PKF_LONG = 1, PKF_FARPOS = 2, PKF_BADREAD = 4, PKF_EVENT_OVF = 8, PKF_CHUNK_OVF = 16, PKF_HEADER_OVF = 32, PKF_WORD_OVF = 64,
       PKF_MULTIREF = 128 };

struct PackTotals {                 // device scalars, copied back to the host
    unsigned long long alg_bytes;
    unsigned long long n_kept, n_words;
    int32_t max_end;
    uint32_t flags;
    uint32_t n_chunks, n_events, n_runs;
    uint32_t word_cursor;
    uint32_t max_len;               // longest reference span of a kept read
    uint32_t n_gen;                 // reads left to the stream-walking tally kernel (longer than TCMI_D_MAXLEN positions)
    unsigned long long n_rec;       // pk_place: alignment records of the decoded range
    // pk_place, a block range: where its first record starts / where the first record behind it starts, as offsets into the range's
    // stream + 1 (0: no record starts in the range / the chain was never fixed) — tcmi_readset_range_anchors
    unsigned long long range_first, range_next;
};

// words a read takes in the plane stream: its pairs, the zero pair behind them, and — for an even number of pairs — one more zero pair,
// so that every read ends on a 16-byte boundary: the reads then lie in ONE contiguous run (a read's place is 2 + the scanned sum of
// the words in front of it, known before any chunk is cut), and the zero pair that closes a read is the one a chunk that starts with
// the next read needs in front of it — 16-byte aligned, as the tally kernel's loads want a chunk's first word.
__device__ inline uint32_t words_of(uint32_t len) { return (2u * ((len + 31u) >> 5) + 2u + 3u) & ~3u; }

// inclusive scan of a pair over the workgroup (PB lanes)
__device__ inline uint2 block_scan2(uint2 v, uint2 *wave_tot /* LDS [PB / 64] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t ox = (uint32_t)__shfl_up((int)v.x, d, 64), oy = (uint32_t)__shfl_up((int)v.y, d, 64);
        if (lane >= d) { v.x += ox; v.y += oy; }
    }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < PB / 64; ++w)
        if (w < wave) { v.x += wave_tot[w].x; v.y += wave_tot[w].y; }
    return v;
}

// ---- 1: classify ----------------------------------------------------------------------------------------------
// what one read is to the packer: its per-read word (0: not kept), the words it takes in the plane stream, its SURVEY §8-d bytes,
// its end; `longread`: left to tally_stream_kernel (a device-decoded stream only; from flat arrays PKF_LONG is raised instead)
struct Classified { uint32_t word, nwords, len; unsigned long long alg; int32_t end; bool longread; };

__device__ inline Classified classify_view(const ReadView &v, int32_t mode, int32_t pos_shift, const uint8_t *rec, bool stream_long_ok, PackTotals *tot)
{
    Classified c = {0u, 0u, 0u, 0ull, 0, false};
    bool kept = !(v.flag & 0x4u) && v.tid == 0 && v.pos >= 0 && !v.bad;
    if (v.broken || (v.bad && !(v.flag & 0x4u) && v.tid == 0 && v.pos >= 0)) atomicOr(&tot->flags, (uint32_t)PKF_BADREAD);
    if (!(v.flag & 0x4u) && v.tid > 0) atomicOr(&tot->flags, (uint32_t)PKF_MULTIREF);   // (the host packer words the error)
    if (!kept) return c;
    // one walk over the CIGAR: reference span, and is it [H]*[S]* (M|=|X)+ [S]*[H]* ?
    int64_t span = 0, m = 0, y0 = 0;
    int ph = 0;                     // 0 start / leading H, 1 leading S, 2 match run, 3 trailing S, 4 trailing H
    bool simple = true;
    for (uint32_t k = 0; k < v.n_cigar; ++k) {
        const uint32_t cw = ld_u32(v.cigar + 4 * (size_t)k), op = cw & 0xFu, len = cw >> 4;
        if (consumes_ref(op)) span += len;
        if (op == 5) { if (ph >= 2) ph = 4; else if (ph == 1) simple = false; }
        else if (op == 4) { if (ph <= 1) { ph = 1; y0 += len; } else if (ph <= 3) ph = 3; else simple = false; }
        else if (is_match(op)) { if (ph <= 2) { ph = 2; m += len; } else simple = false; }
        else simple = false;
    }
    simple = simple && ph >= 2 && m > 0 && y0 < (1 << 20);
    if (mode == 1 && v.n_cigar == 2 && v.l_seq > 0) {   // <l_seq>S<n>N + a CG:B tag: the real CIGAR lives in the tag (SAM spec §4.2.2)
        const uint32_t c0 = ld_u32(v.cigar), c1 = ld_u32(v.cigar + 4);
        if ((c0 & 0xFu) == 4 && (c0 >> 4) == (uint32_t)v.l_seq && (c1 & 0xFu) == 3) {
            if (has_cg_tag(v.seq + ((size_t)v.l_seq + 1) / 2 + (size_t)v.l_seq, rec + 4 + ld_u32(rec)))
                atomicOr(&tot->flags, (uint32_t)PKF_BADREAD);   // (the host reader words the refusal)
        }
    }
    if (span <= 0) return c;
    const int64_t end = (int64_t)v.pos + pos_shift + span;
    const int64_t len = simple ? m : span;
    if (end >= (int64_t)TCMI_F_EVPOS) { atomicOr(&tot->flags, (uint32_t)PKF_FARPOS); return c; }
    c.alg = (unsigned long long)(12 + 4 * (int64_t)v.n_cigar + ((int64_t)v.l_seq + 1) / 2);
    c.end = (int32_t)end;
    if (len > TCMI_D_MAXLEN) {
        // A read that spans more positions than a chunk's window (long-read platforms).  From the flat arrays: the host
        // packer cuts it into pieces.  In a device-decoded stream: it stays out of the packed set and is walked where it
        // lies, CIGAR op by CIGAR op, by tally_stream_kernel (one wavefront per such read).
        if (mode == 1 && stream_long_ok) c.longread = true;
        else { atomicOr(&tot->flags, (uint32_t)PKF_LONG); c.alg = 0; c.end = 0; }
        return c;
    }
    c.word = (uint32_t)len | (simple ? 0u : INFO_PROJ) | INFO_KEPT | (simple ? (uint32_t)y0 << 12 : 0u);
    c.nwords = words_of((uint32_t)len);
    c.len = (uint32_t)len;
    return c;
}

__global__ __launch_bounds__(PB) void pk_classify(PackSrc s, uint32_t *info, uint2 *rd_seq, int32_t *rd_pos, uint2 *blk_sum, unsigned long long *blk_alg,
                                                  int32_t *blk_end, PackTotals *tot, uint32_t *gen_idx)
{
    __shared__ uint2 s_w[PB / 64];
    __shared__ unsigned long long s_alg[PB / 64];
    __shared__ int32_t s_end[PB / 64];
    const int64_t i = (int64_t)blockIdx.x * PB + threadIdx.x;
    uint32_t word = 0, nwords = 0;
    unsigned long long my_alg = 0;
    int32_t my_end = 0;
    uint32_t my_len = 0;
    if (i < s.n) {
        const ReadView v = view(s, i);
        const Classified c = classify_view(v, s.mode, s.pos_shift, s.mode == 1 ? s.stream + s.rec_off[i] : nullptr, gen_idx != nullptr, tot);
        if (c.longread) gen_idx[atomicAdd(&tot->n_gen, 1u)] = (uint32_t)i;
        word = c.word; nwords = c.nwords; my_alg = c.alg; my_end = c.end; my_len = c.len;
        info[i] = word;
        // where the read's SEQ starts (bytes from the stream's / the SEQ array's first byte; low word | high byte) and l_seq:
        // pk_pack goes straight there instead of chasing record offset -> header -> CIGAR -> SEQ through four dependent loads
        const unsigned long long so = (unsigned long long)(v.seq - (s.mode == 0 ? s.seq : s.stream));
        rd_seq[i] = make_uint2((uint32_t)so, ((uint32_t)(so >> 32) & 0xFFu) | ((uint32_t)min(v.l_seq, 0xFFFFFF) << 8));
        rd_pos[i] = v.pos;                      // (pk_scatter's copy: it need not go back to the record)
    }
    const uint2 incl = block_scan2(make_uint2(word ? 1u : 0u, nwords), s_w);
    if (threadIdx.x == PB - 1) blk_sum[blockIdx.x] = incl;
    // algorithmic bytes and extent: per-workgroup partials that pk_scan folds (same-address atomics serialise in L2:
    // 31 000 of them cost 0.39 ms)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        my_alg += (unsigned long long)__shfl_xor((long long)my_alg, d, 64);
        my_end = max(my_end, __shfl_xor(my_end, d, 64));
        my_len = max(my_len, (uint32_t)__shfl_xor((int)my_len, d, 64));
    }
    // (the longest span rides in the top 16 bits of the byte sum: a workgroup's reads are < 2^48 bytes)
    if ((threadIdx.x & 63) == 0) { s_alg[threadIdx.x >> 6] = my_alg | ((unsigned long long)my_len << 48); s_end[threadIdx.x >> 6] = my_end; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0;
        int32_t e = 0;
        unsigned long long ml = 0;
        for (int w = 0; w < PB / 64; ++w) { a += s_alg[w] & 0xFFFFFFFFFFFFull; ml = max(ml, s_alg[w] >> 48); e = max(e, s_end[w]); }
        blk_alg[blockIdx.x] = a | (ml << 48);
        blk_end[blockIdx.x] = e;
    }
}

// ---- 2: exclusive scan of the workgroup sums (one workgroup, any number of entries) ---------------------------------
__global__ __launch_bounds__(1024) void pk_scan(uint2 *blk_sum, const unsigned long long *blk_alg, const int32_t *blk_end, int64_t n_blk,
                                                PackTotals *tot)
{
    __shared__ unsigned long long s_x[16], s_y[16];
    const int t = threadIdx.x;
    const int64_t per = (n_blk + 1023) / 1024, b0 = t * per, b1 = min(b0 + per, n_blk);
    unsigned long long sx = 0, sy = 0, alg = 0;
    int32_t mend = 0;
    uint32_t mlen = 0;
    for (int64_t b = b0; b < b1; ++b) {
        sx += blk_sum[b].x; sy += blk_sum[b].y; alg += blk_alg[b] & 0xFFFFFFFFFFFFull; mlen = max(mlen, (uint32_t)(blk_alg[b] >> 48)); mend = max(mend, blk_end[b]);
    }
    if (alg) atomicAdd(&tot->alg_bytes, alg);                  // (at most 1024 of these)
    if (mlen) atomicMax(&tot->max_len, mlen);
    if (mend) atomicMax(&tot->max_end, mend);
    // inclusive scan of the 1024 partials: within the wavefronts by shuffles, then the sixteen wavefront totals (two barriers;
    // ten rounds of Hillis-Steele through LDS cost twenty, and this kernel is one workgroup's latency from end to end)
    unsigned long long ix = sx, iy = sy;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long ox = (unsigned long long)__shfl_up((long long)ix, d, 64), oy = (unsigned long long)__shfl_up((long long)iy, d, 64);
        if ((t & 63) >= d) { ix += ox; iy += oy; }
    }
    if ((t & 63) == 63) { s_x[t >> 6] = ix; s_y[t >> 6] = iy; }
    __syncthreads();
    for (int w = 0; w < (t >> 6); ++w) { ix += s_x[w]; iy += s_y[w]; }
    unsigned long long bx = ix - sx, by = iy - sy;          // exclusive base of this lane's range
    for (int64_t b = b0; b < b1; ++b) {
        const uint2 v = blk_sum[b];
        blk_sum[b] = make_uint2((uint32_t)bx, (uint32_t)by);
        bx += v.x; by += v.y;
    }
    if (t == 1023) { tot->n_kept = ix; tot->n_words = iy; tot->n_chunks = 0; tot->n_events = 0; tot->n_runs = 0; tot->word_cursor = 0; }   // (the counters pk_pack / pk_planes take from)
}

// ---- 3: scatter: compacted index + word offset of every kept read -----------------------------------------------------
__global__ __launch_bounds__(PB) void pk_scatter(PackSrc s, const uint32_t *info, const uint2 *rd_seq, const int32_t *rd_pos, const uint2 *blk_base, uint32_t *c_idx,
                                                 int32_t *c_pos, uint32_t *c_info, uint32_t *c_woff, uint2 *c_seq)
{
    __shared__ uint2 s_w[PB / 64];
    const int64_t i = (int64_t)blockIdx.x * PB + threadIdx.x;
    const uint32_t w = i < s.n ? info[i] : 0u;
    const uint2 mine = make_uint2(w ? 1u : 0u, w ? words_of(w & 1023u) : 0u);
    const uint2 incl = block_scan2(mine, s_w);
    if (w) {
        const uint2 base = blk_base[blockIdx.x];
        const uint32_t j = base.x + incl.x - 1u;
        c_idx[j] = (uint32_t)i;
        c_pos[j] = rd_pos[i] + s.pos_shift;
        c_info[j] = w;
        c_woff[j] = base.y + incl.y - mine.y;
        c_seq[j] = rd_seq[i];
    }
}

// ---- 4: pack ------------------------------------------------------------------------------------------------------------
struct PackOut {
    uint32_t *lenoff;               // [n_kept]
    uint32_t *seq;                  // [word_cap]
    tcmi_fast_chunk *chunks;        // [chunk_cap]
    uint32_t *covrun;               // [n_kept]: the runs of a chunk start at its first read's index
    uint32_t *events;               // [event_cap]
    uint32_t word_cap, chunk_cap, event_cap;
    uint32_t *slack;                // 64 words behind the last array
};

__device__ inline void push_event(const PackOut &o, PackTotals *tot, uint32_t w)
{
    const uint32_t slot = atomicAdd(&tot->n_events, 1u);
    if (slot < o.event_cap) o.events[slot] = w;
}

// Any boolean function of three words in one instruction (v_bitop3_b32): TT(f) is its truth table, bit (a << 2 | b << 1 | c).  These
// kernels are bound by vector issue, and the compiler finds only some of the three-input forms by itself.
template <typename F> constexpr uint32_t truth_table(F f)
{
    uint32_t t = 0;
    for (uint32_t i = 0; i < 8; ++i) t |= (f((i >> 2) & 1u, (i >> 1) & 1u, i & 1u) & 1u) << i;
    return t;
}
#define BITOP3(a_, b_, c_, ...) ((uint32_t)__builtin_amdgcn_bitop3_b32((a_), (b_), (c_), truth_table([](uint32_t a, uint32_t b, uint32_t c) { return (__VA_ARGS__); })))

// eight 4-bit BAM codes (base k in nibble k) -> one bit per base: C|T, G|T, "is one of A C G T"
__device__ inline void classify8(uint32_t n, uint32_t &lo, uint32_t &hi, uint32_t &ok)
{
    // (bits 0, 4, .. 28 of n, n >> 1, n >> 2, n >> 3 are a base's four code bits; the other bits are masked out of `ok` and so of all three)
    const uint32_t b = n >> 1, c = n >> 2, d = n >> 3;
    const uint32_t one3 = BITOP3(n, b, c, (a ^ b ^ c) & ~(a & b & c)), none3 = BITOP3(n, b, c, ~(a | b | c));
    ok = BITOP3(one3, none3, d, c ? b : a) & NIB;               // exactly one bit set: A=1 C=2 G=4 T=8
    lo = BITOP3(b, d, ok, (a | b) & c);
    hi = BITOP3(c, d, ok, (a | b) & c);
}
__device__ inline uint32_t squeeze8(uint32_t t)                 // bits 0,4,..,28 -> bits 0..7
{
    t = (t | (t >> 3)) & 0x03030303u;
    t = (t | (t >> 6)) & 0x000F000Fu;
    return (t | (t >> 12)) & 0xFFu;
}

// bases [yy, yy + nb) of a read (nb <= 32) as bit planes; bases at or beyond l_seq are "not A/C/G/T"
__device__ inline void fetch32(const uint8_t *seq, int32_t l_seq, int32_t yy, int nb, uint32_t &lo, uint32_t &hi, uint32_t &ok)
{
    lo = hi = ok = 0;
    const int have = min(nb, l_seq - yy);
    if (have <= 0) return;
    const uintptr_t a = reinterpret_cast<uintptr_t>(seq + (yy >> 1));
    const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3) * 8u;
    uint32_t d[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) d[k] = q[k];
    uint32_t w[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint32_t b = sh ? __builtin_amdgcn_alignbit(d[k + 1], d[k], sh) : d[k];
        w[k] = ((b & 0x0F0F0F0Fu) << 4) | ((b >> 4) & 0x0F0F0F0Fu);   // BAM keeps the first base of a byte in the high nibble
    }
    const bool odd = yy & 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t n = odd ? __builtin_amdgcn_alignbit(w[k + 1], w[k], 4) : w[k];
        uint32_t l, h, v;
        classify8(n, l, h, v);
        lo |= squeeze8(l) << (8 * k);
        hi |= squeeze8(h) << (8 * k);
        ok |= squeeze8(v) << (8 * k);
    }
    const uint32_t mask = have >= 32 ? 0xFFFFFFFFu : ((1u << have) - 1u);
    lo &= mask; hi &= mask; ok &= mask;
}

// htslib resolve_cigar2's peek at the last reference base of op k: is an insertion reported there?
__device__ inline bool ins_after(const uint8_t *cg, uint32_t n, uint32_t k)
{
    if (k + 1 >= n) return false;
    const uint32_t c2 = ld_u32(cg + 4 * (size_t)(k + 1)), op2 = c2 & 0xFu;
    uint32_t tot = 0;
    if (op2 == 1) {
        tot = c2 >> 4;
        for (uint32_t j = k + 2; j < n; ++j) {
            const uint32_t c = ld_u32(cg + 4 * (size_t)j), o = c & 0xFu;
            if (o == 1) tot += c >> 4;
            else if (o != 6) break;
        }
    } else if (op2 == 6 && k + 2 < n) {
        for (uint32_t j = k + 2; j < n; ++j) {
            const uint32_t c = ld_u32(cg + 4 * (size_t)j), o = c & 0xFu;
            if (o == 1) tot += c >> 4;
            else if (consumes_ref(o)) break;
        }
    }
    return tot > 0;
}

// 32 consecutive bases of a read that lies on the reference as it is ([H][S]M[S][H]): one unaligned 16-byte load (+ the byte behind
// it for a read whose first aligned base sits in a low nibble) -> the pair {C|T, G|T} and the "is A/C/G/T" plane.
// p: the byte that holds base y (= y0 + 32 q), odd: y is odd, have: bases of the pair the read really has (< 32: masked)
__device__ inline void pair_planes(const uint8_t *p, bool odd, int have, uint32_t &lo, uint32_t &hi, uint32_t &ok)
{
    uint32_t d[5];
    __builtin_memcpy(d, p, 16);                                 // (any byte address: unaligned access mode, one global_load_dwordx4)
    d[4] = p[16];                                               // (what lies behind a read's SEQ — its QUAL, the arrays' slack — is masked below)
    uint32_t w[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) w[k] = BITOP3(d[k] << 4, d[k] >> 4, 0xF0F0F0F0u, c ? a : b);        // BAM keeps the first base of a byte in the high nibble
    // The four words' planes (bits at 0, 4, .. 28: base 8 k + i of word k at bit 4 i) are squeezed TOGETHER: two bits per byte
    // within each word, the four words interleaved into one (byte b: bases 8 k + 2 b + e at bits 2 k + e), then the bytes'
    // 2-bit fields transposed with two masked swaps — 19 instructions a plane where four 3-step squeezes took 39, and this
    // kernel is bound by vector issue.
    uint32_t ul = 0, uh = 0, uv = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t n = odd ? __builtin_amdgcn_alignbit(w[k + 1], w[k], 4) : w[k];
        uint32_t l, h, v;
        classify8(n, l, h, v);
        ul |= BITOP3(l, l >> 3, 0x03030303u, (a | b) & c) << (2 * k);
        uh |= BITOP3(h, h >> 3, 0x03030303u, (a | b) & c) << (2 * k);
        uv |= BITOP3(v, v >> 3, 0x03030303u, (a | b) & c) << (2 * k);
    }
    auto swap_fields = [](uint32_t x, int s, uint32_t m) { const uint32_t t = BITOP3(x >> s, x, m, (a ^ b) & c); return BITOP3(x, t, t << s, a ^ b ^ c); };
    lo = swap_fields(swap_fields(ul, 12, 0x0000F0F0u), 6, 0x00CC00CCu);
    hi = swap_fields(swap_fields(uh, 12, 0x0000F0F0u), 6, 0x00CC00CCu);
    ok = swap_fields(swap_fields(uv, 12, 0x0000F0F0u), 6, 0x00CC00CCu);
    const uint32_t mask = have >= 32 ? 0xFFFFFFFFu : have > 0 ? ((1u << have) - 1u) : 0u;
    lo &= mask; hi &= mask; ok &= mask;
}

// one PROJECTED read (anything but [H][S]M[S][H]) -> its plane pairs (out: 2 * ceil(len / 32) words, then the zero pair) and its event words
__device__ inline void pack_read(const ReadView &v, const PackOut &o, PackTotals *tot, uint32_t info, int32_t gpos, uint32_t *out)
{
    const int len = (int)(info & 1023u), npair = (len + 31) >> 5;
    {
        // Walk the CIGAR: matched bases land on their reference offset (bit-field copies into the pair being built), D / N
        // leave empty positions, and the tokens that are not plain bases become events: X for a deleted base
        // whose token is exactly "*", I on the last reference base before an insertion (also "*+..": I but not X).
        int q_cur = 0;
        uint32_t lo = 0, hi = 0, ok = 0;
        auto flush_to = [&](int q_new) {            // store the pairs [q_cur, q_new), all but the first of them empty
            while (q_cur < q_new && q_cur < npair) {
                const int nb = min(32, len - 32 * q_cur);
                *reinterpret_cast<uint2 *>(out + 2 * q_cur) = make_uint2(lo, hi);
                uint32_t miss = (nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u)) & ~ok;
                while (miss) {
                    const int b = __builtin_ctz(miss);
                    push_event(o, tot, (uint32_t)(gpos + 32 * q_cur + b) | TCMI_F_EV_OTHER);
                    miss &= miss - 1;
                }
                lo = hi = ok = 0;
                ++q_cur;
            }
        };
        int x = 0, y = 0;
        for (uint32_t k = 0; k < v.n_cigar && x < len; ++k) {
            const uint32_t c = ld_u32(v.cigar + 4 * (size_t)k), op = c & 0xFu;
            const int oplen = (int)(c >> 4);
            if (consumes_ref(op)) {
                const bool ins = oplen > 0 && ins_after(v.cigar, v.n_cigar, k);
                if (is_match(op)) {
                    int t = 0;
                    while (t < oplen) {
                        const int q = (x + t) >> 5, b0 = (x + t) & 31, nb = min(32 - b0, oplen - t);
                        flush_to(q);
                        uint32_t l, h, g;
                        fetch32(v.seq, v.l_seq, y + t, nb, l, h, g);
                        lo |= l << b0; hi |= h << b0; ok |= g << b0;
                        t += nb;
                    }
                } else if (op == 2) {
                    const int nx = ins ? oplen - 1 : oplen;
                    for (int t = 0; t < nx; ++t) push_event(o, tot, (uint32_t)(gpos + x + t) | TCMI_F_EV_X);
                }
                if (ins) push_event(o, tot, (uint32_t)(gpos + x + oplen - 1) | TCMI_F_EV_I);
                x += oplen;
            }
            if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) y += oplen;
        }
        flush_to(npair);
    }
    *reinterpret_cast<uint2 *>(out + 2 * npair) = make_uint2(0u, 0u);
}

// dev_slots > 0 (the one-sync path: nobody has read the totals back): n_kept and n_words are taken from `tot`, and the reads per
// workgroup are worked out here as tcmi_pack_on_device does on the host (dev_slots = the tally kernel's resident workgroups, or
// 1 << 30: no balancing); the grid was sized from the capacity, workgroups beyond the reads leave at once.
__global__ __launch_bounds__(PB) void pk_pack(PackOut o, const int32_t *c_pos, const uint32_t *c_info, const uint32_t *c_woff,
                                              uint32_t n_kept, uint32_t n_words, int reads_per_wg, int n_stages, int stage_cap, PackTotals *tot,
                                              int64_t dev_slots)
{
    if (dev_slots > 0) {
        n_kept = (uint32_t)min(tot->n_kept, (unsigned long long)0xFFFFFFF0u);
        n_words = (uint32_t)min(tot->n_words, (unsigned long long)o.word_cap - 18ull);
        int64_t C = 2048;
        if (dev_slots < (1ll << 30)) {
            const int64_t longest = (int64_t)TCMI_F_MAXSTAGE * 400, nf = n_kept;
            const int64_t k = max((int64_t)1, (nf + dev_slots * longest - 1) / (dev_slots * longest));
            C = max((int64_t)64, (nf + k * dev_slots - 1) / (k * dev_slots));
        }
        reads_per_wg = (int)min(C, (int64_t)PK_CMAX);
    }
    __shared__ int32_t s_pos[PK_CMAX];
    __shared__ uint32_t s_woff[PK_CMAX + 1];
    __shared__ uint16_t s_len[PK_CMAX];
    __shared__ uint16_t s_run[PK_CMAX + 1];     // chunk-relative index of every run's first read
    __shared__ int s_red[3][PB / 64];
    __shared__ int s_scan[PB / 64];
    __shared__ uint32_t s_slot[1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 0 && tid < 64) {                          // what the last stage's last 16-byte load takes along, and the blob's slack
        if (tid < 16) o.seq[2u + n_words + tid] = 0u;
        o.slack[tid] = 0u;
    }
    const uint32_t r0 = (uint32_t)blockIdx.x * (uint32_t)reads_per_wg;
    if ((unsigned long long)blockIdx.x * (unsigned long long)reads_per_wg >= n_kept) return;     // (a grid sized from the capacity)
    const int n = (int)min((uint32_t)reads_per_wg, n_kept - r0);
    for (int t = tid; t < n; t += PB) {
        s_pos[t] = c_pos[r0 + t];
        s_len[t] = (uint16_t)(c_info[r0 + t] & 1023u);
        s_woff[t] = c_woff[r0 + t];
    }
    if (tid == 0) s_woff[n] = r0 + n < n_kept ? c_woff[r0 + n] : n_words;
    __syncthreads();

    int cur = 0;
    while (cur < n) {                                            // uniform: one chunk per turn
        const int P0 = s_pos[cur] & ~7;
        // the chunk's reads: the longest prefix of [cur, n) inside a window of MAXPOS positions that starts at P0
        int viol = n;
        for (int t = cur + 1 + tid; t < n; t += PB) {
            const int rel = s_pos[t] - P0;
            if (rel < 0 || rel + (int)s_len[t] > MAXPOS) { viol = t; break; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) viol = min(viol, __shfl_xor(viol, d, 64));
        if (lane == 0) s_red[0][wave] = viol;
        __syncthreads();
        int e1 = s_red[0][0];
#pragma unroll
        for (int w = 1; w < PB / 64; ++w) e1 = min(e1, s_red[0][w]);
        int mend = 0, mlen = 0;
        for (int t = cur + tid; t < e1; t += PB) {
            mend = max(mend, s_pos[t] - P0 + (int)s_len[t]);
            mlen = max(mlen, (int)s_len[t]);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { mend = max(mend, __shfl_xor(mend, d, 64)); mlen = max(mlen, __shfl_xor(mlen, d, 64)); }
        if (lane == 0) { s_red[1][wave] = mend; s_red[2][wave] = mlen; }
        __syncthreads();
        mend = s_red[1][0]; mlen = s_red[2][0];
#pragma unroll
        for (int w = 1; w < PB / 64; ++w) { mend = max(mend, s_red[1][w]); mlen = max(mlen, s_red[2][w]); }
        // stage and chunk sizes exactly as the tally kernel will derive them from (Wn, sub_reads): readset.cpp's rules
        const int Wn = (mend + 7) >> 3;
        const int S = FB / max(2, (Wn * 8 + 31) >> 5);
        int cap = min((int)TCMI_P_SUB, (TCMI_F_SEQCAP - 16 - 2) / (int)words_of((uint32_t)mlen));
        if (stage_cap > 0) cap = min(cap, max(stage_cap, S * 4));
        int sub = S * 4 * max(1, cap / (S * 4));
        if (sub > cap) sub = max(S, cap / S * S);
        const int whole = max(sub, min(((1 << TCMI_P_NPL) - 1) * S, n_stages * sub) / sub * sub);
        const int nc = min(e1 - cur, whole);
        if (tid == 0) s_slot[0] = atomicAdd(&tot->n_chunks, 1u);
        // ---- coverage runs: reads of equal (position, length) follow each other in a sorted BAM ---------------------
        int n_runs = 0;
        for (int t0 = 0; t0 < nc; t0 += PB) {
            const int t = t0 + tid;
            bool start = false;
            if (t < nc) {
                const int j = cur + t;
                start = t == 0 || (t & 2047) == 0 || s_pos[j] != s_pos[j - 1] || s_len[j] != s_len[j - 1];
            }
            const int incl = block_scan_incl(start ? 1 : 0, s_scan);    // (two barriers inside: s_slot is visible after them)
            if (start) s_run[n_runs + incl - 1] = (uint16_t)t;
            __syncthreads();
            n_runs += s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
            __syncthreads();
        }
        static_assert(PB / 64 == 4, "run count sums four wave totals");
        if (tid == 0) s_run[n_runs] = (uint16_t)nc;
        __syncthreads();
        // the chunk's words: from the zero pair that closes the read in front of its first one (the stream's own first pair for read 0)
        const uint32_t ci = s_slot[0], w0 = s_woff[cur];
        const bool fits = ci < o.chunk_cap;
        if (!fits && tid == 0) atomicOr(&tot->flags, (uint32_t)PKF_CHUNK_OVF);
        if (fits) {
            const uint32_t g0 = r0 + (uint32_t)cur;             // compacted index of the chunk's first read
            for (int k = tid; k < n_runs; k += PB) {
                const int t = s_run[k], cnt = (int)s_run[k + 1] - t, j = cur + t;
                o.covrun[g0 + k] = (uint32_t)(s_pos[j] - P0) | ((uint32_t)s_len[j] << 10) | ((uint32_t)cnt << 20);
            }
            if (tid == 0) {
                tcmi_fast_chunk c;
                c.read0 = g0; c.word0 = w0; c.n_reads = nc; c.P0 = P0; c.Wn = Wn; c.sub_reads = sub;
                for (int st = 0; st < TCMI_F_MAXSTAGE; ++st) {
                    const int last = min(nc, (st + 1) * sub);
                    c.stage_end[st] = st * sub < nc ? (int32_t)(2u + s_woff[cur + last] - s_woff[cur]) : 0;
                }
                c.run0 = g0; c.n_runs = n_runs; c.reserved_ = 0;
                o.chunks[ci] = c;
                atomicAdd(&tot->n_runs, (uint32_t)n_runs);
                if (w0 == 0u) { o.seq[0] = 0u; o.seq[1] = 0u; }  // the zero pair in front of the very first read (the others: pk_planes)
            }
        }
        // ---- per read: the header word, and where its planes go (pk_planes writes them, one lane per 32 bases) ----------------
        {
            const uint32_t g0 = r0 + (uint32_t)cur;
            for (int t = tid; t < nc; t += PB) {
                const int j = cur + t, st = t / sub;
                const uint32_t base = 2u + s_woff[j] - s_woff[cur];                          // words from word0
                const uint32_t sb = st == 0 ? 0u : s_woff[cur + st * sub] - s_woff[cur];     // the stage starts on the zero pair in front of its first read
                const uint32_t rel = (uint32_t)(s_pos[j] - P0), len = s_len[j], poff = (base - sb) >> 1;
                if (fits) {
                    if (rel > 1023u || len > 1023u || poff > 4095u) atomicOr(&tot->flags, (uint32_t)PKF_HEADER_OVF);
                    o.lenoff[g0 + t] = rel | (len << 10) | (poff << 20);
                }
            }
        }
        cur += nc;
        __syncthreads();
    }
}

// ---- 5: planes: one lane per 32 bases --------------------------------------------------------------------------------------
// A workgroup takes PB consecutive kept reads.  Every read's lane notes, for each of the read's pairs (and the zero pair behind
// them), "read t, pair q" in an LDS list indexed by the pair's place in the workgroup's stretch of the word stream; then the
// lanes take the places of that list in order: a lane loads ITS 16 bytes of SEQ, makes its pair and stores it — consecutive
// lanes read consecutive 16-byte pieces of a read and store consecutive 8-byte pairs (a read per lane, five pairs one after
// the other, kept 22 uncoalesced dword loads per lane in flight and the kernel at 16 waves a CU waiting for them).
// Reads that need their CIGAR walked (INFO_PROJ) are packed by their own lane, as before, straight to where they go.
constexpr int PL_SLOTS = PB * (TCMI_D_MAXLEN / 32 + 2);      // a read's pairs, its zero pair, and the second one that ends it on 16 bytes
__global__ __launch_bounds__(PB) void pk_planes(PackSrc src, PackOut o, const uint32_t *c_idx, const int32_t *c_pos, const uint32_t *c_info,
                                                const uint32_t *c_woff, const uint2 *c_seq, uint32_t n_kept, uint32_t n_words,
                                                PackTotals *tot)
{
    __shared__ uint32_t s_info[PB], s_word[PB];
    __shared__ int32_t s_pos[PB];
    __shared__ uint2 s_seq[PB];
    __shared__ uint16_t s_owner[PL_SLOTS];
    const int tid = threadIdx.x;
    const uint32_t r0 = (uint32_t)blockIdx.x * PB;
    const int n = (int)min((uint32_t)PB, n_kept - r0);
    const uint32_t w_first = c_woff[r0], w_end = r0 + (uint32_t)n < n_kept ? c_woff[r0 + n] : n_words;
    if (tid < n) {
        const uint32_t g = r0 + (uint32_t)tid, info = c_info[g], word = 2u + c_woff[g];      // (the read's place in the plane stream)
        const int32_t pos = c_pos[g];
        const int npair = (int)((info & 1023u) + 31u) >> 5;
        const uint32_t slot0 = (c_woff[g] - w_first) >> 1;
        const bool own = (info & INFO_PROJ) != 0;                       // (its lane writes its pairs and the zero pair behind them)
        const int n_slot = (int)(words_of(info & 1023u) >> 1);
        s_info[tid] = info; s_word[tid] = word; s_pos[tid] = pos; s_seq[tid] = c_seq[g];
        for (int q = 0; q < n_slot; ++q) s_owner[slot0 + q] = own && q <= npair ? (uint16_t)0xFFFFu : (uint16_t)(tid | (q << 8));
        if (own) pack_read(view(src, c_idx[g]), o, tot, info, pos, o.seq + word);
    }
    __syncthreads();
    const uint8_t *bytes = src.mode == 0 ? src.seq : src.stream;
    const uint32_t n_slots = (w_end - w_first) >> 1;
    for (uint32_t k = tid; k < n_slots; k += PB) {
        const uint32_t ow = s_owner[k];
        if (ow == 0xFFFFu) continue;
        const int t = (int)(ow & 255u), q = (int)(ow >> 8);
        const uint32_t info = s_info[t];
        const int len = (int)(info & 1023u), npair = (len + 31) >> 5;
        uint2 *dst = reinterpret_cast<uint2 *>(o.seq + s_word[t]) + q;                  // (even word offsets: 8-byte aligned)
        if (q >= npair) { *dst = make_uint2(0u, 0u); continue; }
        const uint2 where = s_seq[t];
        const int l_seq = (int)(where.y >> 8), y = (int)(info >> 12) + 32 * q;
        const int nb = min(32, len - 32 * q), have = min(nb, l_seq - y);
        uint32_t lo, hi, ok;
        pair_planes(bytes + ((((unsigned long long)(where.y & 0xFFu) << 32) | where.x) + (uint32_t)(y >> 1)), y & 1, have, lo, hi, ok);
        *dst = make_uint2(lo, hi);
        uint32_t miss = (nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u)) & ~ok;
        while (miss) {
            const int b = __builtin_ctz(miss);
            push_event(o, tot, (uint32_t)(s_pos[t] + 32 * q + b) | TCMI_F_EV_OTHER);
            miss &= miss - 1;
        }
    }
}


// ---- a device-decoded stream -> the packed read set, without a host round trip and without a scan kernel -------------------------
// What rec_scan + the host's chain check + rec_compact + pk_classify + pk_scan + pk_scatter + pk_planes do in seven steps with three
// host round trips, as TWO kernels with none, one workgroup per BGZF block each:
//   pk_index   the records that start in the block (bgzf_copy listed them) get their place in the dense record index — the block's
//              base is the sum of the record counts of the blocks in front of it, which every workgroup adds up for itself (a few
//              thousand 4-byte words from L2: no scan kernel, nobody waits for anybody) —, are classified (classify_view, as
//              pk_classify), and leave 16 bytes each for pk_place; the block leaves its aggregate: kept reads, plane words, and the
//              record chain's transfer function across it
//   pk_place   adds up the aggregates in front of its block the same way, checks that the chain of records arrives at its block
//              where its first record starts (what the host checks block by block on the other path), and writes the kept reads'
//              entries and their bit planes (pk_scatter's and pk_planes' work) at their final places
// A first version did all of this in ONE kernel with a decoupled look-back between the workgroups (Merrill & Garland's single-pass
// scan); alone on the GPU it took 96 us, but a workgroup that waits for its predecessors holds registers and LDS that the kernels
// of other streams want: with eight contexts the pipeline lost 8 %, and with eight hardware queues it collapsed (16 M positions/s).
// Nothing here waits for another workgroup.
// The record chain across block boundaries as a function of "offset in this block at which a record must start": identity (a
// header-only block), subtract the block's length (no record starts in it), a constant (the block's last record runs that far into
// the next).
enum { CH_ID = 0, CH_SUB = 1, CH_CONST = 2 };
constexpr long long CH_BIAS = 1ll << 44;
__device__ inline unsigned long long ch_pack(int kind, long long v) { return ((unsigned long long)kind << 56) | (unsigned long long)(v + CH_BIAS); }
__device__ inline int ch_kind(unsigned long long p) { return (int)((p >> 56) & 3u); }
__device__ inline long long ch_val(unsigned long long p) { return (long long)(p & ((1ull << 56) - 1ull)) - CH_BIAS; }
constexpr uint32_t PKF_CHAIN = 256, PKF_STAT = 512, PKF_REC_OVF = 2048;

struct FusedArgs {
    const uint8_t *stream;
    uint64_t stream_len;
    const BlockDesc *blocks;
    const uint32_t *rec_slot;       // [n_blocks][MAX_REC_PER_BLOCK] (bgzf_copy)
    const uint32_t *n_rec;          // [n_blocks]
    const uint32_t *first;          // [n_blocks] offset of the first record start the block found in itself (0xFFFFFFFF: none)
    const int32_t *over;            // [n_blocks] bytes the block's last record runs into the next blocks (0x7FFFFFFF: read its size here)
    const uint32_t *stat;           // [n_blocks] ST_*
    int32_t n_blocks, n_own, ranged;
    uint2 *agg;                     // [n_blocks] pk_index -> pk_place: {kept reads, plane words} of the block
    unsigned long long *fn;         // [n_blocks] ... the chain's transfer function across the block (ch_pack)
    uint32_t *rec_base;             // [n_blocks] ... index of the block's first record
    // files of very many blocks (every workgroup adding up everything in front of it is quadratic): exclusive prefix sums made by
    // pk_prefix between the kernels — records in front of a block (for pk_index), kept reads and plane words (for pk_place); else null
    const unsigned long long *pre_rec, *pre_k, *pre_w;
    uint4 *rrec;                    // [rec_cap] ... per record {word, where SEQ lies (2 words), pos}
    uint64_t *rec_off;              // out [rec_cap]: dense record offsets
    uint32_t rec_cap;
    uint32_t *c_idx; int32_t *c_pos; uint32_t *c_info; uint32_t *c_woff; uint2 *c_seq;     // out [rec_cap]: the kept reads, compacted
    uint32_t *gen_idx;              // out [rec_cap]: reads left to tally_stream_kernel
    PackOut o;                      // plane stream + events (caps inside)
    unsigned long long *blk_alg;    // out [n_blocks]: algorithmic bytes | longest span << 48
    int32_t *blk_end;               // out [n_blocks]: max end
    PackTotals *tot;
};

// the sum of v[0 .. n) over the workgroup (every lane gets it); s4: LDS [PB / 64]
__device__ inline unsigned long long block_sum_u32(const uint32_t *v, int n, unsigned long long *s4)
{
    unsigned long long acc = 0;
    for (int i = threadIdx.x; i < n; i += PB) acc += v[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += (unsigned long long)__shfl_xor((long long)acc, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = acc;
    __syncthreads();
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < PB / 64; ++w) t += s4[w];
    __syncthreads();
    return t;
}

// exclusive prefix sums of v[i * stride] over i < n (entries from n_lim on count as nothing), one workgroup: out[i], out[n] = the total
__global__ __launch_bounds__(1024) void pk_prefix(const uint32_t *v, int stride, int n, int n_lim, unsigned long long *out)
{
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const unsigned long long x = i < n && i < n_lim ? v[(size_t)i * (size_t)stride] : 0ull;
        unsigned long long incl = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long y = (unsigned long long)__shfl_up((long long)incl, d, 64);
            if (lane >= d) incl += y;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        unsigned long long before = s_run;
        for (int w = 0; w < wave; ++w) before += s_w[w];
        if (i < n) out[i] = before + incl - x;
        __syncthreads();
        if (tid == 1023) s_run = before + incl;
        __syncthreads();
    }
    if (tid == 0) out[n] = s_run;
}

__global__ __launch_bounds__(PB) void pk_index(FusedArgs a)
{
    __shared__ uint2 s_w[PB / 64];
    __shared__ unsigned long long s_alg[PB / 64];
    __shared__ int32_t s_end[PB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (int)blockIdx.x;
    const BlockDesc d = a.blocks[b];
    const bool mine = b < a.n_own;                                  // (a block taken along for the tail of the last record: its records are not ours)
    const uint32_t n = mine ? a.n_rec[b] : 0u;
    const uint32_t *slots = a.rec_slot + (size_t)b * MAX_REC_PER_BLOCK;
    const uint32_t st = a.stat[b];
    const uint32_t first = mine ? a.first[b] : 0xFFFFFFFFu;
    int32_t over = mine ? a.over[b] : 0;
    if (tid == 0 && st != ST_OK) atomicOr(&a.tot->flags, PKF_STAT);
    // the block's first record's index: the records of the blocks in front of it (all of them ours: b < n_own, or n = 0)
    const unsigned long long base = a.pre_rec ? a.pre_rec[b] : block_sum_u32(a.n_rec, min(b, a.n_own), s_alg);
    if (tid == 0) a.rec_base[b] = (uint32_t)min(base, (unsigned long long)0xFFFFFFFFu);
    const bool room = base + n <= a.rec_cap;
    if (!room && tid == 0) atomicOr(&a.tot->flags, PKF_REC_OVF);
    unsigned long long my_alg = 0;
    int32_t my_end = 0;
    uint32_t my_len = 0, tot_k = 0, tot_w = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += PB) {
        const uint32_t t = t0 + (uint32_t)tid;
        uint32_t word = 0, nwords = 0;
        if (t < n) {
            const uint64_t roff = d.uout + slots[t];
            const uint8_t *rec = a.stream + roff;
            ReadView v = view_rec(rec);
            // (a record that claims to end behind the stream: nothing of it is followed — the chain check will refuse the file)
            if (roff + 4ull + ld_u32(rec) > a.stream_len) { v.bad = true; v.broken = true; v.n_cigar = 0; }
            const Classified c = classify_view(v, 1, 0, rec, true, a.tot);
            word = c.word; nwords = c.nwords;
            my_alg += c.alg; my_end = max(my_end, c.end); my_len = max(my_len, c.len);
            if (room) {
                const uint64_t ig = base + t;
                const unsigned long long so = (unsigned long long)(v.seq - a.stream);
                a.rec_off[ig] = roff;
                a.rrec[ig] = make_uint4(word, (uint32_t)so, ((uint32_t)(so >> 32) & 0xFFu) | ((uint32_t)min(v.l_seq, 0xFFFFFF) << 8), (uint32_t)v.pos);
                if (c.longread) a.gen_idx[atomicAdd(&a.tot->n_gen, 1u)] = (uint32_t)ig;
            }
        }
        const uint2 incl = block_scan2(make_uint2(word ? 1u : 0u, nwords), s_w);
        __syncthreads();
        if (tid == PB - 1) s_w[0] = incl;
        __syncthreads();
        tot_k += s_w[0].x; tot_w += s_w[0].y;
        __syncthreads();
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) {
        my_alg += (unsigned long long)__shfl_xor((long long)my_alg, dd, 64);
        my_end = max(my_end, __shfl_xor(my_end, dd, 64));
        my_len = max(my_len, (uint32_t)__shfl_xor((int)my_len, dd, 64));
    }
    if (lane == 0) { s_alg[wave] = my_alg | ((unsigned long long)my_len << 48); s_end[wave] = my_end; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long al = 0, ml = 0;
        int32_t e = 0;
        for (int w = 0; w < PB / 64; ++w) { al += s_alg[w] & 0xFFFFFFFFFFFFull; ml = max(ml, s_alg[w] >> 48); e = max(e, s_end[w]); }
        a.blk_alg[b] = al | (ml << 48);
        a.blk_end[b] = e;
        a.agg[b] = make_uint2(tot_k, tot_w);
        if (mine && over == 0x7FFFFFFF && n) {                      // the last record's size field straddles the block's end: read it from the stream
            const uint32_t at = slots[n - 1];
            const uint32_t bs = ld_u32(a.stream + d.uout + at);
            over = bs - 32u > (1u << 28) - 32u ? -1 : (int32_t)(at + 4u + bs - d.ulen);
        }
        unsigned long long fn;
        if (!mine || d.entry == -1) fn = ch_pack(CH_ID, 0);
        else if (first != 0xFFFFFFFFu) fn = ch_pack(CH_CONST, over);
        else if (d.entry >= 0) fn = ch_pack(CH_CONST, (long long)d.entry - (long long)d.ulen);
        else fn = ch_pack(CH_SUB, d.ulen);
        a.fn[b] = fn;
    }
}

__global__ __launch_bounds__(PB) void pk_place(FusedArgs a)
{
    __shared__ uint2 s_w[PB / 64];
    __shared__ unsigned long long s_sum[2 * (PB / 64)];
    __shared__ uint32_t s_info[PB], s_word[PB];
    __shared__ int32_t s_pos[PB];
    __shared__ uint2 s_seq[PB];
    __shared__ uint16_t s_owner[PL_SLOTS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (int)blockIdx.x;
    const BlockDesc d = a.blocks[b];
    const bool mine = b < a.n_own;
    const uint32_t n = mine ? a.n_rec[b] : 0u;
    const uint32_t st = a.stat[b];
    // ---- the kept reads and words in front of this block: every workgroup adds the aggregates up for itself -------------------------
    unsigned long long ex_k = 0, ex_w = 0;
    {
        if (a.pre_k) { if (tid == 0) { ex_k = a.pre_k[b]; ex_w = a.pre_w[b]; } }
        else for (int i = tid; i < b; i += PB) { const uint2 v = a.agg[i]; ex_k += v.x; ex_w += v.y; }
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) { ex_k += (unsigned long long)__shfl_xor((long long)ex_k, dd, 64); ex_w += (unsigned long long)__shfl_xor((long long)ex_w, dd, 64); }
        if (lane == 0) { s_sum[wave] = ex_k; s_sum[PB / 64 + wave] = ex_w; }
    }
    // ---- the record chain (wavefront 0): the nearest block in front that fixes the state, minus the lengths of the record-less blocks behind it
    if (wave == 0) {
        int kind = CH_ID;
        long long val = 0, subs = 0;
        for (int hi = b - 1; hi >= 0 && kind != CH_CONST; hi -= 64) {
            const int e = hi - lane;
            const unsigned long long f = e >= 0 ? a.fn[e] : ch_pack(CH_ID, 0);
            const int k = ch_kind(f);
            const unsigned long long cmask = __ballot(k == CH_CONST);
            const int fc = cmask ? (int)__builtin_ctzll(cmask) : 64;
            long long sb = lane < fc && k == CH_SUB ? ch_val(f) : 0;
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) sb += __shfl_xor(sb, dd, 64);
            subs += sb;
            if (fc < 64) { kind = CH_CONST; val = __shfl(ch_val(f), fc, 64); }
        }
        // in front of the file: expect = -1; in front of a range: OPEN (it starts wherever its first block finds a record)
        const bool open_in = kind != CH_CONST && a.ranged != 0;
        const long long state_in = kind == CH_CONST ? val - subs : -1 - subs;
        if (lane == 0) {
            const uint32_t first = mine ? a.first[b] : 0xFFFFFFFFu;
            const bool has = first != 0xFFFFFFFFu;
            const long long over = ch_val(a.fn[b]);                 // (pk_index resolved a size field that straddles the block's end)
            int ok = 1;
            long long state_out = state_in;
            bool open_out = open_in;
            if (mine && d.entry != -1) {
                if (has) {
                    const long long expect = d.entry >= 0 ? (long long)d.entry : state_in;
                    if (st == ST_BAD_RECORD) ok = 0;
                    if (!open_in || d.entry >= 0) { if ((long long)first != expect) ok = 0; }
                    else a.tot->range_first = d.uout + first + 1ull;      // (the range starts here, unvouched for: the range in front must end here)
                    if (over < 0) ok = 0;
                    state_out = over; open_out = false;
                } else if (!open_in || d.entry >= 0) {
                    const long long expect = d.entry >= 0 ? (long long)d.entry : state_in;
                    if (st == ST_BAD_RECORD) ok = 0;
                    if (expect < (long long)d.ulen) ok = 0;
                    state_out = expect - (long long)d.ulen; open_out = false;
                }
            }
            if (b == a.n_own - 1 || (a.n_own == 0 && b == 0)) {     // the end of the range: the last record must end in it — or in the block taken along
                if (!open_out) {
                    if (a.n_own < a.n_blocks) { if (state_out > (long long)a.blocks[a.n_own].ulen) ok = 0; }
                    else if (state_out > 0) ok = 0;
                    if (mine) a.tot->range_next = (unsigned long long)((long long)(d.uout + d.ulen) + state_out) + 1ull;
                }
            }
            if (!ok) atomicOr(&a.tot->flags, PKF_CHAIN);
        }
    }
    __syncthreads();
    ex_k = 0; ex_w = 0;
#pragma unroll
    for (int w = 0; w < PB / 64; ++w) { ex_k += s_sum[w]; ex_w += s_sum[PB / 64 + w]; }
    const uint2 mine_agg = a.agg[b];
    const uint32_t tot_w = mine_agg.y;
    const unsigned long long ex_r = a.rec_base[b];
    if (b == a.n_blocks - 1 && tid == 0) {
        a.tot->n_rec = ex_r + n; a.tot->n_kept = ex_k + mine_agg.x; a.tot->n_words = ex_w + tot_w;
        if (ex_r + n > 0x7FFFFFFFull || ex_k + mine_agg.x > 0x7FFFFFFFull) atomicOr(&a.tot->flags, PKF_REC_OVF);
    }
    if (n == 0) return;
    // room for this block's records, kept reads and words?  (the arrays are sized from bounds: a file beyond them takes the other path)
    if (ex_r + n > a.rec_cap || ex_w + tot_w + 2ull + 16ull > a.o.word_cap) {
        if (tid == 0) atomicOr(&a.tot->flags, ex_r + n > a.rec_cap ? PKF_REC_OVF : (uint32_t)PKF_WORD_OVF);
        return;
    }
    // ---- the kept reads' entries and their planes ---------------------------------------------------------------------------------------
    uint32_t run_k = 0, run_w = 0;                                  // kept reads / words of the tiles in front of this one
    for (uint32_t t0 = 0; t0 < n; t0 += PB) {
        const uint32_t t = t0 + (uint32_t)tid;
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (t < n) r = a.rrec[ex_r + t];
        const uint32_t word = r.x, nwords = word ? words_of(word & 1023u) : 0u;
        const uint2 incl = block_scan2(make_uint2(word ? 1u : 0u, nwords), s_w);
        __syncthreads();
        if (tid == PB - 1) s_w[0] = incl;
        __syncthreads();
        const uint32_t tile_k = s_w[0].x, tile_w = s_w[0].y;
        const uint32_t w_first = (uint32_t)ex_w + run_w;            // the tile's first word offset (its reads are contiguous from there)
        if (word) {
            const uint32_t lk = incl.x - 1u, lwoff = incl.y - nwords;    // the read's place among the tile's kept reads; its first word, from the tile's
            const uint32_t j = (uint32_t)ex_k + run_k + lk, woff = w_first + lwoff;
            const int32_t pos = (int32_t)r.w;
            const uint2 sq = make_uint2(r.y, r.z);
            a.c_idx[j] = (uint32_t)(ex_r + t); a.c_pos[j] = pos; a.c_info[j] = word; a.c_woff[j] = woff; a.c_seq[j] = sq;
            const int npair = (int)((word & 1023u) + 31u) >> 5;
            const bool own = (word & INFO_PROJ) != 0;               // (its lane writes its pairs and the zero pair behind them)
            s_info[lk] = word; s_word[lk] = 2u + woff; s_pos[lk] = pos; s_seq[lk] = sq;
            const uint32_t slot0 = lwoff >> 1;
            for (int q = 0; q < (int)(nwords >> 1); ++q) s_owner[slot0 + q] = own && q <= npair ? (uint16_t)0xFFFFu : (uint16_t)(lk | ((uint32_t)q << 8));
            if (own) pack_read(view_rec(a.stream + a.rec_off[ex_r + t]), a.o, a.tot, word, pos, a.o.seq + 2u + woff);
        }
        __syncthreads();
        uint2 *dst = reinterpret_cast<uint2 *>(a.o.seq + 2u + w_first);                      // (word offsets are multiples of 4: 8-byte aligned)
        for (uint32_t k = tid; k < (tile_w >> 1); k += PB) {
            const uint32_t ow = s_owner[k];
            if (ow == 0xFFFFu) continue;
            const int tt = (int)(ow & 255u), q = (int)(ow >> 8);
            const uint32_t info = s_info[tt];
            const int len = (int)(info & 1023u), npair = (len + 31) >> 5;
            if (q >= npair) { dst[k] = make_uint2(0u, 0u); continue; }
            const uint2 where = s_seq[tt];
            const int l_seq = (int)(where.y >> 8), y = (int)(info >> 12) + 32 * q;
            const int nb = min(32, len - 32 * q), have = min(nb, l_seq - y);
            uint32_t lo, hi, ok;
            pair_planes(a.stream + ((((unsigned long long)(where.y & 0xFFu) << 32) | where.x) + (uint32_t)(y >> 1)), y & 1, have, lo, hi, ok);
            dst[k] = make_uint2(lo, hi);
            uint32_t miss = (nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u)) & ~ok;
            while (miss) {
                const int bb = __builtin_ctz(miss);
                push_event(a.o, a.tot, (uint32_t)(s_pos[tt] + 32 * q + bb) | TCMI_F_EV_OTHER);
                miss &= miss - 1;
            }
        }
        run_k += tile_k; run_w += tile_w;
        __syncthreads();
    }
}

// ---- insert-candidate columns: every read of a column as an entry for the host's token vote (Events.py:47-82) -----------------
// One lane per (candidate column, read that starts within TCMI_D_MAXLEN positions before it).  The lane applies the samtools
// stepper's filters, finds the CIGAR op that covers the column and builds what pysam's get_query_sequences(add_indels=True)
// would print for it as a packed 64-bit key (insert_tokens.cpp), with the quality pysam tests, the base, the mate fields and a
// hash of the read name; the few thousand entries per column go to the host, which applies the rules that depend on the other
// reads of the column (max_depth admission, overlapping mates, the vote).  The decoded reads themselves never leave the device.
struct InsArgs {
    PackSrc src;
    const uint32_t *c_idx;
    const int32_t *cols;            // [n_cand] 0-based columns
    const int64_t *lo;              // [n_cand] first compacted read index to look at
    const int64_t *off;             // [n_cand + 1] pair offsets: candidate k owns pairs [off[k], off[k+1])
    tcmi_dev_entry *out;            // [off[n_cand]]: pair p's entry at out[p] — in file order; key 0: the read gives none on that column
    int32_t n_cand;
    uint32_t flag_filter;
    int32_t ignore_orphans;
    // insertions of more than 12 bases do not fit the entry's key: their bases (one 4-bit code per byte) go here, the key says where
    uint8_t *long_text;
    uint32_t *long_cursor;          // bytes taken
    uint32_t long_cap;
};


__global__ __launch_bounds__(PB) void ins_entries_kernel(InsArgs a)
{
    const int64_t p = (int64_t)blockIdx.x * PB + threadIdx.x;
    if (p >= a.off[a.n_cand]) return;
    a.out[p].key = 0;                                           // (overwritten below if the read has a token on the column)
    int k = 0;
    while (k + 1 < a.n_cand && p >= a.off[k + 1]) ++k;          // (a handful of candidates)
    const int64_t j = a.lo[k] + (p - a.off[k]);
    const int32_t col = a.cols[k];
    const uint32_t i = a.c_idx[j];
    const ReadView v = view(a.src, i);
    if (v.flag & a.flag_filter) return;
    if (a.ignore_orphans && (v.flag & 0x1u) && !(v.flag & 0x2u)) return;
    // the op that covers the column
    int64_t x = v.pos, y = 0;
    int64_t span = 0;
    for (uint32_t c = 0; c < v.n_cigar; ++c) {
        const uint32_t w = ld_u32(v.cigar + 4 * (size_t)c);
        if (consumes_ref(w & 0xFu)) span += w >> 4;
    }
    if (col < v.pos || col >= v.pos + span) return;
    for (uint32_t c = 0; c < v.n_cigar; ++c) {
        const uint32_t w = ld_u32(v.cigar + 4 * (size_t)c), op = w & 0xFu;
        const int64_t len = w >> 4;
        if (consumes_ref(op)) {
            if (col < x + len) {
                const bool rev = v.flag & 0x10u;
                const int32_t lq = v.l_seq;
                const int64_t qpos = is_match(op) ? y + (col - x) : y;
                const uint8_t *qual = v.seq + ((size_t)lq + 1) / 2;
                tcmi_dev_entry e;
                e.qual = (uint8_t)(qpos < lq ? byte_at(qual + qpos) : 0u);
                const uint32_t nib = qpos < lq ? nib_at(v.seq, (int32_t)qpos) : 15u;
                e.bits = (uint8_t)(nib | (is_match(op) ? 0x10u : 0u));
                // first character: "=ACMGRSVTWYHKDBN", '=' prints as '.' / ',' by strand; '*' for a deleted base, '>' '<' for a skip
                const char *NT = "=ACMGRSVTWYHKDBN";
                char first = is_match(op) ? NT[nib] : (op == 3 ? (rev ? '<' : '>') : '*');
                if (first == '=') first = rev ? ',' : '.';
                // p->indel of htslib's resolve_cigar2 on the last reference base of the op
                int64_t indel = 0;
                if (col == x + len - 1 && c + 1 < v.n_cigar) {
                    const uint32_t w2 = ld_u32(v.cigar + 4 * (size_t)(c + 1)), op2 = w2 & 0xFu;
                    if (op2 == 2 && op != 2) {
                        indel = -(int64_t)(w2 >> 4);
                        for (uint32_t t = c + 2; t < v.n_cigar; ++t) { const uint32_t wt = ld_u32(v.cigar + 4 * (size_t)t); if ((wt & 0xFu) != 2) break; indel -= wt >> 4; }
                    } else if (op2 == 1) {
                        indel = w2 >> 4;
                        for (uint32_t t = c + 2; t < v.n_cigar; ++t) {
                            const uint32_t wt = ld_u32(v.cigar + 4 * (size_t)t), o = wt & 0xFu;
                            if (o == 1) indel += wt >> 4; else if (o != 6) break;
                        }
                    } else if (op2 == 6 && c + 2 < v.n_cigar) {
                        for (uint32_t t = c + 2; t < v.n_cigar; ++t) {
                            const uint32_t wt = ld_u32(v.cigar + 4 * (size_t)t), o = wt & 0xFu;
                            if (o == 1) indel += wt >> 4; else if (consumes_ref(o)) break;
                        }
                    }
                }
                uint64_t key = (1ull << 63) | (uint8_t)first;
                if (indel > 12) {
                    // does not fit the key: bits 8-39 where its bases start in the text buffer, bits 40-62 how many (bits |= 0x40;
                    // 0x80: the buffer is full or the insertion absurdly long — the host sweep takes the BAM)
                    e.bits |= 0x40;
                    const uint32_t slot = indel < (1 << 23) ? atomicAdd(a.long_cursor, (uint32_t)indel) : a.long_cap;
                    if (indel < (1 << 23) && (uint64_t)slot + (uint64_t)indel <= a.long_cap) {
                        for (int64_t t = 1; t <= indel; ++t) {
                            const int64_t q2 = qpos + t;
                            a.long_text[slot + (uint32_t)(t - 1)] = (uint8_t)(q2 >= lq ? 15u : nib_at(v.seq, (int32_t)q2));
                        }
                        key |= ((uint64_t)slot << 8) | ((uint64_t)indel << 40);
                    } else e.bits |= 0x80;
                } else if (indel > 0) {
                    key |= (1ull << 8) | ((uint64_t)indel << 10);
                    bool any_eq = false;
                    for (int64_t t = 1; t <= indel; ++t) {
                        const int64_t q2 = qpos + t;
                        const uint32_t nb = q2 >= lq ? 15u : nib_at(v.seq, (int32_t)q2);
                        any_eq |= nb == 0;
                        key |= (uint64_t)nb << (15 + 4 * (t - 1));
                    }
                    if (any_eq && rev) key |= 1ull << 14;
                } else if (indel < 0) {
                    key |= (2ull << 8) | ((uint64_t)(-indel) << 10);
                }
                e.key = key;
                // mate fields and the name (behind block_size: refID 0, pos 4, l_read_name 8, ..., next_refID 20, next_pos 24, tlen 28, name 32)
                const uint8_t *r = a.src.stream + a.src.rec_off[i] + 4;
                const int32_t mtid = (int32_t)ld_u32(r + 20);
                e.mpos = (int32_t)ld_u32(r + 24);
                e.isize = (int32_t)ld_u32(r + 28);
                if (mtid >= 0 && mtid != v.tid) e.bits |= 0x20;
                const uint32_t l_name = ld_u32(r + 8) & 0xFFu;
                uint64_t h = 1469598103934665603ull;
                for (uint32_t t = 0; t + 1 < l_name; ++t) { h ^= byte_at(r + 32 + t); h *= 1099511628211ull; }
                e.name_hash = h ? h : 1;
                e.j = (uint32_t)j; e.pos = v.pos; e.end = (int32_t)(v.pos + span); e.l_qseq = lq; e.flag = (uint16_t)v.flag;
                // a deletion / ref-skip token is tested on the quality of the next query base: a matched one? on which reference position
                // (where the overlap tweak of a pair of mates can reach it: insert_tokens.cpp)
                e.qref = -1;
                if (!is_match(op) && qpos < lq) {
                    int64_t xr = x + len;
                    for (uint32_t t = c + 1; t < v.n_cigar; ++t) {
                        const uint32_t wt = ld_u32(v.cigar + 4 * (size_t)t), o = wt & 0xFu;
                        if (is_match(o)) { if ((wt >> 4) > 0 && xr <= INT32_MAX) e.qref = (int32_t)xr; break; }
                        if ((o == 1 || o == 4) && (wt >> 4) > 0) break;
                        if (consumes_ref(o)) xr += wt >> 4;
                    }
                }
                a.out[p] = e;
                return;
            }
            x += len;
        }
        if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) y += len;
    }
}

// do the kept reads ascend by position?  (the packer also takes input with a few reads out of place; the range search below does not)
__global__ __launch_bounds__(256) void ins_sorted_kernel(const int32_t *c_pos, int64_t nf, uint32_t *unsorted)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x + 1;
    if (i < nf && c_pos[i] < c_pos[i - 1]) atomicOr(unsorted, 1u);
}

// which of the kept reads (ascending positions) can cover column cols[k]: those that start in (col - max_len, col]
__global__ __launch_bounds__(64) void ins_ranges_kernel(const int32_t *c_pos, int64_t nf, const int32_t *cols, int32_t n_cand, int32_t max_len,
                                                        int64_t *lo, int64_t *hi)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= n_cand) return;
    const int64_t first = (int64_t)cols[k] - max_len + 1, last = cols[k];
    int64_t a = 0, b = nf;
    while (a < b) { const int64_t m = (a + b) >> 1; if ((int64_t)c_pos[m] < first) a = m + 1; else b = m; }
    lo[k] = a;
    b = nf;
    while (a < b) { const int64_t m = (a + b) >> 1; if ((int64_t)c_pos[m] <= last) a = m + 1; else b = m; }
    hi[k] = a;
}

// does read j (compacted index) have a matched base on reference position `ref`?  -> matched | base << 8 | quality << 16
struct ProbeArgs {
    PackSrc src;
    const uint32_t *c_idx;
    const int64_t *idx;
    const int32_t *ref;
    uint32_t *out;
    int32_t n;
};

__global__ __launch_bounds__(64) void ins_probe_kernel(ProbeArgs a)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= a.n) return;
    uint32_t res = 15u << 8;
    const ReadView v = view(a.src, a.c_idx[a.idx[t]]);
    const int64_t ref = a.ref[t];
    int64_t x = v.pos, y = 0;
    for (uint32_t c = 0; c < v.n_cigar; ++c) {
        const uint32_t w = ld_u32(v.cigar + 4 * (size_t)c), op = w & 0xFu;
        const int64_t len = w >> 4;
        if (consumes_ref(op)) {
            if (ref < x + len) {
                if (is_match(op) && ref >= x) {
                    const int64_t q = y + (ref - x);
                    if (q < v.l_seq) {
                        const uint8_t *qual = v.seq + ((size_t)v.l_seq + 1) / 2;
                        res = 1u | (nib_at(v.seq, (int32_t)q) << 8) | (byte_at(qual + q) << 16);
                    }
                }
                break;
            }
            x += len;
        }
        if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) y += len;
    }
    a.out[t] = res;
}

} // namespace

// ---- host side ------------------------------------------------------------------------------------------------------------
struct tcmi_dev_arena {             // grow-only device scratch of a context (freed with it): the packers' temporaries
    char *base = nullptr;
    size_t cap = 0, used = 0;
};

static int arena_reserve(tcmi_ctx *ctx, size_t bytes)
{
    if (!ctx->dev_arena) ctx->dev_arena = new tcmi_dev_arena();
    tcmi_dev_arena &A = *ctx->dev_arena;
    A.used = 0;
    ++ctx->arena_epoch;                         // whatever lived in the arena is gone
    if (A.cap >= bytes) return TCMI_OK;
    if (A.base) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(A.base); A.base = nullptr; A.cap = 0; }
    const size_t want = bytes + bytes / 8 + (1 << 20);
    if (hipMalloc((void **)&A.base, want) != hipSuccess) return tcmi_fail(ctx, TCMI_E_NOMEM, "hipMalloc(%zu) for the pack scratch failed", want);
    A.cap = want;
    return TCMI_OK;
}

static void *arena_take(tcmi_ctx *ctx, size_t bytes)
{
    tcmi_dev_arena &A = *ctx->dev_arena;
    const size_t at = (A.used + 255) & ~(size_t)255;
    A.used = at + bytes;
    return A.base + at;
}

void tcmi_dev_arena_free(tcmi_dev_arena *a)
{
    if (!a) return;
    if (a->base) (void)hipFree(a->base);
    delete a;
}

void *tcmi_arena_reserve_take(tcmi_ctx *ctx, size_t total, size_t first)   // (bam_device.hip shares the arena)
{
    if (arena_reserve(ctx, total)) return nullptr;
    return arena_take(ctx, first);
}
void *tcmi_arena_take(tcmi_ctx *ctx, size_t bytes) { return arena_take(ctx, bytes); }

// Pack `n` reads described by `src` (device pointers) into a read set.  Returns TCMI_E_UNSUPPORTED (with `*why` set) when
// the input needs the host packer: entries longer than TCMI_D_MAXLEN, positions beyond 2^29, malformed reads (the host
// packer words the error).  The arena must already hold the source arrays; this takes its temporaries behind them.
int tcmi_pack_on_device(tcmi_ctx *ctx, const void *src_, tcmi_readset *rs, uint32_t *why)
{
    PackSrc src = *static_cast<const PackSrc *>(src_);
    *why = 0;
    const int64_t n = src.n;
    if (n > 0xFFFFFFF0ll) { *why = PKF_LONG; return TCMI_E_UNSUPPORTED; }
    const int64_t n_blk = (n + PB - 1) / PB;
    uint32_t *info = (uint32_t *)arena_take(ctx, (size_t)std::max<int64_t>(n, 1) * 4);
    uint2 *rd_seq = (uint2 *)arena_take(ctx, (size_t)std::max<int64_t>(n, 1) * 8);
    int32_t *rd_pos = (int32_t *)arena_take(ctx, (size_t)std::max<int64_t>(n, 1) * 4);
    uint2 *blk_sum = (uint2 *)arena_take(ctx, (size_t)std::max<int64_t>(n_blk, 1) * 8);
    unsigned long long *blk_alg = (unsigned long long *)arena_take(ctx, (size_t)std::max<int64_t>(n_blk, 1) * 8);
    int32_t *blk_end = (int32_t *)arena_take(ctx, (size_t)std::max<int64_t>(n_blk, 1) * 4);
    PackTotals *d_tot = (PackTotals *)arena_take(ctx, sizeof(PackTotals));
    uint32_t *gen_idx = src.mode == 1 ? (uint32_t *)arena_take(ctx, (size_t)std::max<int64_t>(n, 1) * 4) : nullptr;
    PackTotals *h_tot = (PackTotals *)tcmi_ctx_pinned(ctx, sizeof(PackTotals));     // (pinned: the two read-backs below)
    if (!h_tot) return tcmi_fail(ctx, TCMI_E_NOMEM, "pinned scratch for the packer's totals");
    std::memset(h_tot, 0, sizeof *h_tot);
    PackTotals &tot = *h_tot;
    TCMI_HIP(ctx, hipMemsetAsync(d_tot, 0, sizeof(PackTotals), ctx->stream));
    if (n > 0) {
        (void)hipGetLastError();
        tcmi_prof_begin(ctx, TCMI_K_PACK_CLASSIFY);
        hipLaunchKernelGGL(pk_classify, dim3((unsigned)n_blk), dim3(PB), 0, ctx->stream, src, info, rd_seq, rd_pos, blk_sum, blk_alg, blk_end, d_tot, gen_idx);
        hipLaunchKernelGGL(pk_scan, dim3(1), dim3(1024), 0, ctx->stream, blk_sum, blk_alg, blk_end, n_blk, d_tot);
        tcmi_prof_end(ctx, TCMI_K_PACK_CLASSIFY);
        TCMI_HIP(ctx, hipGetLastError());
    }
    TCMI_HIP(ctx, hipMemcpyAsync(h_tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, ctx->stream));
    TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (tot.flags) { *why = tot.flags; return TCMI_E_UNSUPPORTED; }
    const int64_t nf = (int64_t)tot.n_kept;
    rs->n_piled = nf + (int64_t)tot.n_gen; rs->f_reads = nf; rs->alg_bytes = (int64_t)tot.alg_bytes; rs->max_end = tot.max_end; rs->max_len = (int32_t)tot.max_len;
    rs->packed_on_device = 1;
    if (tot.n_gen) {                        // long reads: tallied from the stream, which stays in the arena until this context's next upload
        rs->d_stream = src.stream; rs->d_rec_off = src.rec_off; rs->d_gen_idx = gen_idx; rs->s_reads = (int64_t)tot.n_gen;
        rs->arena_epoch = ctx->arena_epoch;
    }
    if (nf == 0) return TCMI_OK;
    if (tot.n_words > 0xF0000000ull) { *why = PKF_WORD_OVF; return TCMI_E_UNSUPPORTED; }

    // chunk size: as readset.cpp — long chunks, but a multiple of the resident workgroups of them
    int64_t C = 2048;
    const int n_stages = ctx->chunk_stages > 0 ? std::min(ctx->chunk_stages, TCMI_F_MAXSTAGE) : TCMI_F_MAXSTAGE;
    if (ctx->chunk_stages == 0 && ctx->balance_chunks) {
        const int64_t slots = (int64_t)ctx->n_cu * ctx->wg_per_cu, longest = (int64_t)TCMI_F_MAXSTAGE * 400;
        const int64_t k = (nf + slots * longest - 1) / (slots * longest);
        C = std::max<int64_t>(64, (nf + k * slots - 1) / (k * slots));
    }
    C = std::min<int64_t>(C, PK_CMAX);
    const int64_t n_wg = (nf + C - 1) / C;
    // every workgroup opens at least one chunk; more when a window or a lane's 255-read budget runs out
    const uint32_t chunk_cap = (uint32_t)std::min<int64_t>(nf, 4 * n_wg + (int64_t)tot.max_end / 128 + 64);
    const uint32_t word_cap = (uint32_t)(tot.n_words + 2 + 16);        // the zero pair in front, the reads, slack for the last 16-byte load

    uint32_t *c_idx = (uint32_t *)arena_take(ctx, (size_t)nf * 4);
    int32_t *c_pos = (int32_t *)arena_take(ctx, (size_t)nf * 4);
    uint32_t *c_info = (uint32_t *)arena_take(ctx, (size_t)nf * 4);
    uint32_t *c_woff = (uint32_t *)arena_take(ctx, (size_t)nf * 4);
    uint2 *c_seq = (uint2 *)arena_take(ctx, (size_t)nf * 8);
    if (ctx->dev_arena->used > ctx->dev_arena->cap) { return tcmi_fail(ctx, TCMI_E_NOMEM, "internal: pack scratch under-reserved"); }

    uint32_t event_cap = (uint32_t)std::min<int64_t>(0x7FFFFFF0ll, std::max<int64_t>(1 << 20, nf / 2));
    for (int attempt = 0;; ++attempt) {
        PackOut o = {};
        // one allocation for everything the tally kernel reads: headers | planes | chunk records | runs | events
        const size_t b_len = ((size_t)nf * 4 + 255) & ~(size_t)255, b_seq = ((size_t)word_cap * 4 + 255) & ~(size_t)255,
                     b_chk = ((size_t)chunk_cap * sizeof(tcmi_fast_chunk) + 255) & ~(size_t)255, b_run = b_len,
                     b_ev = ((size_t)event_cap * 4 + 255) & ~(size_t)255;
        char *blob = nullptr;
        const size_t want = b_len + b_seq + b_chk + b_run + b_ev + 256;
        for (size_t k = 0; k < ctx->blob_pool.size(); ++k)       // a freed read set of about this size?
            if (ctx->blob_pool[k].bytes >= want && ctx->blob_pool[k].bytes <= want + want / 2 + (1 << 20)) {
                blob = ctx->blob_pool[k].p;
                rs->blob_bytes = ctx->blob_pool[k].bytes;
                ctx->blob_pool.erase(ctx->blob_pool.begin() + (long)k);
                break;
            }
        if (!blob) {
            rs->blob_bytes = want + want / 16;
            if (hipMalloc((void **)&blob, rs->blob_bytes) != hipSuccess)
                return tcmi_fail(ctx, TCMI_E_NOMEM, "hipMalloc(%zu) for the packed read set failed", rs->blob_bytes);
        }
        rs->d_blob = blob;
        o.lenoff = (uint32_t *)blob;
        o.seq = (uint32_t *)(blob + b_len);
        o.chunks = (tcmi_fast_chunk *)(blob + b_len + b_seq);
        o.covrun = (uint32_t *)(blob + b_len + b_seq + b_chk);
        o.events = (uint32_t *)(blob + b_len + b_seq + b_chk + b_run);
        o.word_cap = word_cap; o.chunk_cap = chunk_cap; o.event_cap = event_cap;
        o.slack = reinterpret_cast<uint32_t *>(blob + b_len + b_seq + b_chk + b_run + b_ev);                // 256 bytes behind the last array: pk_pack zeroes them
        if (attempt > 0) TCMI_HIP(ctx, hipMemsetAsync(&d_tot->n_chunks, 0, 4 * sizeof(uint32_t), ctx->stream));    // n_chunks, n_events, n_runs, word_cursor (the first time: pk_scan)
        (void)hipGetLastError();
        tcmi_prof_begin(ctx, TCMI_K_PACK);
        if (attempt == 0)
            hipLaunchKernelGGL(pk_scatter, dim3((unsigned)n_blk), dim3(PB), 0, ctx->stream, src, info, rd_seq, rd_pos, blk_sum, c_idx, c_pos, c_info, c_woff, c_seq);
        hipLaunchKernelGGL(pk_pack, dim3((unsigned)n_wg), dim3(PB), 0, ctx->stream, o, c_pos, c_info, c_woff, (uint32_t)nf,
                           (uint32_t)tot.n_words, (int)C, n_stages, ctx->stage_cap, d_tot, (int64_t)0);
        hipLaunchKernelGGL(pk_planes, dim3((unsigned)((nf + PB - 1) / PB)), dim3(PB), 0, ctx->stream, src, o, c_idx, c_pos, c_info, c_woff, c_seq,
                           (uint32_t)nf, (uint32_t)tot.n_words, d_tot);
        tcmi_prof_end(ctx, TCMI_K_PACK);
        TCMI_HIP(ctx, hipGetLastError());
        TCMI_HIP(ctx, hipMemcpyAsync(h_tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, ctx->stream));
        TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (tot.n_events > event_cap && attempt == 0) {          // rare: a read set full of N / indel tokens — once more with room for all
            (void)hipFree(blob);
            rs->d_blob = nullptr;
            rs->blob_bytes = 0;
            event_cap = tot.n_events + 1024;
            continue;
        }
        if (tot.flags || tot.n_events > event_cap) {
            *why = tot.flags ? tot.flags : (uint32_t)PKF_EVENT_OVF;
            return TCMI_E_UNSUPPORTED;
        }
        rs->d_flenoff = o.lenoff; rs->d_fseq = o.seq; rs->d_fchunk = o.chunks; rs->d_fcovrun = o.covrun; rs->d_fevent = o.events;
        if (src.mode == 1) {                    // the stream and the index stay in the arena until this context's next upload
            rs->d_stream = src.stream; rs->d_rec_off = src.rec_off; rs->d_cidx = c_idx; rs->d_cpos = c_pos;
            rs->arena_epoch = ctx->arena_epoch;
        }
        rs->f_chunks = tot.n_chunks; rs->f_words = (int64_t)tot.n_words + 4; rs->f_events = tot.n_events;
        rs->dev_bytes = nf * 4 + ((int64_t)tot.n_words + 4) * 4 + (int64_t)tot.n_chunks * (int64_t)sizeof(tcmi_fast_chunk) +
                        (int64_t)tot.n_runs * 4 + (int64_t)tot.n_events * 4;
        return TCMI_OK;
    }
}

// what the host checks after its one wait, stored by the kernel itself into pinned host memory (a copy command between two kernels
// costs the stream 40 - 100 us of hand-over between the copy engine and the compute queue; these stores cross PCIe on their own)
__global__ __launch_bounds__(256) void pk_report(const PackTotals *tot, const unsigned long long *blk_alg, const int32_t *blk_end, const uint32_t *stat,
                                                 int32_t nb, PackTotals *h_tot, unsigned long long *h_alg, int32_t *h_end, uint32_t *h_stat)
{
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < nb) { h_alg[i] = blk_alg[i]; h_end[i] = blk_end[i]; h_stat[i] = stat[i]; }
    if (i == 0) *h_tot = *tot;
}

// ---- the one-sync path: pk_index + pk_place + pk_pack queued from capacities, checked after the caller's one wait ------------------------------
static char *take_blob(tcmi_ctx *ctx, tcmi_readset *rs, size_t want)
{
    for (size_t k = 0; k < ctx->blob_pool.size(); ++k)           // a freed read set of about this size?
        if (ctx->blob_pool[k].bytes >= want && ctx->blob_pool[k].bytes <= want + want / 2 + (1 << 20)) {
            char *blob = ctx->blob_pool[k].p;
            rs->blob_bytes = ctx->blob_pool[k].bytes;
            ctx->blob_pool.erase(ctx->blob_pool.begin() + (long)k);
            return blob;
        }
    char *blob = nullptr;
    rs->blob_bytes = want + want / 16;
    if (hipMalloc((void **)&blob, rs->blob_bytes) != hipSuccess) { (void)hipGetLastError(); rs->blob_bytes = 0; return nullptr; }
    return blob;
}

int tcmi_pack_fused_enqueue(tcmi_ctx *ctx, tcmi_fused_job *job, tcmi_readset *rs)
{
    const int64_t nb = job->n_blocks, cap = std::max<int64_t>(job->rec_cap, 1);
    if (cap > 0x7FFFFFF0ll) return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "too many records for the one-pass packer");
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    uint2 *agg = (uint2 *)arena_take(ctx, al((size_t)nb * 8));
    unsigned long long *fn = (unsigned long long *)arena_take(ctx, al((size_t)nb * 8));
    uint32_t *rec_base = (uint32_t *)arena_take(ctx, al((size_t)nb * 4));
    uint4 *rrec = (uint4 *)arena_take(ctx, (size_t)cap * 16);
    unsigned long long *blk_alg = (unsigned long long *)arena_take(ctx, al((size_t)nb * 8));
    int32_t *blk_end = (int32_t *)arena_take(ctx, al((size_t)nb * 4));
    PackTotals *d_tot = (PackTotals *)arena_take(ctx, sizeof(PackTotals));
    // (every workgroup adds up what lies in front of its block: a few thousand words for a 1M-read file, but quadratic in the blocks —
    //  at 67 000 blocks, a 4 GiB stream, it was 40 % of these kernels' time: from 16 384 blocks on three small scan launches do it)
    const bool prefix = ctx->prefix_kernels > 0 || (ctx->prefix_kernels == 0 && nb >= 16384);
    unsigned long long *pre = prefix ? (unsigned long long *)arena_take(ctx, al(((size_t)nb + 1) * 8) * 3) : nullptr;
    job->d_rec = (uint64_t *)arena_take(ctx, (size_t)cap * 8 + 8);
    job->c_idx = (uint32_t *)arena_take(ctx, (size_t)cap * 4);
    job->c_pos = (int32_t *)arena_take(ctx, (size_t)cap * 4);
    uint32_t *c_info = (uint32_t *)arena_take(ctx, (size_t)cap * 4);
    uint32_t *c_woff = (uint32_t *)arena_take(ctx, (size_t)cap * 4);
    uint2 *c_seq = (uint2 *)arena_take(ctx, (size_t)cap * 8);
    job->gen_idx = (uint32_t *)arena_take(ctx, (size_t)cap * 4);
    if (ctx->dev_arena->used > ctx->dev_arena->cap) return tcmi_fail(ctx, TCMI_E_NOMEM, "internal: one-pass packer scratch under-reserved");
    job->d_tot = d_tot;
    const size_t pin_bytes = al(sizeof(PackTotals)) + al((size_t)nb * 8) + al((size_t)nb * 4) + al((size_t)nb * 4);
    job->h_pin = (char *)tcmi_ctx_pinned(ctx, pin_bytes);
    if (!job->h_pin) return tcmi_fail(ctx, TCMI_E_NOMEM, "pinned scratch for the packer's totals");
    // capacities: what the arrays of the packed read set are sized for (a file beyond them takes the several-kernel path)
    const int n_stages = ctx->chunk_stages > 0 ? std::min(ctx->chunk_stages, TCMI_F_MAXSTAGE) : TCMI_F_MAXSTAGE;
    const bool balance = ctx->chunk_stages == 0 && ctx->balance_chunks;
    const int64_t slots = (int64_t)ctx->n_cu * ctx->wg_per_cu, longest = (int64_t)TCMI_F_MAXSTAGE * 400;
    const int64_t k_cap = std::max<int64_t>(1, (cap + slots * longest - 1) / (slots * longest));
    const int64_t n_wg = balance ? std::max<int64_t>(k_cap * slots, (cap + PK_CMAX - 1) / PK_CMAX) : (cap + PK_CMAX - 1) / PK_CMAX + (cap + 2047) / 2048;
    // words: a read of len positions takes <= len / 16 + 7 words, and a read without long deletions / skips has len <= l_seq,
    // each base of which takes 1.5 bytes of the stream (others overflow the capacity: PKF_WORD_OVF, the other path)
    const uint64_t word_cap64 = job->stream_len / 24 + 8ull * (uint64_t)cap + 64;
    if (word_cap64 > 0xF0000000ull) return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "too many plane words for the one-pass packer");
    job->word_cap = (uint32_t)word_cap64;
    job->chunk_cap = (uint32_t)std::min<int64_t>(cap, 4 * n_wg + job->len_bound / 128 + 64);
    job->event_cap = (uint32_t)std::min<int64_t>(0x7FFFFFF0ll, std::max<int64_t>(1 << 20, cap / 2));
    PackOut o = {};
    const size_t b_len = al((size_t)cap * 4), b_seq = al((size_t)job->word_cap * 4), b_chk = al((size_t)job->chunk_cap * sizeof(tcmi_fast_chunk)),
                 b_run = b_len, b_ev = al((size_t)job->event_cap * 4);
    char *blob = take_blob(ctx, rs, b_len + b_seq + b_chk + b_run + b_ev + 256);
    if (!blob) return tcmi_fail(ctx, TCMI_E_NOMEM, "hipMalloc for the packed read set failed");
    rs->d_blob = blob;
    o.lenoff = (uint32_t *)blob;
    o.seq = (uint32_t *)(blob + b_len);
    o.chunks = (tcmi_fast_chunk *)(blob + b_len + b_seq);
    o.covrun = (uint32_t *)(blob + b_len + b_seq + b_chk);
    o.events = (uint32_t *)(blob + b_len + b_seq + b_chk + b_run);
    o.word_cap = job->word_cap; o.chunk_cap = job->chunk_cap; o.event_cap = job->event_cap;
    o.slack = reinterpret_cast<uint32_t *>(blob + b_len + b_seq + b_chk + b_run + b_ev);
    TCMI_HIP(ctx, hipMemsetAsync(d_tot, 0, sizeof(PackTotals), ctx->stream));
    FusedArgs a = {};
    a.stream = job->d_stream; a.stream_len = job->stream_len; a.blocks = static_cast<const BlockDesc *>(job->d_desc);
    a.rec_slot = job->d_slot; a.n_rec = job->d_nrec; a.first = job->d_first; a.over = job->d_over; a.stat = job->d_stat;
    a.n_blocks = (int32_t)nb; a.n_own = (int32_t)job->n_own; a.ranged = job->ranged;
    a.agg = agg; a.fn = fn; a.rec_base = rec_base; a.rrec = rrec; a.rec_off = job->d_rec; a.rec_cap = (uint32_t)cap;
    a.c_idx = job->c_idx; a.c_pos = job->c_pos; a.c_info = c_info; a.c_woff = c_woff; a.c_seq = c_seq; a.gen_idx = job->gen_idx;
    a.o = o; a.blk_alg = blk_alg; a.blk_end = blk_end; a.tot = d_tot;
    const size_t pre_n = al(((size_t)nb + 1) * 8) / 8;
    if (prefix) { a.pre_rec = pre; a.pre_k = pre + pre_n; a.pre_w = pre + 2 * pre_n; }
    (void)hipGetLastError();
    tcmi_prof_begin(ctx, TCMI_K_PACK_CLASSIFY);
    if (prefix) hipLaunchKernelGGL(pk_prefix, dim3(1), dim3(1024), 0, ctx->stream, job->d_nrec, 1, (int)nb, (int)job->n_own, pre);
    hipLaunchKernelGGL(pk_index, dim3((unsigned)nb), dim3(PB), 0, ctx->stream, a);
    tcmi_prof_end(ctx, TCMI_K_PACK_CLASSIFY);
    TCMI_HIP(ctx, hipGetLastError());
    tcmi_prof_begin(ctx, TCMI_K_PACK);
    if (prefix) {
        hipLaunchKernelGGL(pk_prefix, dim3(1), dim3(1024), 0, ctx->stream, reinterpret_cast<const uint32_t *>(agg), 2, (int)nb, (int)nb, pre + pre_n);
        hipLaunchKernelGGL(pk_prefix, dim3(1), dim3(1024), 0, ctx->stream, reinterpret_cast<const uint32_t *>(agg) + 1, 2, (int)nb, (int)nb, pre + 2 * pre_n);
    }
    hipLaunchKernelGGL(pk_place, dim3((unsigned)nb), dim3(PB), 0, ctx->stream, a);
    hipLaunchKernelGGL(pk_pack, dim3((unsigned)n_wg), dim3(PB), 0, ctx->stream, o, job->c_pos, c_info, c_woff, 0u, 0u, 0, n_stages, ctx->stage_cap, d_tot,
                       balance ? slots : (int64_t)1 << 30);
    tcmi_prof_end(ctx, TCMI_K_PACK);
    TCMI_HIP(ctx, hipGetLastError());
    // (the report of the packer's verdicts is queued by tcmi_pack_fused_report, behind whatever the caller queues behind the packer)
    job->d_blk_alg = blk_alg; job->d_blk_end = blk_end;
    // the read set as the tally launch needs it before anyone has read the totals: capacities + where the real counts lie
    rs->packed_on_device = 1;
    rs->d_flenoff = o.lenoff; rs->d_fseq = o.seq; rs->d_fchunk = o.chunks; rs->d_fcovrun = o.covrun; rs->d_fevent = o.events;
    rs->f_chunks = job->chunk_cap; rs->f_events = job->event_cap;
    rs->d_dev_counts = &d_tot->n_chunks;
    static_assert(offsetof(PackTotals, n_events) == offsetof(PackTotals, n_chunks) + 4, "the tally kernel reads {n_chunks, n_events}");
    rs->n_piled = 1;                            // (unknown yet: "there may be reads")
    rs->max_end = 0;
    rs->d_stream = job->d_stream; rs->d_rec_off = job->d_rec; rs->d_cidx = job->c_idx; rs->d_cpos = job->c_pos; rs->d_gen_idx = job->gen_idx;
    rs->arena_epoch = ctx->arena_epoch;
    return TCMI_OK;
}

// the last launch of the one-sync path's chain: totals and per-block verdicts into the job's pinned buffer
int tcmi_pack_fused_report(tcmi_ctx *ctx, tcmi_fused_job *job)
{
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const int64_t nb = job->n_blocks;
    char *h = job->h_pin;
    (void)hipGetLastError();
    hipLaunchKernelGGL(pk_report, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, ctx->stream, static_cast<const PackTotals *>(job->d_tot),
                       static_cast<const unsigned long long *>(job->d_blk_alg), static_cast<const int32_t *>(job->d_blk_end), job->d_stat, (int32_t)nb,
                       reinterpret_cast<PackTotals *>(h), reinterpret_cast<unsigned long long *>(h + al(sizeof(PackTotals))),
                       reinterpret_cast<int32_t *>(h + al(sizeof(PackTotals)) + al((size_t)nb * 8)),
                       reinterpret_cast<uint32_t *>(h + al(sizeof(PackTotals)) + al((size_t)nb * 8) + al((size_t)nb * 4)));
    TCMI_HIP(ctx, hipGetLastError());
    return TCMI_OK;
}

int tcmi_pack_fused_finish(tcmi_ctx *ctx, tcmi_fused_job *job, tcmi_readset *rs, uint32_t *why)
{
    (void)ctx;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const int64_t nb = job->n_blocks;
    const PackTotals &tot = *reinterpret_cast<const PackTotals *>(job->h_pin);
    const unsigned long long *blk_alg = reinterpret_cast<const unsigned long long *>(job->h_pin + al(sizeof(PackTotals)));
    const int32_t *blk_end = reinterpret_cast<const int32_t *>(job->h_pin + al(sizeof(PackTotals)) + al((size_t)nb * 8));
    const uint32_t *stat = reinterpret_cast<const uint32_t *>(job->h_pin + al(sizeof(PackTotals)) + al((size_t)nb * 8) + al((size_t)nb * 4));
    uint32_t flags = tot.flags;
    for (int64_t b = 0; b < nb; ++b) if (stat[b] != ST_OK) flags |= PKF_STAT;
    if (tot.n_rec > (unsigned long long)job->rec_cap) flags |= PKF_REC_OVF;
    if (tot.n_words + 18ull > job->word_cap) flags |= PKF_WORD_OVF;
    if (tot.n_chunks > job->chunk_cap) flags |= PKF_CHUNK_OVF;
    if (tot.n_events > job->event_cap) flags |= PKF_EVENT_OVF;
    *why = flags;
    rs->d_dev_counts = nullptr;
    if (flags) return TCMI_E_UNSUPPORTED;
    unsigned long long alg = 0, mlen = 0;
    int32_t mend = 0;
    for (int64_t b = 0; b < nb; ++b) { alg += blk_alg[b] & 0xFFFFFFFFFFFFull; mlen = std::max(mlen, blk_alg[b] >> 48); mend = std::max(mend, blk_end[b]); }
    const int64_t nf = (int64_t)tot.n_kept;
    rs->n_reads = (int64_t)tot.n_rec;
    rs->n_piled = nf + (int64_t)tot.n_gen; rs->f_reads = nf; rs->alg_bytes = (int64_t)alg; rs->max_end = mend; rs->max_len = (int32_t)mlen;
    rs->s_reads = (int64_t)tot.n_gen;
    rs->range_first = tot.range_first ? (int64_t)(tot.range_first - 1ull) : -1;
    rs->range_next = tot.range_next ? (int64_t)(tot.range_next - 1ull) : -1;
    rs->f_chunks = nf ? tot.n_chunks : 0; rs->f_words = nf ? (int64_t)tot.n_words + 4 : 0; rs->f_events = tot.n_events;
    rs->dev_bytes = nf * 4 + ((int64_t)tot.n_words + 4) * 4 + (int64_t)tot.n_chunks * (int64_t)sizeof(tcmi_fast_chunk) + (int64_t)tot.n_runs * 4 +
                    (int64_t)tot.n_events * 4;
    return TCMI_OK;
}

// ---- tally_stream_kernel: the reads the packer left out (spans above TCMI_D_MAXLEN), straight from the inflated BAM stream -------
// One wavefront per read.  The CIGAR is walked op by op (wave-uniform), the lanes take the positions of an op 64 at a time and add
// each token to the count matrix with a global atomic — the token rules of tally.hip's tally_read_general: a
// matched base counts by its letter, a deleted position counts X unless an insertion follows the deletion's last base ("*+.."),
// the last reference base in front of an insertion counts I, every position from pos to the end counts coverage (M, =, X, D, N).
namespace {
__device__ inline bool st_ins_after(const uint8_t *cg, int n, int k)        // htslib resolve_cigar2's peek at the last base of op k
{
    if (k + 1 >= n) return false;
    const uint32_t op2 = ld_u32(cg + 4 * (size_t)(k + 1)) & 0xFu;
    int64_t tot = 0;
    if (op2 == 1) {
        tot = ld_u32(cg + 4 * (size_t)(k + 1)) >> 4;
        for (int j = k + 2; j < n; ++j) {
            const uint32_t c = ld_u32(cg + 4 * (size_t)j), o = c & 0xFu;
            if (o == 1) tot += c >> 4;
            else if (o != 6) break;
        }
    } else if (op2 == 6 && k + 2 < n) {
        for (int j = k + 2; j < n; ++j) {
            const uint32_t c = ld_u32(cg + 4 * (size_t)j), o = c & 0xFu;
            if (o == 1) tot += c >> 4;
            else if (consumes_ref(o)) break;
        }
    }
    return tot > 0;
}

__global__ __launch_bounds__(256) void tally_stream_kernel(PackSrc s, const uint32_t *gen_idx, uint32_t n_gen, int32_t *counts, int64_t ld, int32_t L)
{
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= n_gen) return;
    const ReadView v = view(s, (int64_t)gen_idx[w]);
    auto add = [&](int col, int32_t p) { if ((uint32_t)p < (uint32_t)L) atomicAdd(&counts[(int64_t)col * ld + p], 1); };
    int32_t x = v.pos + s.pos_shift, y = 0;
    const int32_t x0 = x;
    for (uint32_t k = 0; k < v.n_cigar; ++k) {
        const uint32_t c = ld_u32(v.cigar + 4 * (size_t)k), op = c & 0xFu;
        const int32_t len = (int32_t)(c >> 4);
        if (consumes_ref(op)) {
            const bool ins = len > 0 && st_ins_after(v.cigar, (int)v.n_cigar, (int)k);
            if (is_match(op)) {
                for (int32_t j = lane; j < len; j += 64) {
                    const int32_t q = y + j;
                    const uint32_t nib = q < v.l_seq ? nib_at(v.seq, q) : 15u;          // past SEQ -> 'N'
                    if (__popc(nib) == 1) { const int b = __ffs(nib) - 1; add(b == 0 ? TCMI_A : b == 1 ? TCMI_C : b == 2 ? TCMI_G : TCMI_T, x + j); }
                }
            } else if (op == 2) {
                const int32_t nx = ins ? len - 1 : len;                              // "*+.." does not count X
                for (int32_t j = lane; j < nx; j += 64) add(TCMI_X, x + j);
            }
            if (ins && lane == 0) add(TCMI_I, x + len - 1);
            x += len;
        }
        if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) y += len;
    }
    for (int32_t p = x0 + lane; p < x; p += 64) add(TCMI_COV, p);
}
} // namespace

// the long reads of a device-decoded read set into the count matrix (tcmi_launch_tally calls it behind the packed set's kernel)
int tcmi_launch_tally_stream(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld, int32_t *d_counts)
{
    if (rs->s_reads <= 0) return TCMI_OK;
    if (!rs->d_stream || rs->arena_epoch != ctx->arena_epoch || rs->device != ctx->device)
        return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "the read set's long reads lie in a decoded stream that is gone (another upload on this context): upload it again");
    PackSrc s = {};
    s.stream = rs->d_stream; s.rec_off = rs->d_rec_off; s.mode = 1; s.n = rs->n_reads; s.pos_shift = 0;
    (void)hipGetLastError();
    tcmi_prof_begin(ctx, TCMI_K_TALLY_GENERAL);
    hipLaunchKernelGGL(tally_stream_kernel, dim3((unsigned)((rs->s_reads + 3) / 4)), dim3(256), 0, ctx->stream, s, rs->d_gen_idx, (uint32_t)rs->s_reads,
                       d_counts, ld, (int32_t)L);
    tcmi_prof_end(ctx, TCMI_K_TALLY_GENERAL);
    TCMI_HIP(ctx, hipGetLastError());
    return TCMI_OK;
}

// the flat arrays of struct tcmi_reads -> device (one arena block) -> tcmi_pack_on_device
int tcmi_upload_and_pack_on_device(tcmi_ctx *ctx, const tcmi_reads *r, tcmi_readset *rs, uint32_t *why)
{
    *why = 0;
    const int64_t n = r->n_reads;
    const size_t n_cig = n ? (size_t)r->cigar_off[n] : 0, n_seq = n ? (size_t)r->seq_off[n] : 0;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t sz[8] = {al((size_t)n * 4), al((size_t)n * 2), al((size_t)n * 4), r->tid ? al((size_t)n * 4) : 0, al((size_t)(n + 1) * 8),
                          al(n_cig * 4 + 64), al((size_t)(n + 1) * 8), al(n_seq + 128)};
    size_t src_bytes = 0;
    for (size_t b : sz) src_bytes += b + 256;
    const size_t tmp_bytes = al((size_t)n * 4) * 11 + al((size_t)((n + PB - 1) / PB + 1) * 8) * 3 + 4096 + 16 * 256;
    int rc = arena_reserve(ctx, src_bytes + tmp_bytes);
    if (rc) return rc;
    PackSrc s = {};
    s.mode = 0; s.n = n; s.pos_shift = 0;
    struct Item { const void *h; size_t bytes, room; const void **d; };
    const Item items[8] = {{r->pos, (size_t)n * 4, sz[0], (const void **)&s.pos}, {r->flag, (size_t)n * 2, sz[1], (const void **)&s.flag},
                           {r->l_qseq, (size_t)n * 4, sz[2], (const void **)&s.l_qseq}, {r->tid, (size_t)n * 4, sz[3], (const void **)&s.tid},
                           {r->cigar_off, (size_t)(n + 1) * 8, sz[4], (const void **)&s.cigar_off}, {r->cigar, n_cig * 4, sz[5], (const void **)&s.cigar},
                           {r->seq_off, (size_t)(n + 1) * 8, sz[6], (const void **)&s.seq_off}, {r->seq, n_seq, sz[7], (const void **)&s.seq}};
    for (const Item &it : items) {
        if (!it.h || it.room == 0) { *it.d = nullptr; continue; }
        char *d = (char *)arena_take(ctx, it.room);
        *it.d = d;
        if (it.bytes) TCMI_HIP(ctx, hipMemcpyAsync(d, it.h, it.bytes, hipMemcpyHostToDevice, ctx->stream));
        TCMI_HIP(ctx, hipMemsetAsync(d + it.bytes, 0, it.room - it.bytes, ctx->stream));   // the 24-byte window loads of fetch32 may run past the last read
    }
    if (n == 0) { rs->packed_on_device = 1; return TCMI_OK; }
    return tcmi_pack_on_device(ctx, &s, rs, why);
}

// the device half of Events.ExtractInserts for a device-decoded read set: every read that can reach a candidate column as a 48-byte
// entry (ins_entries_kernel), in file order per column, in the context's pinned scratch (valid until the context's next call)
static int collect_ins_entries(tcmi_ctx *ctx, const tcmi_readset *rs, int32_t n_pos, const int64_t *positions, uint32_t flag_filter, int ignore_orphans,
                               std::vector<int64_t> &off, std::vector<int32_t> &cnt, const tcmi_dev_entry **ents_out, std::vector<uint8_t> &long_text)
{
    if (!ctx || !rs || n_pos < 0 || (n_pos > 0 && !positions)) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");

    if (rs->s_reads > 0)
        return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "long reads lie outside the packed set: their tokens are not looked at here (host sweep)");
    if (!rs->d_stream || rs->arena_epoch != ctx->arena_epoch || rs->device != ctx->device)
        return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "the read set's decoded stream is no longer (or never was) resident on this context: host sweep");
    for (int32_t k = 1; k < n_pos; ++k)
        if (positions[k] <= positions[k - 1]) return tcmi_fail(ctx, TCMI_E_ARG, "positions must ascend");
    off.assign((size_t)n_pos + 1, 0); cnt.assign((size_t)n_pos, 0); *ents_out = nullptr; long_text.clear();
    if (n_pos == 0) return TCMI_OK;
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nf = rs->f_reads;
    // The kept reads ascend by position (the device packer takes nothing else): which of them can reach each column is a
    // binary search on the device; what comes back is two numbers per column.  Scratch (device + pinned host) belongs to the
    // context and only grows: hipMalloc / hipFree per file cost more than the kernels (hipFree waits for the whole device).
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    auto scratch = [&](size_t dev_bytes, size_t host_bytes) -> int {
        if (ctx->tok_dev_cap < dev_bytes) {
            if (ctx->tok_dev) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->tok_dev); ctx->tok_dev = nullptr; ctx->tok_dev_cap = 0; }
            const size_t want = dev_bytes + dev_bytes / 4 + (1 << 20);
            if (hipMalloc((void **)&ctx->tok_dev, want) != hipSuccess) return tcmi_fail(ctx, TCMI_E_NOMEM, "insert-token scratch (%zu bytes)", want);
            ctx->tok_dev_cap = want;
        }
        if (ctx->tok_host_cap < host_bytes) {
            if (ctx->tok_host) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->tok_host); ctx->tok_host = nullptr; ctx->tok_host_cap = 0; }
            const size_t want = host_bytes + host_bytes / 4 + (1 << 20);
            if (hipHostMalloc((void **)&ctx->tok_host, want, hipHostMallocDefault) != hipSuccess) return tcmi_fail(ctx, TCMI_E_NOMEM, "insert-token host scratch (%zu bytes)", want);
            ctx->tok_host_cap = want;
        }
        return TCMI_OK;
    };
    std::vector<int32_t> cols((size_t)n_pos);
    for (int32_t k = 0; k < n_pos; ++k) cols[(size_t)k] = (int32_t)(positions[k] - 1);
    const size_t b_cols = al((size_t)n_pos * 4), b_lo = al((size_t)n_pos * 8), b_off = al(((size_t)n_pos + 1) * 8);
    {
        const int rc = scratch(b_cols + 2 * b_lo + b_off, 2 * b_lo);
        if (rc) return rc;
    }
    int32_t *d_cols = (int32_t *)ctx->tok_dev;
    int64_t *d_lo = (int64_t *)(ctx->tok_dev + b_cols), *d_hi = (int64_t *)(ctx->tok_dev + b_cols + b_lo), *d_off = (int64_t *)(ctx->tok_dev + b_cols + 2 * b_lo);
    int64_t *h_lo = (int64_t *)ctx->tok_host, *h_hi = (int64_t *)(ctx->tok_host + b_lo);
    const int32_t max_len = (int32_t)std::min<int64_t>(std::max<int64_t>(rs->max_len, 1), TCMI_D_MAXLEN);
    TCMI_HIP(ctx, hipMemcpyAsync(d_cols, cols.data(), (size_t)n_pos * 4, hipMemcpyHostToDevice, ctx->stream));
    (void)hipGetLastError();
    uint32_t *d_unsorted = (uint32_t *)d_off;                   // (the offsets go there later)
    TCMI_HIP(ctx, hipMemsetAsync(d_unsorted, 0, 4, ctx->stream));
    if (nf > 1) hipLaunchKernelGGL(ins_sorted_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, ctx->stream, rs->d_cpos, nf, d_unsorted);
    hipLaunchKernelGGL(ins_ranges_kernel, dim3((unsigned)((n_pos + 63) / 64)), dim3(64), 0, ctx->stream, rs->d_cpos, nf, d_cols, n_pos, max_len, d_lo, d_hi);
    TCMI_HIP(ctx, hipGetLastError());
    TCMI_HIP(ctx, hipMemcpyAsync(h_lo, d_lo, (size_t)n_pos * 8, hipMemcpyDeviceToHost, ctx->stream));
    TCMI_HIP(ctx, hipMemcpyAsync(h_hi, d_hi, (size_t)n_pos * 8, hipMemcpyDeviceToHost, ctx->stream));
    uint32_t unsorted = 0;
    TCMI_HIP(ctx, hipMemcpyAsync(&unsorted, d_unsorted, 4, hipMemcpyDeviceToHost, ctx->stream));
    TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (unsorted) return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "reads are not sorted by position: host sweep");
    const std::vector<int64_t> lo_v(h_lo, h_lo + n_pos), hi_v(h_hi, h_hi + n_pos);   // (the scratch below may move)
    for (int32_t k = 0; k < n_pos; ++k) off[(size_t)k + 1] = off[(size_t)k] + std::max<int64_t>(0, hi_v[(size_t)k] - lo_v[(size_t)k]);
    const int64_t total = off[(size_t)n_pos];
    for (int32_t k = 0; k < n_pos; ++k) cnt[(size_t)k] = (int32_t)(off[(size_t)k + 1] - off[(size_t)k]);
    const tcmi_dev_entry *ents = nullptr;
    constexpr size_t LONG_TEXT_CAP = 4u << 20;                  // bases of insertions longer than 12 on the candidate columns of one call
    if (total > 0) {
        const size_t b_head = b_cols + 2 * b_lo + b_off, b_ent = al((size_t)total * sizeof(tcmi_dev_entry));
        const int rc = scratch(b_head + b_ent + 256 + LONG_TEXT_CAP, std::max(2 * b_lo, b_ent));
        if (rc) return rc;
        // (a regrown device buffer lost the columns and ranges: they are sent again — all tiny)
        d_cols = (int32_t *)ctx->tok_dev;
        d_lo = (int64_t *)(ctx->tok_dev + b_cols); d_off = (int64_t *)(ctx->tok_dev + b_cols + 2 * b_lo);
        InsArgs a;
        a.src = {};
        a.src.stream = rs->d_stream; a.src.rec_off = rs->d_rec_off; a.src.mode = 1; a.src.n = rs->n_reads;
        a.c_idx = rs->d_cidx;
        a.out = (tcmi_dev_entry *)(ctx->tok_dev + b_head);
        a.cols = d_cols; a.lo = d_lo; a.off = d_off;
        a.n_cand = n_pos; a.flag_filter = flag_filter; a.ignore_orphans = ignore_orphans;
        a.long_cursor = (uint32_t *)(ctx->tok_dev + b_head + b_ent); a.long_text = (uint8_t *)(ctx->tok_dev + b_head + b_ent + 256); a.long_cap = (uint32_t)LONG_TEXT_CAP;
        TCMI_HIP(ctx, hipMemsetAsync(a.long_cursor, 0, 4, ctx->stream));
        TCMI_HIP(ctx, hipMemcpyAsync(d_cols, cols.data(), (size_t)n_pos * 4, hipMemcpyHostToDevice, ctx->stream));
        TCMI_HIP(ctx, hipMemcpyAsync(d_lo, lo_v.data(), (size_t)n_pos * 8, hipMemcpyHostToDevice, ctx->stream));
        TCMI_HIP(ctx, hipMemcpyAsync(d_off, off.data(), ((size_t)n_pos + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        (void)hipGetLastError();
        hipLaunchKernelGGL(ins_entries_kernel, dim3((unsigned)((total + PB - 1) / PB)), dim3(PB), 0, ctx->stream, a);
        TCMI_HIP(ctx, hipGetLastError());
        TCMI_HIP(ctx, hipMemcpyAsync(ctx->tok_host, a.out, (size_t)total * sizeof(tcmi_dev_entry), hipMemcpyDeviceToHost, ctx->stream));
        uint32_t long_used = 0;
        TCMI_HIP(ctx, hipMemcpyAsync(&long_used, a.long_cursor, 4, hipMemcpyDeviceToHost, ctx->stream));
        TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));       // (lo_v / off / cols were pageable: their copies are done)
        ents = (const tcmi_dev_entry *)ctx->tok_host;
        if (long_used) {                                        // (rare: a long insertion on a candidate column) its bases
            long_text.resize(std::min<size_t>(long_used, LONG_TEXT_CAP));
            TCMI_HIP(ctx, hipMemcpy(long_text.data(), a.long_text, long_text.size(), hipMemcpyDeviceToHost));
        }
    }
    *ents_out = ents;
    return TCMI_OK;
}

extern "C" int tcmi_readset_modal_tokens(tcmi_ctx *ctx, const tcmi_readset *rs, int32_t n_pos, const int64_t *positions,
                                         int32_t min_base_quality, uint32_t flag_filter, int ignore_orphans, int64_t max_depth,
                                         int ignore_overlaps, char *tokens, int64_t tokens_cap, int64_t *token_off, int64_t *n_tokens,
                                         int32_t *status_flags)
{
    if (!ctx || !rs || n_pos < 0 || (n_pos > 0 && (!positions || !tokens || !token_off || !n_tokens)))
        return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    if (!rs->parts.empty())
        return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "a read set of sub-ranges (tcmi_split_step): collect its entries (tcmi_readset_ins_entries) and vote on them (tcmi_modal_from_entries)");
    if (n_pos == 0) { if (status_flags) *status_flags = 0; return TCMI_OK; }
    std::vector<int64_t> off;
    std::vector<int32_t> cnt;
    std::vector<uint8_t> long_text;
    const tcmi_dev_entry *ents = nullptr;
    {
        const int rc = collect_ins_entries(ctx, rs, n_pos, positions, flag_filter, ignore_orphans, off, cnt, &ents, long_text);
        if (rc) return rc;
    }
    const int64_t nf = rs->f_reads;
    // the other mate of an overlapping pair, looked at on one reference position (rare: a pair with a deletion on a candidate column)
    const tcmi_prober prober = [&](const std::vector<tcmi_probe_req> &req, std::vector<tcmi_probe_res> &res) -> int {
        const size_t n = req.size();
        std::vector<int64_t> idx(n);
        std::vector<int32_t> ref(n);
        std::vector<uint32_t> out(n);
        for (size_t t = 0; t < n; ++t) {
            if (req[t].idx < 0 || req[t].idx >= nf) return tcmi_fail(ctx, TCMI_E_ARG, "internal: probe of read %lld", (long long)req[t].idx);
            idx[t] = req[t].idx; ref[t] = req[t].ref;
        }
        char *buf = nullptr;
        hipError_t e = hipMalloc((void **)&buf, n * 16);
        if (e != hipSuccess) return tcmi_fail(ctx, TCMI_E_NOMEM, "probe buffers: %s", hipGetErrorString(e));
        ProbeArgs a;
        a.src = {};
        a.src.stream = rs->d_stream; a.src.rec_off = rs->d_rec_off; a.src.mode = 1; a.src.n = rs->n_reads;
        a.c_idx = rs->d_cidx;
        a.idx = (const int64_t *)buf; a.ref = (const int32_t *)(buf + n * 8); a.out = (uint32_t *)(buf + n * 12); a.n = (int32_t)n;
        e = hipMemcpyAsync((void *)a.idx, idx.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync((void *)a.ref, ref.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            (void)hipGetLastError();
            hipLaunchKernelGGL(ins_probe_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, a);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out.data(), a.out, n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(buf);
        if (e != hipSuccess) return tcmi_fail(ctx, TCMI_E_HIP, "probe kernel failed: %s", hipGetErrorString(e));
        for (size_t t = 0; t < n; ++t) res[t] = tcmi_probe_res{(uint8_t)(out[t] & 1u), (uint8_t)((out[t] >> 8) & 15u), (uint8_t)(out[t] >> 16)};
        return TCMI_OK;
    };
    return tcmi_modal_from_dev_entries(n_pos, ents, off.data(), cnt.data(), min_base_quality, max_depth, ignore_overlaps, &prober, tokens,
                                       tokens_cap, token_off, n_tokens, status_flags, long_text.data(), long_text.size());
}

// The entries themselves (48 bytes each, opaque to the caller) instead of the vote: ranks that share ONE file (BASELINE configs[4])
// each collect the entries of the candidate columns from the records of their own block range and send them to the rank that
// calls; concatenated in rank order (= file order) they are what tcmi_readset_modal_tokens votes on (tcmi_modal_from_entries).
extern "C" int tcmi_readset_ins_entries(tcmi_ctx *ctx, const tcmi_readset *rs, int32_t n_pos, const int64_t *positions, uint32_t flag_filter,
                                        int ignore_orphans, void *entries, int64_t entries_cap, int64_t *ent_off, uint8_t *long_text,
                                        int64_t long_cap, int64_t *long_used)
{
    if (!ctx || !rs || n_pos < 0 || !ent_off || (n_pos > 0 && !positions)) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    static_assert(sizeof(tcmi_dev_entry) == TCMI_INS_ENTRY_BYTES, "include/tcmi.h promises 48-byte entries");
    std::vector<int64_t> off;
    std::vector<int32_t> cnt;
    std::vector<uint8_t> text;
    const tcmi_dev_entry *ents = nullptr;
    if (!rs->parts.empty()) {
        // a read set of sub-ranges: the parts' entries per column one behind the other — file order —, the text offsets of a part's
        // long insertions moved behind the texts of the parts in front of it (what rank 0 does with the ranks' pieces)
        const size_t P = rs->parts.size();
        std::vector<std::vector<int64_t>> p_off(P);
        std::vector<std::vector<tcmi_dev_entry>> p_ent(P);
        std::vector<int64_t> p_base(P, 0);
        for (size_t p = 0; p < P; ++p) {
            const tcmi_readset::Part &pt = rs->parts[p];
            p_off[p].assign((size_t)n_pos + 1, 0);
            p_base[p] = (int64_t)text.size();
            if (pt.rs->n_piled == 0 || pt.rs->f_reads == 0) continue;
            std::vector<uint8_t> t1;
            const tcmi_dev_entry *e1 = nullptr;
            const int rc = collect_ins_entries(pt.cx, pt.rs, n_pos, positions, flag_filter, ignore_orphans, p_off[p], cnt, &e1, t1);
            if (rc) return tcmi_fail(ctx, rc, "%s", pt.cx->err.c_str());
            const int64_t n1 = p_off[p][(size_t)n_pos];
            if (n1) p_ent[p].assign(e1, e1 + n1);                // (the part's pinned scratch is its context's: copied out before the next call there)
            if (n1 && p_base[p]) {
                const int rc2 = tcmi_ins_entries_rebase(p_ent[p].data(), n1, p_base[p]);
                if (rc2) return rc2;
            }
            text.insert(text.end(), t1.begin(), t1.end());
        }
        ent_off[0] = 0;
        for (int32_t k = 0; k < n_pos; ++k) {
            int64_t n = 0;
            for (size_t p = 0; p < P; ++p) n += p_off[p][(size_t)k + 1] - p_off[p][(size_t)k];
            ent_off[k + 1] = ent_off[k] + n;
        }
        if (long_used) *long_used = (int64_t)text.size();
        if (ent_off[n_pos] > entries_cap || (int64_t)text.size() > long_cap)
            return tcmi_fail(ctx, TCMI_E_ARG, "entry buffer too small: %lld entries, %zu bytes of long insertions (ent_off / long_used say what is needed)",
                             (long long)ent_off[n_pos], text.size());
        tcmi_dev_entry *dst = static_cast<tcmi_dev_entry *>(entries);
        for (int32_t k = 0; k < n_pos; ++k)
            for (size_t p = 0; p < P; ++p) {
                const int64_t a = p_off[p][(size_t)k], b = p_off[p][(size_t)k + 1];
                if (b > a) { std::memcpy(dst, p_ent[p].data() + a, (size_t)(b - a) * sizeof(tcmi_dev_entry)); dst += b - a; }
            }
        if (!text.empty()) std::memcpy(long_text, text.data(), text.size());
        return TCMI_OK;
    }
    if (rs->n_piled == 0 || rs->f_reads == 0) {                 // (no kept reads in this range: no entries)
        for (int32_t k = 0; k <= n_pos; ++k) ent_off[k] = 0;
        if (long_used) *long_used = 0;
        return TCMI_OK;
    }
    const int rc = collect_ins_entries(ctx, rs, n_pos, positions, flag_filter, ignore_orphans, off, cnt, &ents, text);
    if (rc) return rc;
    for (int32_t k = 0; k <= n_pos; ++k) ent_off[k] = off[(size_t)k];
    if (long_used) *long_used = (int64_t)text.size();
    if (off[(size_t)n_pos] > entries_cap || (int64_t)text.size() > long_cap)
        return tcmi_fail(ctx, TCMI_E_ARG, "entry buffer too small: %lld entries, %zu bytes of long insertions (ent_off / long_used say what is needed)",
                         (long long)off[(size_t)n_pos], text.size());
    if (off[(size_t)n_pos]) std::memcpy(entries, ents, (size_t)off[(size_t)n_pos] * sizeof(tcmi_dev_entry));
    if (!text.empty()) std::memcpy(long_text, text.data(), text.size());
    return TCMI_OK;
}

// ... and the vote over entries gathered from several read sets (HOST): per column the concatenation, in file order, of the pieces the
// ranks sent; `long_base[k]` rebases the text offsets of piece k's long insertions (their texts concatenated in `long_text`).
extern "C" int tcmi_modal_from_entries(int32_t n_pos, void *entries, const int64_t *ent_off, int32_t min_base_quality, int64_t max_depth,
                                       int ignore_overlaps, const uint8_t *long_text, int64_t long_bytes, char *tokens, int64_t tokens_cap,
                                       int64_t *token_off, int64_t *n_tokens, int32_t *status_flags)
{
    if (n_pos < 0 || !ent_off || (n_pos > 0 && (!tokens || !token_off || !n_tokens))) return tcmi_fail(nullptr, TCMI_E_ARG, "null argument");
    std::vector<int32_t> cnt((size_t)std::max(n_pos, 0));
    for (int32_t k = 0; k < n_pos; ++k) cnt[(size_t)k] = (int32_t)(ent_off[k + 1] - ent_off[k]);
    return tcmi_modal_from_dev_entries(n_pos, static_cast<const tcmi_dev_entry *>(entries), ent_off, cnt.data(), min_base_quality, max_depth, ignore_overlaps,
                                       nullptr, tokens, tokens_cap, token_off, n_tokens, status_flags, long_text, (size_t)std::max<int64_t>(long_bytes, 0));
}

extern "C" int tcmi_ins_entries_rebase(void *entries, int64_t n_entries, int64_t long_base)
{
    if (n_entries < 0 || (n_entries > 0 && !entries) || long_base < 0) return tcmi_fail(nullptr, TCMI_E_ARG, "bad argument");
    tcmi_dev_entry *e = static_cast<tcmi_dev_entry *>(entries);
    for (int64_t i = 0; i < n_entries; ++i)
        if ((e[i].bits & 0x40) && !(e[i].bits & 0x80)) {        // the key says where the insertion's bases lie: bits 8-39
            const uint64_t at = ((e[i].key >> 8) & 0xFFFFFFFFull) + (uint64_t)long_base;
            if (at > 0xFFFFFFFFull) return tcmi_fail(nullptr, TCMI_E_UNSUPPORTED, "more than 4 GiB of long insertions on the candidate columns");
            e[i].key = (e[i].key & ~(0xFFFFFFFFull << 8)) | (at << 8);
        }
    return TCMI_OK;
}
