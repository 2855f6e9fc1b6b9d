// bgzf_decode.hip — DEVICE: the BGZF blocks of a BAM file -> the inflated BAM byte stream + the record starts of every block
// (pysam / htslib's role for indexing.py:19,96-100; SAM spec §4.1 BGZF, RFC 1951 DEFLATE), in two kernels:
//
//   bgzf_symbols   Huffman symbols -> tokens.  A block's symbols are decoded by 32 lanes (two blocks per workgroup; 64 when a
//                  payload exceeds 4 KB: one block per workgroup, staged a window at a time beyond 16 KB), speculatively in parallel.
//   bgzf_copy      tokens -> bytes: the LZ77 copies through an LDS ring of the recent output, the chain of BAM records, the flush —
//                  and the block's CRC-32 against its trailer, taken from the ring while a segment is flushed.
//
// Why two kernels.  A deflate stream is serial twice over: the position of symbol k + 1 is known only when symbol k is decoded,
// and a match may copy what the previous match produced.  Round 2's one-kernel decoder walked both chains in one wavefront, one
// symbol at a time: ~60 wave-instructions per symbol, all of them issued for a single useful lane, and the kernel was bound by
// instruction issue.  Here the first chain is cut into pieces: lane c starts decoding at bit s_c = start + c * chunk — in the
// middle of nowhere, except for lane 0 — and notes where its symbols cross into each new stretch of bits.  Huffman streams
// resynchronise: after a few dozen bits a decoder that started on a wrong bit starts a symbol on a right one, and from there on it
// IS the serial decoder.  A lane stops when a symbol of its own starts on a position that the lane in front of it has noted: from
// there the two would decode the same (pass A).  Starting from lane 0 the chain of these meeting points says which lane holds the
// true symbols of which bit range, and how many they are; the lanes then put the true tokens in order (gathered in output order, or —
// blocks of thousands of tokens — moved row by row; a lane that overflows its scratch sends the block through pass B: the true
// ranges once more, tokens straight to their places).  Nothing in this depends on luck or timing: a lane that never meets anyone
// simply goes on to the block's end, and
// lane 0 alone is the serial decoder.  (tools/spec_inflate_proto.py: the same scheme in Python.)
//
// Tokens (32 bits): literal 1<<31 | byte — or two literals in one token, 1<<31 | 1<<24 | byte | next byte << 8: a lane whose symbol is a
// literal of at most nine bits takes the next code in the same round if that is such a literal too and starts in the same stretch
// of bits (meeting points stay round starts; literal-heavy chunks are the ones whose lanes take longest) —; match len (9 bits) | (dist - 1) << 9; raw 1<<30 | len << 17 | offset of the bytes from
// the block's payload start (a stored deflate block, in pieces of <= 8 191 bytes).  bgzf_copy takes 64 tokens at a time: an
// inclusive scan of the lengths gives every token its output position, the literals of a stretch go to the ring at once, the
// matches one after the other (TCMI_LM_ASM: a byte a lane up to 64 bytes, an aligned dword a lane beyond — the LDS takes unaligned
// words at about a cycle a LANE —; teams of eight lanes for up to eight independent short matches in files of short tokens; in
// files under 4 : 1 the far matches of up to 8 bytes are finished in the batch's set-up, straight from the flushed stream).
// A launch takes a range of the file's blocks and the token array's base: a file whose tokens do not fit the context's scratch is
// decoded a batch of blocks at a time (bam_device.hip: decode_enqueue).
//
// Bit / byte work, bound by instruction issue and the LDS pipe, not by HBM and not a contraction: no MFMA.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "bgzf_device.h"

namespace {

#ifndef TCMI_COPY_RING
#define TCMI_COPY_RING 8192
#endif
#ifndef TCMI_COPY_SEG
#define TCMI_COPY_SEG 2048
#endif
constexpr int CWIN = TCMI_COPY_RING, CWMASK = CWIN - 1;       // bgzf_copy's ring of recent output
constexpr int CSEG = TCMI_COPY_SEG;
// a round of bgzf_copy writes the literals of up to CSEG + 258 bytes ahead of the match it copies: what a match may still read
// from the ring ends that much earlier; a source further back has been flushed (CWIN >= 2 CSEG + 522)
constexpr int CNEAR = CWIN - CSEG - 264;
static_assert(CWIN >= 2 * CSEG + 528 && (CWIN & (CWIN - 1)) == 0 && CWIN % CSEG == 0, "a far match must find its source flushed");
#ifndef TCMI_COPY_TEAMS
#define TCMI_COPY_TEAMS 1
#endif
constexpr int FAR_WORDS = 128;
constexpr int TEAM_BATCH_BYTES = 1536;               // bgzf_copy: a batch of 64 tokens this short (<= 24 bytes a token) copies its matches in teams                      // bgzf_copy: words of LDS in which the sources of a batch's far matches are parked
constexpr uint32_t TOK_LIT = 1u << 31, TOK_RAW = 1u << 30;
constexpr uint32_t TOK_LIT2 = 1u << 24;              // a literal token that carries TWO bytes (the second in bits 8 - 15): bits 24 - 25 = literals - 1
constexpr uint32_t RAW_PIECE = 8191;
constexpr int CL_SLAB = 496;                        // bit positions of the code-length stream looked up at a time
// bgzf_symbols<NB>: NB BGZF blocks per workgroup — one wavefront each for header and tables, then wavefront 0 decodes all of
// them, 64 / NB lanes per block.  NB = 4 (16 lanes: one row of the data-parallel moves) unless the payloads are too large for
// four of them to share a compute unit's LDS; then NB = 1.

struct BlkTabs {                                    // per block
    tab_t ll[1 << LL_ROOT];                         // (first: the code-length stream's table of all positions, CL_SLAB + 16 entries)
    tab_t dt[1 << D_ROOT];                          // (first: the code-length code's root table)
    tab_t long_ll[288], long_d[32];                 // entries of the codes longer than the root bits, in canonical order
    // per such length, for the look-up by range compare: the end of its codes, left-aligned in 15 bits (ascending: canonical codes are
    // ordered by length), and first code | index of its first entry in long_* << 16
    uint32_t lim_ll[16 - LL_ROOT], fb_ll[16 - LL_ROOT], lim_d[16 - D_ROOT], fb_d[16 - D_ROOT];
    // what the block's wavefront hands to the decoding wavefront and gets back
    uint32_t pos;               // first symbol / behind the end-of-block code
    uint32_t end;               // first bit behind the payload
    uint32_t ntok;              // tokens so far
    uint32_t err;               // ST_*
    uint32_t go;                // 1: symbols to decode at pos
    uint32_t last;              // 1: the stream's last deflate block
};
struct HdrScratch {                                 // per block, while its header is decoded and its tables are built
    uint8_t lens[320];
    uint8_t cll[20];
    uint16_t sym_ll[288], sym_d[32], sym_cl[20];
    uint16_t cnt_ll[16], cnt_d[16], cnt_cl[16];
    uint32_t rs[6];
};
#ifndef TCMI_SYM_MOVE
#define TCMI_SYM_MOVE 8                             // bgzf_symbols: tokens a lane has in flight when the tokens are gathered to their places
#endif
#ifndef TCMI_SYM_ROWS
#define TCMI_SYM_ROWS 32                            // bgzf_symbols: parked rows a turn of the row-by-row mover takes (a load each, all in flight)
#endif
#ifndef TCMI_SYM_WAVES
#define TCMI_SYM_WAVES 5                            // bgzf_symbols: wavefronts per SIMD the register budget is cut for (5: 96 VGPRs; with 4 — 128 —
                                                    // a BAM's 2 094 workgroups fill every CU's register file: 182 us instead of 173)
#endif
constexpr int RING = 8;
struct PassALds {
    uint2 ring[64][RING];       // pass A, per lane: {first symbol start in a stretch, symbols decoded before it}
    uint2 rec[64];              // per lane: {state | target << 8, position}
};
// The header scratch and pass A's notes share their LDS: a workgroup barrier separates the two phases, and with 2.3 KB less a
// workgroup of two blocks stays under the 18.2 KB at which nine of them fit a compute unit.
template <int NB>
struct SymLds {
    BlkTabs b[NB];
    union {
        HdrScratch h[NB];
        PassALds a;
    };
};
static_assert((CL_SLAB + 16) * 4 <= sizeof(tab_t) * (1 << LL_ROOT), "the code-length position table borrows the literal/length table's LDS");

struct SymArgs {
    const uint32_t *__restrict__ file32;
    const BlockDesc *blocks;
    uint32_t *tokens;           // block b's tokens at tokens + blocks[b].tok: tok_cap final ones, then tok_cap of scratch
    uint32_t *n_tok;            // [n_blocks]
    uint32_t *status;           // [n_blocks]
    int32_t n_blocks;           // (the launch's blocks end here)
    int32_t first_block;        // ... and start here: workgroup 0's first block
    uint32_t pay_dwords;        // dwords of dynamic LDS per block behind SymLds: the largest block's payload + slack
    uint32_t win_dwords;        // bgzf_symbols<1, true>: dwords of payload staged at a time (a window that moves along the block)
    uint32_t gather_max;        // a pass with more true tokens than this moves them row by row (else: gathered in output order)
    uint32_t shift_bias;        // (A/B) pass A's stretches this many powers of two shorter than chunk / 4 .. chunk / 2
    uint32_t scratch_div;       // a lane's scratch is cut to 1 / scratch_div of its share (tests: lanes overflow and the block goes through pass B)
    uint64_t *stamps;           // diagnostic (TCMI_INFLATE_STAMPS): 16 words per block, s_memtime at the phase boundaries; or null
};
#define TCMI_STAMP(buf_, blk_, k_) do { if (buf_) { if ((threadIdx.x & 63) == 0) (buf_)[(size_t)(blk_) * 16 + (k_)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define TCMI_STAMP_ADD(buf_, blk_, k_, v_) do { if (buf_) { if ((threadIdx.x & 63) == 0) (buf_)[(size_t)(blk_) * 16 + (k_)] += (v_); } } while (0)

// 32 bits of the staged payload from bit p on (any lane, any position)
__device__ __forceinline__ uint32_t peek32(const uint32_t *pay, uint32_t p)
{
    const uint32_t w = p >> 5;
    return __builtin_amdgcn_alignbit(pay[w + 1], pay[w], p);
}
// ... 64 bits: what one symbol can take (15 + 5 bits of a length, 15 + 13 of a distance)
__device__ __forceinline__ void peek64(const uint32_t *pay, uint32_t p, uint32_t &lo, uint32_t &hi)
{
    const uint32_t w = p >> 5;
    const uint32_t w0 = pay[w], w1 = pay[w + 1], w2 = pay[w + 2];
    lo = __builtin_amdgcn_alignbit(w1, w0, p);
    hi = __builtin_amdgcn_alignbit(w2, w1, p);
}

// The codes longer than the root bits: one ready-made table entry per such code, in canonical order.  A lane finds its code
// without a walk through memory: canonical codes are ordered by length, so the code's first 15 bits, left-aligned, lie below the
// end of exactly the lengths that are long enough — counting the ends at or below them gives the length.
template <int ROOT>
__device__ __forceinline__ void build_long(const uint16_t *cnt, const uint16_t *sym, const uint32_t *rs, int kind, tab_t *out, uint32_t *lim_out,
                                           uint32_t *fb_out)
{
    const int lane = threadIdx.x & 63;
    uint32_t first = uni(rs[0]);
    const uint32_t at = uni(rs[1]);
    uint32_t n = 0;
    uint32_t cs[15 - ROOT];
    uint32_t my_lim = 0x10000u, my_fb = 0;                 // (entry 15 - ROOT: above every code)
#pragma unroll
    for (int len = ROOT + 1; len <= 15; ++len) {
        const uint32_t c = uni(cnt[len]);
        cs[len - ROOT - 1] = c;
        if (lane == len - ROOT - 1) { my_lim = (first + c) << (15 - len); my_fb = (first & 0xFFFFu) | (n << 16); }
        first = (first + c) << 1;
        n += c;
    }
    if (lane <= 15 - ROOT) { lim_out[lane] = my_lim; fb_out[lane] = my_fb; }
    for (uint32_t i = (uint32_t)lane; i < n; i += 64) {
        uint32_t base = 0;
        int mylen = 15;
#pragma unroll
        for (int len = ROOT + 1; len <= 15; ++len) {
            const uint32_t c = cs[len - ROOT - 1];
            if (i >= base && i < base + c) mylen = len;
            base += c;
        }
        out[i] = make_entry(kind, (int)sym[at + i], mylen);
    }
}

template <int ROOT>
__device__ __forceinline__ uint32_t long_lookup(const uint32_t *lim, const uint32_t *fb, const tab_t *tab, uint32_t bits)
{
    const uint32_t r15 = __builtin_bitreverse32(bits) >> 17;
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < 15 - ROOT; ++k) n += r15 >= lim[k] ? 1u : 0u;
    if (n >= (uint32_t)(15 - ROOT)) return 0u;
    const uint32_t f = fb[n];
    const uint32_t code = r15 >> (14 - ROOT - n);
    if (code < (f & 0xFFFFu)) return 0u;                    // (a slot of the root table whose short code names no symbol)
    return tab[(f >> 16) + (code - (f & 0xFFFFu))];
}

// an inclusive sum over each group of 64 / NB lanes (one block's lanes): rows of 16, pairs of rows, the wavefront
template <int NB>
__device__ __forceinline__ uint32_t group_scan_add(uint32_t v)
{
    v += dpp_shift<0x111, 0xF>(v);
    v += dpp_shift<0x112, 0xF>(v);
    v += dpp_shift<0x114, 0xF>(v);
    v += dpp_shift<0x118, 0xF>(v);
    if (NB <= 2) v += dpp_shift<0x142, 0xA>(v);
    if (NB == 1) v += dpp_shift<0x143, 0xC>(v);
    return v;
}

enum { SY_LIT = 0, SY_MATCH = 1, SY_EOB = 2, SY_BAD = 3 };
#ifdef TCMI_HDR_CALL                                  // (A/B: -DTCMI_HDR_CALL gives the header step a register budget of its own: a real call)
#define TCMI_HDR_INLINE __attribute__((noinline))
#else
#define TCMI_HDR_INLINE __forceinline__
#endif

// ---- header of one deflate block and its tables: one wavefront, the block's own (T, pay) ------------------------------------------
// -> T.go = 1 and T.pos at the first symbol (a Huffman block), or the block's stream is finished / damaged (T.go = 0).  Stored
// deflate blocks are turned into raw tokens here and the next header is taken at once.
__device__ TCMI_HDR_INLINE void block_header(BlkTabs &T, HdrScratch &H, const uint32_t *pay, uint32_t base_bit, uint32_t *toks, uint32_t cap, bool &last,
                                          uint64_t *stamps, int blk, bool one_header = false)
{
    const int lane = threadIdx.x & 63;
    uint32_t pos = uni(T.pos), ntok = uni(T.ntok), err = ST_OK;
    const uint32_t end = uni(T.end);
    bool go = false;
    while (!last && err == ST_OK && !go) {
        if (pos + 3u > end) { err = ST_BAD_STREAM; break; }
        const uint32_t h = uni(peek32(pay, pos));
        last = (h & 1u) != 0;
        const uint32_t type = (h >> 1) & 3u;
        pos += 3;
        if (type == 0) {
            // ---- stored block: byte-align, LEN / NLEN, LEN raw bytes -> raw tokens ------------------------------------------
            pos = (pos + 7u) & ~7u;
            if (pos + 32u > end) { err = ST_BAD_STREAM; break; }
            const uint32_t v = uni(peek32(pay, pos));
            const uint32_t len = v & 0xFFFFu;
            if (((v >> 16) ^ len) != 0xFFFFu) { err = ST_BAD_STREAM; break; }
            pos += 32;
            if (pos + len * 8u > end) { err = ST_BAD_STREAM; break; }
            const uint32_t off = (pos - base_bit) >> 3;
            const uint32_t pieces = (len + RAW_PIECE - 1u) / RAW_PIECE;
            if (ntok + pieces > cap) { err = ST_BAD_STREAM; break; }
            if ((uint32_t)lane < pieces) {
                const uint32_t o = (uint32_t)lane * RAW_PIECE;
                toks[ntok + (uint32_t)lane] = TOK_RAW | (min(len - o, RAW_PIECE) << 17) | (off + o);
            }
            ntok += pieces;
            pos += len * 8u;
            if (one_header) break;                              // (the caller stages the payload behind the stored bytes first)
            continue;
        }
        if (type == 3) { err = ST_BAD_STREAM; break; }
        // ---- code lengths -------------------------------------------------------------------------------------------------------
        int nlen = 288, ndist = 32;
        wave_sync();
        if (type == 1) {
            for (int i = lane; i < 320; i += 64) H.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
        } else {
            if (pos + 14u > end) { err = ST_BAD_STREAM; break; }
            const uint32_t hh = uni(peek32(pay, pos));
            nlen = (int)(hh & 31u) + 257;
            ndist = (int)((hh >> 5) & 31u) + 1;
            const int ncode = (int)((hh >> 10) & 15u) + 4;
            pos += 14;
            if (nlen > 286 || ndist > 30) { err = ST_BAD_STREAM; break; }
            if (lane < 19) H.cll[lane] = 0;
            wave_sync();
            // (19 x 3 bits: three looks of up to 8 lengths each, lane k takes the k-th)
            for (int i0 = 0; i0 < ncode; i0 += 8) {
                const uint32_t v = uni(peek32(pay, pos + (uint32_t)i0 * 3u));
                const int k = i0 + lane;
                if (lane < 8 && k < ncode) H.cll[CL_ORDER[k]] = (uint8_t)((v >> (3 * lane)) & 7u);
            }
            pos += (uint32_t)ncode * 3u;
            if (uni(build_table<1, CL_ROOT>(H.cll, 19, H.cnt_cl, H.sym_cl, T.dt, K_CODELEN, H.rs + 4) ? 1u : 0u) == 0u) { err = ST_BAD_STREAM; break; }
            for (int i = lane; i < 320; i += 64) H.lens[i] = 0;
            // The code-length symbols (0 .. 15: a length; 16: the previous length 3 - 6 times; 17 / 18: 3 - 10 / 11 - 138 zeros) are
            // a serial chain too, but a short one over few bits.  Every bit position of a slab is looked up by some lane (what
            // symbol would start here, how many lengths would it give, how many bits would it take: step | rep << 4 | val << 12);
            // the chain is then followed through that table with one scalar look-up per symbol that only notes the entry
            // (lane j keeps the j-th of 64), and what the symbols mean is worked out for 64 of them at a time: a sum scan of
            // the repeat counts places them, a maximum scan finds for every "16" the last symbol in front that names a length.
            uint32_t *const P = T.ll;
            uint32_t got = 0, prev = 0;
            const uint32_t total = (uint32_t)(nlen + ndist);
            bool first = true;
            while (got < total && err == ST_OK) {
                wave_sync();
#pragma unroll 2
                for (int o = lane; o < CL_SLAB + 16; o += 64) {
                    const uint32_t v = peek32(pay, pos + (uint32_t)o);
                    const uint32_t e = T.dt[v & ((1u << CL_ROOT) - 1u)];
                    const uint32_t nb = e & 15u, sym = e >> 16;
                    const uint32_t x = v >> nb;
                    const uint32_t eb = sym < 16u ? 0u : sym == 16u ? 2u : sym == 17u ? 3u : 7u;
                    const uint32_t rep = sym < 16u ? 1u : sym == 18u ? 11u + (x & 127u) : 3u + (x & (sym == 16u ? 3u : 7u));
                    const uint32_t val = sym <= 16u ? sym : 0u;
                    P[o] = nb && o < CL_SLAB ? (nb + eb) | (rep << 4) | (val << 12) : 0u;      // (0 behind the slab: the chain stops there)
                }
                wave_sync();
                // The chain of symbols through the table — a symbol's place is known when its predecessor is decoded — was followed one
                // LDS round trip per symbol (~300 of them per header: a fifth of this kernel's time).  Pointer doubling instead: J holds,
                // for every bit position, where the chain stands 2^k symbols later (a position without a code stays where it is);
                // six squarings of the table — every lane takes eight positions — and lane j, applying the squarings its bits ask for,
                // knows where symbol j starts; symbols j + 64, j + 128, .. lie J_6 further each.
                uint16_t *const J = reinterpret_cast<uint16_t *>(T.long_ll);       // 512 entries (the long-code tables are built later)
                static_assert(sizeof(T.long_ll) >= 512 * sizeof(uint16_t) && CL_SLAB + 16 <= 512, "the doubling table borrows the long-code table's LDS");
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t at0 = (uint32_t)lane + 64u * i;
                    const uint32_t e0 = at0 < (uint32_t)CL_SLAB + 16u ? P[at0] : 0u;
                    J[at0] = (uint16_t)(e0 ? at0 + (e0 & 15u) : at0);
                }
                wave_sync();
                uint32_t pm[5];
                pm[0] = 0;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    if (((uint32_t)lane >> k) & 1u) pm[0] = J[pm[0]];
                    uint32_t t8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) t8[i] = J[lane + 64 * i];
#pragma unroll
                    for (int i = 0; i < 8; ++i) t8[i] = J[t8[i]];
                    wave_sync();                                            // (every lane has read before any lane writes)
#pragma unroll
                    for (int i = 0; i < 8; ++i) J[lane + 64 * i] = (uint16_t)t8[i];
                    wave_sync();
                }
#pragma unroll
                for (int m = 1; m < 5; ++m) pm[m] = J[pm[m - 1]];
                uint32_t o = 0;
                bool stopped = false;
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    if (!(got < total && err == ST_OK && !stopped)) break;
                    uint32_t mine = P[pm[m]];
                    // the symbols of this batch that lie on the chain (a prefix of the lanes), and of them those that are still wanted:
                    // up to and including the one that completes the `total` lengths
                    const unsigned long long zmask = __ballot(mine == 0);
                    const uint32_t nv = zmask ? (uint32_t)__builtin_ctzll(zmask) : 64u;
                    if ((uint32_t)lane >= nv) mine = 0;
                    uint32_t rep = (mine >> 4) & 255u;
                    const uint32_t incl0 = wave_scan_add(rep);
                    const unsigned long long reach = __ballot(mine != 0 && got + incl0 >= total);
                    const uint32_t n_take = reach ? min(nv, (uint32_t)__builtin_ctzll(reach) + 1u) : nv;
                    if ((uint32_t)lane >= n_take) { mine = 0; rep = 0; }
                    // lane j: its symbol's place and value
                    const uint32_t v = mine >> 12;
                    const uint32_t at = got + incl0 - ((mine >> 4) & 255u);
                    const uint32_t named = wave_scan_max(mine != 0 && v != 16u ? (uint32_t)lane + 1u : 0u);     // 1 + the lane whose value a "16" here repeats
                    const uint32_t theirs = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((named - 1u) << 2), (int)v);
                    const uint32_t val = named ? theirs : prev;
                    if (__ballot(mine != 0 && (at + rep > total || (first && lane == 0 && v == 16u)))) { err = ST_BAD_STREAM; break; }
                    if (mine != 0 && val != 0) {
#pragma unroll
                        for (uint32_t i = 0; i < 6; ++i)            // (zeros are not stored, so rep <= 6)
                            if (i < rep) H.lens[at + i] = (uint8_t)val;
                    }
                    if (n_take) {
                        prev = (uint32_t)__builtin_amdgcn_readlane((int)val, (int)(n_take - 1u));
                        got += (uint32_t)__builtin_amdgcn_readlane((int)incl0, (int)(n_take - 1u));
                        // behind the last symbol taken
                        o = (uint32_t)__builtin_amdgcn_readlane((int)(pm[m] + (mine & 15u)), (int)(n_take - 1u));
                        first = false;
                    }
                    if (n_take < 64u && got < total) {                     // the chain ends here: no such code, or the slab's end
                        stopped = true;
                        o = (uint32_t)__builtin_amdgcn_readlane((int)pm[m], (int)n_take);
                        if (o < (uint32_t)CL_SLAB) err = ST_BAD_STREAM;
                    }
                }
                if (err == ST_OK && got < total && !stopped) err = ST_BAD_STREAM;      // (320 symbols give at least 320 lengths: not reached)
                pos += o;
                if (pos > end) err = ST_BAD_STREAM;
            }
            if (err != ST_OK) break;
            wave_sync();
            if (uni(H.lens[256]) == 0) { err = ST_BAD_STREAM; break; }    // no end-of-block code
        }
        TCMI_STAMP(stamps, blk, 2);
        // ---- tables: the root tables as in bgzf_inflate, the longer codes as ready-made entries ------------------------------------
        if (uni(build_table<5, LL_ROOT>(H.lens, nlen, H.cnt_ll, H.sym_ll, T.ll, K_LITLEN, H.rs) ? 1u : 0u) == 0u) { err = ST_BAD_STREAM; break; }
        if (uni(build_table<1, D_ROOT>(H.lens + nlen, ndist, H.cnt_d, H.sym_d, T.dt, K_DIST, H.rs + 2) ? 1u : 0u) == 0u) { err = ST_BAD_STREAM; break; }
        build_long<LL_ROOT>(H.cnt_ll, H.sym_ll, H.rs, K_LITLEN, T.long_ll, T.lim_ll, T.fb_ll);
        build_long<D_ROOT>(H.cnt_d, H.sym_d, H.rs + 2, K_DIST, T.long_d, T.lim_d, T.fb_d);
        if (pos >= end) { err = ST_BAD_STREAM; break; }
        go = true;
        TCMI_STAMP(stamps, blk, 3);
    }
    wave_sync();
    if (lane == 0) { T.pos = pos; T.ntok = ntok; T.err = err; T.go = go && err == ST_OK ? 1u : 0u; T.last = last ? 1u : 0u; }
}

// The rounds of pass A, hand-scheduled (see the comment at their use).  Two insertion points for the kernels that put two literals into
// one token (one block per workgroup: files whose blocks hold thousands of literals): the look-up of the code behind a literal, in
// flight under the match lanes' distance look-up, and its resolution — this symbol a literal of <= 9 bits, the next code a root-table
// literal that starts in the same stretch and ends within the soft end: then the token carries both bytes and the lane moves on
// behind the second.  (For the bench file's blocks — two per workgroup, a thousand tokens each — the 22 instructions cost more than
// the 19 % of rounds they save: 141 -> 150 us; at 2.6 : 1 the rounds fall by 38 %.)
// The round's sections, in order (labels in the text below): LT the loop's head — a lane whose symbols cross into a new stretch of 2^shift
// bits notes {p, total} in its ring (LA1); every fourth round the lanes that have crossed since look their position up in their target's
// ring (LB*: a target that has stopped at or in front of the lane is replaced by the lane it met, or by the next one; equal positions:
// met — back to the meeting point, stop); LC1 one symbol: 64 bits of payload, the 9-bit root look-up (LLl: a longer literal / length
// code by range compare), LDeob an end-of-block code, LC3 the literal's token / the length's base and extra bits, PAIR_LOOK_, the
// 8-bit distance root look-up (LLd: a longer one), LC5 PAIR_RESOLVE_, the new position (LDover: past the end), LC6 the token's store
// into the lane's row of the scratch; LDend lanes at the payload's end without an end-of-block code; LX out.
#define TCMI_PAIR_LOOK "v_lshrrev_b32 v44, v50, v47\n" "v_and_b32 v44, 0x1ff, v44\n" "v_lshl_add_u32 v44, v44, 2, %[tabs]\n" "ds_read_b32 v44, v44\n"
#define TCMI_PAIR_RESOLVE "s_waitcnt lgkmcnt(0)\n" "v_and_b32 v40, v49, v44\n" "v_and_b32 v45, 15, v44\n" "v_add_u32 v46, v52, v45\n" "v_xor_b32 v42, %[p], v52\n" "v_lshrrev_b32 v42, %[shift], v42\n" "v_bfe_u32 v40, v40, 8, 1\n" "v_cmp_gt_u32 vcc, 10, v50\n" "v_cmp_eq_u32 s[86:87], 1, v40\n" "s_and_b64 vcc, vcc, s[86:87]\n" "v_cmp_eq_u32 s[86:87], 0, v42\n" "s_and_b64 vcc, vcc, s[86:87]\n" "v_cmp_le_u32 s[86:87], v46, %[wend]\n" "s_and_b64 vcc, vcc, s[86:87]\n" "v_bfe_u32 v40, v44, 16, 8\n" "v_lshl_or_b32 v40, v40, 8, v51\n" "v_or_b32 v40, 0x1000000, v40\n" "v_cndmask_b32 v51, v51, v40, vcc\n" "v_cndmask_b32 v52, v52, v46, vcc\n"
#define TCMI_PASS_A_ASM(PAIR_LOOK_, PAIR_RESOLVE_) \
                asm volatile( \
                    "s_mov_b64 s[92:93], exec\n" \
                    "LT%=:\n" \
                    "s_mov_b64 exec, %[run]\n" \
                    "s_cbranch_execz LX%=\n" \
                    "v_lshrrev_b32 v40, %[shift], %[p]\n" \
                    "v_cmp_ne_u32 vcc, v40, %[kprev]\n" \
                    "s_and_saveexec_b64 s[80:81], vcc\n" \
                    "s_cbranch_execz LA1%=\n" \
                    "v_mov_b32 %[kprev], v40\n" \
                    "v_and_b32 v41, 7, v40\n" \
                    "v_lshl_add_u32 v41, v41, 3, %[ringb]\n" \
                    "ds_write2_b32 v41, %[p], %[total] offset1:1\n" \
                    "v_mov_b32 %[crossp], %[p]\n" \
                    "v_mov_b32 %[crosst], %[total]\n" \
                    "LA1%=:\n" \
                    "s_mov_b64 exec, %[run]\n" \
                    "s_and_b32 s90, %[rounds], 3\n" \
                    "s_cmp_eq_u32 s90, 3\n" \
                    "s_cbranch_scc0 LC%=\n" \
                    "v_cmp_ne_u32 vcc, -1, %[crossp]\n" \
                    "s_and_saveexec_b64 s[80:81], vcc\n" \
                    "s_cbranch_execz LB9%=\n" \
                    "v_add_u32 v41, -1, %[lim]\n" \
                    "v_min_u32 v41, %[tgt], v41\n" \
                    "v_lshl_add_u32 v42, v41, 3, %[recbase]\n" \
                    "ds_read2_b32 v[44:45], v42 offset1:1\n" \
                    "v_lshrrev_b32 v40, %[shift], %[crossp]\n" \
                    "v_and_b32 v40, 7, v40\n" \
                    "v_lshlrev_b32 v40, 3, v40\n" \
                    "s_waitcnt lgkmcnt(0)\n" \
                    "v_and_b32 v43, 3, v44\n" \
                    "v_cmp_ne_u32 vcc, 0, v43\n" \
                    "v_cmp_ge_u32 s[86:87], %[crossp], v45\n" \
                    "s_and_b64 vcc, vcc, s[86:87]\n" \
                    "v_cmp_lt_u32 s[86:87], %[tgt], %[lim]\n" \
                    "s_and_b64 vcc, vcc, s[86:87]\n" \
                    "s_and_saveexec_b64 s[82:83], vcc\n" \
                    "v_cmp_eq_u32 vcc, 1, v43\n" \
                    "v_add_u32 %[tgt], 1, %[tgt]\n" \
                    "s_and_b64 exec, exec, vcc\n" \
                    "v_lshrrev_b32 %[tgt], 8, v44\n" \
                    "s_mov_b64 exec, s[82:83]\n" \
                    "v_cmp_lt_u32 vcc, %[tgt], %[lim]\n" \
                    "s_and_b64 exec, exec, vcc\n" \
                    "s_cbranch_execz LB8%=\n" \
                    "v_lshl_add_u32 v42, %[tgt], 6, v40\n" \
                    "v_add_u32 v42, %[ringbase], v42\n" \
                    "ds_read2_b32 v[44:45], v42 offset1:1\n" \
                    "s_waitcnt lgkmcnt(0)\n" \
                    "v_cmp_eq_u32 vcc, v44, %[crossp]\n" \
                    "s_and_b64 exec, exec, vcc\n" \
                    "s_cbranch_execz LB8%=\n" \
                    "v_mov_b32 %[midx], v45\n" \
                    "v_mov_b32 %[state], 1\n" \
                    "v_mov_b32 %[p], %[crossp]\n" \
                    "v_mov_b32 %[total], %[crosst]\n" \
                    "v_lshl_or_b32 v43, %[tgt], 8, 1\n" \
                    "ds_write2_b32 %[recb], v43, %[p] offset1:1\n" \
                    "s_andn2_b64 %[run], %[run], exec\n" \
                    "LB8%=:\n" \
                    "s_mov_b64 exec, s[80:81]\n" \
                    "v_mov_b32 %[crossp], -1\n" \
                    "LB9%=:\n" \
                    "s_mov_b64 exec, %[run]\n" \
                    "s_cbranch_execz LX%=\n" \
                    "LC%=:\n" \
                    "v_cmp_lt_u32 vcc, %[p], %[wend]\n" \
                    "s_xor_b64 s[86:87], vcc, exec\n" \
                    "s_cmp_lg_u64 s[86:87], 0\n" \
                    "s_cbranch_scc1 LDend%=\n" \
                    "LC1%=:\n" \
                    "v_lshrrev_b32 v40, 5, %[p]\n" \
                    "v_lshl_add_u32 v40, v40, 2, %[pay]\n" \
                    "ds_read2_b32 v[44:45], v40 offset1:1\n" \
                    "ds_read_b32 v46, v40 offset:8\n" \
                    "s_waitcnt lgkmcnt(0)\n" \
                    "v_alignbit_b32 v47, v45, v44, %[p]\n" \
                    "v_alignbit_b32 v48, v46, v45, %[p]\n" \
                    "v_and_b32 v40, 0x1ff, v47\n" \
                    "v_lshl_add_u32 v40, v40, 2, %[tabs]\n" \
                    "ds_read_b32 v49, v40\n" \
                    "s_waitcnt lgkmcnt(0)\n" \
                    "v_and_b32 v50, 15, v49\n" \
                    "v_cmp_eq_u32 vcc, 0, v50\n" \
                    "s_cbranch_vccnz LLl%=\n" \
                    "LC2%=:\n" \
                    "v_bfe_u32 v41, v49, 8, 3\n" \
                    "v_cmp_eq_u32 vcc, 4, v41\n" \
                    "s_cbranch_vccnz LDeob%=\n" \
                    "LC3%=:\n" \
                    "v_bfe_u32 v51, v49, 16, 8\n" \
                    "v_or_b32 v51, 0x80000000, v51\n" \
                    "v_add_u32 v52, %[p], v50\n" \
                    PAIR_LOOK_ \
                    "s_mov_b64 s[88:89], exec\n" \
                    "v_cmp_eq_u32 vcc, 2, v41\n" \
                    "s_and_b64 exec, exec, vcc\n" \
                    "s_cbranch_execz LC5%=\n" \
                    "v_bfe_u32 v53, v49, 11, 5\n" \
                    "v_alignbit_b32 v54, v48, v47, v53\n" \
                    "v_and_b32 v40, 0xff, v54\n" \
                    "v_lshl_add_u32 v40, v40, 2, %[tabs]\n" \
                    "ds_read_b32 v55, v40 offset:%[odt]\n" \
                    "v_lshrrev_b32 v42, v50, v47\n" \
                    "v_bfe_u32 v43, v49, 16, 4\n" \
                    "v_bfe_u32 v42, v42, 0, v43\n" \
                    "v_bfe_u32 v43, v49, 20, 9\n" \
                    "v_add_u32 v56, v43, v42\n" \
                    "s_waitcnt lgkmcnt(0)\n" \
                    "v_and_b32 v57, 15, v55\n" \
                    "v_cmp_eq_u32 vcc, 0, v57\n" \
                    "s_cbranch_vccnz LLd%=\n" \
                    "LC4%=:\n" \
                    "v_bfe_u32 v43, v55, 4, 4\n" \
                    "v_lshrrev_b32 v42, v57, v54\n" \
                    "v_bfe_u32 v42, v42, 0, v43\n" \
                    "v_lshrrev_b32 v40, 16, v55\n" \
                    "v_add_u32 v42, v42, v40\n" \
                    "v_add_u32 v42, -1, v42\n" \
                    "v_lshl_or_b32 v51, v42, 9, v56\n" \
                    "v_add3_u32 v52, %[p], v53, v57\n" \
                    "v_add_u32 v52, v52, v43\n" \
                    "LC5%=:\n" \
                    "s_and_b64 exec, s[88:89], %[run]\n" \
                    "s_cbranch_execz LT%=\n" \
                    PAIR_RESOLVE_ \
                    "v_mov_b32 %[p], v52\n" \
                    "v_cmp_gt_u32 vcc, %[p], %[end]\n" \
                    "s_cbranch_vccnz LDover%=\n" \
                    "LC6%=:\n" \
                    "s_mov_b64 s[80:81], exec\n" \
                    "v_cmp_ne_u32 vcc, 0, %[room]\n" \
                    "s_and_b64 exec, exec, vcc\n" \
                    "global_store_dword %[sptr], v51, off\n" \
                    "v_add_u32 %[room], -1, %[room]\n" \
                    "v_lshl_add_u64 %[sptr], %[sptr], 0, %[sstride]\n" \
                    "s_mov_b64 exec, s[80:81]\n" \
                    "v_add_u32 %[total], 1, %[total]\n" \
                    "s_add_u32 %[rounds], %[rounds], 1\n" \
                    "s_branch LT%=\n" \
                    "LDend%=:\n" \
                    "s_mov_b64 s[82:83], exec\n" \
                    "s_mov_b64 exec, s[86:87]\n" \
                    "v_mov_b32 %[state], 3\n" \
                    "v_mov_b32 v43, 3\n" \
                    "ds_write2_b32 %[recb], v43, %[p] offset1:1\n" \
                    "s_andn2_b64 %[run], %[run], exec\n" \
                    "s_andn2_b64 exec, s[82:83], s[86:87]\n" \
                    "s_cbranch_execz LT%=\n" \
                    "s_branch LC1%=\n" \
                    "LDeob%=:\n" \
                    "s_mov_b64 s[82:83], exec\n" \
                    "s_and_b64 exec, exec, vcc\n" \
                    "v_add_u32 %[p], %[p], v50\n" \
                    "v_add_u32 %[total], 1, %[total]\n" \
                    "v_mov_b32 %[state], 2\n" \
                    "v_mov_b32 v43, 2\n" \
                    "ds_write2_b32 %[recb], v43, %[p] offset1:1\n" \
                    "s_andn2_b64 %[run], %[run], exec\n" \
                    "s_andn2_b64 exec, s[82:83], exec\n" \
                    "s_cbranch_execz LT%=\n" \
                    "s_branch LC3%=\n" \
                    "LDover%=:\n" \
                    "s_mov_b64 s[82:83], exec\n" \
                    "s_and_b64 exec, exec, vcc\n" \
                    "v_mov_b32 %[state], 3\n" \
                    "v_mov_b32 v43, 3\n" \
                    "ds_write2_b32 %[recb], v43, %[p] offset1:1\n" \
                    "s_andn2_b64 %[run], %[run], exec\n" \
                    "s_andn2_b64 exec, s[82:83], exec\n" \
                    "s_cbranch_execz LT%=\n" \
                    "s_branch LC6%=\n" \
                    "LLl%=:\n" \
                    "s_mov_b64 s[84:85], exec\n" \
                    "s_and_b64 exec, exec, vcc\n" \
                    "v_bfrev_b32 v40, v47\n" \
                    "v_lshrrev_b32 v40, 17, v40\n" \
                    "v_add_u32 v41, %[oliml], %[tabs]\n" \
                    "ds_read2_b32 v[58:59], v41 offset1:1\n" \
                    "ds_read2_b32 v[60:61], v41 offset0:2 offset1:3\n" \
                    "ds_read2_b32 v[62:63], v41 offset0:4 offset1:5\n" \
                    "s_waitcnt lgkmcnt(0)\n" \
                    "v_sub_u32 v58, v40, v58\n" \
                    "v_sub_u32 v59, v40, v59\n" \
                    "v_sub_u32 v60, v40, v60\n" \
                    "v_sub_u32 v61, v40, v61\n" \
                    "v_sub_u32 v62, v40, v62\n" \
                    "v_sub_u32 v63, v40, v63\n" \
                    "v_ashrrev_i32 v58, 31, v58\n" \
                    "v_ashrrev_i32 v59, 31, v59\n" \
                    "v_ashrrev_i32 v60, 31, v60\n" \
                    "v_ashrrev_i32 v61, 31, v61\n" \
                    "v_ashrrev_i32 v62, 31, v62\n" \
                    "v_ashrrev_i32 v63, 31, v63\n" \
                    "v_add3_u32 v58, v58, v59, v60\n" \
                    "v_add3_u32 v61, v61, v62, v63\n" \
                    "v_add3_u32 v42, v58, v61, 6\n" \
                    "v_min_u32 v41, 5, v42\n" \
                    "v_lshl_add_u32 v41, v41, 2, %[tabs]\n" \
                    "ds_read_b32 v43, v41 offset:%[ofbll]\n" \
                    "v_sub_u32 v41, 5, v42\n" \
                    "v_lshrrev_b32 v41, v41, v40\n" \
                    "s_waitcnt lgkmcnt(0)\n" \
                    "v_and_b32 v40, 0xffff, v43\n" \
                    "v_cmp_ge_u32 vcc, v41, v40\n" \
                    "v_cmp_gt_u32 s[86:87], 6, v42\n" \
                    "s_and_b64 vcc, vcc, s[86:87]\n" \
                    "v_sub_u32 v41, v41, v40\n" \
                    "v_lshrrev_b32 v40, 16, v43\n" \
                    "v_add_u32 v41, v41, v40\n" \
                    "v_and_b32 v41, 0x1ff, v41\n" \
                    "v_lshl_add_u32 v41, v41, 2, %[tabs]\n" \
                    "ds_read_b32 v49, v41 offset:%[olongll]\n" \
                    "s_waitcnt lgkmcnt(0)\n" \
                    "v_and_b32 v50, 15, v49\n" \
                    "v_cmp_ne_u32 s[86:87], 0, v50\n" \
                    "s_and_b64 vcc, vcc, s[86:87]\n" \
                    "s_andn2_b64 exec, exec, vcc\n" \
                    "s_cbranch_execz LLl9%=\n" \
                    "v_mov_b32 %[state], 3\n" \
                    "v_mov_b32 v43, 3\n" \
                    "ds_write2_b32 %[recb], v43, %[p] offset1:1\n" \
                    "s_andn2_b64 %[run], %[run], exec\n" \
                    "LLl9%=:\n" \
                    "s_and_b64 exec, s[84:85], %[run]\n" \
                    "s_cbranch_execz LT%=\n" \
                    "s_branch LC2%=\n" \
                    "LLd%=:\n" \
                    "s_mov_b64 s[84:85], exec\n" \
                    "s_and_b64 exec, exec, vcc\n" \
                    "v_bfrev_b32 v40, v54\n" \
                    "v_lshrrev_b32 v40, 17, v40\n" \
                    "v_add_u32 v41, %[olimd], %[tabs]\n" \
                    "ds_read2_b32 v[58:59], v41 offset1:1\n" \
                    "ds_read2_b32 v[60:61], v41 offset0:2 offset1:3\n" \
                    "ds_read2_b32 v[62:63], v41 offset0:4 offset1:5\n" \
                    "ds_read_b32 v42, v41 offset:24\n" \
                    "s_waitcnt lgkmcnt(0)\n" \
                    "v_sub_u32 v58, v40, v58\n" \
                    "v_sub_u32 v59, v40, v59\n" \
                    "v_sub_u32 v60, v40, v60\n" \
                    "v_sub_u32 v61, v40, v61\n" \
                    "v_sub_u32 v62, v40, v62\n" \
                    "v_sub_u32 v63, v40, v63\n" \
                    "v_sub_u32 v42, v40, v42\n" \
                    "v_ashrrev_i32 v58, 31, v58\n" \
                    "v_ashrrev_i32 v59, 31, v59\n" \
                    "v_ashrrev_i32 v60, 31, v60\n" \
                    "v_ashrrev_i32 v61, 31, v61\n" \
                    "v_ashrrev_i32 v62, 31, v62\n" \
                    "v_ashrrev_i32 v63, 31, v63\n" \
                    "v_ashrrev_i32 v42, 31, v42\n" \
                    "v_add3_u32 v58, v58, v59, v60\n" \
                    "v_add3_u32 v61, v61, v62, v63\n" \
                    "v_add3_u32 v42, v58, v61, v42\n" \
                    "v_add_u32 v42, 7, v42\n" \
                    "v_min_u32 v41, 6, v42\n" \
                    "v_lshl_add_u32 v41, v41, 2, %[tabs]\n" \
                    "ds_read_b32 v43, v41 offset:%[ofbd]\n" \
                    "v_sub_u32 v41, 6, v42\n" \
                    "v_lshrrev_b32 v41, v41, v40\n" \
                    "s_waitcnt lgkmcnt(0)\n" \
                    "v_and_b32 v40, 0xffff, v43\n" \
                    "v_cmp_ge_u32 vcc, v41, v40\n" \
                    "v_cmp_gt_u32 s[86:87], 7, v42\n" \
                    "s_and_b64 vcc, vcc, s[86:87]\n" \
                    "v_sub_u32 v41, v41, v40\n" \
                    "v_lshrrev_b32 v40, 16, v43\n" \
                    "v_add_u32 v41, v41, v40\n" \
                    "v_and_b32 v41, 31, v41\n" \
                    "v_lshl_add_u32 v41, v41, 2, %[tabs]\n" \
                    "ds_read_b32 v55, v41 offset:%[olongd]\n" \
                    "s_waitcnt lgkmcnt(0)\n" \
                    "v_and_b32 v57, 15, v55\n" \
                    "v_cmp_ne_u32 s[86:87], 0, v57\n" \
                    "s_and_b64 vcc, vcc, s[86:87]\n" \
                    "s_andn2_b64 exec, exec, vcc\n" \
                    "s_cbranch_execz LLd9%=\n" \
                    "v_mov_b32 %[state], 3\n" \
                    "v_mov_b32 v43, 3\n" \
                    "ds_write2_b32 %[recb], v43, %[p] offset1:1\n" \
                    "s_andn2_b64 %[run], %[run], exec\n" \
                    "LLd9%=:\n" \
                    "s_and_b64 exec, s[84:85], %[run]\n" \
                    "s_cbranch_execz LC5%=\n" \
                    "s_branch LC4%=\n" \
                    "LX%=:\n" \
                    "s_mov_b64 exec, s[92:93]\n" \
                    : [p] "+v"(p), [total] "+v"(total), [tgt] "+v"(tgt_abs), [midx] "+v"(midx), [kprev] "+v"(kprev), [state] "+v"(state), \
                      [room] "+v"(room), [crossp] "+v"(crossp), [crosst] "+v"(crosst), [sptr] "+v"(sptr), [run] "+s"(run), [rounds] "+s"(rounds) \
                    : [tabs] "v"(tabs), [pay] "v"(payb), [end] "v"(b_end), [wend] "v"(b_soft), [shift] "v"(shift), [ringb] "v"(ringb), [recb] "v"(recb), [lim] "v"(lim), \
                      [ringbase] "s"(ringbase), [recbase] "s"(recbase), [sstride] "s"(sstride), [odt] "n"(offsetof(BlkTabs, dt)), [olongll] "n"(offsetof(BlkTabs, long_ll)), \
                      [olongd] "n"(offsetof(BlkTabs, long_d)), [oliml] "n"(offsetof(BlkTabs, lim_ll)), [ofbll] "n"(offsetof(BlkTabs, fb_ll)), \
                      [olimd] "n"(offsetof(BlkTabs, lim_d)), [ofbd] "n"(offsetof(BlkTabs, fb_d)) \
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
                      "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", \
                      "s88", "s89", "s90", "s92", "s93", "vcc", "scc", "memory");

// WIN (one block per workgroup): the payload is staged a window at a time.  A block of a file that compresses 2 - 4 : 1 has 16 - 26 KB of
// payload; staged whole, four workgroups fit a CU and a BAM's blocks take four rounds and a half.  With a window of 6 KB pass A runs
// over the symbols that START in the window, the chain's last lane says where the next window begins, and the tables stay.
template <int NB, bool WIN>
__global__ __launch_bounds__(64 * NB) __attribute__((amdgpu_waves_per_eu(TCMI_SYM_WAVES, TCMI_SYM_WAVES))) void bgzf_symbols(SymArgs a)
{
    static_assert(!WIN || NB == 1, "a window per wavefront");
    constexpr int SYM_BLOCKS = NB, SYM_LANES = 64 / NB;
    constexpr bool PAIRS = NB == 1;                 // two literals in one token: the kernels of one block per workgroup (payloads beyond 4 KB)
    static_assert(NB == 4 || NB == 2 || NB == 1, "a block's lanes: a row of 16, two rows, or the wavefront");
    __shared__ SymLds<NB> L;
    extern __shared__ __attribute__((aligned(8))) uint32_t pay_all[];   // per block: its compressed payload, from the dword that holds its first byte on
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int blk0 = a.first_block + (int)blockIdx.x * SYM_BLOCKS;
    const int blk = blk0 + wave;                    // this wavefront's block (header, tables)
    const bool have = blk < a.n_blocks;
    BlkTabs &T = L.b[wave];
    uint32_t *const pay_lds = pay_all + (size_t)wave * (WIN ? a.win_dwords : a.pay_dwords);
    const uint32_t *pay = pay_lds;                  // (WIN: moved so that pay[dword of the payload] hits the staged window)
    BlockDesc d = {};
    if (have) d = a.blocks[blk];
    const uint32_t *const gsrc = a.file32 + (d.cin >> 2);
    uint32_t *const toks = a.tokens + d.tok;
    const uint32_t base_bit = (uint32_t)(d.cin & 3u) * 8u;
    const uint32_t end = base_bit + d.clen * 8u;                    // first bit behind the payload
    // ---- the payload into LDS (+ 6 dwords: a lane looks up to 48 bits past the end; the file buffer has the slack) ----
    if (have) {
        if (a.stamps && lane < 16) a.stamps[(size_t)blk * 16 + lane] = 0;
        TCMI_STAMP(a.stamps, blk, 0);
        if constexpr (!WIN) {
            const uint32_t n = min(a.pay_dwords, (end + 31u) / 32u + 6u);
            for (uint32_t i = (uint32_t)lane; i < n; i += 64) pay_lds[i] = gsrc[i];
        }
    }
    if (lane == 0) { T.pos = base_bit; T.end = end; T.ntok = 0; T.err = ST_OK; T.go = 0; T.last = 0; }
    wave_sync();
    if (have) TCMI_STAMP(a.stamps, blk, 1);
    bool last = !have;
    bool hdr_due = true;                            // WIN: the next thing at T.pos is a deflate header (else: more symbols of the stream)
    uint32_t soft_end = end;                        // WIN: first bit behind the symbols this window's pass A takes
    for (;;) {
        if constexpr (WIN) {
            // ---- the window: from the dword of T.pos on (a header fits in well under a window; so does a symbol behind the soft end)
            if (have && uni(T.err) == ST_OK && (!last || !hdr_due)) {
                const uint32_t wfirst = uni(T.pos) >> 5, total_dw = (end + 31u) / 32u + 6u;
                const uint32_t n = min(a.win_dwords, total_dw - min(total_dw, wfirst));
                wave_sync();
                {   // (8 bytes a lane, four loads in flight; the window starts on any dword of the file: unaligned access mode)
                    const uint32_t n2 = (n + 1u) >> 1;
                    uint2 *const dst2 = reinterpret_cast<uint2 *>(pay_lds);
                    const uint32_t *const src = gsrc + wfirst;
                    for (uint32_t i = (uint32_t)lane; i < n2; i += 256) {
                        uint2 v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (i + 64u * k < n2) __builtin_memcpy(&v[k], src + 2u * (i + 64u * k), 8);
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (i + 64u * k < n2) dst2[i + 64u * k] = v[k];
                    }
                }
                wave_sync();
                pay = pay_lds - wfirst;
                soft_end = min(end, (wfirst + a.win_dwords - 6u) * 32u);
            }
        }
        // ---- every wavefront: its block's next header and tables ------------------------------------------------------------------
        if (have && uni(T.err) == ST_OK && (!last || (WIN && !hdr_due))) {         // (WIN, !hdr_due: the stream's symbols go on in the new window)
            if (!WIN || hdr_due) block_header(T, L.h[wave], pay, base_bit, toks, d.tok_cap, last, a.stamps, blk, WIN);
        } else if (lane == 0) T.go = 0;
        __syncthreads();
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) any |= L.b[k].go;
        if (uni(any) == 0) {
            if (WIN && have && uni(T.err) == ST_OK && !last) continue;     // (a stored block became tokens: the header behind it is next)
            break;                                  // (all streams finished or failed)
        }
        if constexpr (WIN) hdr_due = false;
        if (wave == 0) {
            // ---- wavefront 0: the symbols of all four blocks, 16 lanes each ------------------------------------------------------
            const int b = lane / SYM_LANES, c = lane % SYM_LANES, lane0 = lane - c;       // block, lane in the block, the block's first lane
            BlkTabs &B = L.b[b];
            const uint32_t *const bp = WIN ? pay : pay_all + (size_t)b * a.pay_dwords;
            const bool on = B.go != 0;
            const uint32_t b_end = B.end, start = B.pos;
            const uint32_t b_soft = WIN ? soft_end : b_end;     // where the lanes stop taking symbols (WIN: the window's end)
            const BlockDesc bd = blk0 + b < a.n_blocks ? a.blocks[blk0 + b] : BlockDesc{};
            uint32_t *const btok = a.tokens + bd.tok;
            const uint32_t bcap = bd.tok_cap;
            const uint32_t lane_cap = bcap / (uint32_t)SYM_LANES / max(a.scratch_div, 1u);     // tokens a lane may park in the scratch half (scratch_div: tests force pass B)
            // A lane's k-th parked token lies at scratch[k * SYM_LANES]: the lanes of a block decode one symbol a round each, so the
            // stores of a round fall into consecutive words (4-byte stores into a region of its own per lane cost a memory transaction
            // each: 115 MB of writes per BAM for 4.4 MB of tokens).
            uint32_t *const scratch = btok + bcap + (uint32_t)c;

            const uint32_t chunk = on ? (b_soft - min(b_soft, start) + (uint32_t)SYM_LANES - 1u) / (uint32_t)SYM_LANES : 1u;      // >= 1
            // stretches of >= 64 bits (a symbol takes <= 48: none is skipped), about chunk / 4: a lane trails its target by about
            // a chunk, RING stretches are kept
            const uint32_t shift = (uint32_t)max(6, 30 - (int)__builtin_clz(chunk | 1u) - (int)a.shift_bias);
            // One literal / length / end-of-block code at bit p; a length is followed by its distance.  A literal of at most nine bits
            // takes the next code along if that is a root-table literal too, starts in the same stretch and ends within the soft end:
            // a rule of the position alone, so that pass B and the hand-scheduled rounds below cut the stream into the same tokens.
            auto symbol = [&](uint32_t &p, uint32_t &tok) __attribute__((always_inline)) -> int {
                uint32_t lo, hi;
                peek64(bp, p, lo, hi);
                uint32_t e = B.ll[lo & ((1u << LL_ROOT) - 1u)];
                if (__builtin_expect(__ballot((e & 15u) == 0) != 0, 0)) {
                    const uint32_t e2 = long_lookup<LL_ROOT>(B.lim_ll, B.fb_ll, B.long_ll, lo);
                    if ((e & 15u) == 0) e = e2;
                }
                if ((e & 15u) == 0) return SY_BAD;
                if (e & E_LIT) {
                    const uint32_t p0 = p, nb1 = e & 15u;
                    p += nb1;
                    tok = TOK_LIT | ((e >> 16) & 0xFFu);
                    if (PAIRS && nb1 < 10u) {
                        const uint32_t e2 = B.ll[(lo >> nb1) & ((1u << LL_ROOT) - 1u)];
                        const uint32_t p3 = p + (e2 & 15u);
                        if ((e2 & E_LIT) && ((p ^ p0) >> shift) == 0u && p3 <= b_soft) { tok |= TOK_LIT2 | (((e2 >> 16) & 0xFFu) << 8); p = p3; }
                    }
                    return SY_LIT;
                }
                if (e & E_EOB) { p += e & 15u; return SY_EOB; }
                const uint32_t k = (e >> 11) & 31u;                 // code + extra bits of the length
                const uint32_t d32 = __builtin_amdgcn_alignbit(hi, lo, k);
                uint32_t f = B.dt[d32 & ((1u << D_ROOT) - 1u)];
                if (__builtin_expect(__ballot((f & 15u) == 0) != 0, 0)) {
                    const uint32_t f2 = long_lookup<D_ROOT>(B.lim_d, B.fb_d, B.long_d, d32);
                    if ((f & 15u) == 0) f = f2;
                }
                if ((f & 15u) == 0) return SY_BAD;
                const uint32_t nd = f & 15u, eb2 = (f >> 4) & 15u;
                p += k + nd + eb2;
                const uint32_t nb = e & 15u, eb = (e >> 16) & 15u;
                const uint32_t len = ((e >> 20) & 0x1FFu) + ((lo >> nb) & ((1u << eb) - 1u));
                const uint32_t dist = (f >> 16) + ((d32 >> nd) & ((1u << eb2) - 1u));
                tok = len | ((dist - 1u) << 9);
                return SY_MATCH;
            };

            // ---- pass A: every lane decodes from its own start until it meets the lane in front; the tokens go to its scratch ----
            // Meeting points are looked for where a lane's symbols cross into a new stretch of 2^shift bits: the lane notes its
            // first symbol start p in the stretch (and how many symbols it had decoded by then) in a ring of its own, and looks
            // p up in the ring of its target — the nearest lane in front that is still decoding, or the lane that one met.  Equal
            // positions are one trajectory from there on: the lane stops, its target's symbols from that one on are the true ones.
            const uint32_t s_c = start + (uint32_t)c * chunk;
            enum { RUN = 0, MERGED = 1, EOB = 2, DEAD = 3 };
            uint32_t state = on && s_c < b_soft ? RUN : DEAD;
            uint32_t tgt = (uint32_t)c + 1u, total = 0, midx = 0;
            uint32_t p = min(s_c, b_soft);
            uint32_t kprev = 0xFFFFFFFFu;
            bool spilled = false;                   // more symbols than the scratch holds: pass B decodes this block again
            uint32_t rounds = 0;
#pragma unroll
            for (int k = 0; k < RING; ++k) L.a.ring[lane][k] = make_uint2(0xFFFFFFFFu, 0u);
            L.a.rec[lane] = make_uint2(state, p);
            wave_sync();
            // The rounds, hand-scheduled (TCMI_PASS_A_ASM: ~85 instructions a round; the compiler's version of the same loop took ~180,
            // half of them bookkeeping of which lanes are in which branch — tools/spec_inflate_proto.py and tools/sym_balance_sim.py
            // are the scheme in Python).  A lane's look at its target's ring waits for the next round that is a multiple of four (the
            // lane goes on decoding meanwhile; if it has met its target, it steps back to the meeting point), and a lane's target is
            // kept as a lane of the wavefront.
            {
                uint32_t tgt_abs = (uint32_t)lane0 + tgt, room = lane_cap, crossp = 0xFFFFFFFFu, crosst = 0;
                uint64_t sptr = reinterpret_cast<uint64_t>(scratch);
                const uint64_t sstride = (uint64_t)SYM_LANES * 4u;
                const uint32_t tabs = (uint32_t)reinterpret_cast<uintptr_t>(&B), payb = (uint32_t)reinterpret_cast<uintptr_t>(bp);
                const uint32_t ringbase = (uint32_t)reinterpret_cast<uintptr_t>(&L.a.ring[0][0]), recbase = (uint32_t)reinterpret_cast<uintptr_t>(&L.a.rec[0]);
                const uint32_t ringb = ringbase + (uint32_t)lane * (RING * 8), recb = recbase + (uint32_t)lane * 8u;
                const uint32_t lim = (uint32_t)lane0 + (uint32_t)SYM_LANES;
                unsigned long long run = __ballot(state == RUN);
                static_assert(LL_ROOT == 9 && D_ROOT == 8 && RING == 8, "masks and counts below");
                if constexpr (PAIRS) { TCMI_PASS_A_ASM(TCMI_PAIR_LOOK, TCMI_PAIR_RESOLVE) } else { TCMI_PASS_A_ASM(, ) }
                tgt = tgt_abs - (uint32_t)lane0;
                spilled = total > lane_cap;
            }
            TCMI_STAMP(a.stamps, blk0, 4);
            TCMI_STAMP_ADD(a.stamps, blk0, 8, rounds);
            // ---- per block the chain of lanes that hold the true symbols: lane 0 from `start`, then whoever it met, ... ----------
            uint32_t before = 0;                    // symbols a lane decoded in front of its true start: they do not count
            bool alive = on && c == 0;
            uint32_t eob_pos = 0, berr = ST_OK;
            bool more = false;                      // WIN: the stream goes on behind this window (no end-of-block code yet)
            for (int bb = 0; bb < SYM_BLOCKS; ++bb) {
                if (uni(L.b[bb].go) == 0) continue;
                uint32_t cc = (uint32_t)bb * SYM_LANES;
                for (;;) {
                    const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)state, (int)cc);
                    if (st == MERGED) {
                        const uint32_t t = (uint32_t)bb * SYM_LANES + (uint32_t)__builtin_amdgcn_readlane((int)tgt, (int)cc);
                        const uint32_t mi = (uint32_t)__builtin_amdgcn_readlane((int)midx, (int)cc);
                        if ((uint32_t)lane == t) { alive = true; before = mi; }
                        cc = t;
                    } else {
                        const uint32_t pp = (uint32_t)__builtin_amdgcn_readlane((int)p, (int)cc);
                        if (b == bb) {
                            if (st == EOB) eob_pos = pp;
                            else if (WIN && st == DEAD && pp >= b_soft && pp <= b_end && b_soft < b_end) { eob_pos = pp; more = true; }   // the window's end: on from there
                            else berr = ST_BAD_STREAM;
                        }
                        break;
                    }
                }
            }
            uint32_t cnt = 0;
            if (alive) {
                cnt = total - before;
                if (state == EOB) --cnt;            // (the end-of-block code is a symbol, not a token)
            }
            // exclusive sum over the block's lanes -> every lane's place among the block's tokens
            const uint32_t incl = group_scan_add<NB>(cnt);
            const uint32_t all = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane0 + SYM_LANES - 1) << 2), (int)incl);
            const uint32_t ntok0 = B.ntok;
            if (on && berr == ST_OK && ntok0 + all > bcap) berr = ST_BAD_STREAM;
            // a lane that parked more than its scratch holds: the whole block goes through pass B
            const unsigned long long spill_mask = __ballot(alive && spilled);
            const bool block_redo = ((spill_mask >> lane0) & (NB == 1 ? ~0ull : (1ull << (SYM_LANES & 63)) - 1ull)) != 0;
            TCMI_STAMP(a.stamps, blk0, 5);
            // The true tokens to their places, in order, behind those the block has already.  They lie where the lanes parked them: lane
            // j's k-th at scratch[k][j], of which [before_j, before_j + cnt_j) are true and belong at excl_j onwards.  The lanes of the
            // block take the OUTPUT tokens in turn (lane c: c, c + SYM_LANES, ..): every lane follows the table {where lane j's tokens
            // end, before_j - excl_j} through LDS — the owner of a lane's next token is the same lane or a later one —, reads the token
            // from the parked rows (this workgroup wrote them a moment ago: L2) and the stores of a turn are consecutive words.
            // (Round 4 left the tokens parked and gave bgzf_copy a list of 32 pieces: its lanes then fetched a batch of 64 tokens from
            // 64 rows — 170 MB of 64-byte sectors per BAM for 4.4 MB of tokens, from HBM: the rows of the 4 000 blocks in flight do
            // not fit the L2s.  Before that every lane moved its own tokens: a store per token and lane into 32 places.)
            L.a.ring[lane][0] = make_uint2(incl, before - (incl - cnt));
            uint32_t rows_max = alive && on && berr == ST_OK ? before + cnt : 0u;      // the parked rows that hold true tokens (the wavefront's: uniform turns below)
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) rows_max = max(rows_max, (uint32_t)__shfl_xor((int)rows_max, dd, 64));
            wave_sync();
            if (on && berr == ST_OK) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the parked tokens are this wavefront's own stores)
                if (!block_redo && all <= a.gather_max) {
                    constexpr int GV = TCMI_SYM_MOVE;
                    const uint32_t *const rows = btok + bcap;
                    uint32_t *const dst = btok + ntok0;
                    uint32_t owner = 0;
                    uint2 e = L.a.ring[lane0][0];
                    for (uint32_t i0 = (uint32_t)c; i0 < all; i0 += GV * SYM_LANES) {      // (GV loads in flight: the loop is all latency)
                        uint32_t at[GV], t[GV];
#pragma unroll
                        for (int k = 0; k < GV; ++k) {
                            const uint32_t i = min(i0 + (uint32_t)(k * SYM_LANES), all - 1u);
                            while (e.x <= i && owner + 1u < (uint32_t)SYM_LANES) { ++owner; e = L.a.ring[lane0 + (int)owner][0]; }
                            at[k] = (i + e.y) * (uint32_t)SYM_LANES + owner;
                        }
#pragma unroll
                        for (int k = 0; k < GV; ++k) t[k] = rows[at[k]];
#pragma unroll
                        for (int k = 0; k < GV; ++k) if (i0 + (uint32_t)(k * SYM_LANES) < all) dst[i0 + (uint32_t)(k * SYM_LANES)] = t[k];
                    }
                } else if (!block_redo) {
                    // A block of many tokens (a file that compresses like real data: thousands per block, its parked rows 50 KB and
                    // more — with every block of the chip's round in flight they are in no L2 any more, and a lane that follows ONE
                    // owner reads one word of every row: sixteen times the bytes).  Here the rows are read as they were written — row
                    // k + r, every lane its own word: whole sectors, each once — R rows at a time, all their loads in flight; a lane's R tokens are
                    // consecutive in the output, so it stores them as four 16-byte words (word-aligned; the ends of its true range word
                    // by word).
                    constexpr int R = TCMI_SYM_ROWS;
                    const uint32_t *const mine = btok + bcap + (uint32_t)c;
                    uint32_t *const dst = btok + ntok0 + (incl - cnt);              // this lane's first true token goes here
                    const uint32_t lo = before, hi = before + cnt;
                    const uint32_t kmax = rows_max;
                    const uint32_t last_row = lane_cap ? lane_cap - 1u : 0u;
                    for (uint32_t k0 = 0; k0 < kmax; k0 += R) {
                        uint32_t t[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) t[r] = mine[(size_t)min(k0 + (uint32_t)r, last_row) * SYM_LANES];
                        if (k0 >= lo && k0 + R <= hi) {
                            // (dst is word-aligned only: the 16-byte store goes through a type that says so)
                            struct __attribute__((packed, aligned(4))) W4 { uint32_t w[4]; };
                            W4 *q = reinterpret_cast<W4 *>(dst + (k0 - lo));
#pragma unroll
                            for (int r = 0; r < R; r += 4) q[r / 4] = W4{{t[r], t[r + 1], t[r + 2], t[r + 3]}};
                        } else {
#pragma unroll
                            for (int r = 0; r < R; ++r) if (k0 + r >= lo && k0 + r < hi) dst[k0 + r - lo] = t[r];
                        }
                    }
                } else {
                    // ---- pass B: the true ranges once more, tokens straight to their places -----------------------------------
                    // (the position of a lane's true start is not kept: decode from the lane's own start and drop `before` symbols)
                    uint32_t *const dst = btok + ntok0 + (incl - cnt);
                    uint32_t pp = min(s_c, b_soft);
                    for (uint32_t i = 0; alive && i < before + cnt; ++i) {
                        uint32_t tok = 0;
                        (void)symbol(pp, tok);
                        if (i >= before) dst[i - before] = tok;
                    }
                }
            }
            TCMI_STAMP(a.stamps, blk0, 6);
            if (on && c == 0) { B.ntok = ntok0 + all; B.pos = eob_pos; B.err = berr; }
            if constexpr (WIN) hdr_due = uni(__ballot(more) != 0 ? 1u : 0u) == 0u;       // (an end-of-block code was reached: a header comes next)
        }
        __syncthreads();
    }
    if (have && lane == 0) {
        a.n_tok[blk] = T.ntok;
        a.status[blk] = T.err;
    }
}

struct CopyArgs {
    const uint8_t *file;        // compressed file (raw tokens copy from it)
    const BlockDesc *blocks;
    const uint32_t *tokens;
    const uint32_t *n_tok;
    uint8_t *out;
    uint32_t *rec_slot;
    uint32_t *n_rec;
    int32_t *overshoot;         // bytes by which the block's last record runs into the next blocks (0x7FFFFFFF: its size field does)
    uint32_t *first_rec;        // offset of the first record start found in the block (0xFFFFFFFF: none)
    uint32_t *status;           // in: bgzf_symbols' verdict; out: the block's
    int32_t n_blocks;           // (the launch's blocks end here)
    int32_t first_block;        // ... and start here
    uint32_t n_ref;             // reference sequences of the BAM header
    uint64_t *stamps;           // diagnostic, as SymArgs::stamps
    uint32_t team_bytes;        // a batch of 64 tokens with at most this many bytes of output copies its matches in teams
    // the CRC-32 of every block's output against the value in its trailer (SAM spec 4.1; htslib checks it on every block it reads), taken
    // while the bytes are flushed from the ring (crc != 0):
    uint32_t crc;
    uint32_t zeros_seg[32];     // zeros_seg[i]: the CRC register with only bit i set, CSEG zero bytes later
    const uint32_t *crc_ops;    // [CRC_NOPS][8][16]: the register after 2^k more zero bytes, by nibble (crc_later)
};

// The copy loop of the matches of a stretch, hand-scheduled.  mm: the matches still to be copied; pm: those of them the inner loop
// takes unasked — plain (source in the ring or parked next to it, source and destination apart by the match's length at least) and
// of eight bytes or more.  14 instructions a match: EIGHT bytes a lane at min(8 lane, len - 8) (the last piece overlaps the one
// before instead of running past the end; LDS takes any byte address); the operands come packed for it (vA2 = (len - 1) << 16 |
// (destination - 7) & 0xffff, vB2 = source - 7): one v_cmpx gives the lane mask (vA2 >= 8 lane << 16  <=>  len > 8 lane), one SDWA
// v_min the piece's place + 7, two adds the addresses (the destination's within 16 bits); the NEXT match's operands are fetched
// while the LDS read is under way.  (A CU of these wavefronts issues about one instruction a cycle, whatever its kind: what counts
// is the number of instructions.)  Then the first other match: a plain one of 3 - 7 bytes goes byte-wise (LMs); a far match
// (source flushed to HBM long ago) is copied here too, 64 bytes a load; anything else leaves with its lane in j (C++ copies it:
// periods shorter than the match, ranges across the ring's end) — or j = -1: all done.
#define TCMI_LM_ASM() \
                    asm volatile( \
                        "s_mov_b64 s[92:93], exec\n" \
                        "LO%=:\n" \
                        "s_andn2_b64 s[80:81], %[mm], %[pm]\n" \
                        "s_ff1_i32_b64 %[j], s[80:81]\n" \
                        "s_mov_b64 s[82:83], %[mm]\n" \
                        "s_cmp_lt_i32 %[j], 0\n" \
                        "s_cbranch_scc1 LR%=\n" \
                        "s_lshl_b64 s[82:83], 1, %[j]\n" \
                        "s_sub_u32 s82, s82, 1\n" \
                        "s_subb_u32 s83, s83, 0\n" \
                        "s_and_b64 s[82:83], s[82:83], %[mm]\n" \
                        "LR%=:\n" \
                        "s_andn2_b64 %[mm], %[mm], s[82:83]\n" \
                        "s_cmp_eq_u64 s[82:83], 0\n" \
                        "s_cbranch_scc1 LN%=\n" \
                        "LM%=:\n" \
                        "s_ff1_i32_b64 s84, s[82:83]\n" \
                        "v_readlane_b32 %[sa], %[vA2], s84\n" \
                        "v_readlane_b32 %[sb], %[vB2], s84\n" \
                        "s_bitset0_b64 s[82:83], s84\n" \
                        "s_cmp_lt_u32 %[sa], 0x400000\n" \
                        "s_cbranch_scc0 LML%=\n" \
                        "v_cmpx_ge_u32 vcc, %[sa], %[vX1]\n" \
                        "v_add_u32 %[t0], %[sb], %[vlane7]\n" \
                        "v_add_u32_sdwa %[t1], %[sa], %[vlane7] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" \
                        "ds_read_u8 %[t2], %[t0]\n" \
                        "s_waitcnt lgkmcnt(0)\n" \
                        "ds_write_b8 %[t1], %[t2]\n" \
                        "s_mov_b64 exec, s[92:93]\n" \
                        "LMe%=:\n" \
                        "s_cmp_lg_u64 s[82:83], 0\n" \
                        "s_cbranch_scc1 LM%=\n" \
                        "LN%=:\n" \
                        "s_cmp_lt_i32 %[j], 0\n" \
                        "s_cbranch_scc1 LMx%=\n" \
                        "v_readlane_b32 %[sb], %[vB], %[j]\n" \
                        "v_readlane_b32 %[sa], %[vA], %[j]\n" \
                        "s_cmp_lt_u32 %[sb], 0x20000\n" \
                        "s_cbranch_scc0 LMx%=\n" \
                        "s_bitset0_b64 %[mm], %[j]\n" \
                        "v_readlane_b32 %[sb], %[vC], %[j]\n" \
                        "s_lshr_b32 %[len], %[sa], 16\n" \
                        "s_and_b32 %[sa], %[sa], 0xffff\n" \
                        "v_add_u32 %[t1], %[sa], %[vlane]\n" \
                        "v_add_u32 %[t0], %[sb], %[vlane]\n" \
                        "LMg%=:\n" \
                        "v_cmp_gt_u32 vcc, %[len], %[vlane]\n" \
                        "s_mov_b64 exec, vcc\n" \
                        "global_load_ubyte %[t2], %[t0], %[outp]\n" \
                        "s_waitcnt vmcnt(0)\n" \
                        "ds_write_b8 %[t1], %[t2]\n" \
                        "s_mov_b64 exec, s[92:93]\n" \
                        "s_cmp_gt_u32 %[len], 64\n" \
                        "s_cbranch_scc0 LF1%=\n" \
                        "s_sub_u32 %[len], %[len], 64\n" \
                        "v_add_u32 %[t0], 64, %[t0]\n" \
                        "v_add_u32 %[t1], 64, %[t1]\n" \
                        "s_branch LMg%=\n" \
                        "LF1%=:\n" \
                        "s_mov_b32 %[j], -1\n" \
                        "s_cmp_lg_u64 %[mm], 0\n" \
                        "s_cbranch_scc1 LO%=\n" \
                        "s_branch LMx%=\n" \
                        "LML%=:\n" \
                        "v_readlane_b32 s85, %[vXl], s84\n" \
                        "v_readlane_b32 s86, %[vYl], s84\n" \
                        "v_readlane_b32 s87, %[vKl], s84\n" \
                        "LMq%=:\n" \
                        "s_lshr_b32 s88, s86, 16\n" \
                        "v_add_u32 v48, s87, %[vlane32]\n" \
                        "v_cmpx_gt_i32 vcc, 32, v48\n" \
                        "v_add_u32_sdwa v49, s86, %[vlane4] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" \
                        "v_add_u32_sdwa v50, s85, %[vlane4] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" \
                        "ds_read2_b32 v[52:53], v49 offset1:1\n" \
                        "ds_read_b32 v51, v50\n" \
                        "v_max_i32 v48, 0, v48\n" \
                        "v_lshrrev_b32_e64 v48, v48, -1\n" \
                        "v_and_b32_sdwa v54, s85, %[vlane0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
                        "v_lshlrev_b32_e64 v54, v54, -1\n" \
                        "v_and_b32 v48, v48, v54\n" \
                        "s_waitcnt lgkmcnt(0)\n" \
                        "v_alignbit_b32 v52, v53, v52, s88\n" \
                        "v_bfi_b32 v51, v48, v52, v51\n" \
                        "ds_write_b32 v50, v51\n" \
                        "s_mov_b64 exec, s[92:93]\n" \
                        "s_cmp_lt_i32 s87, -2016\n" \
                        "s_cbranch_scc0 LMe%=\n" \
                        "s_add_u32 s87, s87, 2048\n" \
                        "s_add_u32 s85, s85, 256\n" \
                        "s_and_b32 s85, s85, 0xffff\n" \
                        "s_add_u32 s86, s86, 256\n" \
                        "s_branch LMq%=\n" \
                        "LMx%=:\n" \
                        "s_mov_b64 exec, s[92:93]\n" \
                        : [mm] "+s"(mm), [j] "=&s"(j), [sa] "=&s"(sa), [sb] "=&s"(sb), [len] "=&s"(len), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2) \
                        : [vA] "v"(vA), [vB] "v"(vB), [vC] "v"(vC), [vlane] "v"(lane), [vX] "v"(lane_hi), [vA2] "v"(vA2), [vB2] "v"(vB2), [vXl] "v"(vXl), [vYl] "v"(vYl), [vKl] "v"(vKl), [vX1] "v"(lane_sh16), [vlane7] "v"(lane_p7), [vlane4] "v"(lane_x4), [vlane32] "v"(lane_x32), [vlane0] "v"(lane0_31), [outp] "s"(out), [pm] "s"(plain_mask) \
                        : "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s92", "s93", "vcc", "scc", "memory", "v48", "v49", "v50", "v51", "v52", "v53", "v54");

// bgzf_copy: CW blocks per workgroup, a wavefront each (they share nothing but the CRC tables); every wavefront has its ring, the 512
// bytes next to it where far matches are parked, and the teams' slots.  LDS addresses stay below 64 K: the copy loops do their
// address arithmetic in 16 bits.
// TCMI_CRC_REG (A/B): the CRC's tables not in LDS but across the lanes of four registers — sixteen 16-entry nibble tables
// (t[k][v] = lo_k[v & 15] ^ hi_k[v >> 4]: the tables are linear in v), looked up with ds_bpermute_b32: twice the look-ups, no LDS,
// and with no tables to share a workgroup is ONE wavefront again (18 a CU instead of 16: a 1M-read BAM's 4 187 blocks in one round).
#ifndef TCMI_CRC_REG
#define TCMI_CRC_REG 0
#endif
#ifndef TCMI_COPY_CW
#define TCMI_COPY_CW (TCMI_CRC_REG ? 1 : 4)           // bgzf_copy: blocks (wavefronts) per workgroup: they share the CRC tables (A/B: 2, 3; 4 x 64 lanes build the tables)
#endif
constexpr int CW = TCMI_COPY_CW;
constexpr int CRC_NOPS = 12;                        // crc_ops: 1, 2, 4, .. 2048 zero bytes
struct CopyLds { uint8_t win[CWIN]; uint32_t far[FAR_WORDS]; uint2 team[8]; };
struct CopyShared {
    CopyLds w[CW];
#if !TCMI_CRC_REG
    uint32_t t[4][256];         // t[k][v]: the CRC register after byte v and k zero bytes ("slicing by 4")
    uint32_t seg[8][16];        // seg[j][n]: the register n << 4 j, CSEG zero bytes later
#endif
};
#if TCMI_CRC_REG
// ---- TCMI_CRC_REG: the same tables as 16-entry nibble tables across the lanes of registers ----------------------------------------
// Four tables a register (lane 16 q + n: table q, entry n).  R.t[0]: tables 0 - 3, R.t[1]: 4 - 7 — table 2 k + h is t[k] of the nibble
// value n << 4 h; R.s[0], R.s[1]: the operator "CSEG zero bytes later" by nibble j = 0 .. 7 of the register.
struct CrcRegs { uint32_t t[2], s[2]; };
__device__ __forceinline__ uint32_t crc_lane(uint32_t reg, uint32_t table_in_reg, uint32_t nib)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((table_in_reg * 16u + nib) << 2), (int)reg);
}
__device__ __forceinline__ CrcRegs crc_regs_make(const uint32_t *zeros_seg)
{
    const uint32_t lane = threadIdx.x & 63u, q = lane >> 4, n = lane & 15u;
    CrcRegs R;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t id = 4u * r + q, k = id >> 1, h = id & 1u;       // t[k] of n << 4 h
        uint32_t c = n << (4u * h);
        for (uint32_t step = 0; step < 8u * (k + 1u); ++step) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));     // the byte, then k zero bytes
        R.t[r] = c;
        const uint32_t j = 4u * r + q;                                  // nibble j of the register, CSEG zero bytes later
        uint32_t m = 0;
#pragma unroll
        for (uint32_t e = 0; e < 32u; ++e)                              // (uniform indices: the array is a kernel argument)
            m ^= zeros_seg[e] & (0u - (uint32_t)((e >> 2) == j && ((n >> (e & 3u)) & 1u)));
        R.s[r] = m;
    }
    return R;
}
// t[k][v] for a byte v
__device__ __forceinline__ uint32_t crc_tk(const CrcRegs &R, uint32_t k, uint32_t v)
{
    const uint32_t lo = 2u * k, hi = 2u * k + 1u;
    return crc_lane(R.t[lo >> 2], lo & 3u, v & 15u) ^ crc_lane(R.t[hi >> 2], hi & 3u, (v >> 4) & 15u);
}
__device__ __forceinline__ uint32_t crc16_reg(const CrcRegs &R, uint32_t c, uint4 v)
{
    auto x3 = [](uint32_t x, uint32_t y, uint32_t z) { return (uint32_t)__builtin_amdgcn_bitop3_b32(x, y, z, 0x96); };
    auto step = [&](uint32_t x) {
        // byte b of x goes through t[3 - b]: nibble 2 b is table 2 (3 - b), nibble 2 b + 1 table 2 (3 - b) + 1
        const uint32_t a0 = crc_lane(R.t[1], 2u, x & 15u), a1 = crc_lane(R.t[1], 3u, (x >> 4) & 15u);            // t[3]
        const uint32_t b0 = crc_lane(R.t[1], 0u, (x >> 8) & 15u), b1 = crc_lane(R.t[1], 1u, (x >> 12) & 15u);    // t[2]
        const uint32_t c0 = crc_lane(R.t[0], 2u, (x >> 16) & 15u), c1 = crc_lane(R.t[0], 3u, (x >> 20) & 15u);   // t[1]
        const uint32_t d0 = crc_lane(R.t[0], 0u, (x >> 24) & 15u), d1 = crc_lane(R.t[0], 1u, x >> 28);            // t[0]
        return x3(x3(a0, a1, b0), x3(b1, c0, c1), d0 ^ d1);
    };
    c = step(c ^ v.x);
    c = step(c ^ v.y);
    c = step(c ^ v.z);
    return step(c ^ v.w);
}
__device__ __forceinline__ uint32_t crc_seg_reg(const CrcRegs &R, uint32_t c)
{
    auto x3 = [](uint32_t x, uint32_t y, uint32_t z) { return (uint32_t)__builtin_amdgcn_bitop3_b32(x, y, z, 0x96); };
    return x3(x3(crc_lane(R.s[0], 0u, c & 15u), crc_lane(R.s[0], 1u, (c >> 4) & 15u), crc_lane(R.s[0], 2u, (c >> 8) & 15u)),
              x3(crc_lane(R.s[0], 3u, (c >> 12) & 15u), crc_lane(R.s[1], 0u, (c >> 16) & 15u), crc_lane(R.s[1], 1u, (c >> 20) & 15u)),
              crc_lane(R.s[1], 2u, (c >> 24) & 15u) ^ crc_lane(R.s[1], 3u, c >> 28));
}
#endif
static_assert(sizeof(CopyLds) % 16 == 0 && CW * sizeof(CopyLds) + CWIN < 65536, "16-bit LDS addresses in the copy loops");
static_assert((160 * 1024 / sizeof(CopyShared)) * CW >= 14, "at least fourteen blocks per compute unit");

#if !TCMI_CRC_REG
// the CRC register (linear form: starts at 0, no final inversion) after the 16 bytes of v, from state c
__device__ __forceinline__ uint32_t crc16(const uint32_t (*t)[256], uint32_t c, uint4 v)
{
    auto x3 = [](uint32_t x, uint32_t y, uint32_t z) { return (uint32_t)__builtin_amdgcn_bitop3_b32(x, y, z, 0x96); };
    auto step = [&](uint32_t x) { return x3(t[3][x & 0xFFu], t[2][(x >> 8) & 0xFFu], t[1][(x >> 16) & 0xFFu]) ^ t[0][x >> 24]; };
    c = step(c ^ v.x);
    c = step(c ^ v.y);
    c = step(c ^ v.z);
    return step(c ^ v.w);
}
#endif
// a linear operator on the register given by nibble tables (tab[j][n] = op(n << 4 j)): LDS or global memory
__device__ __forceinline__ uint32_t crc_apply(const uint32_t (*tab)[16], uint32_t c)
{
    auto x3 = [](uint32_t x, uint32_t y, uint32_t z) { return (uint32_t)__builtin_amdgcn_bitop3_b32(x, y, z, 0x96); };
    return x3(x3(tab[0][c & 15u], tab[1][(c >> 4) & 15u], tab[2][(c >> 8) & 15u]), x3(tab[3][(c >> 12) & 15u], tab[4][(c >> 16) & 15u], tab[5][(c >> 20) & 15u]),
              tab[6][(c >> 24) & 15u] ^ tab[7][c >> 28]);
}
// A lane's column of a segment: CCOL = CSEG / 64 bytes (32 of the 2 KiB segments, 16 of 1 KiB ones).
constexpr int CCOL = CSEG / 64, CCOL_LOG = CCOL == 32 ? 5 : 4;
static_assert(CCOL == 32 || CCOL == 16, "the CRC's columns: 64 lanes x 16 or 32 bytes a segment");
// XOR over the lanes of (x of lane l, CCOL (63 - l) zero bytes later): a lane that starts a span of 2 s columns takes its right
// neighbour's span (CCOL s bytes) behind its own; lane 0 ends up with all of it (ops[k]: 2^k zero bytes, nibble tables in global memory)
__device__ __forceinline__ uint32_t crc_fold(const uint32_t *ops, uint32_t c)
{
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t right = (uint32_t)__shfl_down((int)c, 1 << k, 64);
        c = crc_apply(reinterpret_cast<const uint32_t (*)[16]>(ops + (size_t)(CCOL_LOG + k) * 128), c) ^ right;
    }
    return c;
}
// bytes of a 16-byte piece that starts at position `at`: those in front of `from` count as zeros, those in [inv, inv + 4) are inverted
// (the block's first four: the standard's all-ones start, in the linear form)
__device__ __forceinline__ uint4 crc_masked(uint4 v, int32_t at, int32_t from, int32_t inv)
{
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t keep = 0, flip = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int32_t p = at + 4 * k + b;
            if (p >= from) keep |= 0xFFu << (8 * b);
            if (p >= from && p >= inv && p < inv + 4) flip |= 0xFFu << (8 * b);
        }
        w[k] = (w[k] & keep) ^ flip;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// TEAMS: with the rounds of teams for batches of short tokens (files that compress less than ~12 : 1: the host picks the variant;
// both are right for any input — the lean one is 4 % faster where no batch would use teams)
// DIRECT: with the short far matches of a teams' batch finished in the batch's set-up (files that compress less than ~4 : 1: most of
// their matches are 3 - 8 bytes long and come from anywhere in the 32 KB window; at 6 : 1 few do and the lean set-up is 3 % faster)
template <bool TEAMS, bool DIRECT>
__global__ __launch_bounds__(64 * CW) __attribute__((amdgpu_waves_per_eu(4, TCMI_CRC_REG ? 5 : 4))) void bgzf_copy(CopyArgs a)
{
    __shared__ __attribute__((aligned(16))) CopyShared S;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (uniform, and known to be: the block's fields go to scalar registers)
#if TCMI_CRC_REG
    const CrcRegs CR = crc_regs_make(a.zeros_seg);
#else
    if (a.crc) {                                    // the tables of the workgroup's four wavefronts (CW * 64 = 256 lanes: an entry each)
        for (uint32_t v = threadIdx.x; v < 256u; v += 64u * CW) {      // the reflected CRC-32 table (polynomial 0xEDB88320)
            uint32_t c = v;
#pragma unroll
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            S.t[0][v] = c;
        }
        for (uint32_t v = threadIdx.x; v < 128u; v += 64u * CW) {
            const uint32_t j = v >> 4, n = v & 15u;
            uint32_t m = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) m ^= a.zeros_seg[4 * j + i] & (0u - ((n >> i) & 1u));
            S.seg[j][n] = m;
        }
        __syncthreads();
        for (int k = 1; k < 4; ++k) {               // one more zero byte behind it
            for (uint32_t v = threadIdx.x; v < 256u; v += 64u * CW) { const uint32_t c = S.t[k - 1][v]; S.t[k][v] = S.t[0][c & 0xFFu] ^ (c >> 8); }
            __syncthreads();
        }
    }
#endif
    CopyLds &s_lds = S.w[wave];
    uint8_t *const s_win = s_lds.win;
    const uint32_t B = (uint32_t)reinterpret_cast<uintptr_t>(s_win);    // the ring's LDS address (the copy loops take addresses, not ring indices)
    const int blk = a.first_block + (int)blockIdx.x * CW + wave;
    if (blk >= a.n_blocks) return;
    const BlockDesc d = a.blocks[blk];
    const uint32_t ulen = d.ulen;
    uint32_t err = uni(a.status[blk]);
    const uint32_t ntok = err == ST_OK ? uni(a.n_tok[blk]) : 0u;
    const uint32_t *toks = a.tokens + d.tok;
    // The blocks' outputs follow each other in the stream without gaps, so this block's starts at any byte.  Positions in this
    // kernel count from the 16-byte boundary in front of it (`a0` bytes of the previous block come first and are never touched):
    // ring index and stream address of a byte are then equal modulo 16 and the flush can use 16-byte rows.
    const uint32_t a0 = (uint32_t)(d.uout & 15u);
    uint8_t *const out = a.out + (d.uout - a0);
    const uint32_t vend = a0 + ulen;    // the block's end
    const uint8_t *const payload = a.file + d.cin;
    uint32_t *slots = a.rec_slot + (size_t)blk * MAX_REC_PER_BLOCK;
    const uint32_t *const win32 = reinterpret_cast<const uint32_t *>(s_win);
    const uint32_t lane_hi = ((uint32_t)lane << 16) | 0xFFFFu;  // (len << 16 | anything) > lane_hi  <=>  len > lane: the copy round's lane mask from the packed operand
    // per-lane constants of TCMI_LM_ASM's two copy rounds (bytes: lane + 7, lane << 16; dwords: 4 lane, 32 lane, lane 0's 31)
    const uint32_t lane_p7 = (uint32_t)lane + 7u, lane_sh16 = (uint32_t)lane << 16, lane_x4 = (uint32_t)lane * 4u, lane_x32 = (uint32_t)lane * 32u;
    const uint32_t lane0_31 = lane == 0 ? 31u : 0u;
    // (teams of eight lanes: lane l belongs to team l / 8 and takes that team's piece l % 8)
    const uint32_t team_of = (uint32_t)lane >> 3, team_sub = (uint32_t)lane & 7u, team_sub8 = team_sub * 8u;
    const uint32_t team_base = B + (uint32_t)(CWIN + FAR_WORDS * 4), team_slot = team_base + team_of * 8u;     // s_lds.team, as LDS addresses

    uint32_t op = a0, flushed = 0;
    uint32_t next_rec = d.entry >= 0 ? a0 + (uint32_t)d.entry : 0xFFFFFFF0u;
    bool searching = d.entry == -2;     // the block's first record start is still to be found, from `search_pos` on
    uint32_t search_pos = a0;
    uint32_t first_rec = d.entry >= 0 ? (uint32_t)d.entry : 0xFFFFFFFFu;
    uint32_t rec_size = 0;              // 4 + block_size of the last record listed (0: none yet)
    uint32_t n_rec = 0;
    uint32_t next_evt = 0;
    uint32_t bad = 0;
    bool tail_unknown = false;

    // four bytes of the ring at any position
    auto ring_u32 = [&](uint32_t x) __attribute__((always_inline)) {
        const uint32_t i = (x & CWMASK) >> 2;
        return __builtin_amdgcn_alignbit(win32[(i + 1) & (CWIN / 4 - 1)], win32[i], (x & 3u) * 8u);
    };
    // Could an alignment record start at c (its first 40 bytes are in the ring)?  block_size, refID, pos, l_read_name, the variable
    // lengths against block_size, next_refID — what BAM readers that must find a record in the middle of a file test.  A wrong yes
    // is caught by the host: the chain of records through all blocks must close.
    // (`avail`: bytes of the candidate that lie in this block — at the block's end fewer than the 36 of the fixed fields; what
    // is not there is not tested)
    auto plausible = [&](uint32_t c, uint32_t avail) __attribute__((always_inline)) {
        const uint32_t bs = ring_u32(c), refid = ring_u32(c + 4), pos = ring_u32(c + 8), w2 = ring_u32(c + 12), w3 = ring_u32(c + 16);
        const uint32_t l_seq = ring_u32(c + 20), nref = ring_u32(c + 24), npos = ring_u32(c + 28);
        const uint32_t l_name = w2 & 0xFFu, n_cig = w3 & 0xFFFFu;
        const uint64_t need = 32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + l_seq;
        bool ok = avail >= 4u && bs >= 33u && bs < (1u << 24);
        if (avail >= 8u) ok = ok && refid + 1u <= a.n_ref;
        if (avail >= 12u) ok = ok && (int32_t)pos >= -1;
        if (avail >= 13u) ok = ok && l_name >= 1u;
        if (avail >= 24u) ok = ok && l_seq < (1u << 28) && need <= bs;
        if (avail >= 28u) ok = ok && nref + 1u <= a.n_ref;
        if (avail >= 32u) ok = ok && (int32_t)npos >= -1;
        return ok;
    };

    const bool do_crc = a.crc != 0;
    uint32_t crc_acc = 0;               // this lane's column of the flushed segments (linear form)
    // List the record starts whose block_size field is complete, flush the segments that are complete.  The chain of records is
    // serial (a record's start is known when its predecessor's size is), but the records of a BAM block mostly have one size: 16
    // lanes look at where the next 16 records start if they all have the size of the last one, and the chain advances over all
    // that do (at least one per step: the first candidate is a record start for sure).
    auto housekeeping = [&]() __attribute__((always_inline)) {
        while (searching && search_pos + 40u <= op) {            // 64 candidates at a time
            const uint32_t c = search_pos + (uint32_t)lane;
            const unsigned long long hit = __ballot(c + 40u <= op && c < vend && plausible(c, 40u));
            if (hit) {
                next_rec = search_pos + (uint32_t)__builtin_ctzll(hit);
                first_rec = next_rec - a0;
                searching = false;
            } else {
                search_pos = min(search_pos + 64u, op - 39u);
                if (search_pos >= vend) searching = false;
            }
        }
        while (next_rec + 4 <= op) {
            const uint32_t cand = next_rec + (uint32_t)lane * rec_size;
            const bool look = lane < 16 && (lane == 0 || rec_size != 0) && cand + 4 <= op;
            uint32_t bs = 0;
            if (look) {
                const uint32_t i = (cand & CWMASK) >> 2;
                bs = __builtin_amdgcn_alignbit(win32[(i + 1) & (CWIN / 4 - 1)], win32[i], (cand & 3u) * 8u);
            }
            const uint32_t n_look = (uint32_t)__popcll(__ballot(look));                         // (a prefix of the lanes)
            const uint32_t same = (uint32_t)__builtin_ctzll(~__ballot(look && bs + 4u == rec_size));  // leading candidates of the same size
            uint32_t n_conf;
            if (same < n_look) {
                // candidate `same` starts a record of another size (or the first one at all)
                const uint32_t ubs = (uint32_t)__builtin_amdgcn_readlane((int)bs, (int)same);
                if (__builtin_expect(ubs - 32u > (1u << 28) - 32u, 0)) { err = ST_BAD_RECORD; next_rec = 0xFFFFFFF0u; break; }
                n_conf = same + 1u;
                next_rec += same * rec_size + 4u + ubs;
                rec_size = 4u + ubs;
            } else {
                n_conf = n_look;
                next_rec += n_look * rec_size;
            }
            if (n_rec + n_conf > (uint32_t)MAX_REC_PER_BLOCK) { err = ST_BAD_RECORD; next_rec = 0xFFFFFFF0u; break; }
            if ((uint32_t)lane < n_conf) slots[n_rec + (uint32_t)lane] = cand - a0;
            n_rec += n_conf;
        }
        while (op - flushed >= CSEG) {
            const uint4 *src = reinterpret_cast<const uint4 *>(s_win + (flushed & CWMASK));
            uint4 *dst = reinterpret_cast<uint4 *>(out + flushed);
            if (flushed == 0 && a0 != 0) {                       // the block's first row: its first bytes are the previous block's
                if (lane == 0) { for (uint32_t i = a0; i < 16u; ++i) out[i] = s_win[i]; }
                else dst[lane] = src[lane];
            } else dst[lane] = src[lane];
#pragma unroll
            for (int k = 1; k < CSEG / 16 / 64; ++k) dst[k * 64 + lane] = src[k * 64 + lane];
            if (do_crc) {
                // the segment's CRC while it is in the ring: lane l takes the CCOL bytes at CCOL l (its column: the register of the column's
                // bytes so far, CSEG zero bytes later, plus these)
                uint4 p0 = src[(CCOL / 16) * lane], p1 = CCOL == 32 ? src[2 * lane + 1] : make_uint4(0u, 0u, 0u, 0u);
                if (flushed == 0) { p0 = crc_masked(p0, CCOL * lane, (int32_t)a0, (int32_t)a0); if (CCOL == 32) p1 = crc_masked(p1, 32 * lane + 16, (int32_t)a0, (int32_t)a0); }
#if TCMI_CRC_REG
                uint32_t cs = crc16_reg(CR, 0u, p0);
                if (CCOL == 32) cs = crc16_reg(CR, cs, p1);
                crc_acc = crc_seg_reg(CR, crc_acc) ^ cs;
#else
                uint32_t cs = crc16(S.t, 0u, p0);
                if (CCOL == 32) cs = crc16(S.t, cs, p1);
                crc_acc = crc_apply(S.seg, crc_acc) ^ cs;
#endif
            }
            flushed += CSEG;
        }
        next_evt = flushed + (uint32_t)CSEG;
    };
    // a match of any kind: all lanes; with dist < len the pattern of the last `dist` bytes repeats
    auto copy_any = [&](uint32_t at, uint32_t len, uint32_t dist) __attribute__((always_inline)) {
        if (dist + a0 > at) { bad = 1; return; }                 // before the block's first byte
        if (dist > (uint32_t)CNEAR) {
            const uint8_t *src = out + (at - dist);             // flushed by this wavefront (see CNEAR)
#pragma clang loop vectorize(disable) unroll(disable)
            for (uint32_t i = (uint32_t)lane; i < len; i += 64) s_win[(at + i) & CWMASK] = src[i];
        } else if (dist >= len) {
#pragma clang loop vectorize(disable) unroll(disable)
            for (uint32_t i = (uint32_t)lane; i < len; i += 64) s_win[(at + i) & CWMASK] = s_win[(at + i - dist) & CWMASK];
        } else {
            const float inv = 1.0f / (float)dist;
#pragma clang loop vectorize(disable) unroll(disable)
            for (int i = lane; i < (int)len; i += 64) {
                int qd = (int)((float)i * inv);
                int r = i - qd * (int)dist;
                if (r < 0) r += (int)dist;
                if (r >= (int)dist) r -= (int)dist;
                s_win[(at + i) & CWMASK] = s_win[(at - dist + r) & CWMASK];
            }
        }
    };
    if (a.stamps && lane < 16) a.stamps[(size_t)blk * 16 + lane] = 0;
    TCMI_STAMP(a.stamps, blk, 0);
#ifdef TCMI_COPY_PHASES                 // (diagnostic build: where a block's cycles go — batch set-up, match loop, other matches, housekeeping)
    uint64_t ph_t = __builtin_amdgcn_s_memtime(), ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t hist[6] = {0, 0, 0, 0, 0, 0};     // plain matches of < 8, 8 - 64, 65 - 128, 129 - 192, 193 - 256, 257+ bytes
#define PH(k_) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); ph[k_] += now_ - ph_t; ph_t = now_; } while (0)
#else
#define PH(k_) do { } while (0)
#endif
    uint32_t n_match = 0, n_slow = 0, n_round = 0;
    const uint32_t n_team = 0, n_teamed = 0;
    housekeeping();
    // the tokens of the batch that starts at `base`, a token a lane: consecutive words (bgzf_symbols left them in order).
    // ONE load, of every lane, outside any branch, masked where it is used: a load under a condition makes the compiler wait for it on
    // the spot, and the batch's copy loops would start a trip to memory later (2.5 : 1: 1 035 -> 977 us).
    bool t_has = false;
    auto fetch_tokens = [&](uint32_t base) __attribute__((always_inline)) {
        const uint32_t g = base + (uint32_t)lane;
        t_has = g < ntok;
        return toks[t_has ? g : 0u];                        // (a lane without a token reads word 0 and drops it)
    };
    uint32_t t_ahead = fetch_tokens(0);         // (a batch's tokens are asked for while the batch before is copied: HBM is a microsecond away)
    bool t_ahead_has = t_has;
    for (uint32_t base = 0; base < ntok && err == ST_OK && !bad; base += 64) {
        const uint32_t t = t_ahead_has ? t_ahead : 0u;
        const bool is_lit = (t >> 31) != 0;
        const bool is_raw = !is_lit && (t & TOK_RAW);
        if (__builtin_expect(__ballot(is_raw) != 0, 0)) {
            // ---- a batch with stored bytes in it: token by token (rare: incompressible data, flush markers) --------------------
            const uint32_t nb = min(64u, ntok - base);
            for (uint32_t j = 0; j < nb && err == ST_OK && !bad; ++j) {
                const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)t, (int)j);
                if (tj >> 31) {
                    const uint32_t nl = TEAMS ? 1u + ((tj >> 24) & 3u) : 1u;         // (one literal, or — files of short tokens — two in one token)
                    if (op + nl > vend) { err = ST_BAD_LENGTH; break; }
                    s_win[op & CWMASK] = (uint8_t)tj;
                    if (nl > 1u) s_win[(op + 1u) & CWMASK] = (uint8_t)(tj >> 8);
                    op += nl;
                } else if (tj & TOK_RAW) {
                    uint32_t len = (tj >> 17) & 0x1FFFu;
                    const uint8_t *src = payload + (tj & 0x1FFFFu);
                    if (op + len > vend) { err = ST_BAD_LENGTH; break; }
                    while (len) {
                        const uint32_t n = min(len, (uint32_t)CSEG - (op & (CSEG - 1)));
#pragma clang loop vectorize(disable) unroll(disable)
                        for (uint32_t i = lane; i < n; i += 64) s_win[(op + i) & CWMASK] = src[i];
                        op += n; src += n; len -= n;
                        if (op >= next_evt) { housekeeping(); if (err != ST_OK) break; }
                    }
                } else {
                    const uint32_t len = tj & 511u, dist = ((tj >> 9) & 0x7FFFu) + 1u;
                    if (op + len > vend) { err = ST_BAD_LENGTH; break; }
                    copy_any(op, len, dist);
                    op += len;
                }
                if (op >= next_evt) housekeeping();
            }
            t_ahead = fetch_tokens(base + 64u); t_ahead_has = t_has;
            continue;
        }
#if TCMI_COPY_PHASES >= 2
        PH(5);
#endif
        const uint32_t mylen = is_lit ? (TEAMS ? 1u + ((t >> 24) & 3u) : 1u) : (t & 511u);       // (a literal token carries one byte, or — bgzf_symbols<1, *>, whose files get this kernel's TEAMS variants — two)
        const uint32_t dist = ((t >> 9) & 0x7FFFu) + 1u;
        const uint32_t incl = wave_scan_add(mylen);
        const uint32_t dst = op + incl - mylen;                 // where this lane's token starts
        const uint32_t batch_end = op + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (batch_end > vend) { err = ST_BAD_LENGTH; break; }
        // what the copy loop needs of a match, ready in two registers: ring addresses of its destination and source, its length, and
        // whether it is one of the plain ones — source in the ring, source and destination apart by the match's length at least
        // (a round copies the whole match at once), neither range across the ring's end, and for the dword rounds of a match beyond
        // 64 bytes the source not within the ring's first four bytes.  The others (far, period shorter than the match, across the
        // end) take copy_any.
#if TCMI_COPY_PHASES >= 2
        PH(6);
#endif
        const bool is_match = !is_lit && mylen != 0;
        const uint32_t dm = dst & CWMASK, sm = (dst - dist) & CWMASK;
        const bool plain = dist <= (uint32_t)CNEAR && dist + a0 <= dst && dist >= mylen && dm + mylen <= (uint32_t)CWIN && sm + mylen <= (uint32_t)CWIN && (mylen <= 64u || sm >= 4u);
        const bool far_ok = dist > (uint32_t)CNEAR && dist + a0 <= dst && dm + mylen <= (uint32_t)CWIN;      // (its source is flushed when its turn comes: CNEAR)
        uint32_t vA = (B + dm) | (mylen << 16), vB = (B + sm) | (plain ? 0u : far_ok ? 1u << 16 : 2u << 16);     // (LDS addresses: B + ring index, below 64 K)
        const uint32_t vC = dst - dist;                         // a far match's source, as a position
        // A match that reaches back further than the ring holds reads what this wavefront flushed long ago — from HBM, a microsecond
        // away if it is fetched when the match comes up.  So the far matches of the batch whose sources are flushed already (all of
        // them, unless the batch is several KB of output long) are fetched NOW, every lane its own match's bytes, into a few
        // hundred bytes of LDS next to the ring; to the copy loop below they are plain matches whose source lies there.
#if TCMI_COPY_PHASES == 3
        PH(7);
#endif
        // Teams' batches (short tokens: data that compresses like real data, whose matches are mostly 3 - 8 bytes from anywhere in
        // the 32 KB behind): a far match of up to 8 bytes is FINISHED here — three words from the flushed stream, its bytes straight
        // to their place in the ring (exactly `len` of them: lanes write next to each other) — instead of being parked and copied by
        // a team later: 17 + 12 instructions for all of them, and the teams' rounds are left with the near matches.  (The batch is at
        // most team_bytes long: what these writes replace in the ring was flushed long ago and is further back than CNEAR.)
        const bool use_teams = TEAMS && uni(batch_end - op <= a.team_bytes ? 1u : 0u) != 0u;
        bool done = false;
        const bool far_now = is_match && far_ok && mylen <= 64u && dst - dist + mylen <= flushed;     // far, flushed, and short enough to be fetched here
        const unsigned long long far_mask = __ballot(far_now);
        if (DIRECT && use_teams && far_mask) {
            const uint32_t src = dst - dist;
            done = far_now && mylen <= 8u;
            const unsigned long long dmask = __ballot(done);
            if (dmask) {
                uint32_t w0 = 0, w1 = 0, w2 = 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (this wavefront's own flush stores)
                if (done) {
                    const uint32_t *g = reinterpret_cast<const uint32_t *>(out + (src & ~3u));
                    w0 = g[0]; w1 = g[1];
                    if ((src & 3u) + mylen > 8u) w2 = g[2];
                }
                const uint32_t sh = (src & 3u) * 8u;
                const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
                uint32_t lo8, hi8;
                asm volatile(
                    "s_mov_b64 s[92:93], exec\n"
                    "s_mov_b64 exec, %[dmask]\n"
                    "v_lshrrev_b32 %[lo8], 8, %[lo]\n"
                    "v_lshrrev_b32 %[hi8], 8, %[hi]\n"
                    "ds_write_b8 %[at], %[lo]\n"
                    "ds_write_b8 %[at], %[lo8] offset:1\n"
                    "ds_write_b8_d16_hi %[at], %[lo] offset:2\n"
                    "v_cmpx_lt_u32 vcc, 3, %[len]\n"
                    "ds_write_b8_d16_hi %[at], %[lo8] offset:3\n"
                    "v_cmpx_lt_u32 vcc, 4, %[len]\n"
                    "ds_write_b8 %[at], %[hi] offset:4\n"
                    "v_cmpx_lt_u32 vcc, 5, %[len]\n"
                    "ds_write_b8 %[at], %[hi8] offset:5\n"
                    "v_cmpx_lt_u32 vcc, 6, %[len]\n"
                    "ds_write_b8_d16_hi %[at], %[hi] offset:6\n"
                    "v_cmpx_lt_u32 vcc, 7, %[len]\n"
                    "ds_write_b8_d16_hi %[at], %[hi8] offset:7\n"
                    "s_mov_b64 exec, s[92:93]\n"
                    : [lo8] "=&v"(lo8), [hi8] "=&v"(hi8)
                    : [dmask] "s"(dmask), [lo] "v"(lo), [hi] "v"(hi), [at] "v"(B + dm), [len] "v"(mylen)
                    : "s92", "s93", "vcc", "memory");
            }
        }
        {
            const uint32_t src = dst - dist;                    // (position of the source's first byte)
            const bool fetch = far_now && !done;                // (longer ones: 64 bytes an instruction in the loop)
            if (far_mask && __ballot(fetch)) {
                const uint32_t words = fetch ? (mylen + 3u) >> 2 : 0u;
                const uint32_t end_w = wave_scan_add(words);
                const bool take = fetch && end_w <= (uint32_t)FAR_WORDS;
                if (__ballot(take)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (this wavefront's own flush stores)
                    const uint32_t *g = reinterpret_cast<const uint32_t *>(out + (src & ~3u));
                    const uint32_t sh = (src & 3u) * 8u;
                    uint32_t *park = s_lds.far + (end_w - words);
                    const uint32_t n = take ? words : 0u;
                    for (uint32_t k = 0; __ballot(k < n); k += 8) {           // eight words a turn, nine loads in flight: a turn is a trip to HBM
                        uint32_t w[9];                                      // (32 bytes and less — most far matches — in one)
#pragma unroll
                        for (int i = 0; i < 9; ++i) w[i] = k + i <= n && k < n ? g[k + i] : 0u;
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (k + i < n) park[k + i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], sh);
                    }
                    if (take) vB = B + (uint32_t)CWIN + 4u * (end_w - words);
                }
            }
        }
        // Matches whose source lies wholly in front of the first match still to be copied do not depend on it: up to eight of them are
        // copied at a time, by a TEAM of eight lanes each (eight bytes a lane and step).  `srcend`: the position behind a match's source
        // (a parked one's lies in flushed output: 0, always ready; one that is not plain never joins a team).
        // (Worth it where tokens are short: a round of teams costs about three single matches' instructions, and in a batch of long
        // matches — the same record 289 bytes back, say — only two or three matches at a time are independent.)
#if TCMI_COPY_PHASES == 2
        PH(7);
#elif TCMI_COPY_PHASES == 3
        PH(3);
#endif
        const bool teamable = is_match && !done && (vB >> 16) == 0u;
        const unsigned long long plain_mask = __ballot(teamable);  // (plain matches, parked far ones included)
        // TCMI_LM_ASM's operands of a plain match.  Up to 64 bytes, a byte a lane: (len - 1) << 16 | (destination - 7) & 0xffff and
        // source - 7 (lane + 7 is added to both).  Longer ones, an ALIGNED destination dword a lane (LDS takes unaligned words at a
        // fifth of the rate): the first dword's address | 8 (destination & 3) << 16; the aligned address of the source dword that
        // holds the first dword's byte 0 | 8 (its place in it) << 16; and 32 - 8 (bytes from the first dword's start to the match's
        // end): + 32 lane = how far a lane's mask of bytes is to be shifted down (< 32: the lane has bytes at all).
        const uint32_t vA2 = ((mylen - 1u) << 16) | ((B + dm - 7u) & 0xFFFFu), vB2 = vB - 7u;
        const uint32_t hoff = dm & 3u, s0 = sm - hoff;
        const uint32_t vXl = ((B + dm) & ~3u) | (hoff * 8u) << 16, vYl = ((B + s0) & 0xFFFCu) | ((s0 & 3u) * 8u) << 16, vKl = 32u - 8u * (hoff + mylen);
        const uint32_t srcend = teamable ? (vB >= B + (uint32_t)CWIN ? 0u : dst - dist + mylen) : 0xFFFFFFFFu;
        const unsigned long long team_mask = plain_mask;
        uint32_t t_cur = 0;
        // (asked for HERE, behind the batch's set-up and its waits for earlier loads, in front of the copy loops: the load is under
        //  way while the batch is copied)
        t_ahead = fetch_tokens(base + 64u); t_ahead_has = t_has;
#ifdef TCMI_COPY_PHASES
        hist[0] += __popcll(__ballot(teamable && mylen < 8u)); hist[1] += __popcll(__ballot(teamable && mylen >= 8u && mylen <= 64u));
        hist[2] += __popcll(__ballot(teamable && mylen > 64u && mylen <= 128u)); hist[3] += __popcll(__ballot(teamable && mylen > 128u && mylen <= 192u));
        hist[4] += __popcll(__ballot(teamable && mylen > 192u && mylen <= 256u)); hist[5] += __popcll(__ballot(teamable && mylen > 256u));
#endif
        PH(0);
        while (t_cur < 64u) {
            // the tokens [t_cur, t_stop) start in front of the next housekeeping stop: their literals at once, their matches in order
            const unsigned long long from = ~0ull << t_cur;
            const unsigned long long ge = __ballot(dst >= next_evt) & from;
            const uint32_t t_stop = ge ? (uint32_t)__builtin_ctzll(ge) : 64u;
            const unsigned long long rng = t_stop < 64u ? from & ~(~0ull << t_stop) : from;
            const bool mine = (rng >> lane) & 1ull;
            if (mine && is_lit) {
                s_win[dm] = (uint8_t)t;
                if (TEAMS && (t & TOK_LIT2)) s_win[(dm + 1u) & CWMASK] = (uint8_t)(t >> 8);
            }
            unsigned long long mm = __ballot(mine && is_match && !done);
            n_match += (uint32_t)__popcll(mm);
            while (mm) {
                if (use_teams) {
                    // Rounds of teams, hand-scheduled (about 50 instructions a round + 10 per further 64 bytes of the longest match; the
                    // compiler's version of the same took ~90): r = the matches whose source ends in front of F, where the first match
                    // still to be copied starts (+ that one itself, if it is plain: its steps of 64 bytes come in order); the first eight
                    // of them leave {vA, vB} in a slot each, every lane reads its team's slot; a match of up to 8 bytes is copied byte by
                    // byte (lane s of the team: byte s), a longer one in 8-byte pieces at min(8 s + 64 k, len - 8).  Leaves when fewer
                    // than two matches are ready (the single-match loop below takes the first one).
                    uint32_t f_, F_, n_;
                    asm volatile(
                        "s_mov_b64 s[92:93], exec\n"
                        "TL%=:\n"
                        "s_ff1_i32_b64 %[f], %[mm]\n"
                        "v_readlane_b32 %[F], %[vdst], %[f]\n"
                        "s_lshl_b64 s[84:85], 1, %[f]\n"
                        "s_and_b64 s[84:85], s[84:85], %[tmask]\n"
                        "v_cmp_ge_u32 vcc, %[F], %[vsrcend]\n"
                        "s_or_b64 s[80:81], vcc, s[84:85]\n"
                        "s_and_b64 s[80:81], s[80:81], %[mm]\n"
                        "s_bcnt1_i32_b64 %[n], s[80:81]\n"
                        "s_cmp_lt_u32 %[n], 2\n"
                        "s_cbranch_scc1 TX%=\n"
                        "v_mbcnt_lo_u32_b32 v48, s80, 0\n"
                        "v_mbcnt_hi_u32_b32 v48, s81, v48\n"
                        "v_cmp_gt_u32 vcc, 8, v48\n"
                        "s_and_b64 s[82:83], vcc, s[80:81]\n"
                        "s_mov_b64 exec, s[82:83]\n"
                        "v_lshl_add_u32 v49, v48, 3, %[sK]\n"
                        "ds_write2_b32 v49, %[vA], %[vB] offset1:1\n"
                        "s_mov_b64 exec, s[92:93]\n"
                        "s_bcnt1_i32_b64 %[n], s[82:83]\n"
                        "s_andn2_b64 %[mm], %[mm], s[82:83]\n"
                        "ds_read2_b32 v[56:57], %[vslot] offset1:1\n"
                        "s_waitcnt lgkmcnt(0)\n"
                        "v_lshrrev_b32 v58, 16, v56\n"
                        "v_and_b32 v59, 0xffff, v56\n"
                        "v_cmp_gt_u32 vcc, %[n], %[vT]\n"
                        "v_cmp_gt_u32 s[84:85], 9, v58\n"
                        "v_cmp_gt_u32 s[86:87], v58, %[vsub]\n"
                        "v_cmp_gt_u32 s[88:89], v58, %[vsub8]\n"
                        "s_and_b64 s[86:87], s[86:87], s[84:85]\n"
                        "s_andn2_b64 s[88:89], s[88:89], s[84:85]\n"
                        "s_and_b64 s[86:87], s[86:87], vcc\n"
                        "s_and_b64 s[88:89], s[88:89], vcc\n"
                        "s_mov_b64 exec, s[86:87]\n"
                        "v_add_u32 v60, v57, %[vsub]\n"
                        "v_add_u32 v61, v59, %[vsub]\n"
                        "ds_read_u8 v62, v60\n"
                        "s_mov_b64 exec, s[88:89]\n"
                        "v_subrev_u32 v50, 8, v58\n"
                        "v_min_u32 v51, v50, %[vsub8]\n"
                        "v_add_u32 v52, v57, v51\n"
                        "v_add_u32 v53, v59, v51\n"
                        "ds_read_b64 v[54:55], v52\n"
                        "s_waitcnt lgkmcnt(0)\n"
                        "ds_write_b64 v53, v[54:55]\n"
                        "s_mov_b64 exec, s[86:87]\n"
                        "ds_write_b8 v61, v62\n"
                        "s_mov_b64 exec, s[88:89]\n"
                        "v_mov_b32 v60, %[vsub8]\n"
                        "TW%=:\n"
                        "v_add_u32 v60, 64, v60\n"
                        "v_cmp_gt_u32 vcc, v58, v60\n"
                        "s_and_b64 exec, exec, vcc\n"
                        "s_cbranch_scc0 TE%=\n"
                        "v_min_u32 v51, v50, v60\n"
                        "v_add_u32 v52, v57, v51\n"
                        "v_add_u32 v53, v59, v51\n"
                        "ds_read_b64 v[54:55], v52\n"
                        "s_waitcnt lgkmcnt(0)\n"
                        "ds_write_b64 v53, v[54:55]\n"
                        "s_branch TW%=\n"
                        "TE%=:\n"
                        "s_mov_b64 exec, s[92:93]\n"
                        "s_cmp_lg_u64 %[mm], 0\n"
                        "s_cbranch_scc1 TL%=\n"
                        "TX%=:\n"
                        "s_mov_b64 exec, s[92:93]\n"
                        : [mm] "+s"(mm), [f] "=&s"(f_), [F] "=&s"(F_), [n] "=&s"(n_)
                        : [vA] "v"(vA), [vB] "v"(vB), [vdst] "v"(dst), [vsrcend] "v"(srcend), [tmask] "s"(team_mask), [vT] "v"(team_of), [vsub] "v"(team_sub),
                          [vsub8] "v"(team_sub8), [vslot] "v"(team_slot), [sK] "s"(team_base)
                        : "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s92", "s93", "vcc", "scc", "memory", "v48", "v49", "v50", "v51", "v52",
                          "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62");
                    if (!mm) break;
                }
                // the matches of the stretch, one after the other (TCMI_LM_ASM), until one comes up that C++ copies: j says which (-1: none
                // left).  (With teams: one.)
                int j;
                {
                    uint32_t sa, sb, len, t0, t1, t2;
                    // (with teams: this match only — the loop is handed a set of one)
                    const unsigned long long rest = use_teams ? mm & (mm - 1ull) : 0ull;
                    mm ^= rest;
                    PH(1);
                    TCMI_LM_ASM()
                    PH(2);
                    mm |= rest;
                }
                if (j < 0) { if (use_teams) continue; break; }
                mm &= ~(1ull << j);
                copy_any((uint32_t)__builtin_amdgcn_readlane((int)dst, j), (uint32_t)__builtin_amdgcn_readlane((int)mylen, j),
                         (uint32_t)__builtin_amdgcn_readlane((int)dist, j));
                ++n_slow;
                PH(3);
            }
            op = t_stop < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)dst, (int)t_stop) : batch_end;
            t_cur = t_stop;
            ++n_round;
            PH(1);
            if (op >= next_evt) { housekeeping(); if (err != ST_OK) break; }
            PH(4);
            if (bad) break;
        }
    }
    if (bad && err == ST_OK) err = ST_BAD_STREAM;
    if (err == ST_OK && op != vend) err = ST_BAD_LENGTH;
    if (err == ST_OK) {
        housekeeping();
        while (searching && search_pos + 4u <= vend) {           // the block's last 39 bytes: what there is of a record's fixed fields
            const uint32_t c = search_pos + (uint32_t)lane;
            const unsigned long long hit = __ballot(c + 4u <= vend && plausible(c, vend - c));
            if (hit) {
                next_rec = search_pos + (uint32_t)__builtin_ctzll(hit);
                first_rec = next_rec - a0;
                searching = false;
                housekeeping();                                 // (its chain, as far as the block goes)
            } else search_pos += 64u;
        }
        // a record that starts within the block's last three bytes: its start is listed, its size is read from the stream later
        if (next_rec < vend && next_rec + 4 > vend) {
            if (n_rec < (uint32_t)MAX_REC_PER_BLOCK) { if (lane == 0) slots[n_rec] = next_rec - a0; ++n_rec; tail_unknown = true; }
            else err = ST_BAD_RECORD;
        }
        for (uint32_t i = max(flushed, a0) + (uint32_t)lane; i < op; i += 64) out[i] = s_win[i & CWMASK];
        if (do_crc) {
            // ---- the block's CRC-32 against its trailer.  In the linear form crc(A || B) = later(crc(A), |B|) ^ crc(B) and zero bytes in
            // front of a message leave the register at zero: the columns of the flushed segments are joined across the lanes, moved
            // past the tail, and the tail — what lies in the ring behind the last whole segment, cut into CCOL-byte pieces from its END,
            // lane l the piece that ends CCOL (63 - l) bytes in front of the block's end — is joined the same way.
            const uint8_t *e = a.file + d.cin + d.clen;         // the block's trailer: CRC32, ISIZE (little endian)
            const uint32_t want = (uint32_t)e[0] | ((uint32_t)e[1] << 8) | ((uint32_t)e[2] << 16) | ((uint32_t)e[3] << 24);
            uint32_t got;
            if (ulen < 128u) {                                  // (short blocks — the end-of-file marker's is empty — byte by byte)
                uint32_t t = 0xFFFFFFFFu;
#if TCMI_CRC_REG
                for (uint32_t i = 0; i < ulen; ++i) t = crc_tk(CR, 0u, (t ^ s_win[(a0 + i) & CWMASK]) & 0xFFu) ^ (t >> 8);
#else
                for (uint32_t i = 0; i < ulen; ++i) t = S.t[0][(t ^ s_win[(a0 + i) & CWMASK]) & 0xFFu] ^ (t >> 8);
#endif
                got = ~t;
            } else {
                const int32_t from = (int32_t)max(flushed, a0);
                const int32_t ps = (int32_t)vend - CCOL * (64 - lane);                  // where this lane's piece of the tail starts
                uint32_t tl = 0;
#if TCMI_CRC_REG
                {   // (every lane goes through the look-ups — they read the tables from each other's registers; a piece wholly in front of `from` is zeros)
                    const uint4 q0 = make_uint4(ring_u32((uint32_t)ps), ring_u32((uint32_t)ps + 4u), ring_u32((uint32_t)ps + 8u), ring_u32((uint32_t)ps + 12u));
                    tl = crc16_reg(CR, 0u, crc_masked(q0, ps, from, (int32_t)a0));
                    if (CCOL == 32) {
                        const uint4 q1 = make_uint4(ring_u32((uint32_t)ps + 16u), ring_u32((uint32_t)ps + 20u), ring_u32((uint32_t)ps + 24u), ring_u32((uint32_t)ps + 28u));
                        tl = crc16_reg(CR, tl, crc_masked(q1, ps + 16, from, (int32_t)a0));
                    }
                }
#else
                if (ps + CCOL > from) {
                    const uint4 q0 = make_uint4(ring_u32((uint32_t)ps), ring_u32((uint32_t)ps + 4u), ring_u32((uint32_t)ps + 8u), ring_u32((uint32_t)ps + 12u));
                    tl = crc16(S.t, 0u, crc_masked(q0, ps, from, (int32_t)a0));
                    if (CCOL == 32) {
                        const uint4 q1 = make_uint4(ring_u32((uint32_t)ps + 16u), ring_u32((uint32_t)ps + 20u), ring_u32((uint32_t)ps + 24u), ring_u32((uint32_t)ps + 28u));
                        tl = crc16(S.t, tl, crc_masked(q1, ps + 16, from, (int32_t)a0));
                    }
                }
#endif
                uint32_t full = uni(crc_fold(a.crc_ops, crc_acc));
                const uint32_t tail_len = vend - flushed;       // < CSEG
                for (int k = 0; k < 11; ++k)
                    if ((tail_len >> k) & 1u) full = crc_apply(reinterpret_cast<const uint32_t (*)[16]>(a.crc_ops + (size_t)k * 128), full);
                got = ~(full ^ uni(crc_fold(a.crc_ops, tl)));
            }
            if (got != want) err = ST_BAD_CRC;
        }
    }
    if (lane == 0) {
        a.status[blk] = err;
        a.n_rec[blk] = n_rec;
        a.first_rec[blk] = first_rec;
        a.overshoot[blk] = tail_unknown ? 0x7FFFFFFF : first_rec != 0xFFFFFFFFu && next_rec < 0xFFFFFFF0u ? (int32_t)(next_rec - vend) : 0;
    }
    if (a.stamps && lane == 0) {
        uint64_t *st = a.stamps + (size_t)blk * 16;
#ifdef TCMI_COPY_PHASES
        for (int k = 0; k < 5; ++k) st[10 + k] = ph[k];
#if TCMI_COPY_PHASES >= 2
        st[2] = ph[5]; st[3] = ph[6]; st[15] = ph[7];
#else
        st[2] = hist[0] | (uint64_t)hist[1] << 32; st[3] = hist[2] | (uint64_t)hist[3] << 32; st[15] = hist[4] | (uint64_t)hist[5] << 32;
#endif
#endif
        st[1] = __builtin_amdgcn_s_memtime(); st[4] = n_slow; st[5] = n_match; st[6] = n_round; st[7] = ntok; st[8] = n_team; st[9] = n_teamed;
    }
}

// the CRC's operators: "append n zero bytes" is linear on the register — a 32 x 32 matrix over GF(2), built zlib's crc32_combine way
// (one zero bit, squared up).  ops[k][j][n]: the register n << 4 j, 2^k zero bytes later; zeros_seg[i]: bit i, CSEG zero bytes later.
struct CrcTables { uint32_t ops[CRC_NOPS][8][16]; uint32_t zeros_seg[32]; };
static const CrcTables &crc_tables()
{
    static const CrcTables T = [] {
        CrcTables t;
        auto times = [](const uint32_t *mat, uint32_t vec) { uint32_t r = 0; for (int i = 0; vec; vec >>= 1, ++i) if (vec & 1u) r ^= mat[i]; return r; };
        uint32_t a[32], b[32];
        a[0] = 0xEDB88320u;                                     // one zero BIT
        for (int i = 1; i < 32; ++i) a[i] = 1u << (i - 1);
        uint32_t *cur = a, *nxt = b;
        for (int bits = 1; bits <= 8 * 2048; bits <<= 1) {      // `cur` appends `bits` zero bits
            for (int k = 0; k < CRC_NOPS; ++k)
                if (bits == 8 << k)
                    for (int j = 0; j < 8; ++j)
                        for (uint32_t n = 0; n < 16; ++n) t.ops[k][j][n] = times(cur, n << (4 * j));
            if (bits == 8 * CSEG) std::memcpy(t.zeros_seg, cur, sizeof t.zeros_seg);
            for (int i = 0; i < 32; ++i) nxt[i] = times(cur, cur[i]);
            std::swap(cur, nxt);
        }
        return t;
    }();
    return T;
}
// ... in device memory, once per device
static const uint32_t *crc_ops_on_device(tcmi_ctx *ctx)
{
    static std::mutex mu;
    static std::vector<std::pair<int, uint32_t *>> per_device;
    std::lock_guard<std::mutex> lk(mu);
    for (auto &e : per_device) if (e.first == ctx->device) return e.second;
    uint32_t *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(CrcTables::ops)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, crc_tables().ops, sizeof(CrcTables::ops), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    per_device.emplace_back(ctx->device, d);
    return d;
}

} // namespace

int tcmi_bgzf_decode_launch(tcmi_ctx *ctx, const tcmi_bgzf_decode_args &g)
{
    const size_t nb_all = g.n_blocks;
    const size_t b_first = std::min(g.first_block, nb_all), nb = std::min(g.count, nb_all - b_first), b_end = b_first + nb;       // this launch's blocks
    if (nb == 0) return TCMI_OK;
    static const char *stamp_path = std::getenv("TCMI_INFLATE_STAMPS");      // diagnostic: phase clocks of both kernels, per block
    uint64_t *d_stamps = nullptr;
    if (stamp_path && nb == nb_all) TCMI_HIP(ctx, hipMalloc((void **)&d_stamps, nb * 16 * 8 * 2));       // (whole-file launches only)
    SymArgs sa;
    sa.stamps = d_stamps;
    sa.file32 = reinterpret_cast<const uint32_t *>(g.d_file);
    sa.blocks = static_cast<const BlockDesc *>(g.d_desc);
    sa.tokens = g.d_tok - g.tok_base; sa.n_tok = g.d_ntok; sa.status = g.d_stat; sa.n_blocks = (int32_t)b_end; sa.first_block = (int32_t)b_first;
    sa.pay_dwords = g.pay_dwords;
    sa.win_dwords = 0;
    static const int gmax_env = std::getenv("TCMI_SYM_GATHER_MAX") ? std::atoi(std::getenv("TCMI_SYM_GATHER_MAX")) : -1;      // (A/B)
    sa.gather_max = gmax_env >= 0 ? (uint32_t)gmax_env : 2048u;
    sa.scratch_div = (uint32_t)std::max(g.scratch_div, 1);
    static const int sbias_env = std::getenv("TCMI_SYM_SHIFT_BIAS") ? std::atoi(std::getenv("TCMI_SYM_SHIFT_BIAS")) : -1;      // (A/B)
    static const int forced = std::getenv("TCMI_SYM_BLOCKS") ? std::atoi(std::getenv("TCMI_SYM_BLOCKS")) : 0;      // (A/B measurements)
    // (measured on one 4 187-block file, kernel alone: 4 blocks per workgroup 372 us, 2: 285 us, 1: 325 us; on the harder file —
    //  4 611 blocks of 10.7 KB — 1 634 / 1 036 / 698 us: with larger payloads more lanes per block pay)
    const size_t pay = (size_t)g.pay_dwords * 4;
    const int per_wg = forced == 1 || forced == 2 || forced == 4 ? forced : pay <= 4096 ? 2 : 1;
    // payloads of more than 16 KB (files that compress less than ~4 : 1) are staged a window of 5 KB at a time, bgzf_symbols<1, true>:
    // 2.5 : 1 (26 KB a block): 2 766 -> 1 875 us per 1M-read file in round 3 (8 KB windows); at 6 : 1 (11 KB, eight workgroups per CU
    // staged whole) windows cost more than they bring (413 -> 580 us): every window is a pass of its own (ring set-up, chain, tokens
    // moved to their places).  Round 4, same file: windows of 16 / 12 / 8 / 6 / 5 / 4 / 3 KB 1 959 / 1 826 / 1 436 / 1 395 / 1 348 /
    // 1 381 / 1 448 us — the smaller the window the more workgroups a CU holds (12 at 5 KB), the more passes a block takes.
    static const int win_env = std::getenv("TCMI_SYM_WINDOW") ? std::atoi(std::getenv("TCMI_SYM_WINDOW")) : -1;      // (A/B: 0 = never, else the window's bytes)
    const size_t win_bytes = per_wg == 1 ? (win_env >= 0 ? (size_t)win_env : pay > 16384 ? 5120u : 0u) : 0u;
    const bool windowed = win_bytes >= 2048 && win_bytes + 24 < pay;
    sa.win_dwords = windowed ? (uint32_t)((win_bytes / 4 + 6 + 1) & ~(size_t)1) : 0u;    // (even: the window loader stores 8 bytes a lane)
    // a window's chunks are short (5 KB over 64 lanes: 640 bits): with stretches of chunk / 4 .. chunk / 2 bits a lane decodes a third of
    // a chunk into its neighbour's before it can meet it; half as long there (2.5 : 1: 1 258 -> 1 200 us; the bench file, whole payloads: 142 -> 163)
    sa.shift_bias = sbias_env >= 0 ? (uint32_t)sbias_env : windowed ? 1u : 0u;
    const size_t dyn = windowed ? (size_t)sa.win_dwords * 4 : (size_t)g.pay_dwords * 4 * per_wg;
    static const bool attr_once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bgzf_symbols<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - sizeof(SymLds<4>)));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bgzf_symbols<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - sizeof(SymLds<2>)));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bgzf_symbols<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - sizeof(SymLds<1>)));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bgzf_symbols<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - sizeof(SymLds<1>)));
        return true;
    }();
    (void)attr_once;
    if (stamp_path) {                           // (diagnostic) how many workgroups of each kernel a compute unit really holds
        int occ_s = 0, occ_c = 0;
        if (per_wg == 2) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_s, reinterpret_cast<const void *>(bgzf_symbols<2, false>), 128, dyn);
        else if (windowed) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_s, reinterpret_cast<const void *>(bgzf_symbols<1, true>), 64, dyn);
        else if (per_wg == 1) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_s, reinterpret_cast<const void *>(bgzf_symbols<1, false>), 64, dyn);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_c, reinterpret_cast<const void *>(bgzf_copy<false, false>), 64 * CW, 0);
        occ_c *= CW;
        std::fprintf(stderr, "[tcmi inflate] %zu blocks, payload %zu B + slack; bgzf_symbols<%d%s>: %zu B of LDS per workgroup, %d workgroups per CU; bgzf_copy: %d per CU\n",
                     nb, pay, per_wg, windowed ? ", windowed" : "", dyn + (per_wg == 2 ? sizeof(SymLds<2>) : per_wg == 1 ? sizeof(SymLds<1>) : sizeof(SymLds<4>)), occ_s, occ_c);
    }
    if (ctx->ev_before_sym) { TCMI_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_before_sym, 0)); ctx->ev_before_sym = nullptr; }      // (sub-ranges of a split step: skewed starts)
    tcmi_prof_begin(ctx, TCMI_K_INFLATE);
    if (per_wg == 4) hipLaunchKernelGGL((bgzf_symbols<4, false>), dim3((unsigned)((nb + 3) / 4)), dim3(256), dyn, ctx->stream, sa);
    else if (per_wg == 2) hipLaunchKernelGGL((bgzf_symbols<2, false>), dim3((unsigned)((nb + 1) / 2)), dim3(128), dyn, ctx->stream, sa);
    else if (windowed) hipLaunchKernelGGL((bgzf_symbols<1, true>), dim3((unsigned)nb), dim3(64), dyn, ctx->stream, sa);
    else hipLaunchKernelGGL((bgzf_symbols<1, false>), dim3((unsigned)nb), dim3(64), dyn, ctx->stream, sa);
    tcmi_prof_end(ctx, TCMI_K_INFLATE);
    TCMI_HIP(ctx, hipGetLastError());
    if (ctx->after_sym) {
        if (ctx->ev_after_sym) (void)hipEventRecord(ctx->ev_after_sym, ctx->stream);
        auto fn = std::move(ctx->after_sym);
        ctx->after_sym = nullptr;
        fn();
    }
    CopyArgs ca;
    ca.file = g.d_file; ca.blocks = sa.blocks; ca.tokens = sa.tokens; ca.n_tok = g.d_ntok; ca.out = g.d_out; ca.rec_slot = g.d_slot;
    ca.n_rec = g.d_nrec; ca.overshoot = g.d_over; ca.first_rec = g.d_first; ca.status = g.d_stat; ca.n_blocks = (int32_t)b_end; ca.first_block = (int32_t)b_first; ca.n_ref = g.n_ref;
    ca.stamps = d_stamps ? d_stamps + nb * 16 : nullptr;
    static const int team_env = std::getenv("TCMI_TEAM_BYTES") ? std::atoi(std::getenv("TCMI_TEAM_BYTES")) : -1;      // (A/B measurements)
    ca.team_bytes = team_env >= 0 ? (uint32_t)team_env : (uint32_t)TEAM_BATCH_BYTES;
    ca.crc = g.verify_crc ? 1u : 0u;
    ca.crc_ops = nullptr;
    if (ca.crc) {
        ca.crc_ops = crc_ops_on_device(ctx);
        if (!ca.crc_ops) return tcmi_fail(ctx, TCMI_E_NOMEM, "device memory for the CRC operators");
        std::memcpy(ca.zeros_seg, crc_tables().zeros_seg, sizeof ca.zeros_seg);
    }
    const unsigned copy_grid = (unsigned)((nb + CW - 1) / CW);
    tcmi_prof_begin(ctx, TCMI_K_INFLATE_COPY);
    // (bgzf_symbols<1, *> — payloads beyond 4 KB — puts two literals into one token: only the TEAMS variants read those)
    static_assert(TCMI_COPY_TEAMS, "bgzf_symbols<1, *> writes tokens of two literals: the copy kernel's TEAMS variants read them");
    if (g.short_tokens >= 2) hipLaunchKernelGGL((bgzf_copy<true, true>), dim3(copy_grid), dim3(64 * CW), 0, ctx->stream, ca);
    else if (g.short_tokens || per_wg == 1) hipLaunchKernelGGL((bgzf_copy<true, false>), dim3(copy_grid), dim3(64 * CW), 0, ctx->stream, ca);
    else hipLaunchKernelGGL((bgzf_copy<false, false>), dim3(copy_grid), dim3(64 * CW), 0, ctx->stream, ca);
    tcmi_prof_end(ctx, TCMI_K_INFLATE_COPY);
    TCMI_HIP(ctx, hipGetLastError());
    if (d_stamps) {
        std::vector<uint64_t> h(nb * 32);
        TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        TCMI_HIP(ctx, hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(d_stamps);
        if (FILE *fp = std::fopen(stamp_path, "wb")) { std::fwrite(h.data(), 8, h.size(), fp); std::fclose(fp); }
    }
    return TCMI_OK;
}
