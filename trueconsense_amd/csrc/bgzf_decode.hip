// bgzf_decode.hip — DEVICE: the BGZF blocks of a BAM file -> the inflated BAM byte stream + the record starts of every block
// (pysam / htslib's role for indexing.py:19,96-100; SAM spec §4.1 BGZF, RFC 1951 DEFLATE), in two kernels:
//
//   bgzf_symbols   Huffman symbols -> tokens.  One wavefront per BGZF block, and ALL 64 LANES DECODE THAT ONE BLOCK, speculatively
//                  in parallel.
//   bgzf_copy      tokens -> bytes: the LZ77 copies through an LDS ring of the recent output, the chain of BAM records, the flush.
//
// Why two kernels.  A deflate stream is serial twice over: the position of symbol k + 1 is known only when symbol k is decoded,
// and a match may copy what the previous match produced.  Round 2's one-kernel decoder (bam_device.hip: bgzf_inflate) walks both
// chains in one wavefront, one symbol at a time: ~60 wave-instructions per symbol, all of them issued for a single useful lane,
// and the kernel is bound by instruction issue (one instruction per compute unit and cycle).  Here the first chain is cut into
// 64 pieces: lane c starts decoding at bit s_c = start + c * chunk — in the middle of nowhere, except for lane 0 — and notes, in
// a window of WBITS bits behind s_c, every bit position on which it starts a symbol.  Huffman streams resynchronise: after a
// few dozen bits a decoder that started on a wrong bit starts a symbol on a right one, and from there on it IS the serial
// decoder.  A lane stops when a symbol of its own starts on a position that a lane in front of it has noted: from there the
// two would decode the same (pass A).  Starting from lane 0 the chain of these meeting points says which lane holds the true
// symbols of which bit range, and how many they are; an exclusive sum gives every such lane its place in the block's token
// array, and it decodes its range once more, for real (pass B).  Nothing in this depends on luck or timing: a lane that never
// meets anyone simply goes on to the block's end, and lane 0 alone is the serial decoder.  On the bench files
// (tools/spec_inflate_proto.py, the same scheme in Python): 1 600 symbols in 96 + 87 lock-step rounds, 6 900 in 298 + 291.
//
// Tokens (32 bits): literal 1<<31 | byte; match len (9 bits) | (dist - 1) << 9; raw 1<<30 | len << 17 | offset of the bytes from
// the block's payload start (a stored deflate block, in pieces of <= 8 191 bytes).  bgzf_copy takes 64 tokens at a time: an
// inclusive scan of the lengths gives every token its output position, all literals of a round go to the ring at once, the
// matches one after the other (each copied by all 64 lanes).
//
// Bit / byte work, bound by instruction issue and LDS latency, not by HBM and not a contraction: no MFMA.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bgzf_device.h"

namespace {

constexpr int WBITS = 256;                          // bits behind its start in which a lane notes the symbols it starts
constexpr int MARK_W = WBITS / 32 + 1;              // (+1: odd stride, lanes c and c + 4 would share banks otherwise)
constexpr int CWIN = 8192, CWMASK = CWIN - 1;       // bgzf_copy's ring of recent output
constexpr int CSEG = 2048;
// a round of bgzf_copy writes the literals of up to CSEG + 258 bytes ahead of the match it copies: what a match may still read
// from the ring ends that much earlier; a source further back has been flushed (CWIN >= 2 CSEG + 522)
constexpr int CNEAR = CWIN - CSEG - 264;
static_assert(CWIN >= 2 * CSEG + 528 && (CWIN & (CWIN - 1)) == 0 && CWIN % CSEG == 0, "a far match must find its source flushed");
constexpr uint32_t TOK_LIT = 1u << 31, TOK_RAW = 1u << 30;
constexpr uint32_t RAW_PIECE = 8191;
constexpr int CL_SLAB = 496;                        // bit positions of the code-length stream looked up at a time

struct SymLds {
    tab_t ll[1 << LL_ROOT];                         // (first: the code-length stream's table of all positions, CL_SLAB entries)
    tab_t dt[1 << D_ROOT];                          // (first: the code-length code's root table)
    tab_t long_ll[288], long_d[32];                 // entries of the codes longer than the root bits, in canonical order
    union {
        uint32_t marks[64][MARK_W];                 // pass A
        struct {                                    // header and table building
            uint8_t lens[320];
            uint8_t cll[20];
            uint16_t sym_ll[288], sym_d[32], sym_cl[20];
            uint16_t cnt_ll[16], cnt_d[16], cnt_cl[16];
            uint32_t rs[6];
        } h;
    };
};
static_assert((CL_SLAB + 16) * 4 <= sizeof(tab_t) * (1 << LL_ROOT), "the code-length position table borrows the literal/length table's LDS");

struct SymArgs {
    const uint32_t *__restrict__ file32;
    const BlockDesc *blocks;
    uint32_t *tokens;           // block b's tokens at tokens + blocks[b].tok
    uint32_t *n_tok;            // [n_blocks]
    uint32_t *status;           // [n_blocks]
    int32_t n_blocks;
    uint32_t pay_dwords;        // dwords of dynamic LDS behind SymLds: the largest block's payload + slack
    uint64_t *stamps;           // diagnostic (TCMI_INFLATE_STAMPS): 16 words per block, s_memtime at the phase boundaries; or null
};
#define TCMI_STAMP(buf_, blk_, k_) do { if (buf_) { if (threadIdx.x == 0) (buf_)[(size_t)(blk_) * 16 + (k_)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define TCMI_STAMP_ADD(buf_, blk_, k_, v_) do { if (buf_) { if (threadIdx.x == 0) (buf_)[(size_t)(blk_) * 16 + (k_)] += (v_); } } while (0)

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

// The codes longer than the root bits: per length first code | count << 16 (wave-uniform, in scalar registers), and one
// ready-made table entry per such code in canonical order.  A lane looks its code up by comparing the bit-reversed stream bits
// with each length's code range — no walk through memory.
template <int ROOT>
struct LongCodes { uint32_t fc[15 - ROOT]; };

template <int ROOT>
__device__ __forceinline__ void build_long(const uint16_t *cnt, const uint16_t *sym, const uint32_t *rs, int kind, tab_t *out, LongCodes<ROOT> &lc)
{
    uint32_t first = uni(rs[0]);
    const uint32_t at = uni(rs[1]);
    uint32_t n = 0;
#pragma unroll
    for (int len = ROOT + 1; len <= 15; ++len) {
        const uint32_t c = uni(cnt[len]);
        lc.fc[len - ROOT - 1] = (first & 0xFFFFu) | (c << 16);
        first = (first + c) << 1;
        n += c;
    }
    for (uint32_t i = threadIdx.x; i < n; i += 64) {
        uint32_t base = 0;
        int mylen = 15;
#pragma unroll
        for (int len = ROOT + 1; len <= 15; ++len) {
            const uint32_t c = lc.fc[len - ROOT - 1] >> 16;
            if (i >= base && i < base + c) mylen = len;
            base += c;
        }
        out[i] = make_entry(kind, (int)sym[at + i], mylen);
    }
}

template <int ROOT>
__device__ __forceinline__ uint32_t long_lookup(const LongCodes<ROOT> &lc, const tab_t *tab, uint32_t bits)
{
    const uint32_t r = __builtin_bitreverse32(bits);
    uint32_t idx = 0xFFFFFFFFu, base = 0;
#pragma unroll
    for (int len = ROOT + 1; len <= 15; ++len) {
        const uint32_t fc = lc.fc[len - ROOT - 1];
        const uint32_t c = fc >> 16;
        if (c) {                                            // (wave-uniform)
            const uint32_t d = (r >> (32 - len)) - (fc & 0xFFFFu);
            if (d < c) idx = base + d;
            base += c;
        }
    }
    return idx != 0xFFFFFFFFu ? tab[idx] : 0u;
}

enum { SY_LIT = 0, SY_MATCH = 1, SY_EOB = 2, SY_BAD = 3 };

__global__ __launch_bounds__(64) void bgzf_symbols(SymArgs a)
{
    __shared__ SymLds L;
    extern __shared__ uint32_t pay[];               // the block's compressed payload, from the dword that holds its first byte on
    const int lane = threadIdx.x;
    const int blk = blockIdx.x;
    if (blk >= a.n_blocks) return;
    const BlockDesc d = a.blocks[blk];
    uint32_t *const toks = a.tokens + d.tok;
    const uint32_t cap = d.tok_cap;
    // ---- the payload into LDS (+ 6 dwords: a lane looks up to 48 bits past the end; the file buffer has the slack) ----
    const uint32_t base_bit = (uint32_t)(d.cin & 3u) * 8u;
    const uint32_t end = base_bit + d.clen * 8u;                    // first bit behind the payload
    if (a.stamps && lane < 16) a.stamps[(size_t)blk * 16 + lane] = 0;
    TCMI_STAMP(a.stamps, blk, 0);
    {
        const uint32_t *src = a.file32 + (d.cin >> 2);
        const uint32_t n = min(a.pay_dwords, (end + 31u) / 32u + 6u);
        for (uint32_t i = (uint32_t)lane; i < n; i += 64) pay[i] = src[i];
    }
    __syncthreads();
    TCMI_STAMP(a.stamps, blk, 1);
    uint32_t pos = base_bit;            // wave-uniform
    uint32_t ntok = 0;
    uint32_t err = ST_OK;
    bool last = false;
    while (!last && err == ST_OK) {
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
            continue;
        }
        if (type == 3) { err = ST_BAD_STREAM; break; }
        // ---- code lengths -------------------------------------------------------------------------------------------------------
        int nlen = 288, ndist = 32;
        __syncthreads();                            // (the header arrays share their LDS with pass A's notes)
        if (type == 1) {
            for (int i = lane; i < 320; i += 64) L.h.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
        } else {
            if (pos + 14u > end) { err = ST_BAD_STREAM; break; }
            const uint32_t hh = uni(peek32(pay, pos));
            nlen = (int)(hh & 31u) + 257;
            ndist = (int)((hh >> 5) & 31u) + 1;
            const int ncode = (int)((hh >> 10) & 15u) + 4;
            pos += 14;
            if (nlen > 286 || ndist > 30) { err = ST_BAD_STREAM; break; }
            if (lane < 19) L.h.cll[lane] = 0;
            __syncthreads();
            // (19 x 3 bits: three looks of up to 8 lengths each, lane k takes the k-th)
            for (int i0 = 0; i0 < ncode; i0 += 8) {
                const uint32_t v = uni(peek32(pay, pos + (uint32_t)i0 * 3u));
                const int k = i0 + lane;
                if (lane < 8 && k < ncode) L.h.cll[CL_ORDER[k]] = (uint8_t)((v >> (3 * lane)) & 7u);
            }
            pos += (uint32_t)ncode * 3u;
            if (uni(build_table<1, CL_ROOT>(L.h.cll, 19, L.h.cnt_cl, L.h.sym_cl, L.dt, K_CODELEN, L.h.rs + 4) ? 1u : 0u) == 0u) { err = ST_BAD_STREAM; break; }
            for (int i = lane; i < 320; i += 64) L.h.lens[i] = 0;
            // The code-length symbols (0 .. 15: a length; 16: the previous length 3 - 6 times; 17 / 18: 3 - 10 / 11 - 138 zeros) are
            // a serial chain too, but a short one over few bits.  Every bit position of a slab is looked up by some lane (what
            // symbol would start here, how many lengths would it give, how many bits would it take: step | rep << 4 | val << 12);
            // the chain is then followed through that table with one scalar look-up per symbol that only notes the entry
            // (lane j keeps the j-th of 64), and what the symbols mean is worked out for 64 of them at a time: a sum scan of
            // the repeat counts places them, a maximum scan finds for every "16" the last symbol in front that names a length.
            uint32_t *const T = L.ll;
            uint32_t got = 0, prev = 0;
            const uint32_t total = (uint32_t)(nlen + ndist);
            bool first = true;
            while (got < total && err == ST_OK) {
                __syncthreads();
#pragma unroll 2
                for (int o = lane; o < CL_SLAB + 16; o += 64) {
                    const uint32_t v = peek32(pay, pos + (uint32_t)o);
                    const uint32_t e = L.dt[v & ((1u << CL_ROOT) - 1u)];
                    const uint32_t nb = e & 15u, sym = e >> 16;
                    const uint32_t x = v >> nb;
                    const uint32_t eb = sym < 16u ? 0u : sym == 16u ? 2u : sym == 17u ? 3u : 7u;
                    const uint32_t rep = sym < 16u ? 1u : sym == 18u ? 11u + (x & 127u) : 3u + (x & (sym == 16u ? 3u : 7u));
                    const uint32_t val = sym <= 16u ? sym : 0u;
                    T[o] = nb && o < CL_SLAB ? (nb + eb) | (rep << 4) | (val << 12) : 0u;      // (0 behind the slab: the chain stops there)
                }
                __syncthreads();
                uint32_t o = 0;
                while (got < total && err == ST_OK) {
                    uint32_t mine = 0, j = 0, g = got, e;
                    do {
                        e = uni(T[o]);
                        if (e == 0) break;
                        asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(mine) : "s"(e), "s"(j) : "m0");
                        g += (e >> 4) & 255u;
                        o += e & 15u;
                        ++j;
                    } while (g < total && j < 64u);
                    // lane j: its symbol's place and value
                    const uint32_t rep = (mine >> 4) & 255u, v = mine >> 12;
                    const uint32_t incl = wave_scan_add(rep);
                    const uint32_t at = got + incl - rep;
                    const uint32_t named = wave_scan_max(mine != 0 && v != 16u ? (uint32_t)lane + 1u : 0u);     // 1 + the lane whose value a "16" here repeats
                    const uint32_t theirs = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((named - 1u) << 2), (int)v);
                    const uint32_t val = named ? theirs : prev;
                    if (__ballot(mine != 0 && (at + rep > total || (first && lane == 0 && v == 16u)))) { err = ST_BAD_STREAM; break; }
                    if (mine != 0 && val != 0) {
#pragma unroll
                        for (uint32_t i = 0; i < 6; ++i)            // (zeros are not stored, so rep <= 6)
                            if (i < rep) L.h.lens[at + i] = (uint8_t)val;
                    }
                    if (j) prev = (uint32_t)__builtin_amdgcn_readlane((int)val, (int)(j - 1u));
                    got = g;
                    first = false;
                    if (e == 0) { if (o < (uint32_t)CL_SLAB) err = ST_BAD_STREAM; break; }      // no such code / the slab's end
                }
                pos += o;
                if (pos > end) err = ST_BAD_STREAM;
            }
            if (err != ST_OK) break;
            __syncthreads();
            if (uni(L.h.lens[256]) == 0) { err = ST_BAD_STREAM; break; }    // no end-of-block code
        }
        TCMI_STAMP(a.stamps, blk, 2);
        // ---- tables: the root tables as in bgzf_inflate, the longer codes as ready-made entries ------------------------------------
        LongCodes<LL_ROOT> lcl;
        LongCodes<D_ROOT> lcd;
        if (uni(build_table<5, LL_ROOT>(L.h.lens, nlen, L.h.cnt_ll, L.h.sym_ll, L.ll, K_LITLEN, L.h.rs) ? 1u : 0u) == 0u) { err = ST_BAD_STREAM; break; }
        if (uni(build_table<1, D_ROOT>(L.h.lens + nlen, ndist, L.h.cnt_d, L.h.sym_d, L.dt, K_DIST, L.h.rs + 2) ? 1u : 0u) == 0u) { err = ST_BAD_STREAM; break; }
        build_long<LL_ROOT>(L.h.cnt_ll, L.h.sym_ll, L.h.rs, K_LITLEN, L.long_ll, lcl);
        build_long<D_ROOT>(L.h.cnt_d, L.h.sym_d, L.h.rs + 2, K_DIST, L.long_d, lcd);
        if (pos >= end) { err = ST_BAD_STREAM; break; }
        __syncthreads();
        TCMI_STAMP(a.stamps, blk, 3);

        // One literal / length / end-of-block code at bit p; a length is followed by its distance.  VALUES: the token is made
        // (pass B); otherwise only the bits are counted (pass A).
        auto symbol = [&](uint32_t &p, uint32_t &tok, const bool VALUES) __attribute__((always_inline)) -> int {
            uint32_t lo, hi;
            peek64(pay, p, lo, hi);
            uint32_t e = L.ll[lo & ((1u << LL_ROOT) - 1u)];
            if (__builtin_expect(__ballot((e & 15u) == 0) != 0, 0)) {
                const uint32_t e2 = long_lookup<LL_ROOT>(lcl, L.long_ll, lo);
                if ((e & 15u) == 0) e = e2;
            }
            if ((e & 15u) == 0) return SY_BAD;
            if (e & E_LIT) { p += e & 15u; if (VALUES) tok = TOK_LIT | ((e >> 16) & 0xFFu); return SY_LIT; }
            if (e & E_EOB) { p += e & 15u; return SY_EOB; }
            const uint32_t k = (e >> 11) & 31u;                 // code + extra bits of the length
            const uint32_t d32 = __builtin_amdgcn_alignbit(hi, lo, k);
            uint32_t f = L.dt[d32 & ((1u << D_ROOT) - 1u)];
            if (__builtin_expect(__ballot((f & 15u) == 0) != 0, 0)) {
                const uint32_t f2 = long_lookup<D_ROOT>(lcd, L.long_d, d32);
                if ((f & 15u) == 0) f = f2;
            }
            if ((f & 15u) == 0) return SY_BAD;
            const uint32_t nd = f & 15u, eb2 = (f >> 4) & 15u;
            p += k + nd + eb2;
            if (VALUES) {
                const uint32_t nb = e & 15u, eb = (e >> 16) & 15u;
                const uint32_t len = ((e >> 20) & 0x1FFu) + ((lo >> nb) & ((1u << eb) - 1u));
                const uint32_t dist = (f >> 16) + ((d32 >> nd) & ((1u << eb2) - 1u));
                tok = len | ((dist - 1u) << 9);
            }
            return SY_MATCH;
        };

        // ---- pass A: every lane decodes from its own start until it meets a lane in front ----------------------------------------
        const uint32_t start = pos;
        const uint32_t chunk = (end - start + 63u) / 64u;           // >= 1
        const uint32_t chunk_m = 0xFFFFFFFFu / chunk;               // (x * chunk_m) >> 32 = x / chunk or one less, for x < 2^20
        const uint32_t s_c = start + (uint32_t)lane * chunk;
        enum { RUN = 0, MERGED = 1, EOB = 2, DEAD = 3 };
        uint32_t state = s_c < end ? RUN : DEAD;
        uint32_t tgt = (uint32_t)lane + 1u, total = 0;
        uint32_t p = min(s_c, end);
#pragma unroll
        for (int k = 0; k < MARK_W; ++k) L.marks[lane][k] = 0;
        __syncthreads();
        while (__ballot(state == RUN)) {
            if (state == RUN) {
                const uint32_t rel = p - s_c;
                if (rel < (uint32_t)WBITS) atomicOr(&L.marks[lane][rel >> 5], 1u << (rel & 31u));
            }
            __syncthreads();                        // (one wavefront: orders the notes before the looks)
            if (state == RUN) {
                // the nearest lane in front whose window still reaches p (the quotient may be one short: then r >= WBITS below and
                // this round looks at nobody)
                const uint32_t x = p - start;
                const uint32_t q = x >= (uint32_t)WBITS ? __umulhi(x - (uint32_t)WBITS, chunk_m) + 1u : 0u;
                tgt = max(tgt, q);
                const uint32_t r = p - (start + tgt * chunk);
                if (tgt < 64u && r < (uint32_t)WBITS && ((L.marks[tgt][r >> 5] >> (r & 31u)) & 1u)) state = MERGED;
            }
            if (state == RUN) {
                uint32_t tok;
                const int k = p >= end ? SY_BAD : symbol(p, tok, false);
                if (k == SY_BAD || p > end) state = DEAD;
                else {
                    ++total;
                    if (k == SY_EOB) state = EOB;
                }
            }
            TCMI_STAMP_ADD(a.stamps, blk, 8, 1);
        }
        TCMI_STAMP(a.stamps, blk, 4);
        // ---- the chain of lanes that hold the true symbols: lane 0 from `start`, then whoever it met, ... ---------------------
        uint32_t myP = start;
        bool alive = lane == 0;
        uint32_t eob_pos = 0;
        {
            uint32_t c = 0;
            for (;;) {
                const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)state, (int)c);
                if (st == MERGED) {
                    const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)p, (int)c);
                    const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tgt, (int)c);
                    if ((uint32_t)lane == t) { alive = true; myP = m; }
                    c = t;
                } else {
                    if (st == EOB) eob_pos = (uint32_t)__builtin_amdgcn_readlane((int)p, (int)c);
                    else err = ST_BAD_STREAM;
                    break;
                }
            }
        }
        if (err != ST_OK) break;
        // symbols a lane decoded in front of its true start do not count (all of them are noted: the start lies in its window)
        uint32_t cnt = 0;
        if (alive) {
            const uint32_t lim = myP - s_c;         // < WBITS for every lane but 0, where it is 0
            uint32_t before = 0;
#pragma unroll
            for (int k = 0; k < WBITS / 32; ++k) {
                const uint32_t w = L.marks[lane][k];
                const uint32_t lo = (uint32_t)k * 32u;
                const uint32_t m = lim >= lo + 32u ? 0xFFFFFFFFu : lim > lo ? (1u << (lim - lo)) - 1u : 0u;
                before += (uint32_t)__popc(w & m);
            }
            cnt = total - before;
            if (state == EOB) --cnt;                // (the end-of-block code is a symbol, not a token)
        }
        // exclusive sum over the lanes -> every lane's place in the token array
        const uint32_t incl = wave_scan_add(cnt);
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (ntok + all > cap) { err = ST_BAD_STREAM; break; }
        TCMI_STAMP(a.stamps, blk, 5);
        // ---- pass B: the true ranges once more, tokens out --------------------------------------------------------------------------
        {
            uint32_t *dst = toks + ntok + (incl - cnt);
            uint32_t q = myP;
            uint32_t i = 0;
            while (__ballot(alive && i < cnt)) {
                if (alive && i < cnt) {
                    uint32_t tok = 0;
                    (void)symbol(q, tok, true);
                    dst[i] = tok;
                    ++i;
                }
                TCMI_STAMP_ADD(a.stamps, blk, 9, 1);
            }
        }
        TCMI_STAMP(a.stamps, blk, 6);
        ntok += all;
        pos = eob_pos;
    }
    if (lane == 0) {
        a.n_tok[blk] = ntok;
        a.status[blk] = err;
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
    int32_t *overshoot;
    uint32_t *status;           // in: bgzf_symbols' verdict; out: the block's
    int32_t n_blocks;
    uint64_t *stamps;           // diagnostic, as SymArgs::stamps
};

__global__ __launch_bounds__(64) void bgzf_copy(CopyArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_win[CWIN];
    const int lane = threadIdx.x;
    const int blk = blockIdx.x;
    if (blk >= a.n_blocks) return;
    const BlockDesc d = a.blocks[blk];
    const uint32_t ulen = d.ulen;
    uint32_t err = uni(a.status[blk]);
    const uint32_t ntok = err == ST_OK ? uni(a.n_tok[blk]) : 0u;
    const uint32_t *toks = a.tokens + d.tok;
    uint8_t *const out = a.out + d.uout;
    const uint8_t *const payload = a.file + d.cin;
    uint32_t *slots = a.rec_slot + (size_t)blk * MAX_REC_PER_BLOCK;

    uint32_t op = 0, flushed = 0;
    uint32_t next_rec = d.entry >= 0 ? (uint32_t)d.entry : 0xFFFFFFF0u;
    uint32_t n_rec = 0, bad_rec = 0, rec_buf = 0;
    uint32_t next_evt = 0;
    uint32_t bad = 0;

    // list the record starts whose block_size field is complete, flush the segments that are complete
    auto housekeeping = [&]() __attribute__((always_inline)) {
        while (next_rec + 4 <= op) {
            const uint32_t at = next_rec & CWMASK;
            const uint32_t w = reinterpret_cast<const uint32_t *>(s_win)[((at >> 2) + (uint32_t)lane) & (CWIN / 4 - 1)];
            const uint64_t two = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)w, 1) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)w, 0);
            const uint32_t ubs = (uint32_t)(two >> ((at & 3u) * 8u));
            if (__builtin_expect(ubs - 32u > (1u << 28) - 32u, 0)) { bad_rec = 1; break; }
            rec_buf = (uint32_t)lane == (n_rec & 63u) ? next_rec : rec_buf;
            if ((n_rec & 63u) == 63u) slots[(n_rec & ~63u) + (uint32_t)lane] = rec_buf;
            ++n_rec;
            next_rec += 4u + ubs;
        }
        if (bad_rec) { err = ST_BAD_RECORD; next_rec = 0xFFFFFFF0u; }
        while (op - flushed >= CSEG) {
            const uint4 *src = reinterpret_cast<const uint4 *>(s_win + (flushed & CWMASK));
            uint4 *dst = reinterpret_cast<uint4 *>(out + flushed);
#pragma unroll
            for (int k = 0; k < CSEG / 16 / 64; ++k) dst[k * 64 + lane] = src[k * 64 + lane];
            flushed += CSEG;
        }
        next_evt = flushed + (uint32_t)CSEG;
    };
    // a match: all lanes; with dist < len the pattern of the last `dist` bytes repeats
    auto copy_match = [&](uint32_t at, uint32_t len, uint32_t dist) __attribute__((always_inline)) {
        if (dist > at) { bad = 1; return; }                      // before the block's first byte
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
    uint64_t tk0 = 0, t_prep = 0, t_match = 0, t_house = 0, n_match = 0, n_round = 0;
    housekeeping();
    for (uint32_t base = 0; base < ntok && err == ST_OK && !bad; base += 64) {
        if (a.stamps) tk0 = __builtin_amdgcn_s_memtime();
        const uint32_t t = base + (uint32_t)lane < ntok ? toks[base + (uint32_t)lane] : 0u;
        const bool is_lit = (t >> 31) != 0;
        const bool is_raw = !is_lit && (t & TOK_RAW);
        if (__ballot(is_raw)) {
            // ---- a batch with stored bytes in it: token by token (rare: incompressible data, flush markers) --------------------
            const uint32_t nb = min(64u, ntok - base);
            for (uint32_t j = 0; j < nb && err == ST_OK && !bad; ++j) {
                const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)t, (int)j);
                if (tj >> 31) {
                    if (op + 1 > ulen) { err = ST_BAD_LENGTH; break; }
                    s_win[op & CWMASK] = (uint8_t)tj;
                    ++op;
                } else if (tj & TOK_RAW) {
                    uint32_t len = (tj >> 17) & 0x1FFFu;
                    const uint8_t *src = payload + (tj & 0x1FFFFu);
                    if (op + len > ulen) { err = ST_BAD_LENGTH; break; }
                    while (len) {
                        const uint32_t n = min(len, (uint32_t)CSEG - (op & (CSEG - 1)));
#pragma clang loop vectorize(disable) unroll(disable)
                        for (uint32_t i = lane; i < n; i += 64) s_win[(op + i) & CWMASK] = src[i];
                        op += n; src += n; len -= n;
                        if (op >= next_evt) { housekeeping(); if (err != ST_OK) break; }
                    }
                } else {
                    const uint32_t len = tj & 511u, dist = ((tj >> 9) & 0x7FFFu) + 1u;
                    if (op + len > ulen) { err = ST_BAD_LENGTH; break; }
                    copy_match(op, len, dist);
                    op += len;
                }
                if (op >= next_evt) housekeeping();
            }
            continue;
        }
        const uint32_t mylen = is_lit ? 1u : (t & 511u);
        const uint32_t dist = ((t >> 9) & 0x7FFFu) + 1u;
        const uint32_t incl = wave_scan_add(mylen);
        const uint32_t dst = op + incl - mylen;                 // where this lane's token starts
        const uint32_t batch_end = op + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (batch_end > ulen) { err = ST_BAD_LENGTH; break; }
        uint32_t t_cur = 0;
        if (a.stamps) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_prep += now - tk0; tk0 = now; }
        while (t_cur < 64u) {
            // the tokens [t_cur, t_stop) start in front of the next housekeeping stop: their literals at once, their matches in order
            const unsigned long long from = ~0ull << t_cur;
            const unsigned long long ge = __ballot(dst >= next_evt) & from;
            const uint32_t t_stop = ge ? (uint32_t)__builtin_ctzll(ge) : 64u;
            const unsigned long long rng = t_stop < 64u ? from & ~(~0ull << t_stop) : from;
            const bool mine = (rng >> lane) & 1ull;
            if (mine && is_lit) s_win[dst & CWMASK] = (uint8_t)t;
            unsigned long long mm = __ballot(mine && !is_lit && mylen != 0);
            while (mm) {
                const int j = __builtin_ctzll(mm);
                mm &= mm - 1;
                const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)dst, j);
                const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)mylen, j);
                const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)dist, j);
                copy_match(at, len, dj);
                ++n_match;
            }
            op = t_stop < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)dst, (int)t_stop) : batch_end;
            t_cur = t_stop;
            ++n_round;
            if (a.stamps) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_match += now - tk0; tk0 = now; }
            if (op >= next_evt) { housekeeping(); if (err != ST_OK) break; }
            if (a.stamps) { const uint64_t now = __builtin_amdgcn_s_memtime(); t_house += now - tk0; tk0 = now; }
            if (bad) break;
        }
    }
    if (bad && err == ST_OK) err = ST_BAD_STREAM;
    if (err == ST_OK && op != ulen) err = ST_BAD_LENGTH;
    if (err == ST_OK) {
        housekeeping();
        const uint32_t rest = op - flushed;
        for (uint32_t i = lane; i < rest; i += 64) out[flushed + i] = s_win[(flushed + i) & CWMASK];
    }
    if ((uint32_t)lane < (n_rec & 63u)) slots[(n_rec & ~63u) + (uint32_t)lane] = rec_buf;
    if (lane == 0) {
        a.status[blk] = err;
        a.n_rec[blk] = n_rec;
        a.overshoot[blk] = d.entry >= 0 && next_rec < 0xFFFFFFF0u ? (int32_t)(next_rec - ulen) : 0;
    }
    if (a.stamps && lane == 0) {
        uint64_t *st = a.stamps + (size_t)blk * 16;
        st[1] = __builtin_amdgcn_s_memtime(); st[2] = t_prep; st[3] = t_match; st[4] = t_house; st[5] = n_match; st[6] = n_round; st[7] = ntok;
    }
}

} // namespace

int tcmi_bgzf_decode_launch(tcmi_ctx *ctx, const tcmi_bgzf_decode_args &g)
{
    const size_t nb = g.n_blocks;
    static const char *stamp_path = std::getenv("TCMI_INFLATE_STAMPS");      // diagnostic: phase clocks of both kernels, per block
    uint64_t *d_stamps = nullptr;
    if (stamp_path) TCMI_HIP(ctx, hipMalloc((void **)&d_stamps, nb * 16 * 8 * 2));
    SymArgs sa;
    sa.stamps = d_stamps;
    sa.file32 = reinterpret_cast<const uint32_t *>(g.d_file);
    sa.blocks = static_cast<const BlockDesc *>(g.d_desc);
    sa.tokens = g.d_tok; sa.n_tok = g.d_ntok; sa.status = g.d_stat; sa.n_blocks = (int32_t)nb;
    sa.pay_dwords = g.pay_dwords;
    const size_t dyn = (size_t)g.pay_dwords * 4;
    static std::atomic<size_t> dyn_allowed{48 * 1024};
    if (dyn > dyn_allowed.load()) {
        TCMI_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(bgzf_symbols), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(80 * 1024)));
        dyn_allowed.store(80 * 1024);
    }
    tcmi_prof_begin(ctx, TCMI_K_INFLATE);
    hipLaunchKernelGGL(bgzf_symbols, dim3((unsigned)nb), dim3(64), dyn, ctx->stream, sa);
    tcmi_prof_end(ctx, TCMI_K_INFLATE);
    TCMI_HIP(ctx, hipGetLastError());
    CopyArgs ca;
    ca.file = g.d_file; ca.blocks = sa.blocks; ca.tokens = g.d_tok; ca.n_tok = g.d_ntok; ca.out = g.d_out; ca.rec_slot = g.d_slot;
    ca.n_rec = g.d_nrec; ca.overshoot = g.d_over; ca.status = g.d_stat; ca.n_blocks = (int32_t)nb;
    ca.stamps = d_stamps ? d_stamps + nb * 16 : nullptr;
    tcmi_prof_begin(ctx, TCMI_K_INFLATE_COPY);
    hipLaunchKernelGGL(bgzf_copy, dim3((unsigned)nb), dim3(64), 0, ctx->stream, ca);
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
