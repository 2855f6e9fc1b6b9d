// bgzf_device.h — what the device BGZF decoder's translation units share (bam_device.hip: host side, CRC-32, record index;
// bgzf_decode.hip: the decode kernels): block descriptors, status words, Huffman table entries and the table builder.
// Wire format: SAM spec §4.1 (BGZF), RFC 1951 (DEFLATE); pysam / htslib's role for indexing.py:19,96-100.
#pragma once
#include "tcmi_internal.h"

namespace {

#ifndef TCMI_INFLATE_LL_ROOT
#define TCMI_INFLATE_LL_ROOT 9
#endif
#ifndef TCMI_INFLATE_D_ROOT
#define TCMI_INFLATE_D_ROOT 8
#endif
constexpr int LL_ROOT = TCMI_INFLATE_LL_ROOT, D_ROOT = TCMI_INFLATE_D_ROOT, CL_ROOT = 7;   // root-table bits; longer codes take slow_decode
static_assert(D_ROOT >= CL_ROOT, "the code-length table borrows the distance table's LDS");
constexpr int MAX_REC_PER_BLOCK = 65536 / 36 + 2;   // a record is at least 36 bytes (block_size + 32 fixed + 1 name byte ..)

struct BlockDesc {
    uint64_t cin;        // first byte of the deflate payload in the file
    uint64_t uout;       // first byte of its output in the inflated stream
    uint32_t clen;       // payload bytes
    uint32_t ulen;       // ISIZE
    int32_t entry;       // offset of the first record start inside this block (>= 0), or -1: no record walk (header blocks)
    uint32_t tok_cap;    // tokens this block may produce at most (bgzf_symbols)
    uint64_t tok;        // its first token in the token array
};

// status word of a block
enum { ST_OK = 0, ST_BAD_STREAM = 1, ST_BAD_LENGTH = 2, ST_BAD_RECORD = 3, ST_BAD_CRC = 4 };

static __constant__ uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// LDS written by some lanes of a wavefront and read by others of the same wavefront: the hardware keeps a wavefront's LDS
// operations in order, so all this has to do is keep the compiler from moving them (no barrier: a workgroup may hold several
// wavefronts that each work on a block of their own)
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// Inclusive scans over the 64 lanes of a wavefront with data-parallel-primitive moves (no LDS, no waits): four shifts inside
// the rows of 16 lanes, then the last lane of row 0 / 2 into row 1 / 3 and lane 31 into the upper half.  Lanes shifted in
// from outside a row read 0, the identity of both operators below (the maximum is over unsigned values).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_shift(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)
{
    v += dpp_shift<0x111, 0xF>(v);          // row_shr:1
    v += dpp_shift<0x112, 0xF>(v);          // row_shr:2
    v += dpp_shift<0x114, 0xF>(v);          // row_shr:4
    v += dpp_shift<0x118, 0xF>(v);          // row_shr:8
    v += dpp_shift<0x142, 0xA>(v);          // row_bcast:15 into rows 1 and 3
    v += dpp_shift<0x143, 0xC>(v);          // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v)
{
    v = max(v, dpp_shift<0x111, 0xF>(v));
    v = max(v, dpp_shift<0x112, 0xF>(v));
    v = max(v, dpp_shift<0x114, 0xF>(v));
    v = max(v, dpp_shift<0x118, 0xF>(v));
    v = max(v, dpp_shift<0x142, 0xA>(v));
    v = max(v, dpp_shift<0x143, 0xC>(v));
    return v;
}

// Root-table entries (32 bits): code length in bits 0-3 (0: not in the root table — a longer code or none), kind in bits 8-10;
// a literal in bits 16-23; a distance: extra bits in bits 4-7, base in bits 16-31; a length: see make_entry.  The hot loop needs
// no arithmetic on symbols.
constexpr uint32_t E_LIT = 1u << 8, E_BASE = 1u << 9, E_EOB = 1u << 10;
enum { K_LITLEN = 0, K_DIST = 1, K_CODELEN = 2 };

typedef uint32_t tab_t;             // (16-bit entries with base / extra bits computed per symbol: half the table LDS, measured slower)

__device__ inline uint32_t make_entry(int kind, int sym, int nbits)
{
    if (sym < 0) return 0u;
    if (kind == K_CODELEN) return (uint32_t)nbits | ((uint32_t)sym << 16);
    if (kind == K_LITLEN) {
        if (sym < 256) return (uint32_t)nbits | E_LIT | ((uint32_t)sym << 16);
        if (sym == 256) return (uint32_t)nbits | E_EOB;
        const int s = sym - 257;
        if (s > 28) return 0u;                                  // 286, 287: not a symbol
        int eb = 0, base = 3 + s;
        if (s == 28) base = 258;
        else if (s >= 8) { eb = (s >> 2) - 1; base = 3 + ((4 + (s & 3)) << eb); }
        // a LENGTH entry is laid out for the ISA loop: extra-bit count in bits 16-19 (with the code length in bits 0-3 that is an
        // s_bfe_u32 operand: entry & 0x000F000F), code + extra bits in bits 11-15, base length in bits 20-28
        return (uint32_t)nbits | ((uint32_t)(nbits + eb) << 11) | E_BASE | ((uint32_t)eb << 16) | ((uint32_t)base << 20);
    }
    if (sym > 29) return 0u;                                    // 30, 31: not a distance
    int eb = 0, base = 1 + sym;
    if (sym >= 4) { eb = (sym >> 1) - 1; base = 1 + ((2 + (sym & 1)) << eb); }
    return (uint32_t)nbits | ((uint32_t)eb << 4) | E_BASE | ((uint32_t)base << 16);
}

// lens[0 .. n) -> cnt[1 .. 15], canonically ordered symbols, and the root table.  Returns false for an over-subscribed code.
// The kernel is bound by instruction issue, and a quarter of its instructions were spent here: so the per-length counters,
// offsets and first codes live in scalar registers (both loops over the 15 lengths are unrolled: the indices are constants),
// no read-modify-write goes through LDS, and a table slot is decoded by nine compare-and-select steps against those scalars
// instead of a bit-by-bit walk with an LDS read per bit.  MAXG: groups of 64 symbols (5 for the 286 literal/length codes).
template <int MAXG, int ROOT>
__device__ __forceinline__ bool build_table(const uint8_t *lens, int n, uint16_t *cnt, uint16_t *sym, tab_t *tab, int kind, uint32_t *rs)
{
    const int lane = threadIdx.x & 63;
    wave_sync();                                // (LDS hand-offs between the lanes of this wavefront)
    int l[MAXG];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) l[g] = g * 64 + lane < n ? (int)lens[g * 64 + lane] : 0;
    // histogram of the code lengths: one ballot per length and group (wave-uniform counters)
    int c[16];
#pragma unroll
    for (int len = 0; len < 16; ++len) c[len] = 0;
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
#pragma unroll
        for (int len = 1; len <= 15; ++len) c[len] += (int)__popcll(__ballot(l[g] == len));
    }
    // ... to LDS for the long-code walk (long_code(), slow_decode())
    {
        int mine = 0;
#pragma unroll
        for (int len = 1; len <= 15; ++len) mine = lane == len ? c[len] : mine;
        if (lane < 16) cnt[lane] = (uint16_t)mine;
    }
    // offsets of each length in the sorted symbol array, first code of each length; over-subscription check
    int off[16], first[16];
    int left = 1, o = 0, f = 0;
    bool ok = true;
#pragma unroll
    for (int len = 1; len <= 15; ++len) {
        left = (left << 1) - c[len];
        if (left < 0) ok = false;
        off[len] = o;
        first[len] = f;
        o += c[len];
        f = (f + c[len]) << 1;
        if (len == ROOT && lane == 0) { rs[0] = (uint32_t)f; rs[1] = (uint32_t)o; }    // long_code() starts here
    }
    // rank of every symbol among those of its length, in symbol order -> its slot in the sorted array
    {
        int nx[16];
#pragma unroll
        for (int len = 1; len <= 15; ++len) nx[len] = off[len];
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
#pragma unroll
            for (int len = 1; len <= 15; ++len) {
                const unsigned long long m = __ballot(l[g] == len);
                if (m) {
                    const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (l[g] == len) sym[nx[len] + below] = (uint16_t)(g * 64 + lane);
                    nx[len] += (int)__popcll(m);
                }
            }
        }
    }
    wave_sync();
    // root table: slot i holds the symbol whose code is a prefix of the bits of i (first stream bit = bit 0): the code of
    // length len that the slot starts with is p = reverse(i)'s top len bits; it exists if first[len] <= p < first[len] + c[len]
    // (for a prefix code at most one length answers)
    for (int i = lane; i < (1 << ROOT); i += 64) {
        const uint32_t r = __builtin_bitreverse32((uint32_t)i) >> (32 - ROOT);
        int L = 0, si = 0;
#pragma unroll
        for (int len = 1; len <= ROOT; ++len) {
            const uint32_t d = (r >> (ROOT - len)) - (uint32_t)first[len];
            const bool hit = d < (uint32_t)c[len];
            L = hit ? len : L;
            si = hit ? off[len] + (int)d : si;
        }
        tab[i] = L ? make_entry(kind, (int)sym[si], L) : 0u;
    }
    wave_sync();
    return uni(ok ? 1u : 0u) != 0;
}


} // namespace

// bgzf_decode.hip: compressed file + block table in HBM -> inflated stream, record starts per block, a status word per block
// (everything on ctx->stream; the arrays are the caller's, sized as bam_device.hip's decode_on_device sizes them)
struct tcmi_bgzf_decode_args {
    const uint8_t *d_file;          // compressed file, 16-byte aligned, >= 4 KiB of slack behind it
    const void *d_desc;             // BlockDesc [n_blocks]
    uint32_t *d_tok;                // token array (BlockDesc::tok / tok_cap)
    uint32_t *d_ntok;               // [n_blocks]
    uint8_t *d_out;                 // inflated stream (BlockDesc::uout)
    uint32_t *d_slot;               // [n_blocks][MAX_REC_PER_BLOCK] record starts
    uint32_t *d_nrec;               // [n_blocks]
    int32_t *d_over;                // [n_blocks] bytes by which a block's last record runs into the next blocks
    uint32_t *d_first;              // [n_blocks] offset of the first record start the block found in itself (0xFFFFFFFF: none)
    uint32_t *d_stat;               // [n_blocks] ST_*
    size_t n_blocks;                // of the file (or range): the arrays' length
    size_t first_block = 0, count = ~(size_t)0;    // the blocks this launch decodes ([first_block, first_block + count), clipped): a batch
    uint64_t tok_base = 0;          // d_tok holds the tokens from BlockDesc::tok = tok_base on (the batch's first block's)
    uint32_t pay_dwords;            // the largest block's payload in dwords + slack
    uint32_t n_ref;                 // reference sequences of the BAM header (a record's refID must be one of them)
    int verify_crc = 1;             // bgzf_copy checks every block's CRC-32 against its trailer while it flushes the bytes (ST_BAD_CRC)
    int scratch_div = 1;            // (tests) a lane of bgzf_symbols may park 1 / scratch_div of its share of the token scratch: overflowing lanes send their block through pass B
    int short_tokens;               // the file compresses less than ~12 : 1 (many short matches): bgzf_copy's variant with teams; 2: less than ~4 : 1: ... and short far matches finished in the set-up
};
int tcmi_bgzf_decode_launch(tcmi_ctx *ctx, const tcmi_bgzf_decode_args &a);
