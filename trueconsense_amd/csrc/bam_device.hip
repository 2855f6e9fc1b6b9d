// bam_device.hip — DEVICE: a BAM file's BGZF blocks -> the inflated BAM byte stream + the offset of every alignment
// record, in HBM, without the host ever decoding the file (pysam / htslib's role for indexing.py:19,96-100; wire format
// SAM spec §4.1 BGZF, §4.2 BAM, RFC 1951 DEFLATE).
//
// The host only reads the file, walks the gzip member headers (18 bytes per block) and inflates the first block(s) far
// enough to parse the BAM header; the compressed bytes (a few MB .. tens of MB) cross PCIe once.
//
//   bgzf_inflate   ONE WAVEFRONT PER BGZF BLOCK (blocks are independent deflate streams of <= 64 KiB).  All decoder state
//                  is wave-uniform: the bit buffer, the Huffman root tables (9 / 8 bits, in LDS, built in parallel from
//                  the canonical code: every lane decodes its table indices bit by bit; longer codes resume the canonical
//                  walk behind the root bits), the output position.  Literals and LZ77 matches go through a 4 KiB LDS ring of
//                  the most recent output (a match is copied by all 64 lanes at once; the match that reaches further back —
//                  the deflate window is 32 KiB — reads what was already flushed); finished 2 KiB segments are flushed to HBM
//                  with 16-byte stores.  8.8 KiB of LDS per wavefront, on purpose: 18 blocks are resident per CU, so ALL
//                  4 187 blocks of a 1M-read BAM run at once (with 12.9 KiB — 10 / 9-bit tables, a 1 KiB input ring — 11 fit
//                  and the kernel took a second, two-thirds empty round: 1.19 ms; now 0.81 ms).  The kernel is bound by
//                  instruction issue: ~120 k wave-instructions per block, most of them scalar, one per SIMD turn.  While it
//                  inflates, the wave also follows the chain of BAM records through its block (block_size fields, read from
//                  the ring as soon as they are complete) and lists the record starts: htslib-written BAMs start every BGZF
//                  block on a record boundary, which the chain check (`overshoot` of a block = 0) verifies; files that do
//                  not are left to the host reader.
//   bgzf_crc32     the CRC-32 of every block's output against its trailer (one wavefront per block, coalesced rows)
//   rec_compact    per-block record lists -> one dense array of record offsets (block scan + copy)
//
// Serial-latency-bound bit / byte work, not HBM-bound and not a contraction: no MFMA.
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bgzf_device.h"

namespace {

#ifndef TCMI_INFLATE_WIN
#define TCMI_INFLATE_WIN 4096
#endif
constexpr int WIN = TCMI_INFLATE_WIN, WMASK = WIN - 1;   // LDS ring: the most recent output.  The deflate window is 32 KiB: a match that
                                                         // reaches further back than the ring reads what was already flushed to HBM
constexpr int SEG = TCMI_INFLATE_WIN >= 4096 ? 2048 : TCMI_INFLATE_WIN / 2;   // flush granularity
constexpr int NEAR = WIN - 264;                 // matches up to this distance are copied LDS -> LDS
static_assert(SEG <= WIN - 528 && (WIN & (WIN - 1)) == 0 && WIN % SEG == 0 && SEG % 1024 == 0, "a far match must find its source flushed");




struct InflateArgs {
    const uint32_t *__restrict__ file32;     // the file as 4-byte words (16-byte aligned base, >= 64 bytes of slack behind it)
    const BlockDesc *blocks;
    uint8_t *out;               // inflated stream
    uint32_t *rec_slot;         // [n_blocks][MAX_REC_PER_BLOCK]: record starts relative to the block's first byte
    uint32_t *n_rec;            // [n_blocks]
    int32_t *overshoot;         // [n_blocks]: bytes by which the block's last record runs into the next block
    uint32_t *status;           // [n_blocks]
    int32_t n_blocks;
};

#ifndef TCMI_INFLATE_INRING
#define TCMI_INFLATE_INRING 128
#endif
constexpr int IN_RING = TCMI_INFLATE_INRING;    // dwords of compressed input staged in LDS (two halves)


// Canonical Huffman decode (puff.c's loop) of a code that is known to be longer than `root` bits (the root table said so),
// wave-uniform: the canonical decoder's state after `root` bits depends on the counts alone (`rs` = {first, index} at that point, left by build_table), and
// the first `root` bits of the code are the bit-reversed low bits of `v` — so the walk starts at length root + 1 and takes
// one to three rounds for the codes that occur.  -> symbol | code length << 16, or -1
__device__ inline int long_code(const uint16_t *cnt, const uint16_t *sym, const uint32_t *rs, uint32_t v, int root)
{
    int first = (int)uni(rs[0]), index = (int)uni(rs[1]);
    int code = (int)((__builtin_bitreverse32(v) >> (32 - root)) << 1);
    v >>= root;
#pragma unroll 1
    for (int len = root + 1; len <= 15; ++len) {
        code |= (int)(v & 1u);
        v >>= 1;
        const int c = (int)uni(cnt[len]);
        if (code - c < first) return (int)uni(sym[index + (code - first)]) | (len << 16);
        index += c;
        first = (first + c) << 1;
        code <<= 1;
    }
    return -1;
}


struct InflateLds {
    __attribute__((aligned(16))) uint8_t win[WIN];      // the most recent output
    tab_t ll[1 << LL_ROOT];
    tab_t dt[1 << D_ROOT];
    uint8_t lens[320];                          // code lengths: literal/length [0, nlen), distance [nlen, nlen + ndist)
    uint16_t sym_ll[288], sym_d[32], sym_cl[20];
    uint16_t cnt_ll[16], cnt_d[16], cnt_cl[16], nxt[16];
    uint8_t cll[20];
    uint32_t rs[6];                             // long_code()'s starting state per table: {first, index} after the root bits
    __attribute__((aligned(16))) uint32_t in[IN_RING];      // compressed input, two halves
};
static_assert(offsetof(InflateLds, win) == 0, "the window's ring index is its LDS address");

struct Bits {                   // wave-uniform bit reader over the file's dwords, staged through an LDS ring
    const uint32_t *__restrict__ w;
    uint32_t *ring;             // LDS [IN_RING]: dword k of the file sits in ring[k % IN_RING] while hi - IN_RING <= k < hi
    uint32_t idx;               // the dword that `next` holds: the next one to enter the bit buffer (files < 16 GiB)
    uint32_t hi;                // a multiple of IN_RING / 2: dwords [hi - IN_RING, hi - IN_RING/2) are in the ring for sure,
                                // [hi - IN_RING/2, hi) were requested at the last stage and are waited for at the next
    uint32_t next;              // ring[idx], read one refill ahead (per lane the same value; made scalar when it is used)
    uint64_t bb;
    int bc;
};
constexpr int IN_HALF = IN_RING / 2;
static_assert(IN_HALF == 64, "one half of the input ring = one dword per lane");

// The half of the ring that has been consumed is requested anew: 64 dwords straight from HBM into LDS (global_load_lds_dword:
// no staging registers, nothing for the wave to do when they arrive).  Nothing reads that half before the NEXT stage, which
// starts by waiting for this request — by then it is a half ring of decoding old.  (Issued and awaited in ISA: the compiler
// would wait for it before the very next LDS read.)
__device__ inline void stage_input(Bits &b)
{
    const uint32_t *src = b.w + b.hi;
    const uint32_t lds = (uint32_t)offsetof(InflateLds, in) + (b.hi & (uint32_t)(IN_RING - 1)) * 4u;
    asm volatile("s_waitcnt vmcnt(0)\n"
                 "s_mov_b32 m0, %[lds]\n"
                 "s_nop 0\n"
                 "global_load_lds_dword %[voff], %[base]\n"
                 :: [lds] "s"(lds), [voff] "v"((uint32_t)threadIdx.x * 4u), [base] "s"(src) : "memory");       // (m0 is not the compiler's to allocate)
    b.hi += IN_HALF;
}

__device__ inline void seek_bits(Bits &b, uint64_t byte)       // start reading bits at this byte of the file
{
    b.idx = uni((uint32_t)(byte >> 2));
    b.hi = b.idx & ~(uint32_t)(IN_HALF - 1);
    stage_input(b);
    stage_input(b);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int skip = (int)(byte & 3) * 8;
    b.bb = (uint64_t)(uni(b.ring[b.idx & (IN_RING - 1)]) >> skip);
    b.bc = 32 - skip;
    ++b.idx;
    if (b.idx + IN_HALF >= b.hi) stage_input(b);
    b.next = b.ring[b.idx & (IN_RING - 1)];
}

__device__ inline void refill(Bits &b)
{
    if (b.bc <= 32) {
        b.bb |= (uint64_t)uni(b.next) << b.bc;
        b.bc += 32;
        ++b.idx;
        if (b.idx + IN_HALF >= b.hi) stage_input(b);
        b.next = b.ring[b.idx & (IN_RING - 1)];
    }
}
__device__ inline uint32_t take(Bits &b, int n)     // n <= 25, after a refill
{
    const uint32_t v = (uint32_t)b.bb & ((1u << n) - 1u);
    b.bb >>= n;
    b.bc -= n;
    return v;
}


// ---- the symbol loop's fast path, hand-scheduled -------------------------------------------------------------------------
// Literals and matches whose codes sit in the root tables, decoded and — the common kind of match: source inside the LDS ring,
// no overlap with the destination — copied, until something else comes up.  Exit codes:
//   0  `op` reached `next_evt`: housekeeping is due
//   1  at a symbol's first bit: the input ring needs its next half, or a long code / none / the end-of-block code is next
//      -> the C++ step below takes that ONE symbol
//   2  a match whose length is decoded (`len`, bits consumed); its distance code is a long one, or the ring needs its next half
//      first -> C++ decodes the distance and copies
//   3  a match with length and distance decoded (bits consumed) whose source overlaps its destination, or that is impossible
//      -> C++ copies / flags it.  (A FAR match — source already flushed to HBM — is copied here too, through global loads.)
// (Measured and dropped: long matches copied four bytes per lane with unaligned ds_read_b32 / ds_write_b32 — fewer instructions,
// 643 vs 619 us: unaligned LDS dwords are not cheap.  And: the next symbol's table entry requested as soon as the current symbol's
// bits are consumed, a whole copy ahead of its use — 609 vs 610 us: the look-up's latency is not what the kernel waits for, the
// issue slots are.)
// Written in ISA because the kernel is bound by instruction issue and the compiler's version of this loop spends a third of
// its instructions on flags that say which path it came along (35 instructions per literal, 95 per match; here 20 and 60).
// All state is wave-uniform, in scalar registers.
static_assert(LL_ROOT == 9 && D_ROOT == 8 && IN_RING == 128, "masks 0x1ff / 0xff / 127 below");
__device__ __forceinline__ uint32_t fast_symbols(Bits &b, uint32_t &op, uint32_t next_evt, int lane, const uint8_t *out, uint32_t &len_out,
                                                  uint32_t &dist_out)
{
    uint32_t code, t0, t1, t2, e, f, nb, len, dist;
    uint32_t vt, ve, vf, vto, vfrom, vb, vi;
#define TCMI_ASM_REFILL(exit_label_)                                                                                           \
        "s_add_u32 %[t0], %[idx], 65\n"        /* the word after next must be in the ring (idx + 1 + half < hi) */               \
        "s_cmp_lt_u32 %[t0], %[hi]\n"                                                                                          \
        "s_cbranch_scc0 " exit_label_ "%=\n"                                                                                   \
        /* (the read of vnext was issued a table look-up ago: its s_waitcnt covered it) */                                      \
        "v_readfirstlane_b32 s98, %[vnext]\n"                                                                                  \
        "s_add_u32 %[idx], %[idx], 1\n"                                                                                        \
        "s_and_b32 %[t0], %[idx], 127\n"                                                                                       \
        "s_lshl_b32 %[t0], %[t0], 2\n"                                                                                         \
        "v_mov_b32 %[vt], %[t0]\n"                                                                                             \
        "ds_read_b32 %[vnext], %[vt] offset:%[oin]\n"                                                                            \
        "s_mov_b32 s99, 0\n"                                                                                                   \
        "s_lshl_b64 s[98:99], s[98:99], %[bc]\n"                                                                               \
        "s_or_b64 s[96:97], s[96:97], s[98:99]\n"                                                                              \
        "s_add_u32 %[bc], %[bc], 32\n"
    asm volatile(
        "LS%=:\n"                                                  // ---- next symbol
        "s_cmp_gt_i32 %[bc], 32\n"
        "s_cbranch_scc1 LK%=\n"
        TCMI_ASM_REFILL("LX1")
        "LK%=:\n"                                                  // ---- literal / length code
        "s_and_b32 %[t0], s96, 0x1ff\n"
        "s_lshl_b32 %[t0], %[t0], 2\n"
        "v_mov_b32 %[vt], %[t0]\n"
        "ds_read_b32 %[ve], %[vt] offset:%[oll]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_readfirstlane_b32 %[e], %[ve]\n"
        "s_and_b32 %[nb], %[e], 15\n"
        "s_bitcmp1_b32 %[e], 8\n"
        "s_cbranch_scc0 LM%=\n"
        "s_and_b32 %[t0], %[op], %[wmask]\n"                           // a literal: bits 16-23 of the entry
        "v_mov_b32 %[vt], %[t0]\n"
        "ds_write_b8_d16_hi %[vt], %[ve]\n"
        "s_lshr_b64 s[96:97], s[96:97], %[nb]\n"
        "s_sub_u32 %[bc], %[bc], %[nb]\n"
        "s_add_u32 %[op], %[op], 1\n"
        "s_cmp_lt_u32 %[op], %[evt]\n"
        "s_cbranch_scc1 LS%=\n"
        "s_branch LX0%=\n"
        "LM%=:\n"                                                  // ---- a match?  (not: a long code or none — entry 0 —, end of block -> C++)
        "s_bitcmp1_b32 %[e], 9\n"
        "s_cbranch_scc0 LX1%=\n"
        "s_and_b32 %[t1], %[e], 0x000f000f\n"                       // (extra-bit count << 16 | code length: the field of the extra bits)
        "s_bfe_u32 %[t0], s96, %[t1]\n"
        "s_bfe_u32 %[len], %[e], 0x90014\n"                         // base length
        "s_add_u32 %[len], %[len], %[t0]\n"
        "s_bfe_u32 %[t1], %[e], 0x5000b\n"                          // code + extra bits
        "s_lshr_b64 s[96:97], s[96:97], %[t1]\n"
        "s_sub_u32 %[bc], %[bc], %[t1]\n"
        "s_cmp_gt_i32 %[bc], 32\n"                                  // (from here on the length is consumed: leaving = code 2)
        "s_cbranch_scc1 LF%=\n"
        TCMI_ASM_REFILL("LX2")
        "LF%=:\n"                                                  // ---- distance code
        "s_and_b32 %[t0], s96, 0xff\n"
        "s_lshl_b32 %[t0], %[t0], 2\n"
        "v_mov_b32 %[vt], %[t0]\n"
        "ds_read_b32 %[vf], %[vt] offset:%[odt]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_readfirstlane_b32 %[f], %[vf]\n"
        "s_and_b32 %[t1], %[f], 15\n"
        "s_cmp_eq_u32 %[t1], 0\n"
        "s_cbranch_scc1 LX2%=\n"
        "s_bfe_u32 %[t2], %[f], 0x40004\n"
        "s_lshr_b32 %[t0], s96, %[t1]\n"
        "s_add_u32 %[t1], %[t1], %[t2]\n"
        "s_bfm_b32 %[t2], %[t2], 0\n"
        "s_and_b32 %[t0], %[t0], %[t2]\n"
        "s_lshr_b32 %[dist], %[f], 16\n"
        "s_add_u32 %[dist], %[dist], %[t0]\n"
        "s_lshr_b64 s[96:97], s[96:97], %[t1]\n"
        "s_sub_u32 %[bc], %[bc], %[t1]\n"
        "s_cmp_gt_u32 %[dist], %[near]\n"                              // beyond the LDS ring  (length and distance consumed: leaving = code 3)
        "s_cbranch_scc1 LG%=\n"
        "s_cmp_lt_u32 %[dist], %[len]\n"                            // source overlaps destination
        "s_cbranch_scc1 LO%=\n"
        "LN%=:\n"
        "s_cmp_gt_u32 %[dist], %[op]\n"                             // before the block's first byte: C++ flags it
        "s_cbranch_scc1 LX3%=\n"
        "v_add_u32 %[vto], %[op], %[vlane]\n"                       // the copy: 64 bytes per round, lane i byte i.  Whole rounds need no
        "v_subrev_u32 %[vfrom], %[dist], %[vto]\n"                  // lane mask; the last one (1 .. 64 bytes) does
        "s_mov_b32 %[t1], %[len]\n"                                 // bytes of the last round: all of them for a match of up to 64
        "s_cmp_le_u32 %[len], 64\n"
        "s_cbranch_scc1 LE%=\n"
        "s_add_u32 %[t0], %[len], -1\n"
        "s_lshr_b32 %[t0], %[t0], 6\n"                              // whole rounds in front of the last one
        "s_lshl_b32 %[t1], %[t0], 6\n"
        "s_sub_u32 %[t1], %[len], %[t1]\n"
        "LC%=:\n"
        "v_and_b32 %[vt], %[wmask], %[vfrom]\n"
        "ds_read_u8 %[vb], %[vt]\n"
        "v_and_b32 %[vt], %[wmask], %[vto]\n"
        "v_add_u32 %[vfrom], 64, %[vfrom]\n"
        "v_add_u32 %[vto], 64, %[vto]\n"
        "s_sub_u32 %[t0], %[t0], 1\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_write_b8 %[vt], %[vb]\n"
        "s_cmp_lg_u32 %[t0], 0\n"
        "s_cbranch_scc1 LC%=\n"
        "LE%=:\n"
        "v_cmp_gt_u32 vcc, %[t1], %[vlane]\n"
        "s_and_saveexec_b64 s[94:95], vcc\n"
        "v_and_b32 %[vt], %[wmask], %[vfrom]\n"
        "ds_read_u8 %[vb], %[vt]\n"
        "v_and_b32 %[vt], %[wmask], %[vto]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_write_b8 %[vt], %[vb]\n"
        "s_mov_b64 exec, s[94:95]\n"
        "LD%=:\n"
        "s_add_u32 %[op], %[op], %[len]\n"
        "s_cmp_lt_u32 %[op], %[evt]\n"
        "s_cbranch_scc1 LS%=\n"
        "s_branch LX0%=\n"
        "LO%=:\n"                                                  // ---- overlap: rounds of 64 bytes, one after the other, are still right when
        "s_cmp_lt_u32 %[dist], 64\n"                                //      the source lies a whole round back; a shorter period goes to C++
        "s_cbranch_scc1 LX3%=\n"
        "s_branch LN%=\n"
        "LG%=:\n"                                                  // ---- a far match: its source has left the ring, this wave flushed it to `out` earlier
        "s_cmp_gt_u32 %[dist], %[op]\n"
        "s_cbranch_scc1 LX3%=\n"
        "s_sub_u32 %[t0], %[op], %[dist]\n"
        "s_add_u32 s98, s92, %[t0]\n"
        "s_addc_u32 s99, s93, 0\n"
        "v_add_u32 %[vto], %[op], %[vlane]\n"
        "v_mov_b32 %[vi], %[vlane]\n"
        "s_mov_b32 %[t0], 64\n"
        "LH%=:\n"
        "v_cmp_gt_u32 vcc, %[len], %[vi]\n"
        "s_and_saveexec_b64 s[94:95], vcc\n"
        "global_load_ubyte %[vb], %[vi], s[98:99]\n"
        "v_and_b32 %[vt], %[wmask], %[vto]\n"
        "s_waitcnt vmcnt(0)\n"
        "ds_write_b8 %[vt], %[vb]\n"
        "s_mov_b64 exec, s[94:95]\n"
        "s_cmp_lt_u32 %[t0], %[len]\n"
        "s_cbranch_scc0 LD%=\n"
        "s_add_u32 %[t0], %[t0], 64\n"
        "v_add_u32 %[vi], 64, %[vi]\n"
        "v_add_u32 %[vto], 64, %[vto]\n"
        "s_branch LH%=\n"
        "LX0%=:\n"
        "s_mov_b32 %[code], 0\n"
        "s_branch LX%=\n"
        "LX1%=:\n"
        "s_mov_b32 %[code], 1\n"
        "s_branch LX%=\n"
        "LX2%=:\n"
        "s_mov_b32 %[code], 2\n"
        "s_branch LX%=\n"
        "LX3%=:\n"
        "s_mov_b32 %[code], 3\n"
        "LX%=:\n"
        "s_waitcnt lgkmcnt(0)\n"
        : "+{s[96:97]}"(b.bb), [bc] "+s"(b.bc), [op] "+s"(op), [idx] "+s"(b.idx), [vnext] "+v"(b.next), [code] "=&s"(code),
          [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [e] "=&s"(e), [f] "=&s"(f), [nb] "=&s"(nb), [len] "=&s"(len),
          [dist] "=&s"(dist), [vt] "=&v"(vt), [ve] "=&v"(ve), [vf] "=&v"(vf), [vto] "=&v"(vto), [vfrom] "=&v"(vfrom),
          [vb] "=&v"(vb), [vi] "=&v"(vi)
        : [evt] "s"(next_evt), [hi] "s"(b.hi), [vlane] "v"(lane), "{s[92:93]}"(out), [wmask] "n"(WMASK), [near] "n"(NEAR),
          [oll] "n"(offsetof(InflateLds, ll)), [odt] "n"(offsetof(InflateLds, dt)), [oin] "n"(offsetof(InflateLds, in))
        : "s94", "s95", "s98", "s99", "vcc", "scc", "memory");
#undef TCMI_ASM_REFILL
    len_out = len;
    dist_out = dist;
    return code;
}

// ---- the code-length symbols of a dynamic block (RFC 1951 3.2.7), hand-scheduled like the symbol loop: 0 .. 15 = the next
// symbol's code length, 16 = repeat the previous length 3 - 6 times, 17 / 18 = 3 - 10 / 11 - 138 zeros.  The lengths go to
// lens[got ..] (pre-zeroed: zeros are not stored), literal/length and distance lengths in one run.  -> 0: all `total` lengths
// are in; 1: the input ring needs its next half first — C++ takes ONE symbol; 2: damaged stream.
__device__ __forceinline__ uint32_t cl_symbols(Bits &b, uint32_t &got, uint32_t &prev, uint32_t total, int lane)
{
    uint32_t code, t0, t1, t2, e, nb, vt, ve, vb;
#define TCMI_ASM_REFILL(exit_label_)                                                                                           \
        "s_add_u32 %[t0], %[idx], 65\n"                                                                                        \
        "s_cmp_lt_u32 %[t0], %[hi]\n"                                                                                          \
        "s_cbranch_scc0 " exit_label_ "%=\n"                                                                                   \
        "s_waitcnt lgkmcnt(0)\n"                                                                                               \
        "v_readfirstlane_b32 s98, %[vnext]\n"                                                                                  \
        "s_add_u32 %[idx], %[idx], 1\n"                                                                                        \
        "s_and_b32 %[t0], %[idx], 127\n"                                                                                       \
        "s_lshl_b32 %[t0], %[t0], 2\n"                                                                                         \
        "v_mov_b32 %[vt], %[t0]\n"                                                                                             \
        "ds_read_b32 %[vnext], %[vt] offset:%[oin]\n"                                                                          \
        "s_mov_b32 s99, 0\n"                                                                                                   \
        "s_lshl_b64 s[98:99], s[98:99], %[bc]\n"                                                                               \
        "s_or_b64 s[96:97], s[96:97], s[98:99]\n"                                                                              \
        "s_add_u32 %[bc], %[bc], 32\n"
    asm volatile(
        "LA%=:\n"
        "s_cmp_lt_u32 %[got], %[total]\n"
        "s_cbranch_scc0 LZ0%=\n"
        "s_cmp_gt_i32 %[bc], 32\n"
        "s_cbranch_scc1 LB%=\n"
        TCMI_ASM_REFILL("LZ1")
        "LB%=:\n"
        "s_and_b32 %[t0], s96, 0x7f\n"
        "s_lshl_b32 %[t0], %[t0], 2\n"
        "v_mov_b32 %[vt], %[t0]\n"
        "ds_read_b32 %[ve], %[vt] offset:%[ocl]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_readfirstlane_b32 %[e], %[ve]\n"
        "s_and_b32 %[nb], %[e], 15\n"
        "s_cmp_eq_u32 %[nb], 0\n"
        "s_cbranch_scc1 LZ2%=\n"
        "s_lshr_b64 s[96:97], s[96:97], %[nb]\n"
        "s_sub_u32 %[bc], %[bc], %[nb]\n"
        "s_lshr_b32 %[t1], %[e], 16\n"                              // the symbol
        "s_cmp_lt_u32 %[t1], 16\n"
        "s_cbranch_scc0 LR%=\n"
        "s_cmp_eq_u32 %[t1], 0\n"                                   // ---- one length
        "s_cbranch_scc1 LN%=\n"
        "s_add_u32 %[t0], %[got], %[olens]\n"
        "v_mov_b32 %[vt], %[t0]\n"
        "v_mov_b32 %[vb], %[t1]\n"
        "ds_write_b8 %[vt], %[vb]\n"
        "LN%=:\n"
        "s_mov_b32 %[prev], %[t1]\n"
        "s_add_u32 %[got], %[got], 1\n"
        "s_branch LA%=\n"
        "LR%=:\n"
        "s_cmp_eq_u32 %[t1], 16\n"
        "s_cbranch_scc0 LP%=\n"
        "s_cmp_eq_u32 %[got], 0\n"                                  // ---- 16: the previous length 3 - 6 times
        "s_cbranch_scc1 LZ2%=\n"
        "s_and_b32 %[t2], s96, 3\n"
        "s_add_u32 %[t2], %[t2], 3\n"
        "s_lshr_b64 s[96:97], s[96:97], 2\n"
        "s_sub_u32 %[bc], %[bc], 2\n"
        "s_add_u32 %[t0], %[got], %[t2]\n"
        "s_cmp_gt_u32 %[t0], %[total]\n"
        "s_cbranch_scc1 LZ2%=\n"
        "s_cmp_eq_u32 %[prev], 0\n"
        "s_cbranch_scc1 LQ%=\n"
        "v_cmp_gt_u32 vcc, %[t2], %[vlane]\n"
        "s_and_saveexec_b64 s[94:95], vcc\n"
        "s_add_u32 %[t1], %[got], %[olens]\n"
        "v_add_u32 %[vt], %[t1], %[vlane]\n"
        "v_mov_b32 %[vb], %[prev]\n"
        "ds_write_b8 %[vt], %[vb]\n"
        "s_mov_b64 exec, s[94:95]\n"
        "LQ%=:\n"
        "s_mov_b32 %[got], %[t0]\n"
        "s_branch LA%=\n"
        "LP%=:\n"
        "s_cmp_eq_u32 %[t1], 17\n"
        "s_cbranch_scc0 LO%=\n"
        "s_and_b32 %[t2], s96, 7\n"                                 // ---- 17: 3 - 10 zeros
        "s_add_u32 %[t2], %[t2], 3\n"
        "s_lshr_b64 s[96:97], s[96:97], 3\n"
        "s_sub_u32 %[bc], %[bc], 3\n"
        "s_branch LY%=\n"
        "LO%=:\n"
        "s_and_b32 %[t2], s96, 0x7f\n"                              // ---- 18: 11 - 138 zeros
        "s_add_u32 %[t2], %[t2], 11\n"
        "s_lshr_b64 s[96:97], s[96:97], 7\n"
        "s_sub_u32 %[bc], %[bc], 7\n"
        "LY%=:\n"
        "s_add_u32 %[t0], %[got], %[t2]\n"
        "s_cmp_gt_u32 %[t0], %[total]\n"
        "s_cbranch_scc1 LZ2%=\n"
        "s_mov_b32 %[prev], 0\n"
        "s_mov_b32 %[got], %[t0]\n"
        "s_branch LA%=\n"
        "LZ0%=:\n"
        "s_mov_b32 %[code], 0\n"
        "s_branch LZ%=\n"
        "LZ1%=:\n"
        "s_mov_b32 %[code], 1\n"
        "s_branch LZ%=\n"
        "LZ2%=:\n"
        "s_mov_b32 %[code], 2\n"
        "LZ%=:\n"
        "s_waitcnt lgkmcnt(0)\n"
        : "+{s[96:97]}"(b.bb), [bc] "+s"(b.bc), [got] "+s"(got), [prev] "+s"(prev), [idx] "+s"(b.idx), [vnext] "+v"(b.next), [code] "=&s"(code),
          [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [e] "=&s"(e), [nb] "=&s"(nb), [vt] "=&v"(vt), [ve] "=&v"(ve), [vb] "=&v"(vb)
        : [total] "s"(total), [hi] "s"(b.hi), [vlane] "v"(lane), [ocl] "n"(offsetof(InflateLds, dt)), [olens] "n"(offsetof(InflateLds, lens)),
          [oin] "n"(offsetof(InflateLds, in))
        : "s94", "s95", "s98", "s99", "vcc", "scc", "memory");
#undef TCMI_ASM_REFILL
    return code;
}

__global__ __launch_bounds__(64) void bgzf_inflate(InflateArgs a)
{
    __shared__ InflateLds L;                    // ONE LDS object: it sits at LDS address 0 and the member offsets are constants
                                                // (the hand-scheduled symbol loop addresses the window, the tables and the input ring by them)
    uint8_t *const s_win = L.win;
    tab_t *const s_ll = L.ll;
    tab_t *const s_dt = L.dt;
    tab_t *const s_cl = L.dt;                   // the code-length code is done with before the distance table is built
    uint8_t *const s_lens = L.lens;
    uint16_t *const s_sym_ll = L.sym_ll, *const s_sym_d = L.sym_d, *const s_sym_cl = L.sym_cl;
    uint16_t *const s_cnt_ll = L.cnt_ll, *const s_cnt_d = L.cnt_d, *const s_cnt_cl = L.cnt_cl;
    uint8_t *const s_cll = L.cll;
    uint32_t *const s_rs = L.rs;
    uint32_t *const s_in = L.in;

    const int lane = threadIdx.x;
    const int blk = blockIdx.x;
    if (blk >= a.n_blocks) return;
    const BlockDesc d = a.blocks[blk];
    const uint32_t ulen = d.ulen;
    uint8_t *out;
    {   // (made scalar by hand: the ISA loop wants the pointer in a scalar register pair)
        const uint64_t o = (uint64_t)reinterpret_cast<uintptr_t>(a.out + d.uout);
        out = reinterpret_cast<uint8_t *>(((uint64_t)uni((uint32_t)(o >> 32)) << 32) | uni((uint32_t)o));
    }
    uint32_t *slots = a.rec_slot + (size_t)blk * MAX_REC_PER_BLOCK;

    Bits b;
    b.w = a.file32;
    b.ring = s_in;
    auto seek = [&](uint64_t byte) __attribute__((always_inline)) { seek_bits(b, byte); };
    seek(d.cin);
    const uint32_t idx_end = (uint32_t)((d.cin + d.clen + 3) >> 2) + 3;      // reading further than this means a corrupt stream

    uint32_t op = 0;                    // bytes produced
    uint32_t flushed = 0;               // bytes already written to HBM (multiple of SEG)
    uint32_t err = ST_OK;
    // the chain of BAM records through this block
    uint32_t next_rec = d.entry >= 0 ? (uint32_t)d.entry : 0xFFFFFFF0u;
    uint32_t n_rec = 0, bad_rec = 0;
    uint32_t rec_buf = 0;               // lane k: the start of record (n_rec & ~63) + k, until 64 are together
    uint32_t next_evt = 0;              // output position at which the housekeeping below has something to do

    // after every symbol that carries `op` to `next_evt`: list the record starts whose block_size field is complete, and flush
    // the 2 KiB segments that are complete.  (The record walk as a kernel of its own — one lane per block over the flushed output —
    // takes 9 % off this kernel and still loses: 229 dependent reads per lane are 0.16 ms on every BAM's critical path, and
    // the file -> FASTA pipeline is bound by that path, not by issue slots: 41.2 vs 42.5 M positions/s, alternating on one box.)
    auto housekeeping = [&]() __attribute__((always_inline)) {
        const bool over = op > ulen;                            // (ring writes are masked: nothing was overwritten; no flush then)
        if (over) err = ST_BAD_LENGTH;
        while (!over && next_rec + 4 <= op) {                   // (op <= ulen here: the header lies inside the block)
            // block_size: the two ring words around it (lanes 0 and 1 of one read), shifted into place.  The starts are gathered
            // in a register, one lane each, and leave 64 at a time.  (At most ulen / 36 + 1 <= 1 821 records: the slots suffice.)
            const uint32_t at = next_rec & WMASK;
            const uint32_t w = reinterpret_cast<const uint32_t *>(s_win)[((at >> 2) + (uint32_t)lane) & (WIN / 4 - 1)];
            const uint64_t two = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)w, 1) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)w, 0);
            const uint32_t ubs = (uint32_t)(two >> ((at & 3u) * 8u));
            if (__builtin_expect(ubs - 32u > (1u << 28) - 32u, 0)) { bad_rec = 1; break; }     // (an impossible size must end the walk here: it
                                                                                               //  could overrun the slots or never end)
            rec_buf = (uint32_t)lane == (n_rec & 63u) ? next_rec : rec_buf;
            if ((n_rec & 63u) == 63u) slots[(n_rec & ~63u) + (uint32_t)lane] = rec_buf;
            ++n_rec;
            next_rec += 4u + ubs;
        }
        if (bad_rec) { err = ST_BAD_RECORD; next_rec = 0xFFFFFFF0u; }
        while (!over && op - flushed >= SEG) {
            const uint4 *src = reinterpret_cast<const uint4 *>(s_win + (flushed & WMASK));
            uint4 *dst = reinterpret_cast<uint4 *>(out + flushed);          // uout is a multiple of 16 (host pads blocks)
#pragma unroll
            for (int k = 0; k < SEG / 16 / 64; ++k) dst[k * 64 + lane] = src[k * 64 + lane];
            flushed += SEG;
        }
        // the next stop: a full segment.  (Not the next record header: the chain is caught up at every stop — a header that is
        // complete now is at most a segment and a match behind `op` then, well inside the ring — and leaving the symbol loop
        // costs as much as a few symbols.)
        next_evt = flushed + (uint32_t)SEG;
    };
    housekeeping();

    bool last = false;
    while (!last && err == ST_OK) {
        if (b.idx > idx_end) { err = ST_BAD_STREAM; break; }
        refill(b);
        last = take(b, 1) != 0;
        const uint32_t type = take(b, 2);
        if (type == 0) {
            // ---- stored block: byte-align, LEN / NLEN, LEN raw bytes ---------------------------------------------
            take(b, b.bc & 7);
            refill(b);
            const uint32_t len = take(b, 16);
            refill(b);
            const uint32_t nlen = take(b, 16);
            if ((len ^ nlen) != 0xFFFFu || op + len > ulen) { err = ST_BAD_STREAM; break; }
            // byte address of the raw data: what the bit buffer holds beyond it is dropped
            const uint64_t at = (uint64_t)b.idx * 4 - (uint64_t)(b.bc >> 3);
            const uint8_t *src = reinterpret_cast<const uint8_t *>(b.w) + at;
            uint32_t done = 0;
            while (done < len && err == ST_OK) {
                const uint32_t n = min(len - done, (uint32_t)SEG - (op & (SEG - 1)));
#pragma clang loop vectorize(disable) unroll(disable)
                for (uint32_t i = lane; i < n; i += 64) s_win[(op + i) & WMASK] = src[done + i];
                op += n;
                done += n;
                housekeeping();
            }
            seek(at + len);
            continue;
        }
        if (type == 3) { err = ST_BAD_STREAM; break; }
        // ---- code lengths of this block ----------------------------------------------------------------------------
        int nlen = 288, ndist = 32;
        if (type == 1) {
            __syncthreads();
            for (int i = lane; i < 320; i += 64) s_lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
        } else {
            refill(b);
            nlen = (int)take(b, 5) + 257;
            ndist = (int)take(b, 5) + 1;
            const int ncode = (int)take(b, 4) + 4;
            if (nlen > 286 || ndist > 30) { err = ST_BAD_STREAM; break; }
            __syncthreads();
            if (lane < 19) s_cll[lane] = 0;
            __syncthreads();
            for (int i = 0; i < ncode; ++i) {
                refill(b);
                const uint32_t v = take(b, 3);
                if (lane == 0) s_cll[CL_ORDER[i]] = (uint8_t)v;
            }
            if (uni(build_table<1, CL_ROOT>(s_cll, 19, s_cnt_cl, s_sym_cl, s_cl, K_CODELEN, s_rs + 4) ? 1u : 0u) == 0u) { err = ST_BAD_STREAM; break; }
            for (int i = lane; i < 320; i += 64) s_lens[i] = 0;
            __syncthreads();
            // (literal/length and distance lengths in one run: lens[0, nlen) and lens[nlen, nlen + ndist))
            uint32_t got = 0, prev = 0;
            const uint32_t total = (uint32_t)(nlen + ndist);
            for (;;) {
                const uint32_t code = cl_symbols(b, got, prev, total, lane);    // hand-scheduled; leaves when the input ring needs its next half
                if (code == 0) break;
                if (code == 2) { err = ST_BAD_STREAM; break; }
                // ONE symbol here (the refill brings the next half of the ring in)
                refill(b);
                const uint32_t e = uni(s_cl[(uint32_t)b.bb & ((1u << CL_ROOT) - 1u)]);
                const int nb = (int)(e & 15u), sym = (int)(e >> 16);
                if (nb == 0) { err = ST_BAD_STREAM; break; }
                take(b, nb);
                uint32_t rep = 1, val = (uint32_t)sym;
                if (sym == 16) { if (got == 0) { err = ST_BAD_STREAM; break; } rep = 3 + take(b, 2); val = prev; }
                else if (sym == 17) { rep = 3 + take(b, 3); val = 0; }
                else if (sym == 18) { rep = 11 + take(b, 7); val = 0; }
                if (got + rep > total) { err = ST_BAD_STREAM; break; }
                if (val != 0 && (uint32_t)lane < rep) s_lens[got + (uint32_t)lane] = (uint8_t)val;
                got += rep;
                prev = val;
            }
            if (err != ST_OK) break;
            __syncthreads();
            if (uni(s_lens[256]) == 0) { err = ST_BAD_STREAM; break; }    // no end-of-block code
        }
        if (uni(build_table<5, LL_ROOT>(s_lens, nlen, s_cnt_ll, s_sym_ll, s_ll, K_LITLEN, s_rs) ? 1u : 0u) == 0u) { err = ST_BAD_STREAM; break; }
        if (uni(build_table<1, D_ROOT>(s_lens + nlen, ndist, s_cnt_d, s_sym_d, s_dt, K_DIST, s_rs + 2) ? 1u : 0u) == 0u) { err = ST_BAD_STREAM; break; }

        // ---- symbols: the hot loop.  Everything in it is wave-uniform (scalar registers); per symbol one LDS table look-up
        //      (two for a match), no arithmetic on symbol numbers (the entries carry base and extra-bit count), one compare
        //      for the housekeeping.  A damaged stream does not leave the loop where it is noticed: it sets `bad`, decoding goes
        //      on with harmless values (every LDS index is masked, a far source stays inside the arena) and the next
        //      housekeeping or end-of-block code ends it — one exit keeps the loop's control flow lean.
        //      (Tried and measured no faster: 64 bit offsets looked up speculatively by the lanes per round and the chain
        //      walked with v_readlane — 19 % fewer scalar instructions, 1.21 vs 1.19 ms.  Tried and measured SLOWER: the bit
        //      buffer, the extra-bit fields and the base + extra sums on the vector pipe (v_lshrrev_b64, v_bfe_u32; only the
        //      branch conditions scalar) — 0.92 vs 0.81 ms: with four or five waves per SIMD the longer dependent chains cost
        //      more than the freed scalar issue slots give.  Also slower: code + extra bits leaving the buffer in one shift inside
        //      each branch (the compiler then copies the prefetched input registers at every refill and waits for their load:
        //      1.07 ms), and a `continue` per path (one latch block, re-split on flags).) ---------------------------------------
        uint32_t bad = 0;
        // a match's copy: all lanes; with dist < len the pattern of the last `dist` bytes repeats
        auto copy_match = [&](uint32_t len, uint32_t dist) __attribute__((always_inline)) {
            bad |= op - dist;                                   // (op < 2^17, dist <= 2^15: the sign bit says dist > op)
            if (dist > (uint32_t)NEAR) {
                // beyond the LDS ring: the source lies in a segment that is complete and was flushed right after the
                // symbol that completed it (same wavefront: its stores are ordered before this load)
                const uint8_t *src = out + ((int64_t)op - (int64_t)dist);
#pragma clang loop vectorize(disable) unroll(disable)
                for (uint32_t i = (uint32_t)lane; i < len; i += 64) s_win[(op + i) & WMASK] = src[i];
            } else if (dist >= len) {
#pragma clang loop vectorize(disable) unroll(disable)
                for (uint32_t i = (uint32_t)lane; i < len; i += 64) s_win[(op + i) & WMASK] = s_win[(op + i - dist) & WMASK];
            } else {
                const float inv = 1.0f / (float)dist;
#pragma clang loop vectorize(disable) unroll(disable)
                for (int i = lane; i < (int)len; i += 64) {
                    int qd = (int)((float)i * inv);
                    int r = i - qd * (int)dist;
                    if (r < 0) r += (int)dist;
                    if (r >= (int)dist) r -= (int)dist;
                    s_win[(op + i) & WMASK] = s_win[(op - dist + r) & WMASK];
                }
            }
            op += len;
        };
        auto distance_and_copy = [&](uint32_t len) __attribute__((always_inline)) {
            refill(b);
            uint32_t f = uni(s_dt[(uint32_t)b.bb & ((1u << D_ROOT) - 1u)]);
            if ((f & 15u) == 0) {
                const int sl = long_code(s_cnt_d, s_sym_d, s_rs + 2, (uint32_t)b.bb, D_ROOT);
                f = sl < 0 ? 0u : make_entry(K_DIST, sl & 0xFFFF, sl >> 16);
                if ((f & 15u) == 0) { bad = 0x80000000u; f = 1u | E_BASE | (1u << 16); }
            }
            take(b, (int)(f & 15u));
            const uint32_t dist = (f >> 16) + take(b, (int)((f >> 4) & 15u));    // (<= 13 extra bits: still in the buffer)
            copy_match(len, dist);
        };
        for (;;) {
            uint32_t len, dist;
            const uint32_t code = fast_symbols(b, op, next_evt, lane, out, len, dist);   // the common symbols, hand-scheduled
            if (code == 3) copy_match(len, dist);
            else if (code == 2) distance_and_copy(len);
            else if (code == 1) {
                // ONE symbol of the other kinds (or one that needs the next half of the input ring first)
                refill(b);
                uint32_t e = uni(s_ll[(uint32_t)b.bb & ((1u << LL_ROOT) - 1u)]);
                if ((e & 15u) == 0) {               // a code longer than the root table (a few % of the symbols of a 9-bit table) or none
                    const int sl = long_code(s_cnt_ll, s_sym_ll, s_rs, (uint32_t)b.bb, LL_ROOT);
                    e = sl < 0 ? 0u : make_entry(K_LITLEN, sl & 0xFFFF, sl >> 16);
                    if ((e & 15u) == 0) { bad = 0x80000000u; e = 1u | E_EOB; }
                }
                take(b, (int)(e & 15u));
                if (e & E_LIT) {
                    s_win[op & WMASK] = (uint8_t)(e >> 16);     // (every lane stores the same byte to the same address: no exec juggling)
                    ++op;
                } else if (e & E_BASE) {
                    distance_and_copy(((e >> 20) & 0x1FFu) + take(b, (int)((e >> 16) & 15u)));
                } else {
                    break;                          // end of block (E_EOB)
                }
            }
            if (op >= next_evt) {
                if ((bad >> 31) | (b.idx > idx_end ? 1u : 0u)) break;
                housekeeping();
                if (err != ST_OK) break;
            }
        }
        if ((bad >> 31) | (b.idx > idx_end ? 1u : 0u)) err = ST_BAD_STREAM;
    }
    if (err == ST_OK && op != ulen) err = ST_BAD_LENGTH;
    // the tail: whole segments, then bytes
    if (err == ST_OK) {
        housekeeping();
        const uint32_t rest = op - flushed;
        for (uint32_t i = lane; i < rest; i += 64) out[flushed + i] = s_win[(flushed + i) & WMASK];
    }
    if ((uint32_t)lane < (n_rec & 63u)) slots[(n_rec & ~63u) + (uint32_t)lane] = rec_buf;
    if (lane == 0) {
        a.status[blk] = err;
        a.n_rec[blk] = n_rec;
        // a block that is walked must end on a record boundary, or say by how much its last record runs over
        a.overshoot[blk] = d.entry >= 0 && next_rec < 0xFFFFFFF0u ? (int32_t)(next_rec - ulen) : 0;
    }
}

// ---- bgzf_crc32: the CRC-32 of every block's inflated bytes against the value in the block's trailer (SAM spec §4.1; htslib
// checks it on every block it reads).  One wavefront per block, four blocks per workgroup.  The block is read in rows of
// 1 KiB, coalesced: lane l takes the 16 bytes at l * 16 of every row (the rows are cut from the block's END, so the short row
// comes first).  CRCs are joined zlib's crc32_combine way: "append n zero bytes" is a linear operator on the CRC register (a
// 32 x 32 matrix over GF(2)).  Down its column a lane needs the operator for 1 KiB once per row (four byte-indexed tables in
// LDS, built from the 32 columns of the matrix the host passes); across the lanes the 64 column CRCs are joined pairwise with
// the operators for 16, 32, .. 512 bytes.
struct CrcArgs {
    const uint8_t *file;        // compressed file (for the trailers)
    const uint8_t *out;         // inflated stream
    const BlockDesc *blocks;
    uint32_t *status;           // [n_blocks]: ST_OK -> ST_BAD_CRC on a mismatch (blocks that already failed are skipped)
    int32_t n_blocks;
    uint32_t zeros[6][32];      // zeros[k][i]: the CRC register with only bit i set, after 16 << k zero bytes
    uint32_t zeros1k[32];       // ... after 1024 zero bytes
};

__global__ __launch_bounds__(256) void bgzf_crc32(CrcArgs a)
{
    __shared__ uint32_t s_tab[256];
    __shared__ uint32_t s_row[4][256];                      // s_row[b][v]: the register (v << 8 b), 1 KiB of zeros later
    __shared__ uint32_t s_op[6][32];
    {
        uint32_t c = threadIdx.x;                           // the reflected CRC-32 table (polynomial 0xEDB88320)
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        s_tab[threadIdx.x] = c;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            uint32_t m = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) m ^= a.zeros1k[8 * b + i] & (0u - ((threadIdx.x >> i) & 1u));
            s_row[b][threadIdx.x] = m;
        }
        if (threadIdx.x < 192) s_op[threadIdx.x >> 5][threadIdx.x & 31] = a.zeros[threadIdx.x >> 5][threadIdx.x & 31];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= a.n_blocks) return;
    if (a.status[blk] != ST_OK) return;
    const BlockDesc d = a.blocks[blk];
    const uint8_t *p = a.out + d.uout;                      // 16-byte aligned (the host pads every block's output)
    const int32_t body = (int32_t)(d.ulen & ~15u);          // whole 16-byte pieces; the last ulen % 16 bytes follow at the end
    const int32_t rows = (body + 1023) >> 10;
    // All of it in the CRC's LINEAR form (register starts at 0, no final inversion: then crc(A || B) = later(crc(A), |B|) ^ crc(B)
    // and pieces may be taken in any order); the standard's all-ones start is the same as inverting the block's first four bytes.
    auto crc16 = [&](uint4 v, int32_t at) {
        uint32_t c = at == 0 ? 0xFFFFFFFFu : 0u;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c ^= w[k];
#pragma unroll
            for (int b = 0; b < 4; ++b) c = s_tab[c & 0xFFu] ^ (c >> 8);
        }
        return c;
    };
    auto later_1k = [&](uint32_t c) { return s_row[0][c & 0xFFu] ^ s_row[1][(c >> 8) & 0xFFu] ^ s_row[2][(c >> 16) & 0xFFu] ^ s_row[3][c >> 24]; };
    uint32_t acc = 0;
    int32_t off = body - rows * 1024 + lane * 16;           // this lane's piece of the first row; negative: the row is short and starts later
    int32_t r = 0;
    for (; r + 4 <= rows; r += 4, off += 4096) {            // four rows at a time: four independent look-up chains
        const uint4 z = make_uint4(0, 0, 0, 0);
        const uint4 v0 = off >= 0 ? *reinterpret_cast<const uint4 *>(p + off) : z;
        const uint4 v1 = *reinterpret_cast<const uint4 *>(p + off + 1024);
        const uint4 v2 = *reinterpret_cast<const uint4 *>(p + off + 2048);
        const uint4 v3 = *reinterpret_cast<const uint4 *>(p + off + 3072);
        const uint32_t c0 = off >= 0 ? crc16(v0, off) : 0u, c1 = crc16(v1, off + 1024), c2 = crc16(v2, off + 2048), c3 = crc16(v3, off + 3072);
        acc = later_1k(later_1k(later_1k(later_1k(acc) ^ c0) ^ c1) ^ c2) ^ c3;
    }
    for (; r < rows; ++r, off += 1024)
        acc = later_1k(acc) ^ (off >= 0 ? crc16(*reinterpret_cast<const uint4 *>(p + off), off) : 0u);
    // pairwise across the lanes: a lane that starts a span of 2s columns takes its right neighbour's span (16 s bytes) behind its own
    uint32_t c = acc;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t right = (uint32_t)__shfl_down((int)c, 1 << k, 64);
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) m ^= s_op[k][i] & (0u - ((c >> i) & 1u));
        c = m ^ right;
    }
    if (lane == 0) {
        uint32_t t = body ? c : 0xFFFFFFFFu;                // the register after the body; the last ulen % 16 bytes one by one
        for (uint32_t i = (uint32_t)body; i < d.ulen; ++i) t = s_tab[(t ^ p[i]) & 0xFFu] ^ (t >> 8);
        t = ~t;
        const uint8_t *e = a.file + d.cin + d.clen;         // the block's trailer: CRC32, ISIZE (little endian)
        const uint32_t want = (uint32_t)e[0] | ((uint32_t)e[1] << 8) | ((uint32_t)e[2] << 16) | ((uint32_t)e[3] << 24);
        if (t != want) a.status[blk] = ST_BAD_CRC;
    }
}

// the operators bgzf_crc32 takes: zlib's crc32_combine construction (one zero bit, squared up)
static void crc_zero_operators(uint32_t zeros[6][32], uint32_t zeros1k[32])
{
    auto times = [](const uint32_t *mat, uint32_t vec) { uint32_t s = 0; for (int i = 0; vec; vec >>= 1, ++i) if (vec & 1u) s ^= mat[i]; return s; };
    uint32_t a[32], b[32];
    a[0] = 0xEDB88320u;                                     // one zero BIT
    for (int i = 1; i < 32; ++i) a[i] = 1u << (i - 1);
    uint32_t *cur = a, *nxt = b;
    for (int bits = 1; bits <= 8 * 1024; bits <<= 1) {      // `cur` appends `bits` zero bits
        for (int k = 0; k < 6; ++k)
            if (bits == 8 * (16 << k)) std::memcpy(zeros[k], cur, 32 * sizeof(uint32_t));
        if (bits == 8 * 1024) std::memcpy(zeros1k, cur, 32 * sizeof(uint32_t));
        for (int i = 0; i < 32; ++i) nxt[i] = times(cur, cur[i]);
        std::swap(cur, nxt);
    }
}

// per-block record lists -> dense offsets into the stream; base[b] = exclusive scan of n_rec (done by one workgroup first)
__global__ __launch_bounds__(1024) void rec_scan(const uint32_t *n_rec, uint64_t *base, int32_t n_blocks, unsigned long long *total)
{
    __shared__ unsigned long long s[1024];
    const int t = threadIdx.x;
    const int per = (n_blocks + 1023) / 1024, b0 = t * per, b1 = min(b0 + per, n_blocks);
    unsigned long long sum = 0;
    for (int b = b0; b < b1; ++b) sum += n_rec[b];
    s[t] = sum;
    __syncthreads();
    for (int dd = 1; dd < 1024; dd <<= 1) {
        const unsigned long long x = t >= dd ? s[t - dd] : 0;
        __syncthreads();
        s[t] += x;
        __syncthreads();
    }
    unsigned long long run = s[t] - sum;
    for (int b = b0; b < b1; ++b) { base[b] = run; run += n_rec[b]; }
    if (t == 1023) *total = s[1023];
}

__global__ __launch_bounds__(256) void rec_compact(const BlockDesc *blocks, const uint32_t *rec_slot, const uint32_t *n_rec,
                                                   const uint64_t *base, uint64_t *rec_off)
{
    const int blk = blockIdx.x;
    const uint32_t n = n_rec[blk];
    const uint64_t b0 = base[blk], u0 = blocks[blk].uout;
    const uint32_t *src = rec_slot + (size_t)blk * MAX_REC_PER_BLOCK;
    for (uint32_t i = threadIdx.x; i < n; i += 256) rec_off[b0 + i] = u0 + src[i];
}

inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

} // namespace

// ---- host side ------------------------------------------------------------------------------------------------------------
struct tcmi_bamfile {                           // a BAM file's bytes in pinned host memory + what the host parsed of it
    uint8_t *bytes = nullptr;                   // hipHostMalloc
    size_t n_bytes = 0, cap = 0;                // cap: bytes that go to the device (file + zeroed slack)
    size_t pool_cap = 0;                        // bytes of the pinned allocation
    std::vector<BlockDesc> blocks;
    size_t inflated = 0;                        // bytes of the stream as laid out on the device (blocks padded to 16 bytes)
    size_t tok_total = 0;                       // tokens reserved for all blocks (bgzf_symbols -> bgzf_copy)
    uint32_t pay_dwords = 0;                    // the largest block's payload in dwords + slack (bgzf_symbols' dynamic LDS)
    std::string text;
    std::vector<std::string> ref_name;
    std::vector<int64_t> ref_len;
    std::string path;
};

namespace {
// pinned file buffers are kept for the next file: hipHostMalloc / hipHostFree cost about as much as reading 8 MB
struct PinnedPool {
    std::mutex mu;
    struct Buf { uint8_t *p; size_t cap; };
    std::vector<Buf> free_;
    uint8_t *take(size_t want, size_t *cap)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t k = 0; k < free_.size(); ++k)
                if (free_[k].cap >= want && free_[k].cap <= 2 * want + (1 << 20)) {
                    uint8_t *p = free_[k].p;
                    *cap = free_[k].cap;
                    free_.erase(free_.begin() + (long)k);
                    return p;
                }
        }
        uint8_t *p = nullptr;
        const size_t c = want + want / 8;
        if (hipHostMalloc((void **)&p, c, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        *cap = c;
        return p;
    }
    void give(uint8_t *p, size_t cap)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (free_.size() < 8) { free_.push_back({p, cap}); return; }
        }
        (void)hipHostFree(p);
    }
};
PinnedPool &pinned_pool() { static PinnedPool *p = new PinnedPool(); return *p; }   // (never destroyed: the HIP runtime may be gone by then)
} // namespace

extern "C" {

int tcmi_bamfile_free(tcmi_bamfile *f)
{
    if (!f) return TCMI_OK;
    if (f->bytes) pinned_pool().give(f->bytes, f->pool_cap);
    delete f;
    return TCMI_OK;
}

// Read the file into pinned memory, walk the BGZF block headers (RFC 1952 + the BC subfield) and parse the BAM header
// (inflating, with zlib on this thread, only as many leading blocks as the header occupies).
int tcmi_bamfile_read(const char *path, tcmi_bamfile **out) { return tcmi_bamfile_read_threads(path, 0, out); }

// read_threads: threads that copy the file in (0 = by size: four for a file of several MB, which takes the latency of one file
// from 1.15 to 0.65 ms; a runner that reads several files at a time passes 1 — its reader threads are parallel already, and more
// threads only take cores from the ones that feed the GPU: 42.9 vs 41.7 M positions/s)
int tcmi_bamfile_read_threads(const char *path, int read_threads, tcmi_bamfile **out)
{
    if (!path || !out) return tcmi_fail(nullptr, TCMI_E_ARG, "null argument");
    *out = nullptr;
    static const bool timing = std::getenv("TCMI_READ_TIMING") != nullptr;
    const auto tt0 = std::chrono::steady_clock::now();
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return tcmi_fail(nullptr, TCMI_E_IO, "cannot open %s", path);
    std::fseek(fp, 0, SEEK_END);
    const long sz = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    if (sz < 0) { std::fclose(fp); return tcmi_fail(nullptr, TCMI_E_IO, "cannot size %s", path); }
    tcmi_bamfile *f = new tcmi_bamfile();
    f->path = path;
    f->n_bytes = (size_t)sz;
    f->cap = ((size_t)sz + 4096 + 15) & ~(size_t)15;            // slack: the inflate kernel stages its input 1 KiB at a time
    f->bytes = pinned_pool().take(f->cap, &f->pool_cap);
    if (!f->bytes) {
        std::fclose(fp);
        delete f;
        return tcmi_fail(nullptr, TCMI_E_NOMEM, "hipHostMalloc(%zu) for %s failed (is a GPU present?)", (size_t)sz + 4096, path);
    }
    // The file's bytes into the pinned buffer: a page-cache copy runs at ~7 GB/s per thread, which for a file of several MB is
    // most of what this function costs — so a few threads take a quarter each (pread on the same descriptor).
    const auto tt1 = std::chrono::steady_clock::now();
    size_t got = 0;
    {
        const int fd = fileno(fp);
        static const int forced = std::getenv("TCMI_READ_THREADS") ? std::atoi(std::getenv("TCMI_READ_THREADS")) : 0;   // (A/B measurements)
        const int n_thr = forced > 0 ? std::min(forced, 16) : read_threads > 0 ? std::min(read_threads, 16) : sz > (4l << 20) ? 4 : sz > (1l << 20) ? 2 : 1;
        std::vector<size_t> part((size_t)n_thr, 0);
        auto piece = [&](int t) {
            const size_t lo = (size_t)sz * (size_t)t / (size_t)n_thr, hi = (size_t)sz * (size_t)(t + 1) / (size_t)n_thr;
            size_t at = lo;
            while (at < hi) {
                const ssize_t r = pread(fd, f->bytes + at, hi - at, (off_t)at);
                if (r <= 0) break;
                at += (size_t)r;
            }
            part[(size_t)t] = at - lo;
        };
        std::vector<std::thread> thr;
        for (int t = 1; t < n_thr; ++t) thr.emplace_back(piece, t);
        piece(0);
        for (auto &t : thr) t.join();
        for (size_t p : part) got += p;
    }
    const auto tt2 = std::chrono::steady_clock::now();
    std::fclose(fp);
    std::memset(f->bytes + f->n_bytes, 0, f->cap - f->n_bytes);
    auto bail = [&](int code, const char *what, size_t at) {
        tcmi_bamfile_free(f);
        return tcmi_fail(nullptr, code, "%s: %s at byte %zu", path, what, at);
    };
    if (got != (size_t)sz) return bail(TCMI_E_IO, "short read", got);
    // ---- block headers ----
    static const bool prefetch_ahead = std::getenv("TCMI_NO_HEADER_PREFETCH") == nullptr;
    size_t off = 0, uout = 0;
    while (off < f->n_bytes) {
        if (f->n_bytes - off < 18) return bail(TCMI_E_FORMAT, "truncated BGZF block header", off);
        const uint8_t *h = f->bytes + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return bail(TCMI_E_FORMAT, "not a BGZF block (is the file a BAM?)", off);
        const size_t xlen = rd16(h + 10);
        if (f->n_bytes - off < 12 + xlen) return bail(TCMI_E_FORMAT, "truncated BGZF extra field", off);
        size_t bsize = 0;
        for (size_t x = 0; x + 4 <= xlen;) {
            const uint8_t *s = h + 12 + x;
            const size_t slen = rd16(s + 2);
            if (s[0] == 'B' && s[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (size_t)rd16(s + 4) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || f->n_bytes - off < bsize) return bail(TCMI_E_FORMAT, "bad BGZF block size", off);
        // (the chain of headers is a chain of cache misses once other threads have copied the file in — each header lies in some
        //  other core's cache or in memory: ask for the lines where the block after next will probably start; blocks of one file
        //  are of similar size)
        if (prefetch_ahead) {
            const size_t guess = off + 3 * bsize;
            if (guess + 512 < f->n_bytes && guess > 512)
                for (size_t x = guess - 384; x < guess + 384; x += 64) __builtin_prefetch(f->bytes + x, 0, 1);
        }
        BlockDesc b;
        b.cin = off + 12 + xlen;
        b.clen = (uint32_t)(bsize - 12 - xlen - 8);
        b.ulen = rd32(h + bsize - 4);
        b.uout = uout;
        b.entry = 0;
        if (b.ulen > 65536) return bail(TCMI_E_FORMAT, "BGZF block inflates to more than 64 KiB", off);
        // tokens: one per literal / match (each gives >= 1 byte and takes >= 1 bit), one per <= 8 191 stored bytes (a stored
        // deflate block takes >= 5 bytes)
        b.tok_cap = std::min(b.ulen, 8u * b.clen) + b.clen / 2 + 8;
        b.tok = f->tok_total;
        f->tok_total += (2u * b.tok_cap + 3u) & ~3u;        // (as many again behind them: bgzf_symbols' scratch)
        f->pay_dwords = std::max(f->pay_dwords, (uint32_t)(((b.cin & 3u) * 8u + b.clen * 8u + 31u) / 32u + 6u));
        uout += ((size_t)b.ulen + 15) & ~(size_t)15;            // every block's output starts 16-byte aligned on the device
        off += bsize;
        f->blocks.push_back(b);
    }
    f->inflated = uout;
    const auto tt3 = std::chrono::steady_clock::now();
    // ---- BAM header: inflate leading blocks on this thread until it is complete ----
    std::vector<uint8_t> head;
    size_t nb = 0;
    auto more = [&]() -> bool {
        if (nb >= f->blocks.size()) return false;
        const BlockDesc &b = f->blocks[nb];
        const size_t at = head.size();
        head.resize(at + b.ulen);
        if (b.ulen) {
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) return false;
            zs.next_in = f->bytes + b.cin; zs.avail_in = b.clen;
            zs.next_out = head.data() + at; zs.avail_out = b.ulen;
            const int rc = inflate(&zs, Z_FINISH);
            const bool ok = rc == Z_STREAM_END && zs.total_out == b.ulen;
            inflateEnd(&zs);
            if (!ok) return false;
        }
        ++nb;
        return true;
    };
    auto need = [&](size_t k) { while (head.size() < k) if (!more()) return false; return true; };
    if (!need(12) || std::memcmp(head.data(), "BAM\1", 4) != 0) return bail(TCMI_E_FORMAT, "BAM magic missing", 0);
    const size_t l_text = rd32(head.data() + 4);
    if (!need(12 + l_text)) return bail(TCMI_E_FORMAT, "truncated header text", 8);
    f->text.assign((const char *)head.data() + 8, l_text);
    size_t o = 8 + l_text;
    const size_t n_ref = rd32(head.data() + o);
    o += 4;
    for (size_t r = 0; r < n_ref; ++r) {
        if (!need(o + 4)) return bail(TCMI_E_FORMAT, "truncated reference list", o);
        const size_t l_name = rd32(head.data() + o);
        o += 4;
        if (!need(o + l_name + 4)) return bail(TCMI_E_FORMAT, "truncated reference name", o);
        f->ref_name.emplace_back((const char *)head.data() + o, l_name ? l_name - 1 : 0);
        o += l_name;
        f->ref_len.push_back((int64_t)rd32(head.data() + o));
        o += 4;
    }
    // records start `o` bytes into the stream: in block k at offset o - (inflated bytes of the blocks before it)
    size_t before = 0;
    size_t k = 0;
    for (; k < f->blocks.size(); ++k) {
        if (o < before + f->blocks[k].ulen) break;
        f->blocks[k].entry = -1;                                  // header only (or empty)
        before += f->blocks[k].ulen;
    }
    if (k < f->blocks.size()) f->blocks[k].entry = (int32_t)(o - before);
    if (timing) {
        const auto tt4 = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        std::fprintf(stderr, "[tcmi] bamfile_read %s: open + pinned buffer %ld us, read %ld us, block table %ld us, header %ld us\n", path, us(tt0, tt1), us(tt1, tt2), us(tt2, tt3), us(tt3, tt4));
    }
    *out = f;
    return TCMI_OK;
}

int tcmi_bamfile_info(const tcmi_bamfile *f, int64_t *file_bytes, int64_t *inflated_bytes, int64_t *n_blocks, int32_t *n_ref,
                      const char **ref0_name, int64_t *ref0_len)
{
    if (!f) return tcmi_fail(nullptr, TCMI_E_ARG, "bamfile is NULL");
    if (file_bytes) *file_bytes = (int64_t)f->n_bytes;
    if (inflated_bytes) { int64_t s = 0; for (const auto &b : f->blocks) s += b.ulen; *inflated_bytes = s; }
    if (n_blocks) *n_blocks = (int64_t)f->blocks.size();
    if (n_ref) *n_ref = (int32_t)f->ref_name.size();
    if (ref0_name) *ref0_name = f->ref_name.empty() ? "" : f->ref_name[0].c_str();
    if (ref0_len) *ref0_len = f->ref_len.empty() ? 0 : f->ref_len[0];
    return TCMI_OK;
}

const char *tcmi_bamfile_text(const tcmi_bamfile *f) { return f ? f->text.c_str() : ""; }

} // extern "C"

// ---- device decode: H2D of the compressed file, bgzf_inflate, chain check, dense record offsets (all in the context's arena) ----
namespace {
struct DeviceBam { uint8_t *d_out = nullptr; uint64_t *d_rec = nullptr; BlockDesc *d_desc = nullptr; size_t n = 0; };

int decode_on_device(tcmi_ctx *ctx, const tcmi_bamfile *f, DeviceBam *D)
{
    const size_t nb = f->blocks.size();
    if (nb == 0) return tcmi_fail(ctx, TCMI_E_FORMAT, "%s: no BGZF blocks", f->path.c_str());
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // bounds on the records for the arena: a record takes at least 36 bytes of the stream; reserve for records of >= 64 bytes
    // (block_size + 32 fixed bytes + name + CIGAR + SEQ + QUAL of a 15-base read) — the arena cannot grow under live data
    const size_t max_rec = f->inflated / 36 + 16;
    const size_t guess_rec = std::min(max_rec, f->inflated / 64 + 1024);
    static const bool legacy = std::getenv("TCMI_INFLATE_LEGACY") != nullptr;       // (A/B: the one-kernel decoder)
    const size_t b_file = al(f->cap), b_desc = al(nb * sizeof(BlockDesc)), b_out = al(f->inflated + 128),
                 b_slot = al(nb * (size_t)MAX_REC_PER_BLOCK * 4), b_small = al(nb * 4) * 4 + al(nb * 8) + al(nb * 512) + 256,
                 b_tok = legacy ? 0 : al(f->tok_total * 4 + 256);
    const size_t b_rest = al(guess_rec * 8 + 8) + al(guess_rec * 4 + 4) * 9 + al((guess_rec / 256 + 2) * 8) * 3 + 8192 + 20 * 256;
    if (!tcmi_arena_reserve_take(ctx, b_file + b_desc + b_out + b_slot + b_small + b_tok + b_rest + 16 * 256, 0)) return TCMI_E_NOMEM;
    uint8_t *d_file = (uint8_t *)tcmi_arena_take(ctx, b_file);
    BlockDesc *d_desc = (BlockDesc *)tcmi_arena_take(ctx, b_desc);
    uint8_t *d_out = (uint8_t *)tcmi_arena_take(ctx, b_out);
    uint32_t *d_slot = (uint32_t *)tcmi_arena_take(ctx, b_slot);
    uint32_t *d_nrec = (uint32_t *)tcmi_arena_take(ctx, al(nb * 4));
    int32_t *d_over = (int32_t *)tcmi_arena_take(ctx, al(nb * 4));
    uint32_t *d_stat = (uint32_t *)tcmi_arena_take(ctx, al(nb * 4));
    uint64_t *d_base = (uint64_t *)tcmi_arena_take(ctx, al(nb * 8));
    unsigned long long *d_total = (unsigned long long *)tcmi_arena_take(ctx, 256);
    uint32_t *d_ntok = (uint32_t *)tcmi_arena_take(ctx, al(nb * 4));
    uint32_t *d_seg = (uint32_t *)tcmi_arena_take(ctx, al(nb * 512));
    uint32_t *d_tok = legacy ? nullptr : (uint32_t *)tcmi_arena_take(ctx, b_tok);

    TCMI_HIP(ctx, hipMemcpyAsync(d_file, f->bytes, f->cap, hipMemcpyHostToDevice, ctx->stream));
    TCMI_HIP(ctx, hipMemcpyAsync(d_desc, f->blocks.data(), nb * sizeof(BlockDesc), hipMemcpyHostToDevice, ctx->stream));
    InflateArgs a;
    a.file32 = reinterpret_cast<const uint32_t *>(d_file);
    a.blocks = d_desc; a.out = d_out; a.rec_slot = d_slot; a.n_rec = d_nrec; a.overshoot = d_over; a.status = d_stat;
    a.n_blocks = (int32_t)nb;
    (void)hipGetLastError();
    static const bool occ_once = [] {
        if (std::getenv("TCMI_INFLATE_OCCUPANCY")) {             // diagnostic: resident wavefronts of bgzf_inflate per CU
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bgzf_inflate, 64, 0) == hipSuccess)
                std::fprintf(stderr, "[tcmi] bgzf_inflate: %d wavefronts per CU\n", n);
        }
        return true;
    }();
    (void)occ_once;
    if (legacy) {
        tcmi_prof_begin(ctx, TCMI_K_INFLATE);
        hipLaunchKernelGGL(bgzf_inflate, dim3((unsigned)nb), dim3(64), 0, ctx->stream, a);
        tcmi_prof_end(ctx, TCMI_K_INFLATE);
        TCMI_HIP(ctx, hipGetLastError());
    } else {
        tcmi_bgzf_decode_args g;
        g.d_file = d_file; g.d_desc = d_desc; g.d_tok = d_tok; g.d_ntok = d_ntok; g.d_seg = d_seg; g.d_out = d_out; g.d_slot = d_slot; g.d_nrec = d_nrec;
        g.d_over = d_over; g.d_stat = d_stat; g.n_blocks = nb; g.pay_dwords = f->pay_dwords;
        const int rc = tcmi_bgzf_decode_launch(ctx, g);
        if (rc) return rc;
    }
    if (ctx->verify_crc) {
        static const CrcArgs proto = [] { CrcArgs c = {}; crc_zero_operators(c.zeros, c.zeros1k); return c; }();
        CrcArgs c = proto;
        c.file = d_file; c.out = d_out; c.blocks = d_desc; c.status = d_stat; c.n_blocks = (int32_t)nb;
        tcmi_prof_begin(ctx, TCMI_K_CRC);
        hipLaunchKernelGGL(bgzf_crc32, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, ctx->stream, c);
        tcmi_prof_end(ctx, TCMI_K_CRC);
        TCMI_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(rec_scan, dim3(1), dim3(1024), 0, ctx->stream, d_nrec, d_base, (int32_t)nb, d_total);
    TCMI_HIP(ctx, hipGetLastError());
    // the verdict of every block comes back to the host: a few bytes per block
    std::vector<uint32_t> stat(nb);
    std::vector<int32_t> over(nb);
    unsigned long long total = 0;
    TCMI_HIP(ctx, hipMemcpyAsync(stat.data(), d_stat, nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    TCMI_HIP(ctx, hipMemcpyAsync(over.data(), d_over, nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    TCMI_HIP(ctx, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t b = 0; b < nb; ++b)
        if (stat[b] == ST_BAD_STREAM || stat[b] == ST_BAD_LENGTH)
            return tcmi_fail(ctx, TCMI_E_FORMAT, "%s: BGZF block %zu failed to inflate (deflate stream or ISIZE damaged)", f->path.c_str(), b);
    for (size_t b = 0; b < nb; ++b)
        if (stat[b] == ST_BAD_CRC)
            return tcmi_fail(ctx, TCMI_E_FORMAT, "%s: CRC32 mismatch in BGZF block %zu", f->path.c_str(), b);
    // The record chain: every block was walked from offset 0 on the assumption that its predecessor ends on a record
    // boundary.  In block order that assumption holds by induction up to the first block that runs over, so a bad
    // record before that point is real, and anything after it is not to be trusted.
    for (size_t b = 0; b < nb; ++b) {
        if (stat[b] == ST_BAD_RECORD)
            return tcmi_fail(ctx, TCMI_E_FORMAT, "%s: alignment record with an impossible block_size in BGZF block %zu", f->path.c_str(), b);
        if (over[b] != 0)
            return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "%s: a record straddles BGZF blocks %zu / %zu (the file was not written the htslib way): host reader",
                             f->path.c_str(), b, b + 1);
    }
    if (total > max_rec) return tcmi_fail(ctx, TCMI_E_FORMAT, "%s: impossible record count", f->path.c_str());
    const size_t n = (size_t)total;
    const size_t need_rest = al(n * 8 + 8) + al(n * 4 + 4) * 9 + al((n / 256 + 2) * 8) * 3 + 8192 + 20 * 256;
    if (need_rest > b_rest)
        return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "%s: %zu very short records need more device scratch than was reserved: host reader", f->path.c_str(), n);
    uint64_t *d_rec = (uint64_t *)tcmi_arena_take(ctx, al(n * 8 + 8));
    tcmi_prof_begin(ctx, TCMI_K_RECORDS);
    if (n) hipLaunchKernelGGL(rec_compact, dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_desc, d_slot, d_nrec, d_base, d_rec);
    tcmi_prof_end(ctx, TCMI_K_RECORDS);
    TCMI_HIP(ctx, hipGetLastError());
    D->d_out = d_out; D->d_rec = d_rec; D->d_desc = d_desc; D->n = n;
    return TCMI_OK;
}
} // namespace

extern "C" {

// BAM bytes -> read set in HBM, everything on the device: H2D of the compressed file, bgzf_inflate, record index,
// pack_device.hip.  TCMI_E_UNSUPPORTED when the file needs the host reader (records that straddle BGZF blocks, entries
// longer than 512 positions, ...): the caller falls back to tcmi_bam_load + tcmi_readset_upload.
int tcmi_readset_from_bamfile(tcmi_ctx *ctx, const tcmi_bamfile *f, tcmi_readset **out, int64_t *n_reads_out)
{
    if (!ctx || !f || !out) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    *out = nullptr;
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    static const bool timing = std::getenv("TCMI_UPLOAD_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    DeviceBam D;
    int rc = decode_on_device(ctx, f, &D);
    if (rc) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    if (n_reads_out) *n_reads_out = (int64_t)D.n;
    tcmi_pack_src s = {};
    s.stream = D.d_out; s.rec_off = D.d_rec; s.n = (int64_t)D.n; s.mode = 1; s.pos_shift = 0;
    static std::atomic<uint64_t> next_uid{1ull << 40};
    tcmi_readset *rs = new tcmi_readset();
    rs->uid = next_uid.fetch_add(1);
    rs->n_reads = (int64_t)D.n;
    rs->device = ctx->device;
    uint32_t why = 0;
    rc = D.n ? tcmi_pack_on_device(ctx, &s, rs, &why) : TCMI_OK;
    if (rc == TCMI_OK) {
        rs->packed_on_device = 2;               // decoded AND packed on the device
        *out = rs;
        if (timing) {
            const auto t2 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[tcmi bamfile] decode on device %.2f ms, pack %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                         std::chrono::duration<double, std::milli>(t2 - t1).count());
        }
        return TCMI_OK;
    }
    tcmi_readset_free(ctx, rs);
    if (rc == TCMI_E_UNSUPPORTED)
        return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "%s: the device packer declined (flags 0x%x: long reads, far positions, a second reference or malformed reads): host reader",
                         f->path.c_str(), why);
    return rc;
}

// For tests and tools: the device-inflated stream (contiguous, as the file inflates) and the record offsets, back on the host.
// stream_cap >= inflated bytes (tcmi_bamfile_info); rec_cap entries of rec_off.
int tcmi_bamfile_decode_to_host(tcmi_ctx *ctx, const tcmi_bamfile *f, uint8_t *stream, int64_t stream_cap, uint64_t *rec_off,
                                int64_t rec_cap, int64_t *n_rec)
{
    if (!ctx || !f || !stream || !n_rec) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    DeviceBam D;
    int rc = decode_on_device(ctx, f, &D);
    if (rc) return rc;
    *n_rec = (int64_t)D.n;
    if ((int64_t)D.n > rec_cap && rec_off) return tcmi_fail(ctx, TCMI_E_ARG, "rec_off holds %lld entries, the file has %zu records", (long long)rec_cap, D.n);
    int64_t at = 0;
    std::vector<uint64_t> rec(D.n);
    if (D.n) TCMI_HIP(ctx, hipMemcpyAsync(rec.data(), D.d_rec, D.n * 8, hipMemcpyDeviceToHost, ctx->stream));
    for (const BlockDesc &b : f->blocks) {      // the device keeps every block's output 16-byte aligned: close the gaps
        if (at + (int64_t)b.ulen > stream_cap) return tcmi_fail(ctx, TCMI_E_ARG, "stream buffer too small");
        if (b.ulen) TCMI_HIP(ctx, hipMemcpyAsync(stream + at, D.d_out + b.uout, b.ulen, hipMemcpyDeviceToHost, ctx->stream));
        at += b.ulen;
    }
    TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (rec_off) {
        // device offsets are in the padded layout: map them back to offsets in the contiguous stream
        size_t k = 0;
        int64_t contiguous = 0;
        for (size_t i = 0; i < D.n; ++i) {
            while (k + 1 < f->blocks.size() && rec[i] >= f->blocks[k].uout + (((uint64_t)f->blocks[k].ulen + 15) & ~15ull)) { contiguous += f->blocks[k].ulen; ++k; }
            rec_off[i] = (uint64_t)contiguous + (rec[i] - f->blocks[k].uout);
        }
    }
    return TCMI_OK;
}

} // extern "C"
