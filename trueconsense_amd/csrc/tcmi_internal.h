// tcmi_internal.h — shared between the translation units of libtcmi.so (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <functional>
#include <vector>

#include "../../include/tcmi.h"

// ---- device read layout -------------------------------------------------------------
// Only reads that pile up (mapped, tid == 0, pos >= 0, reference span > 0)
// are kept, in two sets:
//
//  * ALIGNED set (tally_planes.hip) — every read of every BASELINE config.  A read whose CIGAR is one run
//    of match ops (M / = / X, optionally flanked by S / H clips) is taken as it is; any other CIGAR is
//    PROJECTED onto the reference while it is packed: matched bases land on their reference offset,
//    deleted / skipped positions stay empty, inserted and clipped bases are dropped, and the tokens that
//    are not plain bases ("*", "..+n..") become EVENT words (position | kind) that the tail blocks of the
//    same launch count.  Per entry ONE packed header word (position - window start | len << 10 | pair
//    offset from the stage's first word << 20) and the bases as codes A=0 C=1 G=2 T=3 (anything else 0),
//    32 bases per pair of words {lo plane, hi plane}, one zero pair in front of every read and behind the
//    last of a chunk: 4 + 8*ceil(l/32) + 8 bytes, 52 for a 150-bp read.
//    "Anything else" (N, IUPAC, '=', base beyond SEQ, deleted / skipped positions) is exactly what
//    indexing.py:115-132 puts in no class; those positions are listed as OTHER event words (they
//    count toward coverage but toward no class).
//    Consecutive reads are grouped into CHUNKS (window <= TCMI_F_MAXW grid words of 8 positions,
//    <= 255 reads per lane); one workgroup tallies one chunk in STAGES of <= sub_reads reads.
//    Packed on the DEVICE from the BAM-native arrays (pack_device.hip: sorted input, entries of
//    <= TCMI_D_MAXLEN positions) or on the HOST (readset.cpp: anything, long reads in pieces of
//    TCMI_F_SEG positions, re-sorted).
//  * GENERAL set (CIGAR-walk kernel, tally.hip): what the aligned path does not take (positions >= 2^29,
//    reads with indels under option project_reads = 0, or everything under option tally_variant = 1;
//    the tests use these to cross-check independent implementations), in ROUNDS of TCMI_ROUND reads with
//    per-round offset tables, raw 4-bit codes.
//
// The algorithmic bytes of SURVEY 8-d are 12 + 4*n_cigar + ceil(l/2) per read: 91 for a 150-bp read.
#define TCMI_ROUND 256
#ifndef TCMI_F_BLOCK
#define TCMI_F_BLOCK 256           // lanes per workgroup of the fast kernel (256 or 512; 256 measured faster)
#endif
#define TCMI_F_MAXW 96             // max grid words (8 positions each) in a chunk window
#define TCMI_F_MAXSPAN 600         // longest aligned read the fast kernel takes in one piece
#define TCMI_F_SEG 512             // projected reads longer than this are cut into pieces of this many positions
#define TCMI_D_MAXLEN 512          // longest entry the device packer takes (a window holds MAXW * 8 = 768 positions)
#ifndef TCMI_F_SEQCAP
#define TCMI_F_SEQCAP 6144         // LDS words for staged bases
#endif
#define TCMI_F_MAXSTAGE 8          // stages per chunk
#ifndef TCMI_P_NPL
#define TCMI_P_NPL 8               // counter planes per lane: a lane counts <= 2^NPL - 1 reads per chunk
#endif
#ifndef TCMI_P_WAVES
#define TCMI_P_WAVES 4             // workgroups per CU the kernel's register budget is set for
#endif
#ifndef TCMI_P_SUB
#define TCMI_P_SUB 512             // max reads staged in LDS at a time
#endif
// event word = reference position | kind; kinds may be combined
#define TCMI_F_EVPOS   (1u << 29)  // positions must stay below this for the fast path
#define TCMI_F_EV_OTHER (1u << 29) // a covered position whose token is no A/C/G/T base: was counted as T by subtraction
#define TCMI_F_EV_X     (1u << 30) // token "*"
#define TCMI_F_EV_I     (1u << 31) // token carries an insertion

struct tcmi_fast_chunk {           // 80 bytes
    int64_t read0;                 // first read (index into f_pos / f_lenoff)
    int64_t word0;                 // first word of the chunk's base stream (multiple of 4)
    int32_t n_reads;
    int32_t P0;                    // window start, multiple of 8
    int32_t Wn;                    // window length in grid words
    int32_t sub_reads;             // reads per stage (<= TCMI_P_SUB)
    int32_t stage_end[TCMI_F_MAXSTAGE];   // word offset (from word0) one past stage i, trailing pad included;
                                          // stage i starts at stage_end[i-1] - pad (0 for i = 0)
    // the chunk's coverage as runs of reads with equal (position, length), words of d_fcovrun:
    // position - P0 | len << 10 | (reads in the run, <= 4095) << 20
    int64_t run0;
    int32_t n_runs;
    int32_t reserved_;
};

struct tcmi_readset {
    uint64_t uid = 0;           // unique per upload (graphs are cached against it, not the pointer)
    int64_t n_reads = 0;        // as handed in
    int64_t n_piled = 0;        // kept on device (aligned + general)
    int64_t alg_bytes = 0;      // sum over kept reads of 12 + 4*n_cigar + ceil(l_qseq/2)
    int64_t dev_bytes = 0;
    int64_t max_end = 0;        // max end position (exclusive) of a kept read
    int32_t max_len = 0;        // device-packed sets: the longest reference span of a kept read (0: not recorded)
    int device = -1;
    int packed_on_device = 0;   // 1: pack_device.hip built the aligned set (everything below lives in d_blob)
    // device-decoded read sets: the inflated stream and the packer's index stay in the context's arena until its next upload
    // (arena_epoch tells): the insert-token kernel reads qualities, inserted bases, names and mate fields from there
    const uint8_t *d_stream = nullptr;
    const uint64_t *d_rec_off = nullptr;
    const uint32_t *d_cidx = nullptr;
    const int32_t *d_cpos = nullptr;
    const uint32_t *d_gen_idx = nullptr;   // records of reads too long for the packed set (s_reads of them): tally_stream_kernel walks them in the stream
    int64_t s_reads = 0;
    // a read set of a block RANGE of a file (tcmi_readset_from_bamfile_blocks): where its first record starts when the range began
    // in the middle of the file (nothing in front of it vouches for that start), and where the first record behind the range starts —
    // offsets into the WHOLE file's inflated stream, -1: none.  The range in front must end where this one starts
    // (tcmi_readset_range_anchors; tcmi_split_step sums the differences along with the counts).
    int64_t range_first = -1, range_next = -1;
    uint64_t arena_epoch = 0;
    // the one-sync file path (bam_device.hip, tcmi_bamfile_step): the packer's totals have not been read back yet — f_chunks and
    // f_events hold the CAPACITIES, the tally kernel takes the real counts from here ({n_chunks, n_events} in the context's arena)
    const uint32_t *d_dev_counts = nullptr;
    char *d_blob = nullptr;     // one allocation holding d_flenoff | d_fseq | d_fchunk | d_fcovrun | d_fevent
    size_t blob_bytes = 0;
    // aligned set
    int64_t f_reads = 0, f_chunks = 0, f_words = 0, f_events = 0;
    uint32_t *d_flenoff = nullptr; // [f_reads] the packed header words
    uint32_t *d_fseq = nullptr; // [f_words]
    uint32_t *d_fevent = nullptr;// [f_events] position | TCMI_F_EV_*: tokens that are not plain A/C/G/T bases
    tcmi_fast_chunk *d_fchunk = nullptr;   // [f_chunks]
    uint32_t *d_fcovrun = nullptr;         // coverage runs of all chunks (tcmi_fast_chunk::run0 / n_runs)
    // general set
    int64_t g_reads = 0, n_rounds = 0, n_cigar = 0, n_seqw = 0;
    int32_t *d_pos = nullptr;   // [g_reads]
    uint32_t *d_meta = nullptr; // [g_reads] flag<<16 | n_cigar
    int32_t *d_lseq = nullptr;  // [g_reads]
    uint32_t *d_cigar = nullptr;// [n_cigar]
    uint32_t *d_seq = nullptr;  // [n_seqw] 8 bases per word, base i at bits [4(i%8), +4), raw BAM codes
    int64_t *d_round_cig = nullptr; // [n_rounds+1]
    int64_t *d_round_seq = nullptr; // [n_rounds+1]
    // A rank's block range decoded as several SUB-RANGES side by side (tcmi_split_step, "split_sub"): this read set is then only the
    // sum of its parts — each a read set of its own on a context of its own (its decoded stream lives in that context's arena) — in
    // file order.  It answers tcmi_readset_info / _range_anchors / _ins_entries / tcmi_tally_dev / tcmi_readset_free.
    struct Part { tcmi_ctx *cx; tcmi_readset *rs; };
    std::vector<Part> parts;
};

struct tcmi_ride {                  // a finished matrix waiting for its call (see tally_common.h, call_other_tile)
    int32_t *counts; int64_t ld, L; int32_t mincov; int amb; uint8_t *plain, *alt, *flags; bool taken;
};

struct tcmi_upload_scratch;              // host buffers of tcmi_readset_upload, kept between calls (readset.cpp)
void tcmi_upload_scratch_free(tcmi_upload_scratch *s);
struct tcmi_dev_arena;                   // grow-only device scratch of the device packer / BAM decoder (pack_device.hip)
void tcmi_dev_arena_free(tcmi_dev_arena *a);

struct tcmi_ctx {
    int device = -1;
    tcmi_upload_scratch *upload_scratch = nullptr;
    tcmi_dev_arena *dev_arena = nullptr;
    uint64_t arena_epoch = 0;        // bumped whenever the arena is handed out anew
    // device-packed read sets hand their allocation back when they are freed; the next upload of a similar size takes it
    // (hipMalloc + hipFree cost more than the pack kernels, and hipFree waits for the device)
    struct Blob { char *p; size_t bytes; };
    std::vector<Blob> blob_pool;
    // scratch of tcmi_readset_modal_tokens (columns, ranges, entries): device + pinned host, grow-only
    char *tok_dev = nullptr, *tok_host = nullptr;
    size_t tok_dev_cap = 0, tok_host_cap = 0;
    // small read-backs of the cold path (block verdicts, packer totals) land in pinned memory of the context: a copy into pageable
    // memory makes the runtime pin and unpin the destination's pages on every call
    char *h_pin = nullptr;
    size_t h_pin_cap = 0;
    char *h_desc = nullptr;          // pinned: the re-based block table of a block range on its way to the device (bam_device.hip: decode_enqueue)
    size_t h_desc_cap = 0;
    int verify_crc = 1;              // the device decoder checks the BGZF CRC-32 of every block
    int64_t stat_one_sync_taken = 0, stat_one_sync_declined = 0, stat_last_decline = 0;     // tcmi_ctx_stat
    int64_t stat_h2d_piped = 0;      // decodes whose compressed bytes crossed PCIe in pieces, ahead of the inflate kernels (bam_device.hip: decode_enqueue)
    int64_t stat_split_sub = 0;      // tcmi_split_step calls whose range went through sub-ranges
    int64_t stat_decode_batched = 0; // files the device decoder took in batches of blocks (tcmi_ctx_stat "decode_batched")
    int32_t split_tail[6] = {0, 0, 0, 0, 0, 0};   // tcmi_split_step: this rank's slot of the range table on its way to the device (the copy is asynchronous)
    uint32_t rec_bytes_seen = 0;     // mean record size of the last file this context decoded (sizes the next file's arrays when the file's own first blocks say nothing)
    int64_t decode_token_mb = 4096;  // bam_device.hip decode_enqueue: the token scratch's size; files that need more are decoded in batches of blocks
    int prefix_kernels = 0;          // pk_index / pk_place take the sums in front of a block from scan launches: 0 = from 16 384 blocks on, 1 = always, -1 = never
    int mid_wait = 0;                // 1: the one-sync path waits once more, behind the decode kernels (bam_device.hip: fast_enqueue); default since round 5: ONE wait per file
    int one_sync = 1;                // device-decoded files take the one-sync path (pk_index + pk_place + pk_pack) first; 0: the several-kernel path only
    int device_pack = 1;             // tcmi_readset_upload packs on the device when the input allows it
    int n_cu = 256;                  // compute units of the device
    int wg_per_cu = TCMI_P_WAVES;    // resident workgroups per CU of the tally kernel (its register budget)
    int balance_chunks = 1;          // size the chunks so that their number is a multiple of the resident workgroups
    hipStream_t stream = nullptr;
    bool own_stream = true;
    // Two streams per context (TCMI_STREAM_SPLIT, api.cpp ctx_create): the inflate kernels of a file stay on `stream_lo`, everything
    // behind them (pk_index .. pk_report: short kernels that triple in duration next to a chip full of bgzf_copy workgroups) goes to
    // `stream_hi` — a higher priority, or compute units of its own.  `stream` is the one the launches use at the moment.
    // tcmi_split_step: a rank's block range is decoded as `split_sub` sub-ranges side by side (0 = auto: a true range of a larger file: three from 6 144 blocks on, two
    // from 4 096; the whole file: one), the first on this context, the others on helper contexts this one owns (a stream and an arena each)
    int split_sub = 0;
    std::vector<tcmi_ctx *> helpers;
    // ... started one behind the other: a sub-range's first inflate kernel waits for the bgzf_symbols of the sub-range in front (an event), so
    // that the sub-ranges run skewed — symbols of k + 1 under copy of k under pack of k - 1 — instead of in step (bgzf_decode.hip:
    // tcmi_bgzf_decode_launch records `ev_after_sym` and calls `after_sym` once per decode)
    int sym_scratch_div = 1;         // option "sym_scratch_div" (tests): bgzf_symbols' lanes park 1 / n of their share before they overflow (pass B)
    int h2d_pieces = 0;              // option "h2d_pieces": 0 / -1 = one copy (default), n = n pieces
    hipStream_t copy_stream = nullptr;           // H2D of a large file's / range's compressed bytes in pieces (decode_enqueue)
    std::vector<hipEvent_t> ev_piece;
    hipEvent_t ev_before_sym = nullptr, ev_after_sym = nullptr;
    std::function<void()> after_sym;
    hipStream_t stream_lo = nullptr, stream_hi = nullptr;
    hipEvent_t ev_split = nullptr;
    hipEvent_t step_done = nullptr;  // recorded at the end of tcmi_step_begin
    std::string err;
    // profiling
    bool prof = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    struct Pending { int k; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[TCMI_K_NKERNELS] = {};
    int64_t prof_n[TCMI_K_NKERNELS] = {};
    // workspace of tcmi_step / host-buffer conveniences
    int64_t ws_L = 0, ws_ld = 0;
    int32_t *d_counts = nullptr;
    uint8_t *d_plain = nullptr, *d_alt = nullptr, *d_flags = nullptr;
    uint8_t *h_rec = nullptr;       // pinned: plain | alt | flags, each ws_ld bytes
    int32_t *h_counts = nullptr;    // pinned [7][ws_ld]
    int64_t step_L = 0;             // > 0 between tcmi_step_begin and tcmi_step_end
    bool step_counts = false;
    bool counts_clean = false;      // the workspace matrix was left zeroed by the last call kernel
    // ride-along call (pipeline): the call of this context's step has not been launched yet; the next step on the
    // stream (another workspace) carries it in its tally launch (tcmi_step_begin_deferred / tcmi_step_flush)
    int defer_call = 1;             // pipeline: the call of step k rides in the tally launch of step k + 1
    bool call_pending = false;
    int64_t pend_L = 0;
    int32_t pend_mincov = 0;
    int pend_amb = 0;
    struct tcmi_ride *ride = nullptr;            // set around a tally launch that may carry another context's call
    int prof_every = 1;             // tcmi_step_begin: every n-th step is launched directly and bracketed with events
    int64_t step_tick = 0;
    bool prof_open = false, prof_mute = false;
    int tally_variant = 0;          // 0 = aligned reads through the fast kernel; 1 = every read through the CIGAR-walk kernel
    int rounds_per_wg = 0;          // 0 = auto
    int host_threads = 8;           // threads tcmi_readset_upload packs with
    int stage_cap = 0;              // upper bound on the reads per stage (0 = fill the LDS buffer)
    int chunk_stages = 0;           // stages per chunk, 0 = default (up to 8)
    int project_reads = 1;          // reads with indels / skips go to the fast kernel projected onto the reference
};

int tcmi_fail(tcmi_ctx *ctx, int code, const char *fmt, ...);
void *tcmi_ctx_pinned(tcmi_ctx *ctx, size_t bytes);             // >= bytes of the context's pinned scratch (grow-only; nullptr: no memory)
#define TCMI_HIP(ctx, call)                                                                   \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess)                                                                \
            return tcmi_fail((ctx), TCMI_E_HIP, "%s failed: %s (%s:%d)", #call,               \
                             hipGetErrorString(e__), __FILE__, __LINE__);                     \
    } while (0)

// profiling brackets (no-ops unless enabled)
void tcmi_prof_begin(tcmi_ctx *ctx, int k);
void tcmi_prof_end(tcmi_ctx *ctx, int k);

// what the device packer reads (device pointers)
struct tcmi_pack_src {
    // mode 0: flat arrays (struct tcmi_reads on the device)
    const int32_t *pos; const uint16_t *flag; const int32_t *l_qseq; const int32_t *tid;
    const uint64_t *cigar_off; const uint32_t *cigar; const uint64_t *seq_off; const uint8_t *seq;
    // mode 1: the inflated BAM stream and the offset of every record's block_size field (bam_device.hip)
    const uint8_t *stream; const uint64_t *rec_off;
    int64_t n;
    int32_t mode, pos_shift;
};
// one read on one insert-candidate column, as the device kernel hands it to the host (pack_device.hip -> insert_tokens.cpp)
struct tcmi_dev_entry {
    uint64_t key;               // packed token (insert_tokens.cpp)
    uint64_t name_hash;         // FNV-1a of the read name
    uint32_t j;                 // the read's place in file order (compacted index)
    int32_t pos, end, mpos, isize, l_qseq;
    uint16_t flag;
    uint8_t qual;
    uint8_t bits;               // base code | on_base << 4 | mate on another reference << 5 | insertion too long for the key << 6 (its bases lie in
                                // the call's text buffer: key bits 8-39 where, bits 40-62 how many) | << 7: that buffer was full
    int32_t qref;               // deletion / ref-skip token: reference position of the matched base whose quality is tested, or -1
};
// "does read `idx` have a matched base on reference position `ref`, which, with what quality?" — asked for the other mate of an
// overlapping pair (insert_tokens.cpp); answered from the host arrays or by a kernel over the resident stream
struct tcmi_probe_req { int64_t idx; int32_t ref; };
struct tcmi_probe_res { uint8_t matched, base, qual; };
typedef std::function<int(const std::vector<tcmi_probe_req> &, std::vector<tcmi_probe_res> &)> tcmi_prober;
int tcmi_modal_from_dev_entries(int32_t n_pos, const tcmi_dev_entry *ents, const int64_t *ent_off, const int32_t *ent_cnt,
                                int32_t min_base_quality, int64_t max_depth, int ignore_overlaps, const tcmi_prober *prober, char *tokens,
                                int64_t tokens_cap, int64_t *token_off, int64_t *n_tokens, int32_t *status_flags, const uint8_t *long_text = nullptr,
                                size_t long_bytes = 0);
extern "C" int tcmi_bamfile_read_threads(const char *path, int read_threads, tcmi_bamfile **out);   // (bam_device.hip; 0 = by size)
// device packer (pack_device.hip): struct tcmi_reads -> device -> packed read set; TCMI_E_UNSUPPORTED + *why when the
// input needs the host packer
int tcmi_upload_and_pack_on_device(tcmi_ctx *ctx, const tcmi_reads *r, tcmi_readset *rs, uint32_t *why);
int tcmi_pack_on_device(tcmi_ctx *ctx, const void *pack_src, tcmi_readset *rs, uint32_t *why);
// The one-sync file path (bam_device.hip: tcmi_bamfile_step, tcmi_readset_from_bamfile): everything behind bgzf_copy —
// record index, the chain of records across the blocks, classification, prefix sums, bit planes (pk_index, pk_place), chunk planning
// (pk_pack) — queued on the context's stream from CAPACITIES instead of counts read back; _finish, once the stream has been
// waited for, checks what was deferred (block verdicts, chain, capacities, packer flags) and fills the read set in, or says that
// the file must take the several-kernel path (TCMI_E_UNSUPPORTED: nothing of the job may be used then).
struct tcmi_fused_job {
    // in: device pointers into the context's arena (bgzf_copy)
    const uint8_t *d_stream = nullptr;
    uint64_t stream_len = 0;
    const void *d_desc = nullptr;       // BlockDesc [n_blocks]
    const uint32_t *d_slot = nullptr, *d_nrec = nullptr, *d_first = nullptr, *d_stat = nullptr;
    const int32_t *d_over = nullptr;
    int64_t n_blocks = 0, n_own = 0;
    int ranged = 0;
    int64_t rec_cap = 0;                // records the arrays are sized for
    int64_t len_bound = 0;              // positions the reads may reach (chunk capacity)
    // out (enqueue)
    char *h_pin = nullptr;              // pinned: PackTotals | per-block partials (filled by the copies queued behind the kernels)
    void *d_tot = nullptr;
    uint64_t *d_rec = nullptr;
    uint32_t *c_idx = nullptr, *gen_idx = nullptr;
    int32_t *c_pos = nullptr;
    uint32_t chunk_cap = 0, event_cap = 0, word_cap = 0;
    const void *d_blk_alg = nullptr, *d_blk_end = nullptr;
};
int tcmi_pack_fused_enqueue(tcmi_ctx *ctx, tcmi_fused_job *job, tcmi_readset *rs);
int tcmi_pack_fused_report(tcmi_ctx *ctx, tcmi_fused_job *job);     // queue it LAST (behind the tally and the call, if any): then wait, then _finish
int tcmi_pack_fused_finish(tcmi_ctx *ctx, tcmi_fused_job *job, tcmi_readset *rs, uint32_t *why);
void *tcmi_arena_reserve_take(tcmi_ctx *ctx, size_t total, size_t first);
void *tcmi_arena_take(tcmi_ctx *ctx, size_t bytes);
// kernels (tally.hip / call.hip)
int tcmi_launch_tally(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld, int32_t *d_counts);
int tcmi_launch_tally_fast(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld, int32_t *d_counts);
int tcmi_launch_tally_stream(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld, int32_t *d_counts);
// pipeline-internal: a step whose call kernel rides in the NEXT step's tally launch (api.cpp)
int tcmi_step_begin_deferred(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int32_t mincov, int include_ambig, tcmi_ctx *prev);
int tcmi_step_flush(tcmi_ctx *ctx);
int tcmi_launch_call(tcmi_ctx *ctx, int32_t *d_counts, int64_t L, int64_t ld, int32_t mincov,
                     int include_ambig, int clean, uint8_t *d_plain, uint8_t *d_alt, uint8_t *d_flags,
                     int32_t *d_events, int32_t *d_event_counts);

static inline int64_t tcmi_round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
