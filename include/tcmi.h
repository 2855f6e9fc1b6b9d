/*
 * tcmi.h — C ABI of the MI355X-native consensus hot path (libtcmi.so).
 *
 * The reference (RIVM-bioinformatics/TrueConsense, pure Python) has no FFI; the
 * seam this library replaces is three Python call sites:
 *
 *   BuildIndex(bamfile, ref)                       TrueConsense/indexing.py:75-154
 *   ListInserts(iDict, mincov, bam)                TrueConsense/Events.py:5-44
 *   BuildConsensus(mincov, iDict, GFFdict, ...)    TrueConsense/Sequences.py:168-322
 *
 * Everything here is plain C: pointers, sizes, int status codes.  No torch or
 * HIP types appear in a signature (streams / device buffers are `void*`).
 * All entry points return 0 on success or a negative TCMI_E_* code; the text of
 * the last failure is available from tcmi_last_error().  Nothing throws or
 * aborts across this boundary.
 *
 * Threading: one context per device/stream; contexts are independent; a single
 * context is not thread-safe.  Host-only entry points (tcmi_consensus_walk, tcmi_modal_tokens,
 * tcmi_bam_*) are re-entrant and need no GPU; tcmi_bamfile_read needs the HIP runtime (pinned memory).
 */
#ifndef TCMI_H
#define TCMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCMI_ABI_VERSION 5

/* ---- status codes ------------------------------------------------------- */
#define TCMI_OK            0
#define TCMI_E_NODEVICE   (-1)  /* no usable HIP device / HIP runtime error at init */
#define TCMI_E_HIP        (-2)  /* HIP runtime call failed                         */
#define TCMI_E_ARG        (-3)  /* bad argument (null pointer, negative size, ...) */
#define TCMI_E_NOMEM      (-4)
#define TCMI_E_FORMAT     (-5)  /* malformed BAM / BGZF input                      */
#define TCMI_E_IO         (-6)
#define TCMI_E_KEYERROR   (-7)  /* reference would raise KeyError (walk past the last position,
                                   Sequences.py:47 via :214/:282); *err_pos = missing key */
#define TCMI_E_ZERODIV    (-8)  /* reference would raise ZeroDivisionError (Events.py:102) */
#define TCMI_E_UNSUPPORTED (-9)

/* ---- count-matrix column order (indexing.py:134) ------------------------ */
enum { TCMI_COV = 0, TCMI_A = 1, TCMI_T = 2, TCMI_C = 3, TCMI_G = 4, TCMI_X = 5, TCMI_I = 6,
       TCMI_NCOL = 7 };

/* ---- per-position flags produced by the call kernel --------------------- */
#define TCMI_F_LOWCOV   0x01u  /* cov < mincov                      Sequences.py:191        */
#define TCMI_F_PRIMX    0x02u  /* primary nucleotide is X           Sequences.py:199,277    */
#define TCMI_F_MINDEL   0x04u  /* (X/cov)*100 >= 15, cov != 0       Events.py:85-106        */
#define TCMI_F_INSCAND  0x08u  /* insert candidate                  Events.py:29-36         */
#define TCMI_F_COVGT    0x10u  /* cov > mincov (strict)             Sequences.py:311, ORFs.py:142 */
#define TCMI_F_COVZERO  0x20u  /* cov == 0                          Events.py:102 (division) */
#define TCMI_F_AMBIG    0x40u  /* IsAmbiguous() true                Ambig.py:179-228        */
/* a position is an "event" (needs the sequential host walk) when it carries one of: */
#define TCMI_EVENT_MASK (TCMI_F_PRIMX | TCMI_F_MINDEL | TCMI_F_INSCAND)

/* ---- reads as flat host arrays (what a BAM reader yields) --------------- */
/* BAM field meaning per SAM spec §4.2.  Offsets are element counts.          */
typedef struct tcmi_reads {
    int64_t         n_reads;
    const int32_t  *pos;        /* [n] 0-based leftmost reference position               */
    const uint16_t *flag;       /* [n] BAM FLAG                                          */
    const int32_t  *l_qseq;     /* [n] query length (0 when SEQ is '*')                  */
    const uint64_t *cigar_off;  /* [n+1] offsets into cigar[]                             */
    const uint32_t *cigar;      /* BAM encoding len<<4|op, ops MIDNSHP=X                 */
    const uint64_t *seq_off;    /* [n+1] byte offsets into seq[]                          */
    const uint8_t  *seq;        /* BAM 4-bit packed "=ACMGRSVTWYHKDBN", high nibble first */
    const uint8_t  *qual;       /* optional, Σ l_qseq bytes (offsets = prefix of l_qseq) or NULL */
    const int32_t  *tid;        /* optional [n] reference id; reads with tid<0 never pile up */
    /* optional accelerators for the insert-token sweep (tcmi_modal_tokens); zero / NULL = not given */
    const uint64_t *qual_off;   /* [n+1] offsets into qual[] (= prefix sums of l_qseq)                  */
    int64_t sorted_max_span;    /* > 0 promises: reads ascend by pos (unplaced reads last) and no read
                                   spans more reference positions than this; lets the sweep visit only
                                   the reads around each candidate position (tcmi_bam_reads fills it)   */
    /* optional mate fields and read names (SAM spec §4.2 next_refID, next_pos, tlen, read_name): what pysam's default
     * pileup needs to find overlapping mates (ignore_overlaps, Events.py:66); NULL = not given                         */
    const int32_t  *next_tid;   /* [n] */
    const int32_t  *next_pos;   /* [n] */
    const int32_t  *tlen;       /* [n] */
    const uint64_t *name_off;   /* [n+1] offsets into names[] */
    const char     *names;      /* read names, no terminators */
} tcmi_reads;

typedef struct tcmi_ctx tcmi_ctx;          /* one per device + stream            */
typedef struct tcmi_readset tcmi_readset;  /* reads resident in HBM              */

/* ---- library / context -------------------------------------------------- */
int         tcmi_abi_version(void);
const char *tcmi_last_error(const tcmi_ctx *ctx);  /* ctx may be NULL: last error of this thread */
int         tcmi_device_count(int *out_count);     /* 0 devices is not an error                  */

int  tcmi_ctx_create(int device, tcmi_ctx **out);  /* fails with TCMI_E_NODEVICE without a GPU    */
/* A context on a stream the caller owns (a hipStream_t; NULL = the default stream; e.g. torch's
 * current stream, or the
 * stream of another tcmi_ctx: two contexts on one stream give two workspaces whose launches
 * run back to back, so steps can be queued ahead without overlapping each other).            */
int  tcmi_ctx_create_on_stream(int device, void *stream, tcmi_ctx **out);
int  tcmi_ctx_destroy(tcmi_ctx *ctx);
int  tcmi_ctx_sync(tcmi_ctx *ctx);                 /* wait for the context's stream               */
void *tcmi_ctx_stream(tcmi_ctx *ctx);              /* the hipStream_t all launches go to          */
/* options (tcmi_readset_upload reads the packing ones, the launches the others):
 *   "tally_variant"  0 = reads take the bit-plane kernel (default), 1 = every read takes the CIGAR-walk kernel
 *                    (an independent implementation the tests cross-check with)
 *   "project_reads"  1 = reads with indels / ref-skips are projected onto the reference when they are packed and
 *                    take the bit-plane kernel (default); 0 = they take the CIGAR-walk kernel
 *   "device_pack"    1 = tcmi_readset_upload copies the BAM-native arrays to the device and packs them there
 *                    (pack_device.hip; default; needs reads sorted by position and entries of <= 512 positions,
 *                    anything else takes the host packer); 0 = always pack on the host
 *   "verify_crc"     1 = the device decoder checks every BGZF block's CRC-32 (in bgzf_copy's flush; default, as htslib does);
 *                    0 = ISIZE, stream termination and the record chain only
 *   "one_sync"       1 = a file decoded on the device takes the one-sync path first (pk_index + pk_place + pk_pack: record index,
 *                    record chain, classification, places and planes without a host round trip; capacities instead of counts read back; default), 0 = only the
 *                    several-kernel path with its three waits (the path that words every refusal; the tests cross-check the two)
 *   "mid_wait"       0 = ONE wait of the host per file (default since round 5: with the CRC taken in bgzf_copy's flush — a kernel less
 *                    per file — eight contexts measured 69.3 - 70.0 M positions/s against 68.4 - 69.4 with the second wait, three turns
 *                    each on one box), 1 = the one-sync path waits a second time, behind the decode kernels (round 4's default)
 *   "prefix_kernels" the one-sync path's kernels need, per BGZF block, the records / kept reads / plane words in front of it: 0 = every
 *                    workgroup adds them up for itself below 16 384 blocks and three one-workgroup scan launches do it from there on
 *                    (the sums are quadratic in the blocks; default), 1 = always the scan launches, -1 = never
 *   "decode_token_mb" the device decoder's token scratch, MiB (default 4096): a file whose BGZF blocks need more (tokens take 3 - 10
 *                    times the inflated bytes while a block is decoded) is decoded in batches of blocks that share the scratch —
 *                    the inflated stream stays whole
 *   "sym_scratch_div" (tests) bgzf_symbols' speculating lanes park 1 / n of their share of the token scratch: lanes overflow and their blocks
 *                    are decoded once more in order (pass B), which must cut the stream into the same tokens; default 1
 *   "h2d_pieces"     n >= 2: compressed bytes that come from host memory cross PCIe in n pieces of whole blocks on a copy stream of the
 *                    context's, each piece's blocks inflated as soon as it has arrived; 0 (default): one copy in front of the decoder —
 *                    on these boxes (46 GB/s) pieces lose: 4.5 ms against 3.97 ms for a 6.25 M-read range, DESIGN 7
 *   "split_sub"      tcmi_split_step: a rank's block range is decoded, packed and tallied as this many SUB-RANGES side by side — the first on
 *                    this context, the others on helper contexts it owns (a stream, an arena and a host thread each), so that the inflate of
 *                    one sub-range runs under the pack of another; the sub-ranges must join like ranks' ranges, else the range is taken in
 *                    one piece.  0 = auto (default: for a true range of a larger file 3 from 6 144 blocks on, 2 from 4 096, else 1 — the whole file as
 *                    one range is taken in one piece: measured), 1 = never, up to 8
 *   "chunk_stages"   stages per chunk of the bit-plane kernel: 0 = default (up to 8, capped by "balance_chunks"), or 1..8
 *   "balance_chunks" chunk_stages = 0: size the chunks so that a launch has a multiple of
 *                    (compute units x "wg_per_cu", default 4) of them (default 1)
 *   "stage_cap"      upper bound on the reads per stage (0 = fill the LDS stage buffer; for experiments)
 *   "host_threads"   threads the host packer uses (default min(16, cores))
 *   "rounds_per_wg"  CIGAR-walk kernel: rounds of 256 reads per workgroup (0 = auto)
 *   "defer_call"     (read from the pipeline's first workspace) 1 = tcmi_pipeline_run attaches the call of step k to
 *                    the tally launch of step k + 1 — one launch per step (default); 0 = tally launch + call launch
 *   "profile_every"  with profiling enabled, every n-th tcmi_step_begin has its kernels bracketed by events,
 *                    the others go out unmeasured (default 1) */
int  tcmi_ctx_set_option(tcmi_ctx *ctx, const char *key, int value);
/* counters of a context: "one_sync_taken" / "one_sync_declined" — files (or block ranges) the one-sync path delivered / handed to the
 * several-kernel path; "one_sync_last_decline_flags" — why the last one was handed over (packer flags; 0: it was not a packer flag);
 * "decode_batched" — files (or ranges) whose blocks the device decoder took in batches ("decode_token_mb");
 * "split_sub_taken" — tcmi_split_step calls whose range went through sub-ranges ("split_sub"); "h2d_piped" — decodes whose
 * compressed bytes crossed PCIe in pieces on a copy stream ("h2d_pieces") */
int  tcmi_ctx_stat(tcmi_ctx *ctx, const char *key, int64_t *value);

/* per-kernel device timing (hipEvents on the context's stream); kernel ids below */
enum { TCMI_K_TALLY = 0 /* bit-plane tally kernel */, TCMI_K_CALL = 1, TCMI_K_ZERO = 2,
       TCMI_K_TALLY_GENERAL = 3 /* CIGAR-walk tally kernel */,
       TCMI_K_PACK_CLASSIFY = 4 /* device packer: classify + scan */, TCMI_K_PACK = 5 /* device packer: scatter + pack */,
       TCMI_K_INFLATE = 6 /* device BGZF inflate */, TCMI_K_RECORDS = 7 /* device BAM record walk */,
       TCMI_K_CRC = 8 /* retired (always 0 since ABI 5): the blocks' CRC-32 is taken in bgzf_copy's flush, filed under TCMI_K_INFLATE_COPY */,
       TCMI_K_INFLATE_COPY = 9 /* device BGZF inflate, second kernel: tokens -> bytes (TCMI_K_INFLATE is the first: symbols -> tokens) */,
       TCMI_K_NKERNELS = 10 };
int  tcmi_profile_enable(tcmi_ctx *ctx, int on);
int  tcmi_profile_reset(tcmi_ctx *ctx);
int  tcmi_profile_get(tcmi_ctx *ctx, int kernel, double *total_ms, int64_t *launches);

/* ---- stage A: pileup tally  (replaces indexing.BuildIndex, indexing.py:75-154;
 *      inner loop parse_query_sequences, indexing.py:102-132; semantics SURVEY §8-P) */

/* Reference length the tally matrix needs: max(ref_len, max end position of any
 * piled-up read) — the reference keeps columns beyond the FASTA length (indexing.py:137-151). */
int tcmi_reads_extent(const tcmi_reads *reads, int64_t ref_len, int64_t *out_L);

/* Reads -> HBM in the tally kernel's layout.  By default the BAM-native arrays are copied to the device as they are
 * and packed there by HIP kernels (CIGAR projection, token classification, coverage runs, 4-bit -> bit planes);
 * inputs the device packer does not take (see option "device_pack") are packed on the host.                        */
int tcmi_readset_upload(tcmi_ctx *ctx, const tcmi_reads *reads, tcmi_readset **out);
/* Several BAMs in ONE read set (BASELINE configs[3], many independent BAMs): BAM b's positions are
 * shifted by b * stride (a multiple of 256, >= the extent of every BAM), so one tally launch and one
 * call launch process the whole batch over n * stride positions; BAM b's counts / records are the
 * slice [b * stride, b * stride + L) of every plane.                                             */
int tcmi_readset_upload_batch(tcmi_ctx *ctx, const tcmi_reads *const *reads, int32_t n, int64_t stride,
                              tcmi_readset **out);
int tcmi_readset_free(tcmi_ctx *ctx, tcmi_readset *rs);
int tcmi_readset_info(const tcmi_readset *rs, int64_t *n_reads, int64_t *n_piled,
                      int64_t *algorithmic_bytes, int64_t *device_bytes, int64_t *max_end);
/* how the reads were split: aligned set (bit-plane kernel; entries, chunks) vs general set (CIGAR-walk kernel) */
int tcmi_readset_sets(const tcmi_readset *rs, int64_t *aligned_reads, int64_t *aligned_chunks,
                      int64_t *general_reads);

/* *packed_on_device = 1 when the HIP packer (pack_device.hip) built the read set, 0 when the host packer did */
int tcmi_readset_origin(const tcmi_readset *rs, int32_t *packed_on_device);

/* Device-resident tally.  d_counts: device int32 [7][ld] (plane order TCMI_COV..TCMI_I,
 * plane p at d_counts + p*ld, ld >= L).  Zeroes the planes first when `zero` != 0,
 * otherwise accumulates (used when one BAM is split over several read sets / GPUs). */
int tcmi_tally_dev(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld,
                   void *d_counts, int zero);

/* Host-buffer convenience: upload, tally, download.  counts: host int32 [L][7] in the
 * reference's row layout (position p-1 at counts + 7*(p-1)).                      */
int tcmi_tally(tcmi_ctx *ctx, const tcmi_reads *reads, int64_t L, int32_t *counts);

/* planes [7][ld] on device  <->  rows [L][7] on host                              */
int tcmi_counts_download(tcmi_ctx *ctx, const void *d_counts, int64_t L, int64_t ld, int32_t *counts);
int tcmi_counts_upload(tcmi_ctx *ctx, const int32_t *counts, int64_t L, int64_t ld, void *d_counts);

/* ---- stage B (position-local part): base calling
 *      replaces Sequences.GetNucleotide/GetDistribution (Sequences.py:119-165),
 *      Ambig.IsAmbiguous (Ambig.py:179-228), Events.MinorityDel (Events.py:85-106),
 *      the candidate test of Events.ListInserts (Events.py:29-36) and the case rule
 *      (Sequences.py:229-235 and copies).  Outputs, one byte per position:
 *        plain[p]  character emitted when the primary nucleotide is not X
 *                  ('N' when cov < mincov)
 *        alt[p]    character of the secondary nucleotide (primary == X, Sequences.py:283-290)
 *        flags[p]  TCMI_F_* bits                                                      */
int tcmi_call_dev(tcmi_ctx *ctx, const void *d_counts, int64_t L, int64_t ld,
                  int32_t mincov, int include_ambig,
                  void *d_plain, void *d_alt, void *d_flags,
                  void *d_events /* int32[ceil(L/256)*256], may be NULL */,
                  void *d_event_counts /* int32[ceil(L/256)], may be NULL */);

/* Host-buffer convenience (counts rows [L][7] in, bytes out). event_idx (capacity L)
 * receives the ascending 0-based indices of positions with flags & TCMI_EVENT_MASK. */
int tcmi_call(tcmi_ctx *ctx, const int32_t *counts, int64_t L, int32_t mincov, int include_ambig,
              uint8_t *plain, uint8_t *alt, uint8_t *flags,
              int32_t *event_idx, int64_t *n_events);

/* Whole resident step: zero + tally + call + copy records to pinned host memory.
 * Returns host pointers valid until the next step on this context.              */
int tcmi_step(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int32_t mincov, int include_ambig,
              const uint8_t **plain, const uint8_t **alt, const uint8_t **flags,
              const int32_t **counts_planes /* host [7][ld] */, int64_t *ld);
/* The same in two halves, so that the host walk of one BAM can overlap the GPU work of the next
 * (two contexts, two streams): _begin only enqueues, _end waits for the context's stream.     */
int tcmi_step_begin(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int32_t mincov,
                    int include_ambig, int want_counts);
int tcmi_step_end(tcmi_ctx *ctx, const uint8_t **plain, const uint8_t **alt, const uint8_t **flags,
                  const int32_t **counts_planes, int64_t *ld);

/* ---- stage B (sequential part, HOST): the consensus walk
 *      replaces the loop of Sequences.BuildConsensus (Sequences.py:179-322) with
 *      ORFs.in_orf / SolveTripletLength / CorrectStartPositions / CorrectGFF
 *      (ORFs.py:1-192) restated in O(L) over the call records.  No GPU needed.
 *
 *  orf_*      : one entry per GFF row (all rows take part in in_orf, ORFs.py:16-26;
 *               only strand '+' rows get their end corrected, ORFs.py:154)
 *  ins_pos    : 1-based positions of accepted inserts (ascending), ins_shift =
 *               int(size_str) of Events.py:75-80 (last digit only), ins_seq/ins_off =
 *               concatenated insert strings.
 *  out_cons   : capacity cons_cap bytes (L + total insert length is enough)
 *  new_start/new_end : corrected GFF coordinates per row
 *  err_pos    : for TCMI_E_KEYERROR the missing key (L+1)                          */
int tcmi_consensus_walk(const uint8_t *plain, const uint8_t *alt, const uint8_t *flags, int64_t L,
                        int32_t n_orf, const int64_t *orf_start, const int64_t *orf_end,
                        const uint8_t *orf_is_plus,
                        int32_t n_ins, const int64_t *ins_pos, const int32_t *ins_shift,
                        const char *ins_seq, const int64_t *ins_off,
                        int include_ins,
                        char *out_cons, int64_t cons_cap, int64_t *out_len,
                        int64_t *new_start, int64_t *new_end, int64_t *err_pos);

/* ---- insert tokens (Events.ExtractInserts, Events.py:47-82), HOST.
 * For each 1-based candidate position (ascending) scans the reads overlapping it under the
 * filters of pysam's default-argument region pileup (Events.py:66) and returns the modal
 * upper-cased token (Counter.most_common, first-seen tie-break, Events.py:71-74).
 *   tokens / token_off : concatenated modal tokens, token k = tokens[token_off[k] .. token_off[k+1])
 *                        (empty when the column has no token)
 *   n_tokens[k]        : tokens in column k after filtering (0: empty pileup -> position dropped)
 *   max_depth          : htslib's maxcnt (pysam's max_depth, default 8000): reads that share their start with the read
 *                        before them are dropped once the column holds this many; 0 = no cap
 *   ignore_overlaps    : pysam's default 1: of two overlapping mates only one keeps its base on the column (needs the
 *                        mate fields and names of tcmi_reads; reads without names are taken as unpaired).  A mate whose
 *                        token on the column is a deletion / ref-skip is tested on its next query base, as htslib does,
 *                        with the tweak evaluated on that base's reference position
 *   status_flags       : bit 0 = max_depth dropped reads (modelled, informational); bit 1 = a pair of overlapping mates
 *                        whose quality tweak could not be evaluated (not raised by the entry points of this library any
 *                        more; kept for callers that test it): the tokens would NOT be what pysam gives — refuse         */
#define TCMI_TOKENS_DEPTH_CAPPED     1
#define TCMI_TOKENS_OVERLAP_UNKNOWN  2
int tcmi_modal_tokens(const tcmi_reads *reads, int32_t n_pos, const int64_t *positions,
                      int32_t min_base_quality, uint32_t flag_filter, int ignore_orphans,
                      int64_t max_depth, int ignore_overlaps, char *tokens, int64_t tokens_cap, int64_t *token_off,
                      int64_t *n_tokens, int32_t *status_flags);

/* ---- many BAMs: native batch runner (BASELINE.json configs[3]: independent BAMs, no collective).
 * The calling thread queues step i+1 behind step i on one stream over `n_slots` workspaces while
 * `n_walkers` host threads turn finished call records into consensus sequences
 * (Events.ListInserts + Sequences.BuildConsensus(..., includeINS=True), the FASTA content).
 *   readsets[i]   reads of BAM i resident in HBM (same device)
 *   host_reads[i] the same reads on the host, needed only to resolve insert tokens
 *                 (Events.py:47-82); the array or an entry may be NULL: an item that then has
 *                 insert candidates fails with TCMI_E_UNSUPPORTED
 *   out_cons      n_items * stride bytes, consensus i at out_cons + i*stride, length out_len[i];
 *                 stride >= L + 1 + total inserted bases
 *   status[i]     TCMI_OK or the error of item i (e.g. TCMI_E_KEYERROR where the reference raises) */
typedef struct tcmi_pipeline tcmi_pipeline;
int tcmi_pipeline_create(int device, int n_slots, int n_walkers, tcmi_pipeline **out);
int tcmi_pipeline_destroy(tcmi_pipeline *p);
int tcmi_pipeline_set_orfs(tcmi_pipeline *p, int32_t n_orf, const int64_t *start, const int64_t *end,
                           const uint8_t *is_plus);
tcmi_ctx *tcmi_pipeline_ctx(tcmi_pipeline *p, int slot);   /* slot contexts (upload reads with slot 0's) */
int tcmi_pipeline_run(tcmi_pipeline *p, int64_t n_items, const tcmi_readset *const *readsets,
                      const tcmi_reads *const *host_reads, int64_t L, int32_t mincov, int include_ambig,
                      char *out_cons, int64_t stride, int64_t *out_len, int32_t *status);
/* The same over BATCHED read sets (tcmi_readset_upload_batch with `batch` BAMs at `pos_stride`):
 * item i yields consensus i*batch .. i*batch + batch - 1 (host_reads, out_len, status likewise).   */
int tcmi_pipeline_run_batched(tcmi_pipeline *p, int64_t n_items, const tcmi_readset *const *readsets,
                              int32_t batch, int64_t pos_stride, const tcmi_reads *const *host_reads,
                              int64_t L, int32_t mincov, int include_ambig, char *out_cons,
                              int64_t stride, int64_t *out_len, int32_t *status);

/* ---- BAM reader (pysam's role; SAM spec §4.2), HOST, zlib inflate -------- */
typedef struct tcmi_bam tcmi_bam;
int  tcmi_bam_load(const char *path, int n_threads, tcmi_bam **out);
int  tcmi_bam_free(tcmi_bam *bam);
/* fills `reads` with pointers owned by `bam`; n_ref / first reference name and length */
int  tcmi_bam_reads(const tcmi_bam *bam, tcmi_reads *reads);
int  tcmi_bam_header(const tcmi_bam *bam, int32_t *n_ref, const char **ref0_name, int64_t *ref0_len);
/* any out pointer may be NULL; sorted = 1 when mapped reads ascend by (tid, pos)               */
int  tcmi_bam_info(const tcmi_bam *bam, int64_t *n_reads, int32_t *sorted, int64_t *file_bytes,
                   int64_t *inflated_bytes, int64_t *n_blocks, int64_t *n_cigar, int64_t *n_qual);
const char *tcmi_bam_text(const tcmi_bam *bam);    /* SAM header text, owned by bam                */

/* ---- BAM decoded ON THE DEVICE (the default file path; the host reader above stays for files it does not take) ----
 * tcmi_bamfile_read: HOST — file bytes into pinned memory, BGZF block table, BAM header (only the leading blocks the
 * header occupies are inflated on the host).  tcmi_readset_from_bamfile: the compressed bytes cross PCIe, HIP kernels
 * inflate every BGZF block, follow the record chain (records may straddle blocks), and pack the reads for the tally (reads
 * spanning more than 512 positions are tallied where they lie in the stream) — the host never sees a decoded read.
 * Returns TCMI_E_UNSUPPORTED for files that need the host reader (a record chain that does not close, a mapped read on a
 * second reference, a CIGAR kept in a CG:B tag, positions beyond 2^29): fall back to tcmi_bam_load + tcmi_readset_upload. */
typedef struct tcmi_bamfile tcmi_bamfile;
int  tcmi_bamfile_read(const char *path, tcmi_bamfile **out);
int  tcmi_bamfile_free(tcmi_bamfile *f);
int  tcmi_bamfile_info(const tcmi_bamfile *f, int64_t *file_bytes, int64_t *inflated_bytes, int64_t *n_blocks, int32_t *n_ref,
                       const char **ref0_name, int64_t *ref0_len);
const char *tcmi_bamfile_text(const tcmi_bamfile *f);
const char *tcmi_bamfile_path(const tcmi_bamfile *f);
/* The file's compressed bytes into HBM, to stay until tcmi_bamfile_free: tcmi_readset_from_bamfile[_blocks] on that device then
 * start from device memory, no PCIe copy per call (a file a peer GPU, a NIC or a storage engine delivered into HBM looks like
 * this; bench.py's headline times this form: "inputs resident in HBM when the timed region starts"). */
int  tcmi_bamfile_to_device(tcmi_ctx *ctx, tcmi_bamfile *f);
int  tcmi_readset_from_bamfile(tcmi_ctx *ctx, const tcmi_bamfile *f, tcmi_readset **out, int64_t *n_reads);
/* ... of the alignment records that START in BGZF blocks [first_block, first_block + n_blocks) only (n_blocks < 0: to the end of the
 * file).  Ranks that share ONE BAM file (BASELINE configs[4]) each decode a contiguous range of its blocks and nothing else; the
 * count matrices of the ranges add up to the file's (what indexing.py:96-100 piles up in one pass). */
int  tcmi_readset_from_bamfile_blocks(tcmi_ctx *ctx, const tcmi_bamfile *f, int64_t first_block, int64_t n_blocks, tcmi_readset **out,
                                      int64_t *n_reads);
/* A range that starts in the middle of the file starts at the first PLAUSIBLE record its first block finds — nothing in front of
 * it vouches for that offset.  *first: where that record starts, *next: where the first record behind the range starts, both as
 * offsets into the whole file's inflated stream (-1: the range starts with the file or holds no record start / its chain was never
 * fixed).  Ranges that tile a file are all true record chains iff every range's *next equals the *first of the next range that has
 * one (by induction from the header: what any reader of a BAM file relies on, htslib's bam_read1 included); tcmi_split_step checks
 * exactly that along with its reduce, trueconsense_amd/distributed.py before its collective. */
int  tcmi_readset_range_anchors(const tcmi_readset *rs, int64_t *first, int64_t *next);
/* BAM file -> call records with ONE wait of the host: what indexing.BuildIndex (indexing.py:75-154) and the position-local part of
 * Sequences.BuildConsensus (Sequences.py:119-165 via Ambig.py / Events.py) come to for one file.  Decode, record index, record chain,
 * classification, packing, tally and call are queued back to back on the context's stream from capacities instead of counts read
 * back (the several calls above wait three + one times); one wait; then everything that was deferred is checked, and a file the
 * one-pass packer cannot vouch for (a damaged block, a record chain that does not close, reads it does not take, a file beyond the
 * capacities) is taken through tcmi_readset_from_bamfile + tcmi_step instead — same results, same errors.  The step covers
 * L = max(ref_len, the reads' extent, 1) positions (*L_out).  Results as tcmi_step's (valid until the context's next step);
 * *rs_out is the caller's (tcmi_readset_free), with the inflated stream resident for tcmi_readset_modal_tokens. */
int  tcmi_bamfile_step(tcmi_ctx *ctx, const tcmi_bamfile *f, int64_t ref_len, int32_t mincov, int include_ambig, tcmi_readset **rs_out,
                       int64_t *L_out, const uint8_t **plain, const uint8_t **alt, const uint8_t **flags,
                       const int32_t **counts_planes /* host [7][ld]; NULL: not wanted */, int64_t *ld);
/* Events.ExtractInserts for a read set the DEVICE decoded (tcmi_readset_from_bamfile): same arguments and results as
 * tcmi_modal_tokens, but the reads of every candidate column are examined by a HIP kernel where they lie (the inflated
 * stream stays resident on the context until its next upload) and only a few thousand 48-byte entries per column reach
 * the host (+ the bases of insertions too long for an entry's 64-bit token key, from a 4 MB text buffer).  TCMI_E_UNSUPPORTED when the
 * stream is gone (another upload happened on the context), the read set was not decoded on the device, or that text buffer
 * overflows: use tcmi_bam_load + tcmi_modal_tokens. */
int  tcmi_readset_modal_tokens(tcmi_ctx *ctx, const tcmi_readset *rs, int32_t n_pos, const int64_t *positions,
                               int32_t min_base_quality, uint32_t flag_filter, int ignore_orphans, int64_t max_depth,
                               int ignore_overlaps, char *tokens, int64_t tokens_cap, int64_t *token_off, int64_t *n_tokens,
                               int32_t *status_flags);
/* ---- ONE BAM file over several GPUs (BASELINE configs[4]; what the ranks jointly replace is the single-pass pile-up of
 * indexing.py:96-100 and, for the insert candidates, the region pile-ups of Events.py:47-82).
 * The library links no collective library: the one exchange of the path — a sum of the int32 count matrix to the rank that calls —
 * is a hook of the caller's (RCCL's ncclReduce on the given stream from C: include/tcmi_rccl.h, libtcmi_rccl.so; torch.distributed from Python:
 * trueconsense_amd/distributed.py).
 *   tcmi_split_step   this rank's part of one step, in C (rank 0 is the root): decode + pack + tally the alignment records that
 *                     start in BGZF blocks [first_block, first_block + n_blocks) into d_counts — device int32 [7][ld] +
 *                     TCMI_SPLIT_TAIL_WORDS(world) more int32 behind it: six words per rank, written by that rank into its own
 *                     slot and zero in everybody else's, so that the sum hands the root the table {first_block, n_blocks, first,
 *                     next (two words each)} of all ranges — a range in the middle of the file starts at the first offset its
 *                     first block finds plausible for a record, and only the range in front can vouch for it: the root checks
 *                     PAIRWISE that every range starts where the one in front ends and that the last one ends with the stream
 *                     (tcmi_readset_range_anchors; the same rule as distributed.check_range_anchors) —, and one word that counts
 *                     the ranks that failed: a rank that cannot do its share (a range that is refused, a workspace that cannot
 *                     be allocated) still takes part in the exchange, with zeros, so nobody waits for it forever;
 *                     reduce(user, d_counts, 7 * ld + TCMI_SPLIT_TAIL_WORDS(world), stream) — the hook must sum over the ranks,
 *                     at least to rank 0 —, and on rank 0 the call kernel: results as tcmi_step's.  *rs_out (any rank; NULL on
 *                     failure) keeps the rank's decoded stream resident for tcmi_readset_ins_entries.  Returns the rank's own
 *                     error, or on rank 0 TCMI_E_UNSUPPORTED when another rank failed or the ranges' record chains do not join.
 *   tcmi_readset_ins_entries   the entries of the candidate columns (TCMI_INS_ENTRY_BYTES each, opaque; per column in file order)
 *                     from this rank's records: what the ranks send to the root.  ent_off[n_pos + 1]; insertions of more than 12
 *                     bases leave their bases in long_text (long_used bytes).  TCMI_E_ARG with ent_off / long_used filled in when
 *                     a buffer is too small.
 *   tcmi_ins_entries_rebase    before concatenating the pieces of several ranks: piece k's long-insertion texts lie `long_base`
 *                     bytes into the concatenated text buffer
 *   tcmi_modal_from_entries    HOST: the vote of tcmi_readset_modal_tokens over per-column concatenations (rank order = file
 *                     order) of such pieces.  status_flags bit 1 (TCMI_TOKENS_OVERLAP_UNKNOWN): a pair of overlapping mates whose
 *                     other mate had to be looked at on another position — not possible across pieces: use the host sweep
 *                     (tcmi_bam_load + tcmi_modal_tokens) for that file. */
#define TCMI_INS_ENTRY_BYTES 48
typedef int (*tcmi_reduce_fn)(void *user, void *d_counts, int64_t n_int32, void *stream);
#define TCMI_SPLIT_TAIL_WORDS(world) (6 * (world) + 1)
int  tcmi_split_step(tcmi_ctx *ctx, const tcmi_bamfile *f, int64_t first_block, int64_t n_blocks, int64_t L, int64_t ld, void *d_counts,
                     int32_t mincov, int include_ambig, tcmi_reduce_fn reduce, void *user, int rank, int world, tcmi_readset **rs_out,
                     const uint8_t **plain, const uint8_t **alt, const uint8_t **flags);
int  tcmi_readset_ins_entries(tcmi_ctx *ctx, const tcmi_readset *rs, int32_t n_pos, const int64_t *positions, uint32_t flag_filter,
                              int ignore_orphans, void *entries, int64_t entries_cap, int64_t *ent_off, uint8_t *long_text,
                              int64_t long_cap, int64_t *long_used);
int  tcmi_ins_entries_rebase(void *entries, int64_t n_entries, int64_t long_base);
int  tcmi_modal_from_entries(int32_t n_pos, void *entries, const int64_t *ent_off, int32_t min_base_quality, int64_t max_depth,
                             int ignore_overlaps, const uint8_t *long_text, int64_t long_bytes, char *tokens, int64_t tokens_cap,
                             int64_t *token_off, int64_t *n_tokens, int32_t *status_flags);
/* for tests and tools: the device-inflated stream and the record offsets copied back to the host */
int  tcmi_bamfile_decode_to_host(tcmi_ctx *ctx, const tcmi_bamfile *f, uint8_t *stream, int64_t stream_cap,
                                 uint64_t *rec_off, int64_t rec_cap, int64_t *n_rec);

/* ---- BAM files -> FASTA text: native runner of the whole path for many inputs (TrueConsense.py:212-264 per file;
 * BASELINE configs[1] / [3]).  Three stages on their own threads, consecutive BAMs overlapping: read (host: file bytes,
 * block table, header), gpu (one thread per context: device decode + pack + tally + call; the host reader for files
 * the device decoder declines), walk (host: insert tokens if a candidate exists, consensus walk, FASTA text).
 *   out_text        n * stride bytes; text i (">name mincov=N\n<consensus>\n", Outputs.py:182-183) at out_text + i*stride
 *   stage_seconds   [4] busy seconds summed over the items: read, decode + pack, step, walk (may be NULL)
 *   decoded_on      [2] items decoded on the device / by the host reader (may be NULL)                                */
typedef struct tcmi_filerunner tcmi_filerunner;
int tcmi_filerunner_create(int device, int n_readers, int n_gpu, int n_walkers, int host_decode_threads, tcmi_filerunner **out);
int tcmi_filerunner_destroy(tcmi_filerunner *r);
int tcmi_filerunner_set_orfs(tcmi_filerunner *r, int32_t n_orf, const int64_t *start, const int64_t *end, const uint8_t *is_plus);
tcmi_ctx *tcmi_filerunner_ctx(tcmi_filerunner *r, int k);
int tcmi_filerunner_run(tcmi_filerunner *r, int64_t n, const char *const *paths, const char *const *names, int64_t ref_len,
                        int32_t mincov, int include_ambig, int device_decode, char *out_text, int64_t stride, int64_t *out_len,
                        int32_t *status, double *stage_seconds, int64_t *decoded_on);
/* ... of files read before (tcmi_bamfile_read; they stay the caller's): the read stage has nothing to do, and with
 * tcmi_bamfile_to_device the GPU stage starts from HBM.  Everything else as tcmi_filerunner_run. */
int tcmi_filerunner_run_resident(tcmi_filerunner *r, int64_t n, tcmi_bamfile *const *files, const char *const *names, int64_t ref_len,
                                 int32_t mincov, int include_ambig, char *out_text, int64_t stride, int64_t *out_len, int32_t *status,
                                 double *stage_seconds, int64_t *decoded_on);
/* The command line's other three outputs for many samples (Outputs.py:13-71, 104-180; Coverage.py:1-16), written by the runner's
 * walker threads.  tcmi_filerunner_set_outputs: the reference (id and sequence of its first FASTA record), the complete VCF header
 * text (Outputs.py:115-127), the GFF header text and per GFF row (the rows of tcmi_filerunner_set_orfs, same order) six strings:
 * source, type, score, strand, phase, attributes.  tcmi_filerunner_run_files: per sample the paths of its FASTA and (optionally;
 * an array or an entry may be NULL) VCF, corrected GFF and coverage TSV. */
int tcmi_filerunner_set_outputs(tcmi_filerunner *r, const char *ref_id, const char *ref_seq, const char *vcf_head, const char *gff_head,
                                int32_t n_rows, const char *const *row_cols);
int tcmi_filerunner_run_files(tcmi_filerunner *r, int64_t n, const char *const *paths, const char *const *names, const char *const *fasta,
                              const char *const *vcf, const char *const *gff, const char *const *doc, int64_t ref_len, int32_t mincov,
                              int include_ambig, int device_decode, int32_t *status, double *stage_seconds, int64_t *decoded_on);

#ifdef __cplusplus
}
#endif
#endif /* TCMI_H */
