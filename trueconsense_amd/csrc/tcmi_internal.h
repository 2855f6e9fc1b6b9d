// tcmi_internal.h — shared between the translation units of libtcmi.so (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/tcmi.h"

// ---- device read layout -------------------------------------------------------------
// Only reads that pile up (mapped, tid >= 0, pos >= 0, reference span > 0)
// are kept.  They are grouped in ROUNDS of TCMI_ROUND consecutive reads; the per-round
// offset tables replace per-read offsets (the kernel rebuilds those with a block scan),
// so HBM traffic is the algorithmic 12 + 4*n_cigar + ceil(l/2) bytes per read plus
// 16 bytes per round and <= 3 bytes of word padding per read.
#define TCMI_ROUND 256

struct tcmi_readset {
    int64_t n_reads = 0;        // as handed in
    int64_t n_piled = 0;        // kept on device
    int64_t n_rounds = 0;
    int64_t n_cigar = 0;        // total ops of kept reads
    int64_t n_seqw = 0;         // total 32-bit SEQ words of kept reads
    int64_t alg_bytes = 0;      // sum over kept reads of 12 + 4*n_cigar + ceil(l_qseq/2)
    int64_t dev_bytes = 0;
    int64_t max_end = 0;        // max end position (exclusive) of a kept read
    int32_t max_span = 0;       // max reference span of a kept read
    int device = -1;
    int32_t *d_pos = nullptr;   // [n_piled]
    uint32_t *d_meta = nullptr; // [n_piled] flag<<16 | n_cigar
    int32_t *d_lseq = nullptr;  // [n_piled]
    uint32_t *d_cigar = nullptr;// [n_cigar]
    uint32_t *d_seq = nullptr;  // [n_seqw] 8 bases per word, base i at bits [4(i%8), +4)
    int64_t *d_round_cig = nullptr; // [n_rounds+1]
    int64_t *d_round_seq = nullptr; // [n_rounds+1]
};

struct tcmi_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;
    // profiling
    bool prof = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    struct Pending { int k; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[TCMI_K_NKERNELS] = {0, 0, 0};
    int64_t prof_n[TCMI_K_NKERNELS] = {0, 0, 0};
    // workspace of tcmi_step / host-buffer conveniences
    int64_t ws_L = 0, ws_ld = 0;
    int32_t *d_counts = nullptr;
    uint8_t *d_plain = nullptr, *d_alt = nullptr, *d_flags = nullptr;
    uint8_t *h_rec = nullptr;       // pinned: plain | alt | flags, each ws_ld bytes
    int32_t *h_counts = nullptr;    // pinned [7][ws_ld]
    int tally_variant = 0;          // 0 = default; see tally.hip
    int rounds_per_wg = 0;          // 0 = auto
};

int tcmi_fail(tcmi_ctx *ctx, int code, const char *fmt, ...);
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

// kernels (tally.hip / call.hip)
int tcmi_launch_tally(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld, int32_t *d_counts);
int tcmi_launch_call(tcmi_ctx *ctx, const int32_t *d_counts, int64_t L, int64_t ld, int32_t mincov,
                     int include_ambig, uint8_t *d_plain, uint8_t *d_alt, uint8_t *d_flags,
                     int32_t *d_events, int32_t *d_event_counts);

static inline int64_t tcmi_round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
