// tally.hip — stage A on gfx950: reads (CIGAR + 4-bit SEQ) -> per-position counters.
//
// Replaces indexing.BuildIndex (TrueConsense/indexing.py:75-154): htslib's pileup plus the
// per-token Python loop parse_query_sequences (indexing.py:102-132).  Semantics: SURVEY §8-P.
//
// Two kernels, one per device read set (tcmi_internal.h):
//   tally_fast_kernel    ALIGNED reads (one match op): bit-sliced register accumulation,
//                        no atomics in the inner loop — see the comment above it.
//   tally_atomic_kernel  GENERAL reads (any CIGAR): token-by-token walk, LDS atomics.
//
// tally_atomic_kernel decomposition (read-major; the count matrix is a commutative integer sum):
//   workgroup  = `rounds_per_wg` consecutive ROUNDS of 256 coordinate-sorted reads
//   LDS window = counters for WIN positions starting at the first read's position:
//                planes A,C,G,T,X,I (u32) + a coverage difference array
//   flush      = prefix-sum of the difference array, then one global atomic per touched
//                (plane, position), 64 consecutive positions per wave instruction.
// Tokens that fall outside the window (sparse input, long reads, unsorted input) go straight
// to global atomics: always correct, only slower.
//
// Integer work, HBM-streaming: no MFMA anywhere (BASELINE.json north_star).
#include "tcmi_internal.h"

namespace {

constexpr int WIN = 512;        // positions per LDS window
constexpr int NPLANE = 6;       // A C G T X I
constexpr int BLOCK = 256;      // == TCMI_ROUND: one thread per read of a round

enum { PL_A = 0, PL_C = 1, PL_G = 2, PL_T = 3, PL_X = 4, PL_I = 5 };
__device__ constexpr int plane_col(int pl)
{
    return pl == PL_A ? TCMI_A : pl == PL_C ? TCMI_C : pl == PL_G ? TCMI_G : pl == PL_T ? TCMI_T : pl == PL_X ? TCMI_X : TCMI_I;
}

struct TallyArgs {
    const int32_t *pos;
    const uint32_t *meta;
    const int32_t *lseq;
    const uint32_t *cigar;
    const uint32_t *seq;
    const int64_t *round_cig;
    const int64_t *round_seq;
    int64_t n_piled;
    int64_t n_rounds;
    int64_t ld;
    int32_t *counts;
    int32_t L;
    int32_t rounds_per_wg;
};

__device__ inline bool op_consumes_ref(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
__device__ inline bool op_is_match(uint32_t op) { return op == 0 || op == 7 || op == 8; }
__device__ inline bool op_consumes_query(uint32_t op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }

// htslib resolve_cigar2's peek at the last reference base of op k
__device__ inline bool ins_after(const uint32_t *cg, int n, int k)
{
    if (k + 1 >= n) return false;
    const uint32_t op2 = cg[k + 1] & 0xF;
    int64_t tot = 0;
    if (op2 == 1) {
        tot = cg[k + 1] >> 4;
        for (int j = k + 2; j < n; ++j) {
            const uint32_t o = cg[j] & 0xF;
            if (o == 1) tot += cg[j] >> 4;
            else if (o != 6) break;
        }
    } else if (op2 == 6 && k + 2 < n) {
        for (int j = k + 2; j < n; ++j) {
            const uint32_t o = cg[j] & 0xF;
            if (o == 1) tot += cg[j] >> 4;
            else if (op_consumes_ref(o)) break;
        }
    }
    return tot > 0;
}

// inclusive block scan of a 64-bit value over 256 threads (4 waves of 64)
__device__ inline uint64_t block_scan_incl(uint64_t v, uint64_t *wave_tot /* LDS [4] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    __syncthreads();                       // wave_tot may still be read from a previous scan
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
    uint64_t base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) base += wave_tot[w];
    return v + base;
}

__device__ inline void add_token(uint32_t *win, const TallyArgs &a, int plane, int32_t p, int32_t P0)
{
    const uint32_t w = (uint32_t)(p - P0);
    if (w < (uint32_t)WIN) atomicAdd(&win[plane * WIN + w], 1u);
    else if ((uint32_t)p < (uint32_t)a.L) atomicAdd(&a.counts[(int64_t)plane_col(plane) * a.ld + p], 1);
}

// coverage of [x0, x1): in-window part through the difference array, the rest per position
__device__ inline void add_coverage(int32_t *covd, const TallyArgs &a, int32_t x0, int32_t x1, int32_t P0)
{
    const int32_t lo = max(x0, P0), hi = min(x1, P0 + WIN);
    if (lo < hi) {
        atomicAdd(&covd[lo - P0], 1);
        atomicAdd(&covd[hi - P0], -1);
        for (int32_t p = x0; p < min(x1, P0); ++p)
            if ((uint32_t)p < (uint32_t)a.L) atomicAdd(&a.counts[(int64_t)TCMI_COV * a.ld + p], 1);
        for (int32_t p = max(x0, P0 + WIN); p < x1; ++p)
            if ((uint32_t)p < (uint32_t)a.L) atomicAdd(&a.counts[(int64_t)TCMI_COV * a.ld + p], 1);
    } else {
        for (int32_t p = x0; p < x1; ++p)
            if ((uint32_t)p < (uint32_t)a.L) atomicAdd(&a.counts[(int64_t)TCMI_COV * a.ld + p], 1);
    }
}

// one read, token by token (general CIGAR)
__device__ inline void tally_read_general(uint32_t *win, int32_t *covd, const TallyArgs &a, int32_t pos, int nc,
                                          int32_t lq, const uint32_t *cg, const uint32_t *sq, int32_t P0)
{
    int32_t x = pos, y = 0;
    for (int k = 0; k < nc; ++k) {
        const uint32_t c = cg[k], op = c & 0xF;
        const int32_t len = (int32_t)(c >> 4);
        if (op_consumes_ref(op)) {
            const bool ins = len > 0 && ins_after(cg, nc, k);
            if (op_is_match(op)) {
                uint32_t word = 0;
                for (int32_t j = 0; j < len; ++j) {
                    const int32_t q = y + j;
                    uint32_t nib = 15u;                              // past SEQ -> 'N'
                    if (q < lq) {
                        if (j == 0 || (q & 7) == 0) word = sq[q >> 3];
                        nib = (word >> ((q & 7) * 4)) & 15u;
                    }
                    if (__popc(nib) == 1) add_token(win, a, __ffs(nib) - 1, x + j, P0);   // A=1 C=2 G=4 T=8
                }
            } else if (op == 2) {                                     // D: token "*" counts X ...
                const int32_t nx = ins ? len - 1 : len;              // ... but "*+.." does not
                for (int32_t j = 0; j < nx; ++j) add_token(win, a, PL_X, x + j, P0);
            }
            if (ins) add_token(win, a, PL_I, x + len - 1, P0);        // '+' in the token
            x += len;
        }
        if (op_consumes_query(op)) y += len;
    }
    add_coverage(covd, a, pos, x, P0);                                // M/=/X, D and N all add coverage
}

__global__ __launch_bounds__(BLOCK) void tally_atomic_kernel(TallyArgs a)
{
    __shared__ uint32_t win[NPLANE * WIN];
    __shared__ int32_t covd[WIN + 1];
    __shared__ uint64_t scan_tmp[4];

    const int t = threadIdx.x;
    const int64_t round0 = (int64_t)blockIdx.x * a.rounds_per_wg;
    if (round0 >= a.n_rounds) return;
    const int64_t round1 = min(round0 + a.rounds_per_wg, a.n_rounds);
    const int32_t P0 = a.pos[round0 * BLOCK];

    for (int i = t; i < NPLANE * WIN; i += BLOCK) win[i] = 0;
    for (int i = t; i <= WIN; i += BLOCK) covd[i] = 0;
    __syncthreads();

    for (int64_t rd = round0; rd < round1; ++rd) {
        const int64_t r = rd * BLOCK + t;
        const bool valid = r < a.n_piled;
        int32_t pos = 0, lq = 0;
        uint32_t meta = 0;
        if (valid) { pos = a.pos[r]; meta = a.meta[r]; lq = a.lseq[r]; }
        const uint32_t nc = meta & 0xFFFFu;
        const uint32_t nw = (uint32_t)(lq + 7) >> 3;
        const uint64_t mine = ((uint64_t)nw << 32) | nc;
        const uint64_t excl = block_scan_incl(mine, scan_tmp) - mine;
        if (valid) {
            const uint32_t *cg = a.cigar + a.round_cig[rd] + (uint32_t)excl;
            const uint32_t *sq = a.seq + a.round_seq[rd] + (excl >> 32);
            tally_read_general(win, covd, a, pos, (int)nc, lq, cg, sq, P0);
        }
    }
    __syncthreads();

    // coverage: inclusive prefix sum of the difference array (2 entries per thread)
    {
        const int32_t d0 = covd[2 * t], d1 = covd[2 * t + 1];
        const uint64_t incl = block_scan_incl((uint64_t)(int64_t)(d0 + d1), scan_tmp);
        const int32_t before = (int32_t)(int64_t)incl - (d0 + d1);
        __syncthreads();
        covd[2 * t] = before + d0;
        covd[2 * t + 1] = before + d0 + d1;
    }
    __syncthreads();
    for (int w = t; w < WIN; w += BLOCK) {
        const int32_t p = P0 + w;
        if ((uint32_t)p >= (uint32_t)a.L) continue;
        const int32_t cv = covd[w];
        if (cv) atomicAdd(&a.counts[(int64_t)TCMI_COV * a.ld + p], cv);
#pragma unroll
        for (int pl = 0; pl < NPLANE; ++pl) {
            const uint32_t v = win[pl * WIN + w];
            if (v) atomicAdd(&a.counts[(int64_t)plane_col(pl) * a.ld + p], (int32_t)v);
        }
    }
}

} // namespace

int tcmi_launch_tally(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld, int32_t *d_counts)
{
    if (rs->f_chunks > 0 || (ctx->ride && rs->g_reads == 0)) {   // (an empty fast launch still carries a ride-along call)
        const int rc = tcmi_launch_tally_fast(ctx, rs, L, ld, d_counts);
        if (rc) return rc;
    }
    if (rs->s_reads > 0) {                                    // long reads of a device-decoded stream (pack_device.hip)
        const int rc = tcmi_launch_tally_stream(ctx, rs, L, ld, d_counts);
        if (rc) return rc;
    }
    if (rs->g_reads == 0) return TCMI_OK;
    TallyArgs a;
    a.pos = rs->d_pos; a.meta = rs->d_meta; a.lseq = rs->d_lseq; a.cigar = rs->d_cigar; a.seq = rs->d_seq;
    a.round_cig = rs->d_round_cig; a.round_seq = rs->d_round_seq;
    a.n_piled = rs->g_reads; a.n_rounds = rs->n_rounds; a.ld = ld; a.counts = d_counts; a.L = (int32_t)L;
    int rpw = ctx->rounds_per_wg;
    if (rpw <= 0) rpw = rs->n_rounds >= 4096 ? 4 : rs->n_rounds >= 1024 ? 2 : 1;
    a.rounds_per_wg = rpw;
    const int64_t grid = (rs->n_rounds + rpw - 1) / rpw;
    if (grid <= 0 || grid > INT32_MAX) return tcmi_fail(ctx, TCMI_E_ARG, "bad grid %lld", (long long)grid);
    tcmi_prof_begin(ctx, TCMI_K_TALLY_GENERAL);
    (void)hipGetLastError();                               // drop any stale error of this thread
    hipLaunchKernelGGL(tally_atomic_kernel, dim3((unsigned)grid), dim3(BLOCK), 0, ctx->stream, a);
    tcmi_prof_end(ctx, TCMI_K_TALLY_GENERAL);
    TCMI_HIP(ctx, hipGetLastError());
    return TCMI_OK;
}
