// tally_fast.hip — stage A for ALIGNED reads (one run of match ops) on gfx950, without atomics in
// the inner loop.  Counterpart of the per-token loop of indexing.py:102-132 for the tokens that
// are plain bases: coverage += 1 and, if the base is A/C/G/T, that class += 1.
//
// Data (tcmi_internal.h): per read 8 bytes of header and its aligned bases as one-hot nibbles
// (A=1 C=2 G=4 T=8, anything else 0), 8 bases per 32-bit word, `pad` zero words after each read.
// The genome is cut in GRID WORDS of 8 positions; a CHUNK of <= 1024 coordinate-sorted reads
// touches a window of Wn <= 96 grid words (24 at 5000x coverage with 150-bp reads).
//
// One workgroup (TCMI_F_BLOCK lanes) per chunk:
//   lane (g, s)  owns NW adjacent grid words (8*NW positions) g of the window and DEPTH SLICE s
//                of the reads; S = lanes / ceil(Wn/NW) slices work in parallel on different reads.
//   stage        <= 256 reads at a time: headers and bases -> LDS with coalesced loads (16 bytes
//                per lane for the bases); the loads of stage i+1 are issued before the inner loop
//                of stage i runs and land in registers meanwhile.  Every input byte leaves HBM once.
//   inner loop   per read of the slice: NW+1 LDS words (the zero padding makes bounds tests
//                unnecessary: one v_med3 clamps the word index) -> NW v_alignbit funnel shifts
//                bring the read's nibbles onto the lane's grid words; `(w >> c) & 0x11111111`
//                isolates class c as eight 4-bit counters, added to a register.  Only A, C, G are
//                counted: T = coverage - A - C - G - (covered positions without an A/C/G/T base:
//                rare, kept as event words and subtracted by the tail blocks of the same launch).  Every <= 12
//                reads the 4-bit counters are widened into 8-bit counters.  No atomics, no
//                data-dependent branches.
//   coverage     difference array in LDS, one (+run, -run) pair per run of equal (pos, len)
//                reads found with a wave ballot, then a block prefix sum.
//   reduce       slices are summed through LDS (plain stores / loads) into 16-bit window counters,
//                then ONE coalesced global atomic per touched (class, position) of the window.
//
//   fused call   (FUSED launches, tcmi_step_begin) the matrix is cut in TILES of 256 positions; the read set knows
//                how many workgroups add into each tile (chunk windows and tail blocks).  Every workgroup
//                signs off the tiles it touched (release fence + one atomic per tile); the one that signs
//                off last reads the finished counters of the tile, CALLS its 256 positions (call_device.h),
//                stores the records and leaves counters and sign-off count zeroed for the next launch.
//                No workgroup ever waits for another; the separate call launch and its kernel boundary go.
//
// HBM-streaming integer work: no MFMA (BASELINE.json north_star).
#include <algorithm>

#include "tally_fast_common.h"

namespace {

#ifndef TCMI_ABL
#define TCMI_ABL 0      // diagnostic builds only (tools/build_variant.sh), bit mask: 1 no class extraction, 2 no inner loop,
                        // 4 no global loads / staging, 8 no final reduce + atomics, 16 no coverage runs
#endif
constexpr int UNR = TCMI_F_BLOCK == 512 ? 3 : 4;                          // reads in flight per lane in the inner loop
constexpr int WIDEN = 12;                       // reads between widenings of the 4-bit counters (<= 15, multiple of UNR)

constexpr int CPL = (MAXPOS + FB - 1) / FB;     // coverage entries per lane in the final prefix sum

template <int NW, bool FUSED>
__global__ __launch_bounds__(FB, FB == 512 ? 8 : 4) void tally_fast_kernel(FastArgs a)
{
    constexpr int PAD = NW + 1;
    constexpr int NREG = NW * 3 * 2;                                          // 8-bit counter registers per lane
    __shared__ __attribute__((aligned(16))) uint32_t s_seq[TCMI_F_SEQCAP];   // staged bases; later the slice partials
    __shared__ __attribute__((aligned(8))) uint2 s_hdr[TCMI_F_SUB + 1];      // {pos - P0 | words << 16, byte offset in s_seq}
    __shared__ int32_t s_cov[MAXPOS + 8];                                     // coverage difference array
    __shared__ int s_scan[FB / 64];
    static_assert(NREG * FB <= TCMI_F_SEQCAP, "slice partials must fit the stage buffer");

    const int tid = threadIdx.x, lane = tid & 63;
    if ((int)blockIdx.x < a.n_call2) {           // ride-along call of an earlier step's matrix (first in the grid: done early)
        call_other_tile(a, (int)blockIdx.x);
        return;
    }
    const int bid = (int)blockIdx.x - a.n_call2;
    if (bid >= a.n_chunks) {
        tally_tail_block<FUSED>(a, bid, reinterpret_cast<int *>(s_hdr));
        return;
    }
    const tcmi_fast_chunk *chp = a.chunks + bid;
    const int64_t read0 = chp->read0, word0 = chp->word0;
    const int n_reads = chp->n_reads, P0 = chp->P0, Wn = chp->Wn, sub_reads = chp->sub_reads;
    const int npos = Wn * 8;
    const int Gn = (Wn + NW - 1) / NW;          // lane groups
    const int S = FB / Gn;                      // depth slices
    const int s = tid / Gn, gi = tid - s * Gn;
    const int base8 = gi * (8 * NW);            // first owned position, relative to P0
    const int s_eff = s < S ? s : (1 << 20);     // lanes beyond the last slice only ever see the dummy read
    const int n_stage = (n_reads + sub_reads - 1) / sub_reads;

    for (int i = tid; i <= npos; i += FB) s_cov[i] = 0;
    __syncthreads();                            // before any wave adds coverage runs into it

    uint32_t nib[NW][3];                        // [word][class A,C,G]: eight 4-bit counters
    uint32_t byt[NW][3][2];                     // [word][class][even|odd position]: four 8-bit counters
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int c = 0; c < 3; ++c) nib[w][c] = byt[w][c][0] = byt[w][c][1] = 0;

    // ---- prefetch registers: the next stage's headers and bases ------------------------------------
    int h_pos = 0;
    uint32_t h_lo = 0;
    uint4 pre0 = {}, pre1 = {}, pre2 = {}, pre3 = {}, pre4 = {}, pre5 = {};   // (named registers: an array ended up in scratch)
    // all four stage ends up front (scalar loads with the rest of the chunk record): a load of
    // stage_end[stage + 1] inside the loop put a full memory round trip in front of every prefetch
    const int se0 = chp->stage_end[0], se1 = chp->stage_end[1], se2 = chp->stage_end[2], se3 = chp->stage_end[3];
    static_assert(TCMI_F_MAXSTAGE >= 4, "stage ends are held in four scalars: format 1 has at most four stages per chunk");
    int st_begin = 0, st_end = se0;                              // word range of the stage (from word0)
    // every lane loads (indices clamped into the stage): no exec-masked branch, so the compiler can leave
    // the loads in flight across the inner loop instead of waiting at a branch join.  (A macro, not a
    // lambda: the closure kept `pre` in scratch memory.)
#define TCMI_ISSUE_STAGE(stage_, begin_, end_)                                                        \
    do {                                                                                              \
        const int r_ = min((stage_) * sub_reads + tid, n_reads - 1);                                  \
        h_pos = a.pos[read0 + r_];                                                                    \
        h_lo = a.lenoff[read0 + r_];                                                                  \
        const int mis_ = (int)((word0 + (begin_)) & 3); /* keep the 16-byte loads aligned */          \
        const uint4 *src_ = reinterpret_cast<const uint4 *>(a.seq + (word0 + (begin_) - mis_));       \
        const int last_ = ((end_) - (begin_) + mis_ + 3) / 4 - 1;                                     \
        if (NLD > 0) pre0 = src_[min(0 * FB + tid, last_)];                                           \
        if (NLD > 1) pre1 = src_[min(1 * FB + tid, last_)];                                           \
        if (NLD > 2) pre2 = src_[min(2 * FB + tid, last_)];                                           \
        if (NLD > 3) pre3 = src_[min(3 * FB + tid, last_)];                                           \
        if (NLD > 4) pre4 = src_[min(4 * FB + tid, last_)];                                           \
        if (NLD > 5) pre5 = src_[min(5 * FB + tid, last_)];                                           \
    } while (0)
    TCMI_ISSUE_STAGE(0, st_begin, st_end);

    for (int stage = 0; stage < n_stage; ++stage) {
        const int ns = min(sub_reads, n_reads - stage * sub_reads);
        const int mis = (int)((word0 + st_begin) & 3);
        const int tw = st_end - st_begin + mis;
        // ---- A: headers, coverage runs and bases of this stage -> LDS -------------------------------
        const bool valid = tid < ns;
        int rel = 0, len = 0;
        if (valid) {
            rel = h_pos - P0;
            len = (int)(h_lo & 1023u);
            const int off = (int)(h_lo >> 10) - st_begin + mis;   // word index of the read in s_seq
            s_hdr[tid] = make_uint2((uint32_t)rel | ((uint32_t)(len + 7) >> 3) << 16, (uint32_t)off * 4u);
        }
        if (tid == 0)                                            // dummy: a read far to the right, no words
            s_hdr[ns] = make_uint2(0x7FFFu, (uint32_t)(mis + PAD) * 4u);
        {
            uint4 *dst = reinterpret_cast<uint4 *>(s_seq);
            if (!(TCMI_ABL & 4) && NLD > 0 && (0 * FB + tid) * 4 < tw) dst[0 * FB + tid] = pre0;
            if (!(TCMI_ABL & 4) && NLD > 1 && (1 * FB + tid) * 4 < tw) dst[1 * FB + tid] = pre1;
            if (!(TCMI_ABL & 4) && NLD > 2 && (2 * FB + tid) * 4 < tw) dst[2 * FB + tid] = pre2;
            if (!(TCMI_ABL & 4) && NLD > 3 && (3 * FB + tid) * 4 < tw) dst[3 * FB + tid] = pre3;
            if (!(TCMI_ABL & 4) && NLD > 4 && (4 * FB + tid) * 4 < tw) dst[4 * FB + tid] = pre4;
            if (!(TCMI_ABL & 4) && NLD > 5 && (5 * FB + tid) * 4 < tw) dst[5 * FB + tid] = pre5;
        }
        __syncthreads();
        // ---- B: issue the next stage's loads; they complete while C runs ----------------------------
        if (stage + 1 < n_stage) {
            st_begin = st_end - PAD;
            st_end = stage == 0 ? se1 : stage == 1 ? se2 : se3;
            TCMI_ISSUE_STAGE(stage + 1, st_begin, st_end);
        }
        // coverage: one (+run, -run) pair per run of equal (pos, len) reads inside the wave; placed here so
        // that the LDS atomics complete under the inner loop instead of in front of the barrier
        {
            const int prel = __shfl_up(rel, 1, 64), plen = __shfl_up(len, 1, 64);
            const bool lead = valid && (lane == 0 || rel != prel || len != plen);
            const unsigned long long leads = __ballot(lead), valids = __ballot(valid);
            const unsigned long long above = lane == 63 ? 0ull : (leads >> (lane + 1)) << (lane + 1);
            const int next = above ? (__ffsll((long long)above) - 1) : __popcll(valids);
            if (lead && !(TCMI_ABL & 16)) {
                const int run = next - lane;
                atomicAdd(&s_cov[rel], run);
                atomicAdd(&s_cov[rel + len], -run);
            }
        }
        // ---- C: this lane's slice of the staged reads: r = s, s + S, s + 2S, ...  Branch-free: indices
        //      past the stage are clamped onto a dummy header whose read lies entirely in the zero padding.
        const int Rs = (ns + S - 1) / S;
        const int hbytes_end = ns * 8;
        int hb = s_eff * 8;                                    // byte offset of the lane's next header
        for (int k0 = 0; k0 < ((TCMI_ABL & 2) ? 0 : Rs); k0 += WIDEN) {
            const int k1 = min(k0 + WIDEN, Rs);
            for (int k = k0; k < k1; k += UNR) {
                uint2 h[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    h[u] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(s_hdr) + min(hb, hbytes_end));
                    hb += S * 8;
                }
                uint32_t w[UNR][NW + 1];
                uint32_t c4[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int d = base8 - (int)(h[u].x & 0xFFFFu);   // first owned position relative to the read start
                    // word of the read holding it, clamped into the zero padding on either side
                    const int q = max(-PAD, min(d >> 3, (int)(h[u].x >> 16)));
                    const uint32_t *wp = reinterpret_cast<const uint32_t *>(
                        reinterpret_cast<const char *>(s_seq) + (int)h[u].y + q * 4);
#pragma unroll
                    for (int j = 0; j <= NW; ++j) w[u][j] = wp[j];
                    c4[u] = (uint32_t)d << 2;                   // v_alignbit uses bits [4:0]: 4 * (d mod 8)
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u)
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const uint32_t A = __builtin_amdgcn_alignbit(w[u][j + 1], w[u][j], c4[u]);   // bases d+8j ..
#if TCMI_ABL & 1
                        nib[j][0] ^= A;
#else
#pragma unroll
                        for (int c = 0; c < 3; ++c) nib[j][c] += (A >> c) & 0x11111111u;
#endif
                    }
            }
            // the 4-bit counters may be full (<= WIDEN reads since the last widening): widen
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    byt[w][c][0] += nib[w][c] & 0x0F0F0F0Fu;
                    byt[w][c][1] += (nib[w][c] >> 4) & 0x0F0F0F0Fu;
                    nib[w][c] = 0;
                }
        }
        __syncthreads();                                        // every lane is done with this stage's LDS
    }
#if TCMI_ABL & 8
    if (byt[0][0][0] != 0x12345678u) return;
#endif
    // ---- slice partials -> LDS, layout [register j][lane] (conflict-free both ways) -----------------
    uint32_t *s_part = s_seq;
    // window counters of A, C, G (16-bit), behind the partials in the same buffer: 34 -> 30 KB of LDS per
    // workgroup, i.e. five workgroups per CU instead of four
    uint16_t (*s_fin)[MAXPOS] = reinterpret_cast<uint16_t (*)[MAXPOS]>(s_seq + NREG * FB);
    static_assert((NREG * FB + 3 * MAXPOS / 2) <= TCMI_F_SEQCAP, "partials and window counters must fit the stage buffer");
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) s_part[((w * 3 + c) * 2 + h) * FB + tid] = byt[w][c][h];
    __syncthreads();
    // ---- sum the slices; register j of group g holds 4 positions of one class -----------------------
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
        const int w = j / 6, c = (j >> 1) % 3, h = j & 1;
        const int p = (g * NW + w) * 8 + h;                     // byte i of the register <-> position p + 2*i
        if (p < npos) {                                         // (the last group's trailing words may lie outside)
            uint16_t *f = &s_fin[c][p];
            f[0] = (uint16_t)(e & 0xFFFFu);
            f[4] = (uint16_t)(e >> 16);
            f[2] = (uint16_t)(o & 0xFFFFu);
            f[6] = (uint16_t)(o >> 16);
        }
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
    // ---- one coalesced global atomic per touched (class, position) -----------------------------------
    for (int p = tid; p < npos; p += FB) {
        const int gp = P0 + p;
        if (gp >= a.L) continue;
        const int cv = s_cov[p];
        if (cv == 0) continue;
        const int nA = s_fin[0][p], nC = s_fin[1][p], nG = s_fin[2][p];
        const int nT = cv - nA - nC - nG;                       // includes the "other" bases, subtracted by the tail blocks
        atomicAdd(&a.counts[(int64_t)TCMI_COV * a.ld + gp], cv);
        if (nA) atomicAdd(&a.counts[(int64_t)TCMI_A * a.ld + gp], nA);
        if (nC) atomicAdd(&a.counts[(int64_t)TCMI_C * a.ld + gp], nC);
        if (nG) atomicAdd(&a.counts[(int64_t)TCMI_G * a.ld + gp], nG);
        if (nT) atomicAdd(&a.counts[(int64_t)TCMI_T * a.ld + gp], nT);
    }
    if constexpr (FUSED) {
        const int t0 = P0 / TILE;
        sign_off_and_call(a, (P0 + npos - 1) / TILE - t0 + 1, [&](int k) { return t0 + k; }, reinterpret_cast<int *>(s_hdr));
    }
}

#undef TCMI_ISSUE_STAGE

} // namespace

static int launch(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld, int32_t *d_counts, bool fused, int32_t *d_tile_done,
                  int64_t tile_cap, int32_t mincov, int include_ambig, uint8_t *plain, uint8_t *alt, uint8_t *flags)
{
    FastArgs a = {};
    a.pos = rs->d_fpos; a.lenoff = rs->d_flenoff; a.seq = rs->d_fseq; a.chunks = rs->d_fchunk; a.events = rs->d_fevent; a.covrun = rs->d_fcovrun;
    a.counts = d_counts; a.ld = ld; a.n_events = rs->f_events; a.L = (int32_t)L;
    a.other_col = rs->f_fmt == 2 ? TCMI_A : TCMI_T;
    a.pair_ok = (ld % 2 == 0) && (reinterpret_cast<uintptr_t>(d_counts) % 8 == 0);
    a.n_tail = (int32_t)((rs->f_events + FB - 1) / FB);
    int64_t grid = rs->f_chunks + a.n_tail;
    if (fused) {
        const int64_t tiles_L = (L + TILE - 1) / TILE;
        if (std::max<int64_t>(tiles_L, rs->f_tiles) > tile_cap) return tcmi_fail(ctx, TCMI_E_ARG, "workspace holds %lld tiles, need %lld", (long long)tile_cap, (long long)std::max<int64_t>(tiles_L, rs->f_tiles));
        a.tile_need = rs->d_ftile_need; a.tile_done = d_tile_done; a.ev_tile_off = rs->d_fev_tile_off; a.ev_tile = rs->d_fev_tile;
        a.orphans = rs->d_forphan; a.n_tiles = (int32_t)rs->f_tiles; a.n_orphans = (int32_t)rs->f_orphans;
        a.mincov = mincov; a.include_ambig = include_ambig; a.plain = plain; a.alt = alt; a.flags = flags;
        grid += rs->f_orphans + std::max<int64_t>(0, tiles_L - rs->f_tiles);
    }
    if (ctx->ride && !fused && !ctx->ride->taken) {            // carry another workspace's call in this launch
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
    if (rs->f_nw != 2) return tcmi_fail(ctx, TCMI_E_ARG, "read set was packed for %d grid words per lane", rs->f_nw);
    (void)hipGetLastError();                                   // drop any stale error of this thread
    if (rs->f_fmt == 2) tcmi_dispatch_tally_planes(a, (unsigned)grid, ctx->stream, fused);
    else if (fused) hipLaunchKernelGGL((tally_fast_kernel<2, true>), dim3((unsigned)grid), dim3(FB), 0, ctx->stream, a);
    else hipLaunchKernelGGL((tally_fast_kernel<2, false>), dim3((unsigned)grid), dim3(FB), 0, ctx->stream, a);   // (<4> measured slower)
    tcmi_prof_end(ctx, TCMI_K_TALLY);
    TCMI_HIP(ctx, hipGetLastError());
    return TCMI_OK;
}

int tcmi_launch_tally_fast(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld, int32_t *d_counts)
{
    return launch(ctx, rs, L, ld, d_counts, false, nullptr, 0, 0, 0, nullptr, nullptr, nullptr);
}

// Tally + call in one launch (see "fused call" above).  Only for read sets without a GENERAL part; `d_counts` and
// `d_tile_done` must be zero on entry and are zero again when the kernel has finished.
int tcmi_launch_step_fused(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld, int32_t *d_counts, int32_t *d_tile_done,
                           int64_t tile_cap, int32_t mincov, int include_ambig, uint8_t *plain, uint8_t *alt, uint8_t *flags)
{
    if (rs->g_reads) return tcmi_fail(ctx, TCMI_E_ARG, "fused step needs a read set without a general part");
    return launch(ctx, rs, L, ld, d_counts, true, d_tile_done, tile_cap, mincov, include_ambig, plain, alt, flags);
}
