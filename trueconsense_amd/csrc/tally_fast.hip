// tally_fast.hip — stage A for ALIGNED reads (one match op) on gfx950, without atomics in the
// inner loop.  Counterpart of the per-token loop of indexing.py:102-132 for the tokens that
// are plain bases: coverage += 1 and, if the base is A/C/G/T, that class += 1.
//
// Data (tcmi_internal.h): per read (pos, len) and its aligned bases as one-hot nibbles
// (A=1 C=2 G=4 T=8, anything else 0), 8 bases per 32-bit word.  The genome is cut in GRID WORDS
// of 8 positions; a CHUNK of <= 1024 coordinate-sorted reads touches a window of Wn <= 96 grid
// words (24 at 5000x coverage with 150-bp reads).
//
// One workgroup (256 lanes) per chunk:
//   lane (g, s)  owns TWO adjacent grid words (16 positions) g of the window and DEPTH SLICE s of
//                the reads; S = 256 / ceil(Wn/2) slices work in parallel on different reads.
//   stage        <= 256 reads at a time: headers -> LDS (block scan for word offsets), bases ->
//                LDS with 16-byte coalesced loads: every input byte leaves HBM once.
//   inner loop   per read of the slice: three LDS words -> two v_alignbit funnel shifts bring the
//                read's nibbles onto the lane's grid words; `& 0x11111111` of the word shifted by
//                0..3 isolates one class as eight 4-bit counters, added to 8 registers; every 15
//                reads the 4-bit counters are widened into 8-bit counters (16 registers).
//                No atomics, no branches on the data.
//   coverage     difference array in LDS, one (+run, -run) pair per run of equal (pos, len)
//                reads found with a wave ballot, then a block prefix sum.
//   reduce       slices are summed through LDS (plain stores / loads) into 16-bit window counters,
//                then ONE coalesced global atomic per touched (class, position) of the window.
//
// HBM-streaming integer work: no MFMA (BASELINE.json north_star).
#include "tcmi_internal.h"

namespace {

constexpr int FB = 256;                         // lanes per workgroup
constexpr int MAXPOS = TCMI_F_MAXW * 8;         // positions in the largest window

struct FastArgs {
    const int32_t *pos;
    const int32_t *len;
    const uint32_t *seq;
    const tcmi_fast_chunk *chunks;
    int32_t *counts;
    int64_t ld;
    int32_t L;
};

__device__ constexpr int plane_col(int k) { return k == 0 ? TCMI_A : k == 1 ? TCMI_C : k == 2 ? TCMI_G : TCMI_T; }

// inclusive block scan of one int over 256 lanes (4 waves)
__device__ inline int block_scan_incl(int v, int *wave_tot /* LDS [4] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    __syncthreads();                            // wave_tot may still be read from a previous scan
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) base += wave_tot[w];
    return v + base;
}

__global__ __launch_bounds__(FB) void tally_fast_kernel(FastArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_seq[TCMI_F_SEQCAP];   // staged bases; later the slice partials
    __shared__ uint2 s_hdr[TCMI_F_SUB];                                       // {pos - P0, word offset | nw << 20}
    __shared__ int32_t s_cov[MAXPOS + 8];                                     // coverage difference array
    __shared__ uint16_t s_fin[4][MAXPOS];                                     // window counters per class
    __shared__ int s_scan[4];
    __shared__ int s_total;

    const tcmi_fast_chunk ch = a.chunks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63;
    const int P0 = ch.P0, Wn = ch.Wn, npos = Wn * 8;
    const int Gn = (Wn + 1) >> 1;               // lane groups (two grid words each)
    const int S = FB / Gn;                      // depth slices
    const int s = tid / Gn, gi = tid - s * Gn;
    const bool active = s < S;
    const int base8 = gi * 16;                  // first position of the lane's words, relative to P0

    for (int i = tid; i <= npos; i += FB) s_cov[i] = 0;

    uint32_t nib[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};          // [word][class]: eight 4-bit counters
    uint32_t byt[2][4][2] = {};                                  // [word][class][even|odd position]: four 8-bit counters
    int since_flush = 0;
    int64_t word_base = ch.word0;

    for (int sub0 = 0; sub0 < ch.n_reads; sub0 += ch.sub_reads) {
        const int ns = min(ch.sub_reads, ch.n_reads - sub0);
        // ---- headers: one read per lane ---------------------------------------------------
        const bool valid = tid < ns;
        int rel = 0, len = 0;
        if (valid) {
            rel = a.pos[ch.read0 + sub0 + tid] - P0;
            len = a.len[ch.read0 + sub0 + tid];
        }
        const int nw = (len + 7) >> 3;
        const int incl = block_scan_incl(nw, s_scan);           // (two barriers: previous stage fully consumed)
        const int mis = (int)(word_base & 3);                    // keep the 16-byte loads aligned
        if (valid) s_hdr[tid] = make_uint2((uint32_t)rel, (uint32_t)(incl - nw + mis) | ((uint32_t)nw << 20));
        if (tid == FB - 1) s_total = incl;                       // total words of this stage
        // coverage: one (+run, -run) pair per run of equal (pos, len) inside the wave
        {
            const int prel = __shfl_up(rel, 1, 64), plen = __shfl_up(len, 1, 64);
            const bool lead = valid && (lane == 0 || rel != prel || len != plen);
            const unsigned long long leads = __ballot(lead), valids = __ballot(valid);
            const unsigned long long above = lane == 63 ? 0ull : (leads >> (lane + 1)) << (lane + 1);
            const int next = above ? (__ffsll((long long)above) - 1) : __popcll(valids);
            if (lead) {
                const int run = next - lane;
                atomicAdd(&s_cov[rel], run);
                atomicAdd(&s_cov[rel + len], -run);
            }
        }
        __syncthreads();
        const int total = s_total;
        const int tw = total + mis;
        // ---- bases: coalesced 16-byte loads into LDS ----------------------------------------
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.seq + (word_base - mis));
            uint4 *dst = reinterpret_cast<uint4 *>(s_seq);
            for (int i = tid; i * 4 < tw; i += FB) dst[i] = src[i];
        }
        word_base += total;
        __syncthreads();
        // ---- inner loop: this lane's slice of the staged reads ------------------------------
        const int Rs = (ns + S - 1) / S;
        const int r0 = s * Rs;
        for (int k = 0; k < Rs; ++k) {
            const int r = r0 + k;
            if (active && r < ns) {
                const uint2 h = s_hdr[r];
                const int off = (int)(h.y & 0xFFFFFu), rnw = (int)(h.y >> 20);
                const int d = base8 - (int)h.x;                 // first owned position relative to the read start
                const int q = d >> 3;                           // read word holding it (floor)
                const uint32_t c4 = (uint32_t)(d & 7) << 2;
                const bool v0 = (unsigned)q < (unsigned)rnw, v1 = (unsigned)(q + 1) < (unsigned)rnw,
                           v2 = (unsigned)(q + 2) < (unsigned)rnw;
                uint32_t w0 = s_seq[off + (v0 ? q : 0)];
                uint32_t w1 = s_seq[off + (v1 ? q + 1 : 0)];
                uint32_t w2 = s_seq[off + (v2 ? q + 2 : 0)];
                w0 = v0 ? w0 : 0u;
                w1 = v1 ? w1 : 0u;
                w2 = v2 ? w2 : 0u;
                const uint32_t A0 = __builtin_amdgcn_alignbit(w1, w0, c4);      // bases d .. d+7
                const uint32_t A1 = __builtin_amdgcn_alignbit(w2, w1, c4);      // bases d+8 .. d+15
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    nib[0][c] += (A0 >> c) & 0x11111111u;
                    nib[1][c] += (A1 >> c) & 0x11111111u;
                }
            }
            if (++since_flush == 15) {                          // 4-bit counters are full: widen
                since_flush = 0;
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        byt[w][c][0] += nib[w][c] & 0x0F0F0F0Fu;
                        byt[w][c][1] += (nib[w][c] >> 4) & 0x0F0F0F0Fu;
                        nib[w][c] = 0;
                    }
            }
        }
    }
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            byt[w][c][0] += nib[w][c] & 0x0F0F0F0Fu;
            byt[w][c][1] += (nib[w][c] >> 4) & 0x0F0F0F0Fu;
        }
    __syncthreads();                                            // every lane is done with s_seq
    // ---- slice partials -> LDS, layout [register j][lane] (conflict-free both ways) -----------
    uint32_t *s_part = s_seq;
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) s_part[((w * 4 + c) * 2 + h) * FB + tid] = byt[w][c][h];
    __syncthreads();
    // ---- sum the slices; register j of group gi holds 4 positions of one class ----------------
    for (int item = tid; item < Gn * 16; item += FB) {
        const int j = item / Gn, g = item - j * Gn;
        uint32_t e = 0, o = 0;                                  // bytes 0,2 and bytes 1,3 as 16-bit sums
        for (int t = 0; t < S; ++t) {
            const uint32_t v = s_part[j * FB + t * Gn + g];
            e += v & 0x00FF00FFu;
            o += (v >> 8) & 0x00FF00FFu;
        }
        const int w = j >> 3, c = (j >> 1) & 3, h = j & 1;
        const int p = (g * 2 + w) * 8 + h;                      // byte i of the register <-> position p + 2*i
        if (p < npos) {                                         // (odd Wn: the last lane group's 2nd word is outside)
            uint16_t *f = &s_fin[c][p];
            f[0] = (uint16_t)(e & 0xFFFFu);
            f[4] = (uint16_t)(e >> 16);
            f[2] = (uint16_t)(o & 0xFFFFu);
            f[6] = (uint16_t)(o >> 16);
        }
    }
    // ---- coverage: inclusive prefix sum of the difference array, 3 entries per lane ------------
    {
        const int i0 = tid * 3;
        const int d0 = i0 < npos ? s_cov[i0] : 0, d1 = i0 + 1 < npos ? s_cov[i0 + 1] : 0,
                  d2 = i0 + 2 < npos ? s_cov[i0 + 2] : 0;
        const int incl = block_scan_incl(d0 + d1 + d2, s_scan);
        const int before = incl - (d0 + d1 + d2);
        __syncthreads();
        if (i0 < npos) s_cov[i0] = before + d0;
        if (i0 + 1 < npos) s_cov[i0 + 1] = before + d0 + d1;
        if (i0 + 2 < npos) s_cov[i0 + 2] = before + d0 + d1 + d2;
    }
    __syncthreads();
    // ---- one coalesced global atomic per touched (class, position) ------------------------------
    for (int p = tid; p < npos; p += FB) {
        const int gp = P0 + p;
        if (gp >= a.L) continue;
        const int cv = s_cov[p];
        if (cv) atomicAdd(&a.counts[(int64_t)TCMI_COV * a.ld + gp], cv);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int v = s_fin[c][p];
            if (v) atomicAdd(&a.counts[(int64_t)plane_col(c) * a.ld + gp], v);
        }
    }
}

} // namespace

int tcmi_launch_tally_fast(tcmi_ctx *ctx, const tcmi_readset *rs, int64_t L, int64_t ld, int32_t *d_counts)
{
    FastArgs a;
    a.pos = rs->d_fpos; a.len = rs->d_flen; a.seq = rs->d_fseq; a.chunks = rs->d_fchunk;
    a.counts = d_counts; a.ld = ld; a.L = (int32_t)L;
    if (rs->f_chunks > INT32_MAX) return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "too many chunks");
    tcmi_prof_begin(ctx, TCMI_K_TALLY);
    hipLaunchKernelGGL(tally_fast_kernel, dim3((unsigned)rs->f_chunks), dim3(FB), 0, ctx->stream, a);
    tcmi_prof_end(ctx, TCMI_K_TALLY);
    TCMI_HIP(ctx, hipGetLastError());
    return TCMI_OK;
}
