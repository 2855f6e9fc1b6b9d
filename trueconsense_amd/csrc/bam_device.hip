// bam_device.hip — DEVICE: a BAM file's BGZF blocks -> the inflated BAM byte stream + the offset of every alignment
// record, in HBM, without the host ever decoding the file (pysam / htslib's role for indexing.py:19,96-100; wire format
// SAM spec §4.1 BGZF, §4.2 BAM, RFC 1951 DEFLATE).
//
// The host only reads the file, walks the gzip member headers (18 bytes per block) and inflates the first block(s) far
// enough to parse the BAM header; the compressed bytes (a few MB .. tens of MB) cross PCIe once.
//
//   (bgzf_decode.hip)  bgzf_symbols + bgzf_copy: the compressed blocks -> the inflated stream and every block's record starts
//                  (bgzf_copy also takes every block's CRC-32 against its trailer while it flushes the bytes: no pass of its own)
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

// per-block record lists -> dense offsets into the stream; base[b] = exclusive scan of n_rec (done by one workgroup first)
// A block whose last record starts within its last three bytes could not read that record's size: it is read here, from the stream
// (`overshoot` = TAIL_UNKNOWN on entry: the record's start is the block's last listed one).
constexpr int32_t TAIL_UNKNOWN = 0x7FFFFFFF;
__global__ __launch_bounds__(1024) void rec_scan(uint32_t *n_rec, uint64_t *base, int32_t n_blocks, int32_t n_own, unsigned long long *total,
                                                 const BlockDesc *blocks, const uint32_t *rec_slot, const uint8_t *out, int32_t *overshoot)
{
    __shared__ unsigned long long s[1024];
    const int t = threadIdx.x;
    const int per = (n_blocks + 1023) / 1024, b0 = t * per, b1 = min(b0 + per, n_blocks);
    unsigned long long sum = 0;
    for (int b = b0; b < b1; ++b) {
        if (b >= n_own) n_rec[b] = 0;                           // (a block taken along for the tail of the last record: its records are not ours)
        sum += n_rec[b];
        if (overshoot[b] == TAIL_UNKNOWN && n_rec[b]) {
            const uint32_t at = rec_slot[(size_t)b * MAX_REC_PER_BLOCK + n_rec[b] - 1];
            const uint8_t *q = out + blocks[b].uout + at;
            const uint32_t bs = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
            overshoot[b] = bs - 32u > (1u << 28) - 32u ? -1 : (int32_t)(at + 4u + bs - blocks[b].ulen);     // (-1: no such record)
        }
    }
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
    uint8_t *d_bytes = nullptr;                 // the same `cap` bytes in HBM (tcmi_bamfile_to_device), or null
    size_t desc_at = 0;                         // the block table (BlockDesc[]) lies behind the file's bytes, at this offset of `bytes` / `d_bytes` (0: it does not)
    int d_device = -1;
    size_t n_bytes = 0, cap = 0;                // cap: bytes that go to the device (file + zeroed slack)
    size_t pool_cap = 0;                        // bytes of the pinned allocation
    std::vector<BlockDesc> blocks;
    size_t inflated = 0;                        // bytes of the stream as laid out on the device (blocks padded to 16 bytes)
    size_t tok_total = 0;                       // tokens reserved for all blocks (bgzf_symbols -> bgzf_copy)
    uint32_t pay_dwords = 0;                    // the largest block's payload in dwords + slack (bgzf_symbols' dynamic LDS)
    uint32_t rec_bytes_hint = 0;                // mean bytes of the alignment records behind the header in the blocks the host inflated (0: too few seen)
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
    if (f->d_bytes) (void)hipFree(f->d_bytes);
    delete f;
    return TCMI_OK;
}

// The file's compressed bytes into HBM, to stay there: tcmi_readset_from_bamfile[_blocks] of this file then starts from device
// memory (no PCIe copy per call) — the form in which a file arrives that a peer GPU, a NIC or a storage engine wrote into HBM,
// and the one bench.py's headline times ("inputs resident in HBM when the timed region starts").
int tcmi_bamfile_to_device(tcmi_ctx *ctx, tcmi_bamfile *f)
{
    if (!ctx || !f) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    if (f->d_bytes && f->d_device == ctx->device) return TCMI_OK;
    if (f->d_bytes) { (void)hipFree(f->d_bytes); f->d_bytes = nullptr; }
    if (hipMalloc((void **)&f->d_bytes, f->cap) != hipSuccess) {
        (void)hipGetLastError();
        f->d_bytes = nullptr;
        return tcmi_fail(ctx, TCMI_E_NOMEM, "hipMalloc(%zu) for the bytes of %s failed", f->cap, f->path.c_str());
    }
    f->d_device = ctx->device;
    TCMI_HIP(ctx, hipMemcpyAsync(f->d_bytes, f->bytes, f->cap, hipMemcpyHostToDevice, ctx->stream));
    TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TCMI_OK;
}

const char *tcmi_bamfile_path(const tcmi_bamfile *f) { return f ? f->path.c_str() : ""; }

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
    // (+ room for the block table behind the bytes, so that ONE copy takes both to the device: a block is at least 28 bytes, usually ~ 2 KB and more)
    const size_t table_room = std::min<size_t>(((size_t)sz / 28 + 2) * sizeof(BlockDesc), ((size_t)sz / 512 + 64) * sizeof(BlockDesc));
    f->bytes = pinned_pool().take(f->cap + table_room + 1024, &f->pool_cap);
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
    const size_t table_cap = table_room;
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
        b.entry = -2;                                           // (-2: the block's first record starts where the device finds it)
        if (b.ulen > 65536) return bail(TCMI_E_FORMAT, "BGZF block inflates to more than 64 KiB", off);
        // tokens: one per literal / match (each gives >= 1 byte and takes >= 1 bit), one per <= 8 191 stored bytes (a stored
        // deflate block takes >= 5 bytes)
        b.tok_cap = std::min(b.ulen, 8u * b.clen) + b.clen / 2 + 8;
        b.tok = f->tok_total;
        f->tok_total += (2u * b.tok_cap + 3u) & ~3u;        // (as many again behind them: bgzf_symbols' scratch)
        f->pay_dwords = std::max(f->pay_dwords, (uint32_t)(((b.cin & 3u) * 8u + b.clen * 8u + 31u) / 32u + 6u));
        uout += (size_t)b.ulen;                                 // the blocks' outputs follow each other without gaps: the stream as it inflates
        off += bsize;
        f->blocks.push_back(b);
    }
    f->inflated = uout;
    if (f->blocks.size() * sizeof(BlockDesc) <= table_cap) f->desc_at = (f->cap + 255) & ~(size_t)255;     // (else: blocks of < 512 bytes — the table goes by a copy of its own)
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
    {   // what is left of the inflated bytes behind the header are the file's first records: their mean size sizes the one-sync path's arrays
        size_t at = o, cnt = 0;
        while (at + 4 <= head.size()) {
            const size_t bs = rd32(head.data() + at);
            if (bs < 32 || bs > (1u << 24) || at + 4 + bs > head.size()) break;
            at += 4 + bs;
            ++cnt;
        }
        if (cnt >= 16) f->rec_bytes_hint = (uint32_t)((at - o) / cnt);
    }
    if (f->desc_at) {
        std::memcpy(f->bytes + f->desc_at, f->blocks.data(), f->blocks.size() * sizeof(BlockDesc));
        f->cap = f->desc_at + ((f->blocks.size() * sizeof(BlockDesc) + 255) & ~(size_t)255);      // what goes to the device: bytes, slack, table
    }
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

// ---- device decode: H2D of the compressed file, bgzf_symbols + bgzf_copy (all in the context's arena) ---------------------------------
namespace {
struct DeviceBam { uint8_t *d_out = nullptr; uint64_t *d_rec = nullptr; BlockDesc *d_desc = nullptr; size_t n = 0; int64_t range_first = -1, range_next = -1; };

// a decode in flight: the range as a file of its own, and where its pieces lie in the arena
struct Decoded {
    tcmi_bamfile part;                  // (a block range: its blocks' places in the stream and in the token array start at 0; the bytes stay the whole file's)
    const tcmi_bamfile *f = nullptr;
    bool ranged = false;
    size_t nb = 0, nb_own = 0, max_rec = 0, guess_rec = 0, b_rest = 0;
    uint8_t *d_file = nullptr, *d_out = nullptr;
    BlockDesc *d_desc = nullptr;
    uint32_t *d_slot = nullptr, *d_nrec = nullptr, *d_stat = nullptr, *d_first = nullptr;
    int32_t *d_over = nullptr;
    uint64_t *d_base = nullptr;
    unsigned long long *d_total = nullptr;
};
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
// arena bytes of everything that is sized per record: the several-kernel path's arrays (rec_off + 12 words) — the one-pass packer's are fewer
inline size_t rest_bytes(size_t n_rec, size_t nb) { return al256(n_rec * 8 + 8) + al256(n_rec * 4 + 4) * 12 + al256((n_rec / 256 + 2) * 8) * 3 + al256(nb * 33 + 64) + al256(nb * 8) * 4 + al256(nb * 4) + 8192 + 32 * 256; }

// blocks [first, first + count) of the file (count < 0: to the end): the records that START in them.  A range that does not end
// with the file takes one block more along — the last record may run into it — whose own records are left out.
// Queues the copies and the three kernels on the context's stream; waits for nothing.
int decode_enqueue(tcmi_ctx *ctx, const tcmi_bamfile *whole, Decoded &D, int64_t first_blk = 0, int64_t count = -1)
{
    if (whole->blocks.empty()) return tcmi_fail(ctx, TCMI_E_FORMAT, "%s: no BGZF blocks", whole->path.c_str());
    const int64_t all = (int64_t)whole->blocks.size();
    if (first_blk < 0 || first_blk > all) return tcmi_fail(ctx, TCMI_E_ARG, "block range starts at %lld, the file has %lld blocks", (long long)first_blk, (long long)all);
    const int64_t own = count < 0 ? all - first_blk : std::min<int64_t>(count, all - first_blk);
    const bool ranged = first_blk != 0 || own != all;
    D.ranged = ranged;
    // the range as a file of its own: the blocks' places in the stream and in the token array start at 0
    tcmi_bamfile &part = D.part;
    const tcmi_bamfile *f = whole;
    if (ranged) {
        const int64_t extra = first_blk + own < all ? 1 : 0;
        part.path = whole->path; part.ref_name = whole->ref_name; part.ref_len = whole->ref_len;
        part.blocks.assign(whole->blocks.begin() + first_blk, whole->blocks.begin() + first_blk + own + extra);
        part.pay_dwords = whole->pay_dwords;
        // (only the range's bytes cross PCIe: from the 16-byte boundary in front of its first payload to behind its last trailer)
        const size_t lo = part.blocks.empty() ? 0 : (size_t)part.blocks.front().cin & ~(size_t)15;
        const size_t hi = part.blocks.empty() ? 0 : (size_t)part.blocks.back().cin + part.blocks.back().clen + 8;
        part.bytes = whole->bytes + lo; part.d_bytes = whole->d_bytes ? whole->d_bytes + lo : nullptr; part.d_device = whole->d_device;
        part.n_bytes = hi - lo; part.cap = std::min(whole->cap - lo, ((hi - lo) + 4096 + 15) & ~(size_t)15);
        size_t uout = 0;
        for (BlockDesc &b : part.blocks) {
            b.cin -= lo;
            b.uout = uout; uout += b.ulen;
            b.tok = part.tok_total; part.tok_total += (2u * b.tok_cap + 3u) & ~3u;
        }
        part.inflated = uout;
        f = &part;
    }
    D.f = f;
    const size_t nb = f->blocks.size(), nb_own = ranged ? (size_t)own : nb;
    D.nb = nb; D.nb_own = nb_own;
    if (nb == 0) return TCMI_OK;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // bounds on the records for the arena: a record takes at least 36 bytes of the stream; reserve for records of >= 64 bytes
    // (block_size + 32 fixed bytes + name + CIGAR + SEQ + QUAL of a 15-base read) — the arena cannot grow under live data
    D.max_rec = f->inflated / 36 + 16;
    D.guess_rec = std::min(D.max_rec, f->inflated / 64 + 1024);
    const bool resident = f->d_bytes != nullptr && f->d_device == ctx->device;       // (the compressed bytes are in HBM already)
    // The tokens of a block in flight take 3 - 10 times its inflated bytes (a token per literal, as many again of scratch for the
    // speculating lanes): a file whose blocks need more than the context's token scratch ("decode_token_mb") is decoded in batches
    // of blocks that share it, one pair of launches per batch — the inflated stream, the record slots and the verdicts stay whole.
    const uint64_t tok_cap_words = (uint64_t)std::max<int64_t>(ctx->decode_token_mb, 1) << 18;
    std::vector<size_t> batch_at;                               // first block of every batch (+ the end)
    uint64_t tok_words = f->tok_total;
    if (f->tok_total > tok_cap_words) {
        tok_words = 0;
        uint64_t at = f->blocks[0].tok;
        batch_at.push_back(0);
        for (size_t b = 0; b < nb; ++b) {
            const uint64_t e = b + 1 < nb ? f->blocks[b + 1].tok : f->tok_total;
            if (e - at > tok_cap_words && b > batch_at.back()) { tok_words = std::max(tok_words, f->blocks[b].tok - at); batch_at.push_back(b); at = f->blocks[b].tok; }
        }
        tok_words = std::max(tok_words, f->tok_total - at);
        batch_at.push_back(nb);
    }
    const size_t b_file = resident ? 256 : al(f->cap), b_desc = al(nb * sizeof(BlockDesc)), b_out = al(f->inflated + 128),
                 b_slot = al(nb * (size_t)MAX_REC_PER_BLOCK * 4), b_small = al(nb * 4) * 5 + al(nb * 8) + al(nb * 512) + 256,
                 b_tok = al(tok_words * 4 + 256);
    D.b_rest = rest_bytes(D.guess_rec, nb);
    if (!tcmi_arena_reserve_take(ctx, b_file + b_desc + b_out + b_slot + b_small + b_tok + D.b_rest + 16 * 256, 0)) return TCMI_E_NOMEM;
    uint8_t *d_file = (uint8_t *)tcmi_arena_take(ctx, b_file);
    D.d_desc = (BlockDesc *)tcmi_arena_take(ctx, b_desc);
    D.d_out = (uint8_t *)tcmi_arena_take(ctx, b_out);
    D.d_slot = (uint32_t *)tcmi_arena_take(ctx, b_slot);
    D.d_nrec = (uint32_t *)tcmi_arena_take(ctx, al(nb * 4));
    D.d_over = (int32_t *)tcmi_arena_take(ctx, al(nb * 4));
    D.d_stat = (uint32_t *)tcmi_arena_take(ctx, al(nb * 4));
    D.d_first = (uint32_t *)tcmi_arena_take(ctx, al(nb * 4));
    D.d_base = (uint64_t *)tcmi_arena_take(ctx, al(nb * 8));
    D.d_total = (unsigned long long *)tcmi_arena_take(ctx, 256);
    uint32_t *d_ntok = (uint32_t *)tcmi_arena_take(ctx, al(nb * 4));
    uint32_t *d_tok = (uint32_t *)tcmi_arena_take(ctx, b_tok);

    // Option "h2d_pieces" (off by default): many compressed bytes from the host (a rank's range of a large file: 48 MB for 6.25 M reads, a
    // millisecond of PCIe) go out in pieces of whole blocks on a stream of its own, and the inflate kernels of a piece's blocks wait
    // only for their piece.  Measured on a 6.25 M-read range (profiles/r06d_*): eight pieces 4.5 ms against 3.97 ms in one copy — every
    // piece is a pair of launches with a tail of its own and a hand-over between the copy engine and the compute queue; the
    // sub-ranges of tcmi_split_step ("split_sub") hide the copy better (3.5 ms).  Kept as an option for links slower than this one.
    static const int pipe_env = std::getenv("TCMI_H2D_PIECES") ? std::atoi(std::getenv("TCMI_H2D_PIECES")) : 0;      // (A/B: -1 = never, n = that many pieces)
    const int pipe_opt = pipe_env ? pipe_env : ctx->h2d_pieces;
    int pieces = 0;
    if (!resident && batch_at.empty() && nb >= 64 && pipe_opt >= 0)
        pieces = pipe_opt > 0 ? std::min(pipe_opt, 16) : 0;
    if (pieces >= 2 && !ctx->copy_stream) {
        if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->copy_stream = nullptr; pieces = 0; }
    }
    while (pieces >= 2 && (int)ctx->ev_piece.size() < pieces + 1) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); pieces = 0; break; }
        ctx->ev_piece.push_back(e);
    }
    const bool table_behind = !ranged && f->desc_at;
    if (resident) d_file = f->d_bytes;
    else if (pieces < 2) TCMI_HIP(ctx, hipMemcpyAsync(d_file, f->bytes, f->cap, hipMemcpyHostToDevice, ctx->stream));
    D.d_file = d_file;
    // the block table: behind the file's bytes (one copy took both, or none: a resident file's table is resident too) — a range's
    // table, re-based, goes by a copy of its own
    if (table_behind) {
        D.d_desc = reinterpret_cast<BlockDesc *>(d_file + f->desc_at);
        if (pieces >= 2) TCMI_HIP(ctx, hipMemcpyAsync(d_file + f->desc_at, f->bytes + f->desc_at, f->cap - f->desc_at, hipMemcpyHostToDevice, ctx->stream));
    } else {
        // (a range's re-based table: through pinned memory of the context's own — a copy from the pageable vector is staged by the runtime
        //  while the host waits, 0.1 - 0.2 ms for a 6 M-read range's 26 000 blocks, in every rank's step)
        const size_t tb = nb * sizeof(BlockDesc);
        if (ctx->h_desc_cap < tb) {
            if (ctx->h_desc) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->h_desc); ctx->h_desc = nullptr; ctx->h_desc_cap = 0; }
            const size_t want = tb + tb / 4 + 4096;
            if (hipHostMalloc((void **)&ctx->h_desc, want, hipHostMallocDefault) == hipSuccess) ctx->h_desc_cap = want;
            else { (void)hipGetLastError(); ctx->h_desc = nullptr; }
        }
        const void *src = f->blocks.data();
        if (ctx->h_desc) {
            // (the last range's copy out of this buffer is behind us: every decode ends with a wait of the host for its stream)
            std::memcpy(ctx->h_desc, f->blocks.data(), tb);
            src = ctx->h_desc;
        }
        TCMI_HIP(ctx, hipMemcpyAsync(D.d_desc, src, tb, hipMemcpyHostToDevice, ctx->stream));
    }
    (void)hipGetLastError();
    {
        tcmi_bgzf_decode_args g;
        g.d_file = d_file; g.d_desc = D.d_desc; g.d_tok = d_tok; g.d_ntok = d_ntok; g.d_out = D.d_out; g.d_slot = D.d_slot; g.d_nrec = D.d_nrec;
        g.d_over = D.d_over; g.d_first = D.d_first; g.d_stat = D.d_stat; g.n_blocks = nb; g.pay_dwords = f->pay_dwords;
        g.n_ref = (uint32_t)f->ref_name.size();
        g.verify_crc = ctx->verify_crc;
        g.scratch_div = ctx->sym_scratch_div;
        g.short_tokens = f->inflated < 4 * f->n_bytes ? 2 : f->inflated < 12 * f->n_bytes ? 1 : 0;
        if (pieces >= 2) {
            // (the copy stream starts behind whatever this context's stream still holds: the arena's last user)
            TCMI_HIP(ctx, hipEventRecord(ctx->ev_piece[(size_t)pieces], ctx->stream));
            TCMI_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_piece[(size_t)pieces], 0));
            const size_t top = table_behind ? f->desc_at : f->cap;
            for (int k = 0; k < pieces; ++k) {
                const size_t b0 = nb * (size_t)k / (size_t)pieces, b1 = nb * (size_t)(k + 1) / (size_t)pieces;
                // bytes of blocks [b0, b1): from a little in front of the first payload (16-byte aligned) to a little behind the last trailer —
                // bgzf_symbols stages up to 24 bytes past a payload, bgzf_copy reads the trailer; neighbouring pieces overlap by those bytes
                const size_t lo = k == 0 ? 0 : (f->blocks[b0].cin > 64 ? (f->blocks[b0].cin - 64) & ~(size_t)15 : 0);
                const size_t hi = k == pieces - 1 ? top : std::min(top, (f->blocks[b1 - 1].cin + f->blocks[b1 - 1].clen + 8 + 64 + 15) & ~(size_t)15);
                if (hi > lo) TCMI_HIP(ctx, hipMemcpyAsync(d_file + lo, f->bytes + lo, hi - lo, hipMemcpyHostToDevice, ctx->copy_stream));
                TCMI_HIP(ctx, hipEventRecord(ctx->ev_piece[(size_t)k], ctx->copy_stream));
                TCMI_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_piece[(size_t)k], 0));
                g.first_block = b0; g.count = b1 - b0; g.tok_base = 0;
                if (b1 > b0) { const int rc = tcmi_bgzf_decode_launch(ctx, g); if (rc) return rc; }
            }
            ++ctx->stat_h2d_piped;
        } else if (batch_at.empty()) {
            const int rc = tcmi_bgzf_decode_launch(ctx, g);
            if (rc) return rc;
        } else {
            for (size_t k = 0; k + 1 < batch_at.size(); ++k) {  // (one after the other on the stream: the next batch's tokens overwrite this one's)
                g.first_block = batch_at[k]; g.count = batch_at[k + 1] - batch_at[k]; g.tok_base = f->blocks[batch_at[k]].tok;
                const int rc = tcmi_bgzf_decode_launch(ctx, g);
                if (rc) return rc;
            }
            ++ctx->stat_decode_batched;
        }
    }
    return TCMI_OK;
}

// The several-kernel path behind decode_enqueue: the block verdicts and the record chain come back to the host (one wait), which
// words every refusal; then the dense record offsets.
int decode_on_device(tcmi_ctx *ctx, const tcmi_bamfile *whole, DeviceBam *Dout, int64_t first_blk = 0, int64_t count = -1)
{
    Decoded D;
    int rc = decode_enqueue(ctx, whole, D, first_blk, count);
    if (rc) return rc;
    const tcmi_bamfile *f = D.f;
    const size_t nb = D.nb, nb_own = D.nb_own;
    const bool ranged = D.ranged;
    if (nb == 0) { Dout->d_out = nullptr; Dout->d_rec = nullptr; Dout->d_desc = nullptr; Dout->n = 0; return TCMI_OK; }
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    uint8_t *d_out = D.d_out;
    BlockDesc *d_desc = D.d_desc;
    uint32_t *d_slot = D.d_slot, *d_nrec = D.d_nrec, *d_stat = D.d_stat, *d_first = D.d_first;
    int32_t *d_over = D.d_over;
    uint64_t *d_base = D.d_base;
    unsigned long long *d_total = D.d_total;
    const size_t max_rec = D.max_rec, b_rest = D.b_rest;
    hipLaunchKernelGGL(rec_scan, dim3(1), dim3(1024), 0, ctx->stream, d_nrec, d_base, (int32_t)nb, (int32_t)nb_own, d_total, d_desc, d_slot, d_out, d_over);
    TCMI_HIP(ctx, hipGetLastError());
    // the verdict of every block comes back to the host: a few bytes per block
    char *pin = (char *)tcmi_ctx_pinned(ctx, nb * 12 + 16);
    if (!pin) return tcmi_fail(ctx, TCMI_E_NOMEM, "pinned scratch for %zu block verdicts", nb);
    const uint32_t *stat = (const uint32_t *)(pin + 16), *first = stat + nb;
    const int32_t *over = (const int32_t *)(first + nb);
    TCMI_HIP(ctx, hipMemcpyAsync((void *)stat, d_stat, nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    TCMI_HIP(ctx, hipMemcpyAsync((void *)first, d_first, nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    TCMI_HIP(ctx, hipMemcpyAsync((void *)over, d_over, nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    TCMI_HIP(ctx, hipMemcpyAsync(pin, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long total = *(const unsigned long long *)pin;
    for (size_t b = 0; b < nb; ++b)
        if (stat[b] == ST_BAD_STREAM || stat[b] == ST_BAD_LENGTH)
            return tcmi_fail(ctx, TCMI_E_FORMAT, "%s: BGZF block %zu failed to inflate (deflate stream or ISIZE damaged)", f->path.c_str(), b);
    for (size_t b = 0; b < nb; ++b)
        if (stat[b] == ST_BAD_CRC)
            return tcmi_fail(ctx, TCMI_E_FORMAT, "%s: CRC32 mismatch in BGZF block %zu", f->path.c_str(), b);
    // The record chain.  Every block found the first record start in its own bytes by itself — where the header says (the first
    // record), or the first offset at which a plausible record starts (htslib cuts its blocks on record boundaries: offset 0;
    // other writers fill them to the brim) — and followed the chain of block_size fields from there.  In block order: if every
    // block's find is where its predecessor's last record ends, all of them are record starts, by induction from the header.
    {
        int64_t expect = -1;                                    // offset in the next block at which a record must start
        bool open = ranged;                                     // (a range: wherever its first block found one)
        for (size_t b = 0; b < nb_own; ++b) {
            const BlockDesc &d = f->blocks[b];
            if (d.entry == -1) continue;                        // header only
            if (open && first[b] != 0xFFFFFFFFu) { expect = first[b]; open = false; if (d.entry < 0) Dout->range_first = (int64_t)d.uout + first[b]; }
            if (open) continue;
            if (stat[b] == ST_BAD_RECORD)
                return tcmi_fail(ctx, TCMI_E_FORMAT, "%s: alignment record with an impossible block_size in BGZF block %zu", f->path.c_str(), b);
            if (d.entry >= 0) expect = d.entry;
            if (first[b] == 0xFFFFFFFFu) {                      // no record starts in this block: it lies inside one, or is empty
                if (expect < (int64_t)d.ulen)
                    return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "%s: no alignment record found where one must start in BGZF block %zu: host reader", f->path.c_str(), b);
                expect -= d.ulen;
                continue;
            }
            if ((int64_t)first[b] != expect || over[b] < 0)
                return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "%s: the chain of alignment records does not close at BGZF block %zu (found a start at %u, expected %lld): host reader",
                                 f->path.c_str(), b, first[b], (long long)expect);
            expect = over[b];
        }
        if (!open && nb_own > 0) Dout->range_next = (int64_t)(f->blocks[nb_own - 1].uout + f->blocks[nb_own - 1].ulen) + expect;
        if (nb_own < nb) {                                      // the range's last record must end in the block taken along
            if (expect > (int64_t)f->blocks[nb_own].ulen)
                return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "%s: a record longer than a BGZF block at the end of a block range: host reader", f->path.c_str());
            expect = 0;
        }
        if (expect > 0)
            return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "%s: the last alignment record runs %lld bytes past the end of the file: host reader", f->path.c_str(), (long long)expect);
    }
    if (total > max_rec) return tcmi_fail(ctx, TCMI_E_FORMAT, "%s: impossible record count", f->path.c_str());
    const size_t n = (size_t)total;
    const size_t need_rest = rest_bytes(n, nb);
    if (need_rest > b_rest)
        return tcmi_fail(ctx, TCMI_E_UNSUPPORTED, "%s: %zu very short records need more device scratch than was reserved: host reader", f->path.c_str(), n);
    uint64_t *d_rec = (uint64_t *)tcmi_arena_take(ctx, al(n * 8 + 8));
    tcmi_prof_begin(ctx, TCMI_K_RECORDS);
    if (n) hipLaunchKernelGGL(rec_compact, dim3((unsigned)nb_own), dim3(256), 0, ctx->stream, d_desc, d_slot, d_nrec, d_base, d_rec);
    tcmi_prof_end(ctx, TCMI_K_RECORDS);
    TCMI_HIP(ctx, hipGetLastError());
    Dout->d_out = d_out; Dout->d_rec = d_rec; Dout->d_desc = d_desc; Dout->n = n;
    return TCMI_OK;
}
} // namespace

extern "C" {

// BAM bytes -> read set in HBM, everything on the device: H2D of the compressed file, bgzf_inflate, record index,
// pack_device.hip.  TCMI_E_UNSUPPORTED when the file needs the host reader (records that straddle BGZF blocks, entries
// longer than 512 positions, ...): the caller falls back to tcmi_bam_load + tcmi_readset_upload.
int tcmi_readset_from_bamfile(tcmi_ctx *ctx, const tcmi_bamfile *f, tcmi_readset **out, int64_t *n_reads_out)
{
    return tcmi_readset_from_bamfile_blocks(ctx, f, 0, -1, out, n_reads_out);
}

// ... of the records that start in BGZF blocks [first_block, first_block + n_blocks) only (n_blocks < 0: to the end of the file): ranks
// that share ONE file each take a contiguous range of its blocks (tcmi_bamfile_info says how many there are) and decode nothing
// else; their count matrices add up to the file's (BASELINE configs[4]; indexing.py:96).
//
// Two ways through.  The ONE-SYNC path (default): decode, record index, chain check, classification, prefix sums, planes and chunk
// planning are all queued from capacities (decode_enqueue, tcmi_pack_fused_enqueue: pk_index + pk_place + pk_pack), the host waits once and
// checks what was deferred.  Whatever that path cannot vouch for — a damaged block, a chain that does not close, reads it does not
// take, a file beyond the capacities — is decoded again by the SEVERAL-KERNEL path below it (three waits), which words every refusal.
}   // extern "C"

namespace {
std::atomic<uint64_t> g_next_uid{1ull << 40};

// A context with two streams (TCMI_STREAM_SPLIT): the launches behind the inflate go to the high stream, which waits for the low
// one's decode; the caller's scope ends with the host's wait for the step (or for the high stream), so the next file's launches on
// the low stream find the arena free.
int split_to_high(tcmi_ctx *c)
{
    if (!c->stream_hi || c->stream == c->stream_hi) return TCMI_OK;
    TCMI_HIP(c, hipEventRecord(c->ev_split, c->stream_lo));
    TCMI_HIP(c, hipStreamWaitEvent(c->stream_hi, c->ev_split, 0));
    c->stream = c->stream_hi;
    return TCMI_OK;
}
struct SplitScope {
    tcmi_ctx *c;
    explicit SplitScope(tcmi_ctx *ctx) : c(ctx) {}
    ~SplitScope()
    {
        if (c->stream_hi && c->stream == c->stream_hi) {
            (void)hipStreamSynchronize(c->stream_hi);               // (a no-op behind the step's wait; what an error path needs)
            c->stream = c->stream_lo;
        }
    }
};

// queue everything of the one-sync path up to the packed read set; D and J must outlive the wait
// (safe_caps: the arrays sized from the worst case the arena was reserved for, not from a hint of the records' mean size)
int fast_enqueue(tcmi_ctx *ctx, const tcmi_bamfile *f, int64_t first_block, int64_t n_blocks, Decoded &D, tcmi_fused_job &J, tcmi_readset *rs, bool safe_caps = false)
{
    int rc = decode_enqueue(ctx, f, D, first_block, n_blocks);
    if (rc) return rc;
    if (D.nb == 0) return TCMI_E_UNSUPPORTED;                       // (an empty range: the other path's few lines)
    // A wait of the host behind the decode kernels (option "mid_wait" = 1; round 4's default: with everything queued in one go eight
    // contexts measured 60 - 64 M positions/s on boxes where the wait gave 64 - 66 — four hardware queues that always hold a ready
    // kernel run four kernels at once, and these kernels, each sized to fill the chip alone, get in each other's way).  Round 5: one
    // kernel less per file (the CRC pass), and the single wait wins by 1 % (69.3 - 70.0 against 68.4 - 69.4 M, three turns each).
    static const int split_env = std::getenv("TCMI_ONE_SYNC_SPLIT") ? std::atoi(std::getenv("TCMI_ONE_SYNC_SPLIT")) : -1;    // (A/B: 0 none, 1 behind the decode, 2 behind the packer, 3 both)
    const int split = split_env >= 0 ? split_env : ctx->mid_wait;
    if (split == 1 || split == 3) TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rc = split_to_high(ctx);                                        // (the caller's SplitScope brings the context back)
    if (rc) return rc;
    J.d_stream = D.d_out; J.stream_len = D.f->inflated; J.d_desc = D.d_desc; J.d_slot = D.d_slot; J.d_nrec = D.d_nrec; J.d_first = D.d_first;
    J.d_over = D.d_over; J.d_stat = D.d_stat; J.n_blocks = (int64_t)D.nb; J.n_own = (int64_t)D.nb_own; J.ranged = D.ranged ? 1 : 0;
    // records: from the mean size of the file's first records (+ 25 %) where the host saw enough of them, else as the arena was reserved
    J.rec_cap = (int64_t)D.guess_rec;
    // (a file whose header fills its first block shows the host no record: the last file of this context stands in — files of a
    //  batch are alike, and a file beyond the capacity takes the several-kernel path)
    const uint32_t hint = f->rec_bytes_hint >= 36 ? f->rec_bytes_hint : ctx->rec_bytes_seen;
    if (hint >= 36 && !safe_caps) J.rec_cap = std::min<int64_t>(J.rec_cap, (int64_t)(D.f->inflated / hint) * 5 / 4 + 8192);
    J.len_bound = (f->ref_len.empty() ? 0 : f->ref_len[0]) + 65536;
    rs->uid = g_next_uid.fetch_add(1);
    rs->device = ctx->device;
    rc = tcmi_pack_fused_enqueue(ctx, &J, rs);
    if (rc == TCMI_OK && (split == 2 || split == 3)) TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}
} // namespace

// a range of nothing but header blocks at the start of the file: the first record starts where the header ends (the host parsed it)
static int64_t header_only_next(const tcmi_bamfile *f, int64_t first_block, int64_t own)
{
    if (first_block != 0) return -1;
    const int64_t all = (int64_t)f->blocks.size();
    for (int64_t b = 0; b < all; ++b) {
        if (f->blocks[(size_t)b].entry == -1) continue;            // header only
        if (f->blocks[(size_t)b].entry >= 0 && b >= std::min(own, all)) return (int64_t)f->blocks[(size_t)b].uout + f->blocks[(size_t)b].entry;
        return -1;
    }
    return -1;
}

static int readset_from_blocks(tcmi_ctx *ctx, const tcmi_bamfile *f, int64_t first_block, int64_t n_blocks, tcmi_readset **out, int64_t *n_reads_out, bool try_fused)
{
    if (!ctx || !f || !out) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    *out = nullptr;
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    static const bool timing = std::getenv("TCMI_UPLOAD_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    for (int attempt = 0; ctx->one_sync && try_fused && attempt < 2; ++attempt) {
        SplitScope back(ctx);
        Decoded D;
        tcmi_fused_job J;
        tcmi_readset *rs = new tcmi_readset();
        uint32_t why = 0;
        int rc = fast_enqueue(ctx, f, first_block, n_blocks, D, J, rs, attempt > 0);
        if (rc == TCMI_OK) rc = tcmi_pack_fused_report(ctx, &J);
        if (rc == TCMI_OK) {
            TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
            rc = tcmi_pack_fused_finish(ctx, &J, rs, &why);
        }
        if (rc == TCMI_OK) {
            ++ctx->stat_one_sync_taken;
            if (rs->n_reads > 64) ctx->rec_bytes_seen = (uint32_t)std::min<uint64_t>(J.stream_len / (uint64_t)rs->n_reads, 1u << 24);
            rs->packed_on_device = 2;           // decoded AND packed on the device
            if (D.ranged && first_block < (int64_t)f->blocks.size()) {      // (the anchors: from the range's stream to the file's)
                const int64_t at = (int64_t)f->blocks[(size_t)first_block].uout;
                if (rs->range_first >= 0) rs->range_first += at;
                if (rs->range_next >= 0) rs->range_next += at;
                else rs->range_next = header_only_next(f, first_block, (int64_t)D.nb_own);
            } else rs->range_first = rs->range_next = -1;
            if (n_reads_out) *n_reads_out = rs->n_reads;
            *out = rs;
            if (timing) std::fprintf(stderr, "[tcmi bamfile] one-sync path: decode + pack %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            return TCMI_OK;
        }
        tcmi_readset_free(ctx, rs);
        if (rc != TCMI_E_UNSUPPORTED) return rc;
        if (attempt == 0 && (why & 0x800u)) { ctx->rec_bytes_seen = 0; continue; }     // (PKF_REC_OVF: records shorter than the hint said — once more, sized for the worst case)
        ++ctx->stat_one_sync_declined; ctx->stat_last_decline = why;
        if (timing) std::fprintf(stderr, "[tcmi bamfile] one-sync path declined (flags 0x%x): the several-kernel path\n", why);
        break;
    }
    DeviceBam D;
    int rc = decode_on_device(ctx, f, &D, first_block, n_blocks);
    if (rc) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    if (n_reads_out) *n_reads_out = (int64_t)D.n;
    tcmi_pack_src s = {};
    s.stream = D.d_out; s.rec_off = D.d_rec; s.n = (int64_t)D.n; s.mode = 1; s.pos_shift = 0;
    tcmi_readset *rs = new tcmi_readset();
    rs->uid = g_next_uid.fetch_add(1);
    rs->n_reads = (int64_t)D.n;
    rs->device = ctx->device;
    uint32_t why = 0;
    rc = D.n ? tcmi_pack_on_device(ctx, &s, rs, &why) : TCMI_OK;
    if (rc == TCMI_OK) {
        rs->packed_on_device = 2;               // decoded AND packed on the device
        if ((first_block != 0 || (n_blocks >= 0 && first_block + n_blocks < (int64_t)f->blocks.size())) && first_block >= 0 && first_block < (int64_t)f->blocks.size()) {
            const int64_t at = (int64_t)f->blocks[(size_t)first_block].uout;
            rs->range_first = D.range_first >= 0 ? D.range_first + at : -1;
            rs->range_next = D.range_next >= 0 ? D.range_next + at : header_only_next(f, first_block, n_blocks < 0 ? (int64_t)f->blocks.size() - first_block : n_blocks);
        }
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

extern "C" {

int tcmi_readset_from_bamfile_blocks(tcmi_ctx *ctx, const tcmi_bamfile *f, int64_t first_block, int64_t n_blocks, tcmi_readset **out, int64_t *n_reads_out)
{
    return readset_from_blocks(ctx, f, first_block, n_blocks, out, n_reads_out, true);
}

int tcmi_readset_range_anchors(const tcmi_readset *rs, int64_t *first, int64_t *next)
{
    if (!rs) return TCMI_E_ARG;
    if (first) *first = rs->range_first;
    if (next) *next = rs->range_next;
    return TCMI_OK;
}

// BAM file -> call records with ONE wait of the host (TrueConsense.py:225-237 per file: BuildIndex + the position-local part of
// BuildConsensus): the one-sync path's decode and pack, the tally (its grid sized from the packer's capacities, the counts taken
// from device memory) and the call are queued back to back; then the wait; then everything that was deferred is checked.  The
// step is queued for L = max(ref_len, 1) positions: a file whose reads reach beyond that, or with reads too long for the packed
// set, gets its step once more (the read set is complete by then); a file the one-sync path declines goes through
// tcmi_readset_from_bamfile + tcmi_step.  Results as tcmi_step's; *rs_out is the caller's (tcmi_readset_free), *L_out the positions.
int tcmi_bamfile_step(tcmi_ctx *ctx, const tcmi_bamfile *f, int64_t ref_len, int32_t mincov, int include_ambig, tcmi_readset **rs_out, int64_t *L_out,
                      const uint8_t **plain, const uint8_t **alt, const uint8_t **flags, const int32_t **counts_planes, int64_t *ld_out)
{
    if (!ctx || !f || !rs_out || !L_out) return tcmi_fail(ctx, TCMI_E_ARG, "null argument");
    *rs_out = nullptr;
    TCMI_HIP(ctx, hipSetDevice(ctx->device));
    for (int attempt = 0; ctx->one_sync && ctx->step_L == 0 && !ctx->call_pending && attempt < 2; ++attempt) {
        static const bool timing = std::getenv("TCMI_STEP_TIMING") != nullptr;       // diagnostic: where the host's time goes, per 256 calls
        static thread_local double t_acc[5] = {0, 0, 0, 0, 0};
        static thread_local int t_n = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        const auto t0 = now();
        SplitScope back(ctx);
        Decoded D;
        tcmi_fused_job J;
        tcmi_readset *rs = new tcmi_readset();
        uint32_t why = 0;
        const int64_t L0 = std::max<int64_t>(ref_len, 1);
        int rc = fast_enqueue(ctx, f, 0, -1, D, J, rs, attempt > 0);
        const auto t1 = now();
        bool began = false;
        if (rc == TCMI_OK) {
            rc = tcmi_step_begin(ctx, rs, L0, mincov, include_ambig, counts_planes != nullptr);
            began = rc == TCMI_OK;
        }
        const auto t2 = now();
        if (began) {
            rc = tcmi_pack_fused_report(ctx, &J);
            if (rc == TCMI_OK && hipEventRecord(ctx->step_done, ctx->stream) != hipSuccess)     // (the step's end is behind the report now)
                rc = tcmi_fail(ctx, TCMI_E_HIP, "hipEventRecord failed");
        }
        if (began) {
            if (rc) ctx->step_L = 0;
            else rc = tcmi_step_end(ctx, plain, alt, flags, counts_planes, ld_out);     // THE wait
            const auto t3 = now();
            if (rc == TCMI_OK) rc = tcmi_pack_fused_finish(ctx, &J, rs, &why);
            else ctx->step_L = 0;
            if (timing) {
                static thread_local std::chrono::steady_clock::time_point t_last = t0;
                t_acc[0] += us(t0, t1); t_acc[1] += us(t1, t2); t_acc[2] += us(t2, t3); t_acc[3] += us(t3, now()); t_acc[4] += us(t_last, t0);
                t_last = now();
                if (++t_n == 256) {
                    std::fprintf(stderr, "[tcmi step] per file: enqueue decode + pack %.0f us, enqueue tally + call %.0f us, wait %.0f us, finish %.0f us; between calls %.0f us\n",
                                 t_acc[0] / t_n, t_acc[1] / t_n, t_acc[2] / t_n, t_acc[3] / t_n, t_acc[4] / t_n);
                    t_n = 0; for (double &x : t_acc) x = 0;
                }
            }
        }
        if (rc == TCMI_OK) {
            ++ctx->stat_one_sync_taken;
            if (rs->n_reads > 64) ctx->rec_bytes_seen = (uint32_t)std::min<uint64_t>(J.stream_len / (uint64_t)rs->n_reads, 1u << 24);
            rs->packed_on_device = 2;
            int64_t L = L0;
            if (rs->max_end > L0 || rs->s_reads > 0) {              // reads beyond the reference's end, or long reads that the packed set leaves out
                L = std::max<int64_t>(L0, rs->max_end);
                rc = tcmi_step(ctx, rs, L, mincov, include_ambig, plain, alt, flags, counts_planes, ld_out);
                if (rc) { tcmi_readset_free(ctx, rs); return rc; }
            }
            *rs_out = rs; *L_out = L;
            return TCMI_OK;
        }
        if (began) {                                                // the step ran on a read set that is not to be trusted: leave the workspace as a fresh one
            ctx->counts_clean = false;
        }
        tcmi_readset_free(ctx, rs);
        if (rc != TCMI_E_UNSUPPORTED) return rc;
        if (attempt == 0 && (why & 0x800u)) { ctx->rec_bytes_seen = 0; continue; }     // (PKF_REC_OVF: once more, sized for the worst case)
        ++ctx->stat_one_sync_declined; ctx->stat_last_decline = why;
        break;
    }
    tcmi_readset *rs = nullptr;
    int rc = readset_from_blocks(ctx, f, 0, -1, &rs, nullptr, false);   // (the one-sync path has had its turn)
    if (rc) return rc;
    int64_t max_end = 0;
    tcmi_readset_info(rs, nullptr, nullptr, nullptr, nullptr, &max_end);
    const int64_t L = std::max<int64_t>({ref_len, max_end, 1});
    rc = tcmi_step(ctx, rs, L, mincov, include_ambig, plain, alt, flags, counts_planes, ld_out);
    if (rc) { tcmi_readset_free(ctx, rs); return rc; }
    *rs_out = rs; *L_out = L;
    return TCMI_OK;
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
    int64_t inflated = 0;
    for (const BlockDesc &b : f->blocks) inflated += b.ulen;
    if (inflated > stream_cap) return tcmi_fail(ctx, TCMI_E_ARG, "stream buffer too small");
    if (inflated) TCMI_HIP(ctx, hipMemcpyAsync(stream, D.d_out, (size_t)inflated, hipMemcpyDeviceToHost, ctx->stream));
    if (rec_off && D.n) TCMI_HIP(ctx, hipMemcpyAsync(rec_off, D.d_rec, D.n * 8, hipMemcpyDeviceToHost, ctx->stream));
    TCMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TCMI_OK;
}

} // extern "C"
