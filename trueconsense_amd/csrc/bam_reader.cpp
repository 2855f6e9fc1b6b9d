// bam_reader.cpp — HOST: BGZF / BAM decoding into the flat read arrays of tcmi_reads.
//
// Plays the part pysam.AlignmentFile plays for the reference (TrueConsense/indexing.py:19, 96;
// the wire format is SAM spec §4.1 BGZF and §4.2 BAM).  No index (.bai) is
// needed: the whole file is inflated (BGZF blocks are independent gzip members, so they are
// inflated by `n_threads` workers straight into their final place in one buffer) and every
// record is decoded once into struct-of-arrays form, the layout tcmi_readset_upload consumes.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "tcmi_internal.h"

// a buffer that is not zero-filled on allocation: the decode passes write every byte they hand out, and
// zero-filling hundreds of MB on one thread (page faults included) cost more than the decoding itself
template <class T> struct RawBuf {
    std::unique_ptr<T[]> p;
    size_t n = 0;
    void alloc(size_t count) { p.reset(new T[count]); n = count; }
    T *data() { return p.get(); }
    const T *data() const { return p.get(); }
    size_t size() const { return n; }
};

struct tcmi_bam {
    std::string text;                          // SAM header text
    std::vector<std::string> ref_name;
    std::vector<int64_t> ref_len;
    int64_t n = 0;
    std::vector<int32_t> pos, l_qseq, tid, next_tid, next_pos, tlen;
    std::vector<uint64_t> name_off;
    RawBuf<char> names;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> mapq;
    std::vector<uint64_t> cigar_off, seq_off, qual_off;
    int64_t max_span = 0;                      // over all reads (reference positions consumed by the CIGAR)
    RawBuf<uint32_t> cigar;
    RawBuf<uint8_t> seq, qual;
    int sorted = 1;                            // coordinate-sorted (tid, pos) among mapped reads
    int64_t file_bytes = 0, inflated_bytes = 0, n_blocks = 0;
};

namespace {

struct Block { size_t cin, clen, uout, ulen; uint32_t crc; };

inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// Walk the gzip member headers (RFC 1952 + the BC extra subfield of SAM spec §4.1).
int scan_blocks(const std::vector<uint8_t> &f, std::vector<Block> &blocks, size_t *total)
{
    size_t off = 0, out = 0;
    while (off < f.size()) {
        if (f.size() - off < 18) return tcmi_fail(nullptr, TCMI_E_FORMAT, "truncated BGZF block header at byte %zu", off);
        const uint8_t *h = f.data() + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4))
            return tcmi_fail(nullptr, TCMI_E_FORMAT, "not a BGZF block at byte %zu (is the file a BAM?)", off);
        const size_t xlen = rd16(h + 10);
        if (f.size() - off < 12 + xlen) return tcmi_fail(nullptr, TCMI_E_FORMAT, "truncated BGZF extra field at byte %zu", off);
        size_t bsize = 0;
        for (size_t x = 0; x + 4 <= xlen;) {
            const uint8_t *s = h + 12 + x;
            const size_t slen = rd16(s + 2);
            if (s[0] == 'B' && s[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (size_t)rd16(s + 4) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || f.size() - off < bsize)
            return tcmi_fail(nullptr, TCMI_E_FORMAT, "bad BGZF block size at byte %zu", off);
        Block b;
        b.cin = off + 12 + xlen;
        b.clen = bsize - 12 - xlen - 8;
        b.crc = rd32(h + bsize - 8);
        b.ulen = rd32(h + bsize - 4);
        b.uout = out;
        if (b.ulen > 65536) return tcmi_fail(nullptr, TCMI_E_FORMAT, "BGZF block at byte %zu inflates to %zu bytes (> 64 KiB)", off, b.ulen);
        out += b.ulen;
        off += bsize;
        blocks.push_back(b);
    }
    *total = out;
    return TCMI_OK;
}

bool inflate_block(const uint8_t *in, const Block &b, uint8_t *out)
{
    if (b.ulen == 0) return true;
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef *>(in + b.cin);
    zs.avail_in = (uInt)b.clen;
    zs.next_out = out + b.uout;
    zs.avail_out = (uInt)b.ulen;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == b.ulen;
    inflateEnd(&zs);
    if (!ok) return false;
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), out + b.uout, (uInt)b.ulen) == b.crc;
}

} // namespace

extern "C" {

int tcmi_bam_load(const char *path, int n_threads, tcmi_bam **out)
{
    if (!path || !out) return tcmi_fail(nullptr, TCMI_E_ARG, "null argument");
    *out = nullptr;
    const bool timing = std::getenv("TCMI_BAM_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t_start = now();
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return tcmi_fail(nullptr, TCMI_E_IO, "cannot open %s", path);
    std::vector<uint8_t> file;
    {
        std::fseek(fp, 0, SEEK_END);
        const long sz = std::ftell(fp);
        std::fseek(fp, 0, SEEK_SET);
        if (sz < 0) { std::fclose(fp); return tcmi_fail(nullptr, TCMI_E_IO, "cannot size %s", path); }
        file.resize((size_t)sz);
        const size_t got = sz ? std::fread(file.data(), 1, (size_t)sz, fp) : 0;
        std::fclose(fp);
        if (got != (size_t)sz) return tcmi_fail(nullptr, TCMI_E_IO, "short read on %s", path);
    }
    const auto t_read = now();
    std::vector<Block> blocks;
    size_t total = 0;
    int rc = scan_blocks(file, blocks, &total);
    if (rc) return rc;
    RawBuf<uint8_t> raw;
    raw.alloc(total + 8);

    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    if ((size_t)n_threads > blocks.size()) n_threads = blocks.empty() ? 1 : (int)blocks.size();
    std::atomic<size_t> next{0};
    std::atomic<long long> bad{-1};
    auto worker = [&]() {
        for (;;) {
            const size_t b0 = next.fetch_add(64);
            if (b0 >= blocks.size() || bad.load() >= 0) return;
            const size_t b1 = b0 + 64 < blocks.size() ? b0 + 64 : blocks.size();
            for (size_t b = b0; b < b1; ++b)
                if (!inflate_block(file.data(), blocks[b], raw.data())) { bad.store((long long)b); return; }
        }
    };
    if (n_threads == 1) worker();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(worker);
        for (auto &t : th) t.join();
    }
    if (bad.load() >= 0)
        return tcmi_fail(nullptr, TCMI_E_FORMAT, "BGZF block %lld failed to inflate or its CRC32 does not match", bad.load());

    const auto t_inflate = now();
    // ---- BAM header (SAM spec §4.2) ----
    const uint8_t *p = raw.data();
    const size_t N = total;
    size_t o = 0;
    auto need = [&](size_t k) { return N - o >= k; };
    if (!need(12) || std::memcmp(p, "BAM\1", 4) != 0) return tcmi_fail(nullptr, TCMI_E_FORMAT, "%s: BAM magic missing", path);
    tcmi_bam *bam = new tcmi_bam();
    bam->file_bytes = (int64_t)file.size();
    bam->inflated_bytes = (int64_t)total;
    bam->n_blocks = (int64_t)blocks.size();
    auto fail = [&](const char *what) {
        delete bam;
        return tcmi_fail(nullptr, TCMI_E_FORMAT, "%s: %s at inflated byte %zu", path, what, o);
    };
    const size_t l_text = rd32(p + 4);
    o = 8;
    if (!need(l_text + 4)) return fail("truncated header text");
    bam->text.assign((const char *)p + o, l_text);
    o += l_text;
    const size_t n_ref = rd32(p + o);
    o += 4;
    for (size_t r = 0; r < n_ref; ++r) {
        if (!need(4)) return fail("truncated reference list");
        const size_t l_name = rd32(p + o);
        o += 4;
        if (!need(l_name + 4)) return fail("truncated reference name");
        bam->ref_name.emplace_back((const char *)p + o, l_name ? l_name - 1 : 0);
        o += l_name;
        bam->ref_len.push_back((int64_t)rd32(p + o));
        o += 4;
    }

    // ---- records: pass 1 walks the record lengths (sequential, light) and lays out the flat arrays,
    //      pass 2 fills them with `n_threads` workers over disjoint record ranges ----
    std::vector<size_t> rec_at;
    rec_at.reserve(N / 200 + 16);
    std::vector<uint64_t> &qual_off = bam->qual_off;
    bam->cigar_off.reserve(N / 200 + 16); bam->seq_off.reserve(N / 200 + 16); qual_off.reserve(N / 200 + 16);
    uint64_t co = 0, so = 0, qo = 0, no = 0;
    bam->name_off.reserve(N / 200 + 16);
    while (o < N) {                                              // one walk: record starts and the offsets of the flat arrays
        if (!need(4)) return fail("truncated record length");
        const size_t bs = rd32(p + o);
        if (bs < 32 || !need(4 + bs)) return fail("truncated alignment record");
        const uint8_t *r = p + o + 4;
        const size_t l_name = r[8], n_c = rd16(r + 12), l_seq = rd32(r + 16);
        if (32 + l_name + 4 * n_c + (l_seq + 1) / 2 + l_seq > bs) return fail("alignment record fields overrun block_size");
        if (l_name == 0) return fail("alignment record without a read name (l_read_name = 0; htslib refuses it too)");
        rec_at.push_back(o);
        bam->cigar_off.push_back(co); bam->seq_off.push_back(so); qual_off.push_back(qo); bam->name_off.push_back(no);
        co += n_c; so += (l_seq + 1) / 2; qo += l_seq; no += l_name ? l_name - 1 : 0;
        o += 4 + bs;
    }
    const auto t_walk = now();
    const int64_t n = (int64_t)rec_at.size();
    bam->n = n;
    bam->cigar_off.push_back(co); bam->seq_off.push_back(so); qual_off.push_back(qo); bam->name_off.push_back(no);
    bam->next_tid.resize((size_t)n); bam->next_pos.resize((size_t)n); bam->tlen.resize((size_t)n);
    bam->names.alloc((size_t)no + 1);
    bam->pos.resize((size_t)n); bam->l_qseq.resize((size_t)n); bam->tid.resize((size_t)n);
    bam->flag.resize((size_t)n); bam->mapq.resize((size_t)n);
    {
        bam->cigar.alloc((size_t)co + 1); bam->seq.alloc((size_t)so + 1); bam->qual.alloc((size_t)qo + 1);
        bam->cigar.data()[co] = 0; bam->seq.data()[so] = 0; bam->qual.data()[qo] = 0;
    }
    {
        int nt = n_threads;
        if ((int64_t)nt > n / 4096 + 1) nt = (int)(n / 4096 + 1);
        std::vector<int> unsorted((size_t)nt, 0), long_cigar((size_t)nt, 0);
        std::vector<int64_t> spans((size_t)nt, 0);
        auto fill = [&](int t) {
            const int64_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
            int32_t last_tid = 0, last_pos = -1;
            bool seen_unplaced = false;
            if (i0 > 0) {                                        // order is checked across the range boundary too
                const uint8_t *r = p + rec_at[(size_t)i0 - 1] + 4;
                const int32_t tid = (int32_t)rd32(r);
                if (tid < 0) seen_unplaced = true; else { last_tid = tid; last_pos = (int32_t)rd32(r + 4); }
            }
            for (int64_t i = i0; i < i1; ++i) {
                const uint8_t *r = p + rec_at[(size_t)i] + 4;
                const int32_t tid = (int32_t)rd32(r), pos = (int32_t)rd32(r + 4);
                const size_t l_name = r[8], n_c = rd16(r + 12), l_seq = rd32(r + 16);
                bam->tid[(size_t)i] = tid;
                bam->pos[(size_t)i] = pos;
                bam->mapq[(size_t)i] = r[9];
                bam->flag[(size_t)i] = rd16(r + 14);
                bam->l_qseq[(size_t)i] = (int32_t)l_seq;
                bam->next_tid[(size_t)i] = (int32_t)rd32(r + 20);
                bam->next_pos[(size_t)i] = (int32_t)rd32(r + 24);
                bam->tlen[(size_t)i] = (int32_t)rd32(r + 28);
                if (l_name > 1) std::memcpy(bam->names.data() + bam->name_off[(size_t)i], r + 32, l_name - 1);
                const uint8_t *c = r + 32 + l_name;
                std::memcpy(bam->cigar.data() + bam->cigar_off[(size_t)i], c, 4 * n_c);     // little-endian host
                {
                    int64_t span = 0;
                    const uint32_t *cg = bam->cigar.data() + bam->cigar_off[(size_t)i];
                    // SAM spec §4.2.2: a CIGAR of more than 65 535 ops is stored in the CG:B,I tag behind the placeholder
                    // <l_seq>S<ref_len>N.  htslib (bam_tag2cigar) substitutes the real CIGAR when that tag is present; this
                    // reader does not: a read that has both the placeholder and the tag is refused.
                    if (n_c == 2 && (cg[0] & 0xF) == 4 && (cg[0] >> 4) == l_seq && (cg[1] & 0xF) == 3 && l_seq > 0) {
                        const uint8_t *a = c + 4 * n_c + (l_seq + 1) / 2 + l_seq, *e = r + rd32(r - 4);
                        while (a + 3 <= e) {                     // aux fields: tag[2] type value
                            const char ty = (char)a[2];
                            if (a[0] == 'C' && a[1] == 'G' && ty == 'B') { long_cigar[(size_t)t] = 1; break; }
                            a += 3;
                            if (ty == 'A' || ty == 'c' || ty == 'C') a += 1;
                            else if (ty == 's' || ty == 'S') a += 2;
                            else if (ty == 'i' || ty == 'I' || ty == 'f') a += 4;
                            else if (ty == 'Z' || ty == 'H') { while (a < e && *a) ++a; ++a; }
                            else if (ty == 'B' && a + 5 <= e) {
                                const char st = (char)a[0];
                                const size_t cnt = rd32(a + 1), sz = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                                a += 5 + cnt * sz;
                            } else break;                        // unknown type: stop scanning
                        }
                    }
                    for (size_t k = 0; k < n_c; ++k) {
                        const unsigned op = cg[k] & 0xF;
                        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += cg[k] >> 4;
                    }
                    if (span > spans[(size_t)t]) spans[(size_t)t] = span;
                }
                const uint8_t *sq = c + 4 * n_c;
                std::memcpy(bam->seq.data() + bam->seq_off[(size_t)i], sq, (l_seq + 1) / 2);
                std::memcpy(bam->qual.data() + qual_off[(size_t)i], sq + (l_seq + 1) / 2, l_seq);
                if (tid < 0) seen_unplaced = true;
                else {
                    // (a range that starts inside the unplaced tail sees a placed read only if the file is unsorted)
                    if (seen_unplaced || tid < last_tid || (tid == last_tid && pos < last_pos)) unsorted[(size_t)t] = 1;
                    last_tid = tid;
                    last_pos = pos;
                }
            }
        };
        if (nt <= 1) fill(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back(fill, t);
            for (auto &x : th) x.join();
        }
        for (int u : unsorted)
            if (u) bam->sorted = 0;
        for (int u : long_cigar)
            if (u) {
                delete bam;
                return tcmi_fail(nullptr, TCMI_E_UNSUPPORTED, "%s: a read keeps its CIGAR in the CG tag (more than 65 535 operations): not supported", path);
            }
        for (int64_t sp : spans) bam->max_span = std::max(bam->max_span, sp);
    }
    if (timing)
        std::fprintf(stderr, "[tcmi bam] read %.1f ms, inflate %.1f ms (%d threads, %zu blocks), record walk %.1f ms, fill %.1f ms\n",
                     ms(t_start, t_read), ms(t_read, t_inflate), n_threads, blocks.size(), ms(t_inflate, t_walk), ms(t_walk, now()));
    *out = bam;
    return TCMI_OK;
}

int tcmi_bam_free(tcmi_bam *bam)
{
    delete bam;
    return TCMI_OK;
}

int tcmi_bam_reads(const tcmi_bam *bam, tcmi_reads *reads)
{
    if (!bam || !reads) return tcmi_fail(nullptr, TCMI_E_ARG, "null argument");
    reads->n_reads = bam->n;
    reads->pos = bam->pos.data();
    reads->flag = bam->flag.data();
    reads->l_qseq = bam->l_qseq.data();
    reads->cigar_off = bam->cigar_off.data();
    reads->cigar = bam->cigar.data();
    reads->seq_off = bam->seq_off.data();
    reads->seq = bam->seq.data();
    reads->qual = bam->qual.data();
    reads->tid = bam->tid.data();
    reads->qual_off = bam->qual_off.data();
    reads->sorted_max_span = bam->sorted ? std::max<int64_t>(1, bam->max_span) : 0;
    reads->next_tid = bam->next_tid.data();
    reads->next_pos = bam->next_pos.data();
    reads->tlen = bam->tlen.data();
    reads->name_off = bam->name_off.data();
    reads->names = bam->names.data();
    return TCMI_OK;
}

int tcmi_bam_header(const tcmi_bam *bam, int32_t *n_ref, const char **ref0_name, int64_t *ref0_len)
{
    if (!bam) return tcmi_fail(nullptr, TCMI_E_ARG, "bam is NULL");
    if (n_ref) *n_ref = (int32_t)bam->ref_name.size();
    if (ref0_name) *ref0_name = bam->ref_name.empty() ? "" : bam->ref_name[0].c_str();
    if (ref0_len) *ref0_len = bam->ref_len.empty() ? 0 : bam->ref_len[0];
    return TCMI_OK;
}

int tcmi_bam_info(const tcmi_bam *bam, int64_t *n_reads, int32_t *sorted, int64_t *file_bytes, int64_t *inflated_bytes,
                  int64_t *n_blocks, int64_t *n_cigar, int64_t *n_qual)
{
    if (!bam) return tcmi_fail(nullptr, TCMI_E_ARG, "bam is NULL");
    if (n_reads) *n_reads = bam->n;
    if (sorted) *sorted = bam->sorted;
    if (file_bytes) *file_bytes = bam->file_bytes;
    if (inflated_bytes) *inflated_bytes = bam->inflated_bytes;
    if (n_blocks) *n_blocks = bam->n_blocks;
    if (n_cigar) *n_cigar = (int64_t)bam->cigar_off[(size_t)bam->n];
    if (n_qual) *n_qual = (int64_t)bam->qual.size() - 1;
    return TCMI_OK;
}

const char *tcmi_bam_text(const tcmi_bam *bam) { return bam ? bam->text.c_str() : ""; }

} // extern "C"
