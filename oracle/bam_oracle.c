/*
 * bam_oracle.c — scalar C restatement of the BAM file -> flat read arrays step (pysam's role in
 * indexing.py:19,96: AlignmentFile + the record fields the pileup consumes).  TEST INFRASTRUCTURE ONLY:
 * loaded by tests/ and by bench.py's cpu_baseline leg; the product never links or calls it.
 *
 * Written independently of trueconsense_amd/csrc/bam_reader.cpp so the two can check each other: this one
 * treats the file as a plain multi-member gzip stream (RFC 1952; zlib's gzip-header mode finds every member
 * by itself and never looks at the BGZF "BC" subfield), inflates it sequentially on one thread and walks
 * the records once.  Wire format: SAM spec §4.2.
 *
 * PARITY UNPINNED against pysam/htslib (absent here); pinned to the SAM specification text and to the
 * hand-assembled fixture of tests/test_bam_fixture.py.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

typedef struct orc_bam {
    uint8_t *raw;              /* inflated stream */
    size_t n_raw;
    size_t first_rec;          /* offset of the first alignment record */
    int64_t n, n_cigar, n_seq, n_qual;
    int32_t n_ref;
    int64_t ref0_len;
    char ref0_name[256];
} orc_bam;

static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

/* whole file -> one buffer, member after member (inflateReset between members) */
static int gunzip_all(const uint8_t *in, size_t n_in, uint8_t **out, size_t *n_out)
{
    size_t cap = n_in * 4 + 65536, have = 0, fed = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    if (!buf) return -1;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) { free(buf); return -1; }   /* 15 + 16: expect gzip headers */
    int rc = n_in ? Z_OK : Z_STREAM_END;
    while (n_in) {
        if (zs.avail_in == 0 && fed < n_in) {                            /* avail_in is 32 bits wide: feed in pieces */
            const size_t more = (n_in - fed) > 0x40000000u ? 0x40000000u : (n_in - fed);
            zs.next_in = (Bytef *)(in + fed);
            zs.avail_in = (uInt)more;
            fed += more;
        }
        if (cap - have < 131072) {
            cap *= 2;
            uint8_t *nb = (uint8_t *)realloc(buf, cap);
            if (!nb) { rc = Z_MEM_ERROR; break; }
            buf = nb;
        }
        zs.next_out = buf + have;
        zs.avail_out = (uInt)((cap - have) > 0x40000000u ? 0x40000000u : (cap - have));
        const size_t room = zs.avail_out;
        rc = inflate(&zs, Z_NO_FLUSH);
        have += room - zs.avail_out;
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0 && fed == n_in) break;                  /* the last member ended with the file */
            if (inflateReset(&zs) != Z_OK) { rc = Z_STREAM_ERROR; break; }
            rc = Z_OK;
        } else if (rc == Z_BUF_ERROR || rc == Z_OK) {
            if (zs.avail_in == 0 && fed == n_in) { rc = Z_DATA_ERROR; break; }   /* file ends inside a member */
            rc = Z_OK;
        } else break;
    }
    inflateEnd(&zs);
    if (rc != Z_STREAM_END) { free(buf); return -2; }
    *out = buf;
    *n_out = have;
    return 0;
}

int orc_bam_load(const char *path, orc_bam **out)
{
    *out = NULL;
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    fseek(fp, 0, SEEK_END);
    const long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    uint8_t *file = (uint8_t *)malloc(sz > 0 ? (size_t)sz : 1);
    if (!file || (sz > 0 && fread(file, 1, (size_t)sz, fp) != (size_t)sz)) { fclose(fp); free(file); return -1; }
    fclose(fp);
    orc_bam *b = (orc_bam *)calloc(1, sizeof *b);
    const int rc = gunzip_all(file, (size_t)sz, &b->raw, &b->n_raw);
    free(file);
    if (rc) { free(b); return rc; }
    const uint8_t *p = b->raw;
    const size_t N = b->n_raw;
    if (N < 12 || memcmp(p, "BAM\1", 4) != 0) { free(b->raw); free(b); return -3; }
    size_t o = 8 + le32(p + 4);                                         /* magic, l_text, text */
    if (o + 4 > N) { free(b->raw); free(b); return -3; }
    b->n_ref = (int32_t)le32(p + o);
    o += 4;
    for (int32_t r = 0; r < b->n_ref; ++r) {
        if (o + 4 > N) { free(b->raw); free(b); return -3; }
        const size_t l_name = le32(p + o);
        o += 4;
        if (o + l_name + 4 > N) { free(b->raw); free(b); return -3; }
        if (r == 0) {
            snprintf(b->ref0_name, sizeof b->ref0_name, "%.*s", (int)(l_name ? l_name - 1 : 0), (const char *)p + o);
            b->ref0_len = (int64_t)le32(p + o + l_name);
        }
        o += l_name + 4;
    }
    b->first_rec = o;
    while (o < N) {                                                      /* count */
        if (o + 4 > N) { free(b->raw); free(b); return -3; }
        const size_t bs = le32(p + o);
        if (bs < 32 || o + 4 + bs > N) { free(b->raw); free(b); return -3; }
        const uint8_t *r = p + o + 4;
        const size_t l_seq = le32(r + 16);
        b->n += 1;
        b->n_cigar += le16(r + 12);
        b->n_seq += (int64_t)((l_seq + 1) / 2);
        b->n_qual += (int64_t)l_seq;
        o += 4 + bs;
    }
    *out = b;
    return 0;
}

void orc_bam_dims(const orc_bam *b, int64_t *n, int64_t *n_cigar, int64_t *n_seq, int64_t *n_qual, int32_t *n_ref,
                  int64_t *ref0_len, int64_t *inflated)
{
    *n = b->n; *n_cigar = b->n_cigar; *n_seq = b->n_seq; *n_qual = b->n_qual; *n_ref = b->n_ref; *ref0_len = b->ref0_len;
    *inflated = (int64_t)b->n_raw;
}

const char *orc_bam_ref0(const orc_bam *b) { return b->ref0_name; }

/* fields of SAM spec §4.2: refID, pos, l_read_name, mapq, bin, n_cigar_op, flag, l_seq, next_refID, next_pos, tlen,
 * read_name, cigar, seq, qual */
void orc_bam_fill(const orc_bam *b, int32_t *pos, uint16_t *flag, int32_t *l_qseq, int32_t *tid, uint8_t *mapq,
                  uint64_t *cigar_off, uint32_t *cigar, uint64_t *seq_off, uint8_t *seq, uint64_t *qual_off, uint8_t *qual)
{
    const uint8_t *p = b->raw;
    size_t o = b->first_rec;
    uint64_t co = 0, so = 0, qo = 0;
    for (int64_t i = 0; i < b->n; ++i) {
        const size_t bs = le32(p + o);
        const uint8_t *r = p + o + 4;
        const size_t l_name = r[8], n_c = le16(r + 12), l_seq = le32(r + 16);
        tid[i] = (int32_t)le32(r);
        pos[i] = (int32_t)le32(r + 4);
        mapq[i] = r[9];
        flag[i] = le16(r + 14);
        l_qseq[i] = (int32_t)l_seq;
        cigar_off[i] = co; seq_off[i] = so; qual_off[i] = qo;
        const uint8_t *c = r + 32 + l_name;
        for (size_t k = 0; k < n_c; ++k) cigar[co + k] = le32(c + 4 * k);
        memcpy(seq + so, c + 4 * n_c, (l_seq + 1) / 2);
        memcpy(qual + qo, c + 4 * n_c + (l_seq + 1) / 2, l_seq);
        co += n_c; so += (l_seq + 1) / 2; qo += l_seq;
        o += 4 + bs;
    }
    cigar_off[b->n] = co; seq_off[b->n] = so; qual_off[b->n] = qo;
}

void orc_bam_free(orc_bam *b)
{
    if (!b) return;
    free(b->raw);
    free(b);
}
