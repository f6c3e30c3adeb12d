/* aln_reader.c -- SAM / gzip-SAM / BAM -> structure-of-arrays alignment records.
 *
 * Replaces the reference's use of htslib for this path (sam_open, sam_hdr_read,
 * sam_read1, bam_aux_get, bam_aux2A: src/bam2gtf.c:35-37,92,107;
 * src/update_gtf.c:1065-1070).  Only the fields gen_exon() reads are kept:
 * refID, pos, FLAG, CIGAR words, QNAME, and the XS aux tag.  Formats follow the
 * SAM/BAM specification (SAMv1 sections 1.4, 4.2).
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "l2r_host.h"

typedef h_blob blob;

static blob slurp_gz(const char *fn, const char *who)
{
    /* gzread() passes plain files through and inflates gzip/BGZF (concatenated members) */
    gzFile g = gzopen(fn, "rb");
    if (!g) h_fatal(who, "Can not open \"%s\"\n", fn);
    gzbuffer(g, 1 << 20);
    blob b = {NULL, 0};
    size_t cap = 1 << 22;
    b.p = (uint8_t *)h_malloc(cap);
    for (;;) {
        if (b.n + (1 << 20) > cap) { cap *= 2; b.p = (uint8_t *)h_realloc(b.p, cap); }
        int k = gzread(g, b.p + b.n, 1 << 20);
        if (k < 0) h_fatal(who, "read error in \"%s\"", fn);
        if (k == 0) break;
        b.n += (size_t)k;
    }
    gzclose(g);
    return b;
}

/* ---- BGZF: independent deflate blocks (SAMv1 4.1), inflated on several threads ------------------------
 * Every block is a gzip member with a "BC" extra subfield that holds its total size and ends with CRC32 and the
 * uncompressed size, so the block table and the output offsets come from one pass over the headers. */
#include <pthread.h>
#include <unistd.h>
#include <stdio.h>

typedef struct { size_t in_off, in_len, out_off, out_len; } bgzf_block;
typedef struct { const uint8_t *in; uint8_t *out; const bgzf_block *blk; size_t n_blk; size_t next; pthread_mutex_t mu; int failed; } bgzf_job;

static void *bgzf_worker(void *arg)
{
    bgzf_job *jb = (bgzf_job *)arg;
    z_stream z; memset(&z, 0, sizeof z);
    if (inflateInit2(&z, -15) != Z_OK) { jb->failed = 1; return NULL; }
    for (;;) {
        pthread_mutex_lock(&jb->mu);
        size_t lo = jb->next, hi = lo + 64 < jb->n_blk ? lo + 64 : jb->n_blk;            /* 64 blocks (<= 4 MB) per grab */
        jb->next = hi;
        pthread_mutex_unlock(&jb->mu);
        if (lo >= hi) break;
        for (size_t k = lo; k < hi; ++k) {
            const bgzf_block *b = &jb->blk[k];
            if (b->out_len == 0) continue;
            inflateReset(&z);
            z.next_in = (Bytef *)(jb->in + b->in_off); z.avail_in = (uInt)b->in_len;
            z.next_out = jb->out + b->out_off; z.avail_out = (uInt)b->out_len;
            const int rc = inflate(&z, Z_FINISH);
            if (rc != Z_STREAM_END || z.avail_out != 0) { jb->failed = 1; break; }
            /* the block's CRC32 of the inflated bytes sits before ISIZE (htslib checks it too: a damaged block that still
               inflates to ISIZE bytes must not pass) */
            const uint8_t *t = jb->in + b->in_off + b->in_len;
            const uLong want = (uLong)t[0] | ((uLong)t[1] << 8) | ((uLong)t[2] << 16) | ((uLong)t[3] << 24);
            if (crc32(crc32(0L, Z_NULL, 0), jb->out + b->out_off, (uInt)b->out_len) != want) { jb->failed = 2; break; }
        }
    }
    inflateEnd(&z);
    return NULL;
}

/* Whole file -> memory; BGZF is inflated block-parallel, anything else goes through gzread. */
h_blob h_slurp(const char *fn, const char *who)
{
    FILE *f = fopen(fn, "rb");
    if (!f) h_fatal(who, "Can not open \"%s\"\n", fn);
    uint8_t hd[18];
    const size_t got = fread(hd, 1, sizeof hd, f);
    const int bgzf = got == sizeof hd && hd[0] == 0x1f && hd[1] == 0x8b && hd[2] == 8 && (hd[3] & 4) && hd[12] == 'B' && hd[13] == 'C' &&
                     hd[14] == 2 && hd[15] == 0 && (hd[10] | (hd[11] << 8)) == 6;
    if (!bgzf) { fclose(f); return slurp_gz(fn, who); }
    fseek(f, 0, SEEK_END);
    const long fsz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *raw = (uint8_t *)h_malloc((size_t)fsz + 1);
    if (fread(raw, 1, (size_t)fsz, f) != (size_t)fsz) h_fatal(who, "read error in \"%s\"", fn);
    fclose(f);
    /* block table */
    size_t n_blk = 0, cap = 1024, out_total = 0;
    bgzf_block *blk = (bgzf_block *)h_malloc(cap * sizeof *blk);
    for (size_t p = 0; p < (size_t)fsz;) {
        if ((size_t)fsz - p < 28 || raw[p] != 0x1f || raw[p + 1] != 0x8b || raw[p + 12] != 'B' || raw[p + 13] != 'C') {
            free(blk); free(raw); return slurp_gz(fn, who);                                 /* not plain BGZF after all */
        }
        const size_t bsize = (size_t)(raw[p + 16] | (raw[p + 17] << 8)) + 1;
        if (bsize < 28 || p + bsize > (size_t)fsz) h_fatal(who, "truncated BGZF block in \"%s\"", fn);
        const uint8_t *tail = raw + p + bsize - 4;
        const size_t isize = (size_t)tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
        if (n_blk == cap) { cap *= 2; blk = (bgzf_block *)h_realloc(blk, cap * sizeof *blk); }
        blk[n_blk].in_off = p + 18; blk[n_blk].in_len = bsize - 18 - 8; blk[n_blk].out_off = out_total; blk[n_blk].out_len = isize;
        ++n_blk; out_total += isize; p += bsize;
    }
    blob b; b.n = out_total; b.p = (uint8_t *)h_malloc(out_total + 1);
    bgzf_job jb; memset(&jb, 0, sizeof jb);
    jb.in = raw; jb.out = b.p; jb.blk = blk; jb.n_blk = n_blk; jb.next = 0; pthread_mutex_init(&jb.mu, NULL);
    const char *e = getenv("L2R_THREADS");
    long n_thr = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    if (n_thr > 32) n_thr = 32;
    if (n_thr < 1 || n_blk < 8) n_thr = 1;
    pthread_t th[32];
    for (long k = 1; k < n_thr; ++k) if (pthread_create(&th[k], NULL, bgzf_worker, &jb)) h_fatal(who, "pthread_create failed");
    bgzf_worker(&jb);
    for (long k = 1; k < n_thr; ++k) pthread_join(th[k], NULL);
    pthread_mutex_destroy(&jb.mu);
    if (jb.failed) h_fatal(who, jb.failed == 2 ? "CRC32 mismatch in a BGZF block of \"%s\"" : "corrupt BGZF block in \"%s\"", fn);
    free(blk); free(raw);
    return b;
}

/* What the engine's upload would otherwise walk every CIGAR for (include/lr2rmats_hip.h, l2r_reads::cig_summary): collected op by op
 * where a record's CIGAR is converted anyway -- reference bases, N operations, the shortest N, the longest D, the shortest stretch of
 * reference bases between two N operations (src/bam2gtf.c:41-74: what decides whether an N cuts, a D cuts, an inner exon is dropped). */
typedef struct { uint64_t ref, seg; uint32_t nn, mn, md, ms; int first; } cigsum;
static inline void cigsum_init(cigsum *a) { a->ref = 0; a->seg = 0; a->nn = 0; a->mn = 65535u; a->md = 0; a->ms = 65535u; a->first = 1; }
static inline void cigsum_op(cigsum *a, uint32_t w)
{
    const uint32_t op = w & 15u, len = w >> 4;
    const int adv = (0x18d >> op) & 1;                    /* ops M D N = X advance the reference */
    if (op == 3u) {
        a->nn++; if (len < a->mn) a->mn = len;
        if (!a->first && a->seg < a->ms) a->ms = (uint32_t)a->seg;      /* (the stretch in front of the first N is the first exon: kept whatever its length) */
        a->first = 0; a->seg = 0;
    } else {
        if (op == 2u && len > a->md) a->md = len;
        if (adv) a->seg += len;
    }
    if (adv) a->ref += len;
}
static inline void cigsum_store(const cigsum *a, uint32_t *q)
{
    q[0] = a->ref > 0xffffffffull ? 0xffffffffu : (uint32_t)a->ref;
    q[1] = (a->nn > 65535u ? 65535u : a->nn) | ((a->mn > 65535u ? 65535u : a->mn) << 16);
    q[2] = (a->md > 65535u ? 65535u : a->md) | ((a->ms > 65535u ? 65535u : a->ms) << 16);
}

/* the same for records that are in memory already (synthetic reads, callers with their own reader): out[3 * n] */
void h_cigar_summaries(int64_t n, const int64_t *cig_off, const uint32_t *cig, uint32_t *out)
{
    for (int64_t i = 0; i < n; ++i) {
        cigsum cs; cigsum_init(&cs);
        for (int64_t k = cig_off[i]; k < cig_off[i + 1]; ++k) cigsum_op(&cs, cig[k]);
        cigsum_store(&cs, out + 3 * (size_t)i);
    }
}

static void reads_reserve(h_reads *r, int64_t more_reads, int64_t more_cig)
{
    if (r->n + more_reads + 1 > r->cap) {
        int64_t c = r->cap ? r->cap * 2 : 1 << 16;
        while (c < r->n + more_reads + 1) c *= 2;
        r->tid = (int32_t *)h_realloc(r->tid, (size_t)c * 4); r->pos = (int32_t *)h_realloc(r->pos, (size_t)c * 4);
        r->rev = (uint8_t *)h_realloc(r->rev, (size_t)c); r->cig_off = (int64_t *)h_realloc(r->cig_off, (size_t)(c + 1) * 8);
        r->qname = (uint32_t *)h_realloc(r->qname, (size_t)c * 4);
        r->cig_sum = (uint32_t *)h_realloc(r->cig_sum, (size_t)c * 12);
        r->cap = c;
    }
    if (r->n_cig + more_cig > r->cap_cig) {
        int64_t c = r->cap_cig ? r->cap_cig * 2 : 1 << 20;
        while (c < r->n_cig + more_cig) c *= 2;
        r->cig = (uint32_t *)h_realloc(r->cig, (size_t)c * 4); r->cap_cig = c;
    }
}

void h_reads_free(h_reads *r)
{
    free(r->tid); free(r->pos); free(r->rev); free(r->cig_off); free(r->cig); free(r->cig_sum); free(r->qname); free(r->tid_name); free(r->names.buf);
    memset(r, 0, sizeof *r);
}

static void add_qname(h_reads *r, const char *s, size_t len, const char *who)
{
    char tmp[H_NAME_MAX];
    if (len >= H_NAME_MAX) h_fatal(who, "read name of %zu characters; the reference stores names in char[100]", len);
    memcpy(tmp, s, len); tmp[len] = 0;
    r->qname[r->n] = h_str_add(&r->names, tmp);
}

/* ------------------------------------------------------------------ SAM text */

static void parse_sam_header_line(const char *l, const char *e, h_chroms *chr)
{
    if (e - l < 3 || memcmp(l, "@SQ", 3) != 0) return;
    const char *p = l;
    while (p < e) {
        const char *t = memchr(p, '\t', (size_t)(e - p));
        if (!t) break;
        p = t + 1;
        if (e - p > 3 && memcmp(p, "SN:", 3) == 0) {
            const char *q = memchr(p, '\t', (size_t)(e - p));
            size_t len = (size_t)((q ? q : e) - (p + 3));
            char name[H_NAME_MAX];
            if (len >= H_NAME_MAX) h_fatal("sam_hdr_read", "reference name of 100 or more characters");
            memcpy(name, p + 3, len); name[len] = 0;
            h_chrom_intern(chr, name);
            return;
        }
    }
}

static void parse_sam(const blob *b, h_chroms *chr, h_reads *out, int skip_unmapped, int header_only, const char *who)
{
    const char *p = (const char *)b->p, *end = p + b->n;
    int in_header = 1;
    while (p < end) {
        const char *nl = memchr(p, '\n', (size_t)(end - p));
        const char *e = nl ? nl : end;
        const char *le = e;
        if (le > p && le[-1] == '\r') --le;
        if (le == p) { p = e + 1; continue; }
        if (*p == '@') { if (in_header) parse_sam_header_line(p, le, chr); p = e + 1; continue; }
        if (in_header) { in_header = 0; chr->n_hdr = chr->n; if (header_only) return; }
        /* QNAME FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL [aux...] */
        const char *f[12]; int nf = 0; const char *q = p;
        while (nf < 11) {
            f[nf++] = q;
            const char *t = memchr(q, '\t', (size_t)(le - q));
            if (!t) { q = le; break; }
            q = t + 1;
        }
        if (nf < 11) h_fatal(who, "truncated SAM record");
        f[11] = q;                                           /* start of aux (or le) */
        const int flag = atoi(f[1]);
        if (flag & 4) {
            if (skip_unmapped) { p = e + 1; continue; }
            /* reference: gen_trans() leaves exon_n = 0, then malloc((0-1)*2) aborts (src/bam2gtf.c:82,100) */
            h_fatal_core("read_bam_trans", "Malloc fail!\nSize: -2\n");
        }
        size_t ops_bound = (size_t)(f[6] - f[5]);
        reads_reserve(out, 1, (int64_t)ops_bound);
        add_qname(out, f[0], (size_t)(f[1] - 1 - f[0]), who);
        /* RNAME */
        {
            size_t len = (size_t)(f[3] - 1 - f[2]);
            char name[H_NAME_MAX];
            if (len >= H_NAME_MAX) h_fatal(who, "reference name too long");
            memcpy(name, f[2], len); name[len] = 0;
            int tid = (len == 1 && name[0] == '*') ? -1 : h_chrom_find(chr, name, chr->n_hdr);
            if (tid < 0) h_fatal(who, "record \"%.*s\": reference \"%s\" is not in the header", (int)(f[1] - 1 - f[0]), f[0], name);
            out->tid[out->n] = tid;
        }
        out->pos[out->n] = atoi(f[3]) - 1;
        /* CIGAR */
        out->cig_off[out->n] = out->n_cig;
        cigsum cs; cigsum_init(&cs);
        {
            const char *c = f[5], *ce = f[6] - 1;
            if (!(ce - c == 1 && *c == '*')) {
                while (c < ce) {
                    uint32_t len = 0;
                    while (c < ce && *c >= '0' && *c <= '9') { len = len * 10u + (uint32_t)(*c - '0'); ++c; }
                    if (c >= ce) h_fatal(who, "bad CIGAR");
                    uint32_t op;
                    switch (*c) {
                    case 'M': op = 0; break; case 'I': op = 1; break; case 'D': op = 2; break; case 'N': op = 3; break;
                    case 'S': op = 4; break; case 'H': op = 5; break; case 'P': op = 6; break; case '=': op = 7; break;
                    case 'X': op = 8; break; case 'B': op = 9; break;
                    default: h_fatal(who, "bad CIGAR operator '%c'", *c); op = 0;
                    }
                    out->cig[out->n_cig++] = (len << 4) | op;
                    cigsum_op(&cs, (len << 4) | op);
                    ++c;
                }
            }
        }
        /* strand: XS aux present ? (type 'A' and value '+' ? 0 : 1) : FLAG & 16   (src/bam2gtf.c:35-37;
         * htslib's bam_aux2A returns 0 for a tag that is not of type 'A') */
        {
            uint8_t rev = (flag & 16) ? 1 : 0;
            const char *a = f[11];
            while (a < le) {
                const char *t = memchr(a, '\t', (size_t)(le - a));
                const char *ae = t ? t : le;
                if (ae - a >= 5 && a[0] == 'X' && a[1] == 'S' && a[2] == ':') {
                    rev = (a[3] == 'A' && ae - a >= 6 && a[5] == '+') ? 0 : 1;
                    break;
                }
                if (!t) break;
                a = t + 1;
            }
            out->rev[out->n] = rev;
        }
        cigsum_store(&cs, out->cig_sum + 3 * (size_t)out->n);
        out->n++;
        out->cig_off[out->n] = out->n_cig;
        p = e + 1;
    }
    if (in_header) chr->n_hdr = chr->n;
}

/* ------------------------------------------------------------------ BAM */

static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

/* size of one aux value of the given type at p (SAMv1 4.2.4); 0 on error */
static size_t aux_size(uint8_t type, const uint8_t *p, const uint8_t *end)
{
    switch (type) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'Z': case 'H': { const uint8_t *z = memchr(p, 0, (size_t)(end - p)); return z ? (size_t)(z - p) + 1 : 0; }
    case 'B': {
        if (end - p < 5) return 0;
        size_t w;
        switch (p[0]) { case 'c': case 'C': w = 1; break; case 's': case 'S': w = 2; break; case 'i': case 'I': case 'f': w = 4; break; default: return 0; }
        return 5 + w * (size_t)le32(p + 1);
    }
    default: return 0;
    }
}

typedef struct {
    const blob *b; const size_t *starts; size_t lo, hi; const h_chroms *chr; int skip_unmapped; const char *who; h_reads piece;
    h_reads *dst; int64_t at, cat; size_t nat;               /* where the piece goes in the joined arrays */
} bam_piece;

static void *bam_piece_main(void *arg)
{
    bam_piece *q = (bam_piece *)arg;
    h_reads *out = &q->piece;
    const h_chroms *chr = q->chr; const char *who = q->who; const int skip_unmapped = q->skip_unmapped;
    reads_reserve(out, 1, 1);
    out->cig_off[0] = 0;
    for (size_t ri = q->lo; ri < q->hi; ++ri) {
        const uint8_t *p = q->b->p + q->starts[ri];
        const uint32_t bs = le32(p);
        const uint8_t *rec = p + 4, *rend = rec + bs;
        int32_t refid = (int32_t)le32(rec), pos = (int32_t)le32(rec + 4);
        uint32_t l_read_name = rec[8], n_cig = le16(rec + 12), flag = le16(rec + 14), l_seq = le32(rec + 16);
        if (flag & 4) {
            if (skip_unmapped) continue;
            h_fatal_core("read_bam_trans", "Malloc fail!\nSize: -2\n");
        }
        const uint8_t *name = rec + 32, *cig = name + l_read_name;
        const uint8_t *aux = cig + 4 * (size_t)n_cig + (l_seq + 1) / 2 + l_seq;
        if (aux > rend || l_read_name == 0) h_fatal(who, "corrupt BAM record");
        if (refid < 0 || refid >= chr->n_hdr) h_fatal(who, "BAM record without a valid reference id");
        /* long CIGARs (> 65535 ops) are stored in the CG:B,I tag with a <len>S<reflen>N placeholder */
        const uint8_t *cg = NULL; uint32_t cg_n = 0;
        uint8_t rev = (flag & 16) ? 1 : 0; int xs_seen = 0;
        for (const uint8_t *a = aux; a + 3 <= rend;) {
            uint8_t t = a[2];
            size_t sz = aux_size(t, a + 3, rend);
            if (sz == 0 || a + 3 + sz > rend) h_fatal(who, "corrupt BAM aux field");
            if (!xs_seen && a[0] == 'X' && a[1] == 'S') { xs_seen = 1; rev = (t == 'A' && a[3] == '+') ? 0 : 1; }
            if (a[0] == 'C' && a[1] == 'G' && t == 'B' && a[3] == 'I') { cg_n = le32(a + 4); cg = a + 8; }
            a += 3 + sz;
        }
        const uint8_t *cp = cig; uint32_t cn = n_cig;
        if (cg && n_cig == 2 && (le32(cig) & 15u) == 4 && (le32(cig) >> 4) == l_seq && (le32(cig + 4) & 15u) == 3) { cp = cg; cn = cg_n; }
        reads_reserve(out, 1, cn);
        add_qname(out, (const char *)name, strnlen((const char *)name, l_read_name), who);
        out->tid[out->n] = refid; out->pos[out->n] = pos; out->rev[out->n] = rev;
        out->cig_off[out->n] = out->n_cig;
        cigsum cs; cigsum_init(&cs);
        for (uint32_t k = 0; k < cn; ++k) { const uint32_t w = le32(cp + 4 * (size_t)k); out->cig[out->n_cig++] = w; cigsum_op(&cs, w); }
        cigsum_store(&cs, out->cig_sum + 3 * (size_t)out->n);
        out->n++;
        out->cig_off[out->n] = out->n_cig;
    }
    return NULL;
}

static void *bam_join_main(void *arg)
{
    bam_piece *pc = (bam_piece *)arg;
    h_reads *q = &pc->piece, *out = pc->dst;
    const int64_t at = pc->at, cat = pc->cat; const size_t nat = pc->nat;
    memcpy(out->tid + at, q->tid, (size_t)q->n * 4); memcpy(out->pos + at, q->pos, (size_t)q->n * 4); memcpy(out->rev + at, q->rev, (size_t)q->n);
    for (int64_t i = 0; i < q->n; ++i) { out->cig_off[at + i] = q->cig_off[i] + cat; out->qname[at + i] = q->qname[i] + (uint32_t)nat; }
    memcpy(out->cig + cat, q->cig, (size_t)q->n_cig * 4);
    memcpy(out->cig_sum + 3 * (size_t)at, q->cig_sum, (size_t)q->n * 12);
    memcpy(out->names.buf + nat, q->names.buf, q->names.len);
    h_reads_free(q);
    return NULL;
}

/* the fields of records starts[0 .. n_mine) of the inflated stream b, extracted on several threads (every thread fills a piece of
 * its own) and APPENDED to *out in order (read names: ids are offsets into the joined table) */
static void bam_extract(const blob *b, const size_t *starts, size_t n_mine, h_chroms *chr, h_reads *out, int skip_unmapped, const char *who)
{
    const char *e = getenv("L2R_THREADS");
    long n_thr = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    if (n_thr > 32) n_thr = 32;
    if (n_thr < 1 || (!e && n_mine < 200000)) n_thr = 1;
    if ((size_t)n_thr > n_mine) n_thr = n_mine ? (long)n_mine : 1;
    bam_piece pc[32];
    pthread_t th[32];
    for (long k = 0; k < n_thr; ++k) {
        pc[k].b = b; pc[k].starts = starts; pc[k].lo = n_mine * (size_t)k / (size_t)n_thr; pc[k].hi = n_mine * (size_t)(k + 1) / (size_t)n_thr;
        pc[k].chr = chr; pc[k].skip_unmapped = skip_unmapped; pc[k].who = who;
        memset(&pc[k].piece, 0, sizeof pc[k].piece);
    }
    for (long k = 1; k < n_thr; ++k) if (pthread_create(&th[k], NULL, bam_piece_main, &pc[k])) h_fatal(who, "pthread_create failed");
    bam_piece_main(&pc[0]);
    for (long k = 1; k < n_thr; ++k) pthread_join(th[k], NULL);
    /* join */
    int64_t n = 0, n_cig = 0; size_t names = 0;
    for (long k = 0; k < n_thr; ++k) { n += pc[k].piece.n; n_cig += pc[k].piece.n_cig; names += pc[k].piece.names.len; }
    if (out->names.len + names >= 0xffffffffu) h_fatal(who, "read names exceed 4 GiB");
    reads_reserve(out, n, n_cig);
    if (out->names.len + names + 1 > out->names.cap) {
        size_t c = out->names.cap ? out->names.cap : 1 << 20;
        while (c < out->names.len + names + 1) c *= 2;
        out->names.buf = (char *)h_realloc(out->names.buf, c); out->names.cap = c;
    }
    /* (every piece is copied into place by a thread of its own: the pages of the joined arrays are touched in parallel) */
    for (long k = 0; k < n_thr; ++k) {
        pc[k].dst = out; pc[k].at = out->n; pc[k].cat = out->n_cig; pc[k].nat = out->names.len;
        out->n += pc[k].piece.n; out->n_cig += pc[k].piece.n_cig; out->names.len += pc[k].piece.names.len;
    }
    out->cig_off[out->n] = out->n_cig;
    for (long k = 1; k < n_thr; ++k) if (pthread_create(&th[k], NULL, bam_join_main, &pc[k])) h_fatal(who, "pthread_create failed");
    bam_join_main(&pc[0]);
    for (long k = 1; k < n_thr; ++k) pthread_join(th[k], NULL);
}

/* A one-process-per-GPU run (dist.py) asks every rank for ITS shard only: the ranks agree on the cuts from the records' core
 * fields alone (chromosome, position, number of CIGAR operations: one cheap pass), and extract names / CIGARs / strands just for
 * their own range.  The cuts are the ones of workload.aligned_shard_bounds: equal shares of 4 * ops + 64 bytes per record, moved to
 * the nearest chromosome boundary.  Only coordinate-sorted BAM input is cut this way (anything else: every rank loads all). */
typedef struct { int rank, world; int64_t lo, hi, n_total; int done; } shard_req;
static shard_req *g_shard = NULL;

static void parse_bam(const blob *b, h_chroms *chr, h_reads *out, int skip_unmapped, int header_only, const char *who)
{
    const uint8_t *p = b->p, *end = p + b->n;
    if (end - p < 12 || memcmp(p, "BAM\1", 4) != 0) h_fatal(who, "not a BAM stream");
    uint32_t l_text = le32(p + 4);
    p += 8 + l_text;
    if (p + 4 > end) h_fatal(who, "truncated BAM header");
    uint32_t n_ref = le32(p); p += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (p + 4 > end) h_fatal(who, "truncated BAM header");
        uint32_t l_name = le32(p); p += 4;
        if (p + l_name + 4 > end) h_fatal(who, "truncated BAM header");
        h_chrom_intern(chr, (const char *)p);            /* NUL terminated */
        p += l_name + 4;
    }
    chr->n_hdr = chr->n;
    if (header_only) return;
    /* the records' places (one cheap pass over the block_size words), then the field extraction on several threads: every
     * thread fills a piece of its own, the pieces are joined in order (read names: ids are offsets into the joined table) */
    size_t n_rec = 0, cap_rec = 1 << 16;
    size_t *starts = (size_t *)h_malloc(cap_rec * sizeof *starts);
    while (p + 4 <= end) {
        const uint32_t bs = le32(p);
        if (bs < 32 || p + 4 + bs > end) h_fatal(who, "truncated BAM record");
        if (n_rec == cap_rec) { cap_rec *= 2; starts = (size_t *)h_realloc(starts, cap_rec * sizeof *starts); }
        starts[n_rec++] = (size_t)(p - b->p);
        p += 4 + bs;
    }
    size_t r_lo = 0, r_hi = n_rec;
    if (g_shard && g_shard->world > 1) {
        shard_req *sq = g_shard;
        sq->n_total = (int64_t)n_rec; sq->lo = 0; sq->hi = (int64_t)n_rec; sq->done = 0;
        /* sorted by (tid, pos)?  weights; chromosome starts */
        int sorted = 1;
        double total_w = 0.0;
        for (size_t i = 0; i < n_rec; ++i) {
            const uint8_t *rec = b->p + starts[i] + 4;
            const int32_t tid = (int32_t)le32(rec), pos = (int32_t)le32(rec + 4);
            if (i) {
                const uint8_t *pr = b->p + starts[i - 1] + 4;
                const int32_t pt = (int32_t)le32(pr), pp = (int32_t)le32(pr + 4);
                if (tid < pt || (tid == pt && pos < pp)) { sorted = 0; break; }
            }
            total_w += 4.0 * (double)le16(rec + 12) + 64.0;
        }
        if (sorted && n_rec) {
            const int W = sq->world;
            size_t *cut = (size_t *)h_malloc((size_t)(W + 1) * sizeof *cut);
            /* ideal cuts: first record at which the running weight reaches k / W of the total */
            size_t *ideal = (size_t *)h_malloc((size_t)(W + 1) * sizeof *ideal);
            {   double run = 0.0; int k = 1; ideal[0] = 0;
                for (size_t i = 0; i < n_rec && k < W; ++i) {
                    while (k < W && run >= total_w * (double)k / (double)W) ideal[k++] = i;
                    run += 4.0 * (double)le16(b->p + starts[i] + 4 + 12) + 64.0;
                }
                while (k <= W) ideal[k++] = n_rec; }
            cut[0] = 0;
            for (int k = 1; k < W; ++k) {
                /* nearest chromosome boundary (0 and n_rec count as boundaries) */
                size_t lo = ideal[k], hi = ideal[k];
                while (lo > 0 && lo < n_rec && le32(b->p + starts[lo] + 4) == le32(b->p + starts[lo - 1] + 4)) --lo;
                while (hi < n_rec && hi > 0 && le32(b->p + starts[hi] + 4) == le32(b->p + starts[hi - 1] + 4)) ++hi;
                size_t c = (ideal[k] - lo <= hi - ideal[k]) ? lo : hi;
                if (c < cut[k - 1]) c = cut[k - 1];
                cut[k] = c;
            }
            cut[W] = n_rec;
            r_lo = cut[sq->rank]; r_hi = cut[sq->rank + 1];
            sq->lo = (int64_t)r_lo; sq->hi = (int64_t)r_hi; sq->done = 1;
            free(cut); free(ideal);
        }
    }
    bam_extract(b, starts + r_lo, r_hi - r_lo, chr, out, skip_unmapped, who);
    free(starts);
}

/* ------------------------------------------------------------------ BAM in bounded memory
 * The reference reads record by record (src/bam2gtf.c:150, src/update_gtf.c:1069).  Here a BGZF-compressed BAM goes through WINDOWS
 * of whole BGZF blocks (128 MiB of the file at a time; L2R_READ_WINDOW overrides): read, inflated block-parallel, its complete
 * records extracted on several threads and appended to the arrays; the bytes of a record that continues in the next window are
 * carried over.  Neither the file nor the inflated stream is ever in memory as a whole: the peak is the record arrays (which the
 * engine needs anyway) plus one window.  h_read_alignments() runs this to the end; bam2gtf consumes it batch by batch. */
struct h_aln_stream {
    FILE *f; const char *who; h_chroms *chr; int skip_unmapped;
    uint8_t *carry; size_t carry_n;
    int header_done, eof;
    size_t window;
    int keep_blk; bgzf_block *blk; size_t n_blk;          /* keep_blk: the last window's block table stays (in_off relative to the window's first byte) */
};

static int file_is_bgzf(FILE *f)
{
    uint8_t hd[18];
    const size_t got = fread(hd, 1, sizeof hd, f);
    fseek(f, 0, SEEK_SET);
    return got == sizeof hd && hd[0] == 0x1f && hd[1] == 0x8b && hd[2] == 8 && (hd[3] & 4) && hd[12] == 'B' && hd[13] == 'C' &&
           hd[14] == 2 && hd[15] == 0 && (hd[10] | (hd[11] << 8)) == 6;
}

/* one window: the next whole blocks of the file, inflated behind the carried bytes; returns the buffer (caller frees), *n_out its
 * length; NULL at the end of the file (the carried bytes stay where they are) */
static uint8_t *stream_window(h_aln_stream *s, size_t *n_out)
{
    *n_out = 0;
    if (s->eof) return NULL;
    uint8_t *raw = (uint8_t *)h_malloc(s->window + 1);
    const size_t n_raw = fread(raw, 1, s->window, s->f);
    if (n_raw == 0) { free(raw); s->eof = 1; return NULL; }
    size_t n_blk = 0, cap = 1024, out_total = s->carry_n, p = 0;
    bgzf_block *blk = (bgzf_block *)h_malloc(cap * sizeof *blk);
    while (n_raw - p >= 18) {
        if (raw[p] != 0x1f || raw[p + 1] != 0x8b || raw[p + 12] != 'B' || raw[p + 13] != 'C') h_fatal(s->who, "not a BGZF block where one was expected");
        const size_t bsize = (size_t)(raw[p + 16] | (raw[p + 17] << 8)) + 1;
        if (bsize < 28) h_fatal(s->who, "corrupt BGZF block");
        if (p + bsize > n_raw) break;
        const uint8_t *tail = raw + p + bsize - 4;
        const size_t isize = (size_t)tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
        if (n_blk == cap) { cap *= 2; blk = (bgzf_block *)h_realloc(blk, cap * sizeof *blk); }
        blk[n_blk].in_off = p + 18; blk[n_blk].in_len = bsize - 18 - 8; blk[n_blk].out_off = out_total; blk[n_blk].out_len = isize;
        ++n_blk; out_total += isize; p += bsize;
    }
    if (n_blk == 0) h_fatal(s->who, n_raw < s->window ? "truncated BGZF block at the end of the file" : "BGZF block larger than the read window");
    if (p < n_raw) fseek(s->f, -(long)(n_raw - p), SEEK_CUR);         /* the next window starts at the block that did not fit */
    uint8_t *out = (uint8_t *)h_malloc(out_total + 1);
    if (s->carry_n) memcpy(out, s->carry, s->carry_n);
    free(s->carry); s->carry = NULL; s->carry_n = 0;
    bgzf_job jb; memset(&jb, 0, sizeof jb);
    jb.in = raw; jb.out = out; jb.blk = blk; jb.n_blk = n_blk; jb.next = 0; pthread_mutex_init(&jb.mu, NULL);
    const char *e = getenv("L2R_THREADS");
    long n_thr = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    if (n_thr > 32) n_thr = 32;
    if (n_thr < 1 || n_blk < 8) n_thr = 1;
    pthread_t th[32];
    for (long k = 1; k < n_thr; ++k) if (pthread_create(&th[k], NULL, bgzf_worker, &jb)) h_fatal(s->who, "pthread_create failed");
    bgzf_worker(&jb);
    for (long k = 1; k < n_thr; ++k) pthread_join(th[k], NULL);
    pthread_mutex_destroy(&jb.mu);
    if (jb.failed) h_fatal(s->who, jb.failed == 2 ? "CRC32 mismatch in a BGZF block" : "corrupt BGZF block");
    if (s->keep_blk) { free(s->blk); s->blk = blk; s->n_blk = n_blk; } else free(blk);
    free(raw);
    *n_out = out_total;
    return out;
}

/* NULL: not a BGZF-compressed BAM (SAM text, gzip, BGZF-compressed SAM): the caller reads the file as a whole */
h_aln_stream *h_aln_stream_open(const char *fn, h_chroms *chr, int skip_unmapped, const char *who)
{
    FILE *f = fopen(fn, "rb");
    if (!f) h_fatal(who, "Can not open \"%s\"\n", fn);
    if (!file_is_bgzf(f)) { fclose(f); return NULL; }
    h_aln_stream *s = (h_aln_stream *)calloc(1, sizeof *s);
    s->f = f; s->who = who; s->chr = chr; s->skip_unmapped = skip_unmapped;
    /* the stream's first block must start with the BAM magic */
    s->window = 65536 + 64;
    size_t n = 0;
    uint8_t *w = stream_window(s, &n);
    const int is_bam = w && n >= 4 && memcmp(w, "BAM\1", 4) == 0;
    free(w);
    if (!is_bam) { fclose(f); free(s); return NULL; }
    fseek(f, 0, SEEK_SET); s->eof = 0;
    s->window = (size_t)128 << 20;
    const char *e = getenv("L2R_READ_WINDOW");
    if (e && atoll(e) >= 65536 + 64) s->window = (size_t)atoll(e);
    return s;
}

/* Appends the next batch of records to *out (an initialised h_reads); returns their number (0: a batch of skipped records), -1 at
 * the end of the file.  The first batch also brings the header into chr. */
int64_t h_aln_stream_next(h_aln_stream *s, h_reads *out)
{
    if (!out->cig_off) { reads_reserve(out, 1, 1); out->cig_off[0] = 0; }          /* (a zeroed h_reads) */
    for (;;) {
        size_t n = 0;
        uint8_t *w = stream_window(s, &n);
        if (!w) {
            if (s->carry_n) h_fatal(s->who, s->header_done ? "truncated BAM record at the end of the file" : "truncated BAM header");
            if (!s->header_done) h_fatal(s->who, "not a BAM stream");
            return -1;
        }
        const uint8_t *p = w, *end = w + n;
        if (!s->header_done) {
            /* the header has to be complete in the buffer; if it is not, the buffer is carried into a longer one */
            int ok = end - p >= 12 && memcmp(p, "BAM\1", 4) == 0;
            if (end - p >= 4 && memcmp(p, "BAM\1", 4) != 0) h_fatal(s->who, "not a BAM stream");
            uint32_t n_ref = 0;
            const uint8_t *q = p;
            if (ok) { ok = (size_t)(end - p) >= 12 + (size_t)le32(p + 4); if (ok) { q = p + 8 + le32(p + 4); n_ref = le32(q); q += 4; } }
            for (uint32_t i = 0; ok && i < n_ref; ++i) {
                ok = q + 4 <= end;
                if (ok) { const uint32_t l_name = le32(q); ok = (size_t)(end - q) >= 8 + (size_t)l_name; if (ok) q += 8 + l_name; }
            }
            if (!ok) { s->carry = w; s->carry_n = n; continue; }
            const uint8_t *h = p + 8 + le32(p + 4) + 4;
            for (uint32_t i = 0; i < n_ref; ++i) { const uint32_t l_name = le32(h); h_chrom_intern(s->chr, (const char *)(h + 4)); h += 8 + l_name; }
            s->chr->n_hdr = s->chr->n;
            s->header_done = 1;
            p = q;
        }
        /* the complete records of the buffer; what is left goes in front of the next window */
        size_t n_rec = 0, cap_rec = 1 << 14;
        size_t *starts = (size_t *)h_malloc(cap_rec * sizeof *starts);
        while (p + 4 <= end) {
            const uint32_t bs = le32(p);
            if (bs < 32) h_fatal(s->who, "truncated BAM record");
            if ((size_t)(end - p) < 4 + (size_t)bs) break;
            if (n_rec == cap_rec) { cap_rec *= 2; starts = (size_t *)h_realloc(starts, cap_rec * sizeof *starts); }
            starts[n_rec++] = (size_t)(p - w);
            p += 4 + bs;
        }
        if (p < end) { s->carry_n = (size_t)(end - p); s->carry = (uint8_t *)h_malloc(s->carry_n); memcpy(s->carry, p, s->carry_n); }
        const int64_t before = out->n;
        if (n_rec) { const blob b = {w, n}; bam_extract(&b, starts, n_rec, s->chr, out, s->skip_unmapped, s->who); }
        free(starts); free(w);
        if (n_rec) return out->n - before;
        /* (a window without a complete record -- one record longer than a window: read on) */
    }
}

void h_aln_stream_close(h_aln_stream *s)
{
    if (!s) return;
    if (s->f) fclose(s->f);
    free(s->carry); free(s);
}

/* ------------------------------------------------------------------ one rank's BLOCK RANGE of a coordinate-sorted BAM
 * A rank of a multi-process run (dist.py) that inflates the whole file to find its shard spends what the ranks were meant to
 * share.  Here rank r of W inflates only the BGZF blocks its records are in, plus the stretch it needs to find where they begin:
 *
 *   target(r)  = the first BGZF block that starts at or behind byte (file size * r / W)            (no inflation: block headers)
 *   B(r)       = the first BAM record that starts at or behind the beginning of that block
 *   S(r)       = the first record at or behind B(r) on ANOTHER chromosome than B(r)                 (S(0) = the first record)
 *   the rank's records = [S(r), S(r + 1)),  S(W) = the end of the file
 *
 * so shards are cut at chromosome boundaries (what the partitioned tail needs: merge_trans never looks across one) and every
 * rank finds both of its ends alone.  Its own end is exact: it streams there record by record.  Its start is found WITHOUT the
 * records before it: B(r) is the first offset in the block at which a chain of eight plausible, coordinate-sorted records begins
 * (block_size, refID, pos, name length, a printable NUL-terminated name, n_cigar, l_seq consistent with block_size).  dist.py
 * compares every rank's start with its predecessor's (exact) end and falls back to loading the whole file everywhere when one pair
 * differs.  info: [0,1] start (file offset of the block, offset inside its inflated bytes)  [2,3] end  [4] compressed bytes this
 * rank inflated  [5] file size  [6] records kept.  Returns 1 = done, 0 = not a BGZF-compressed BAM / no boundary found (caller
 * reads the file the other way). */
typedef struct { int64_t coff; int64_t uoff; } vpos;

static int bgzf_magic_at(const uint8_t *p, size_t n)
{
    return n >= 18 && p[0] == 0x1f && p[1] == 0x8b && p[2] == 8 && (p[3] & 4) && p[12] == 'B' && p[13] == 'C' && p[14] == 2 && p[15] == 0 &&
           (p[10] | (p[11] << 8)) == 6 && (size_t)(p[16] | (p[17] << 8)) + 1 >= 28;
}

/* first block that starts at or behind `target` (fsz: none) */
static int64_t bgzf_block_from(FILE *f, int64_t fsz, int64_t target)
{
    if (target <= 0) return 0;
    if (target >= fsz) return fsz;
    const size_t want = 4 * 65536 + 64;
    uint8_t *buf = (uint8_t *)h_malloc(want);
    fseek(f, (long)target, SEEK_SET);
    const size_t n = fread(buf, 1, want, f);
    int64_t found = fsz;
    for (size_t p = 0; p + 18 <= n && p < 65536 + 18; ++p) {
        if (!bgzf_magic_at(buf + p, n - p)) continue;
        /* two more blocks (or the end of the file) must follow where this one says */
        size_t q = p; int ok = 1;
        for (int k = 0; k < 3 && ok; ++k) {
            const size_t bs = (size_t)(buf[q + 16] | (buf[q + 17] << 8)) + 1;
            q += bs;
            if ((int64_t)(target + (int64_t)q) == fsz) break;
            if (q + 18 > n) { ok = (int64_t)(target + (int64_t)q) < fsz; break; }
            ok = bgzf_magic_at(buf + q, n - q);
        }
        if (ok) { found = target + (int64_t)p; break; }
    }
    free(buf);
    return found;
}

/* one block at file offset coff, inflated behind buf[*n] (grown as needed); returns the block's size in the file, 0 at the end */
static size_t bgzf_inflate_one(FILE *f, int64_t coff, uint8_t **buf, size_t *n, size_t *cap, const char *who)
{
    uint8_t raw[65536 + 64];
    fseek(f, (long)coff, SEEK_SET);
    const size_t got = fread(raw, 1, 18, f);
    if (got == 0) return 0;
    if (!bgzf_magic_at(raw, got)) h_fatal(who, "not a BGZF block where one was expected");
    const size_t bsize = (size_t)(raw[16] | (raw[17] << 8)) + 1;
    if (fread(raw + 18, 1, bsize - 18, f) != bsize - 18) h_fatal(who, "truncated BGZF block");
    const uint8_t *tail = raw + bsize - 4;
    const size_t isize = (size_t)tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
    if (*n + isize + 1 > *cap) { while (*n + isize + 1 > *cap) *cap = *cap ? *cap * 2 : 1 << 20; *buf = (uint8_t *)h_realloc(*buf, *cap); }
    if (isize) {
        z_stream z; memset(&z, 0, sizeof z);
        if (inflateInit2(&z, -15) != Z_OK) h_fatal(who, "zlib");
        z.next_in = raw + 18; z.avail_in = (uInt)(bsize - 18 - 8); z.next_out = *buf + *n; z.avail_out = (uInt)isize;
        const int rc = inflate(&z, Z_FINISH);
        inflateEnd(&z);
        if (rc != Z_STREAM_END || z.avail_out != 0) h_fatal(who, "corrupt BGZF block");
    }
    *n += isize;
    return bsize;
}

/* the BAM header at the beginning of the file -> chr; *after = where the first record begins */
static int bam_header_light(FILE *f, h_chroms *chr, vpos *after, const char *who)
{
    uint8_t *buf = NULL; size_t n = 0, cap = 0;
    int64_t coff = 0;
    int64_t blk_off[4096]; size_t blk_out[4096]; size_t n_blk = 0;
    for (;;) {
        if (n_blk == 4096) { free(buf); return 0; }
        blk_off[n_blk] = coff; blk_out[n_blk] = n;
        const size_t bs = bgzf_inflate_one(f, coff, &buf, &n, &cap, who);
        if (!bs) { free(buf); return 0; }
        ++n_blk; coff += (int64_t)bs;
        if (n < 12) continue;
        if (memcmp(buf, "BAM\1", 4) != 0) { free(buf); return 0; }
        const uint8_t *p = buf, *end = buf + n;
        if ((size_t)(end - p) < 12 + (size_t)le32(p + 4)) continue;
        const uint8_t *q = p + 8 + le32(p + 4);
        const uint32_t n_ref = le32(q); q += 4;
        int ok = 1;
        for (uint32_t i = 0; ok && i < n_ref; ++i) {
            ok = q + 4 <= end;
            if (ok) { const uint32_t l_name = le32(q); ok = (size_t)(end - q) >= 8 + (size_t)l_name; if (ok) q += 8 + l_name; }
        }
        if (!ok) continue;
        const uint8_t *h = p + 8 + le32(p + 4) + 4;
        for (uint32_t i = 0; i < n_ref; ++i) { const uint32_t l_name = le32(h); h_chrom_intern(chr, (const char *)(h + 4)); h += 8 + l_name; }
        chr->n_hdr = chr->n;
        /* (the first record begins at offset q - buf: in the last block whose first byte is not behind it; at the very end of a
         *  block it is the next block's first byte) */
        const size_t at = (size_t)(q - buf);
        size_t k = n_blk - 1;
        while (k > 0 && blk_out[k] > at) --k;
        after->coff = blk_off[k]; after->uoff = (int64_t)(at - blk_out[k]);
        if (at == n) { after->coff = coff; after->uoff = 0; }
        free(buf);
        return 1;
    }
}

static int bam_record_plausible(const uint8_t *p, const uint8_t *end, int32_t n_ref, int32_t *tid, int32_t *pos, uint32_t *bs_out)
{
    if (end - p < 36) return -1;                           /* (not enough bytes to say) */
    const uint32_t bs = le32(p);
    const int32_t ref = (int32_t)le32(p + 4), ps = (int32_t)le32(p + 8);
    const uint32_t l_name = p[12], n_cig = le16(p + 16), l_seq = le32(p + 20);
    const int32_t nref = (int32_t)le32(p + 24), npos = (int32_t)le32(p + 28);
    if (bs < 32 || bs > (1u << 27) || ref < -1 || ref >= n_ref || ps < -1 || l_name < 1 || l_seq > (1u << 27) || nref < -1 || nref >= n_ref || npos < -1) return 0;
    const uint64_t fixed = 32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + l_seq;
    if (fixed > bs) return 0;
    if ((size_t)(end - p) < 36 + (size_t)l_name) return -1;
    const uint8_t *nm = p + 36;
    if (nm[l_name - 1] != 0) return 0;
    for (uint32_t i = 0; i + 1 < l_name; ++i) if (nm[i] < 33 || nm[i] > 126) return 0;
    *tid = ref; *pos = ps; *bs_out = bs;
    return 1;
}

/* sorted BAM order: unmapped records (refID -1) come last */
static int bam_key_le(int32_t t0, int32_t p0, int32_t t1, int32_t p1)
{
    const uint32_t a = (uint32_t)t0, b = (uint32_t)t1;     /* -1 -> 0xffffffff */
    return a < b || (a == b && p0 <= p1);
}

/* S(r) for r > 0 (see above): returns 1 and *S (coff == fsz: the end of the file), 0 when no record boundary was found;
 * *next_begins_before = 1 when a record between B(r) and S(r) already lies at or behind block c1, the next rank's first block: then
 * B(r + 1) is on B(r)'s chromosome, S(r + 1) = S(r), and this rank has no records; *inflated += compressed bytes inflated */
static int bam_find_start(FILE *f, int64_t fsz, int64_t c0, int64_t c1, int32_t n_ref, vpos *S, int *next_begins_before, int64_t *inflated, const char *who)
{
    *next_begins_before = 0;
    if (c0 >= fsz) { S->coff = fsz; S->uoff = 0; return 1; }
    uint8_t *buf = NULL; size_t n = 0, cap = 0;
    size_t cap_blk = 1024, n_blk = 0;
    int64_t *blk_off = (int64_t *)h_malloc(cap_blk * sizeof *blk_off);
    size_t *blk_out = (size_t *)h_malloc(cap_blk * sizeof *blk_out);
    int64_t coff = c0;
    int eof = 0;
#define MORE() do { \
        if (n_blk == cap_blk) { cap_blk *= 2; blk_off = (int64_t *)h_realloc(blk_off, cap_blk * sizeof *blk_off); blk_out = (size_t *)h_realloc(blk_out, cap_blk * sizeof *blk_out); } \
        blk_off[n_blk] = coff; blk_out[n_blk] = n; \
        const size_t bs_ = coff < fsz ? bgzf_inflate_one(f, coff, &buf, &n, &cap, who) : 0; \
        if (!bs_) eof = 1; else { ++n_blk; coff += (int64_t)bs_; *inflated += (int64_t)bs_; } } while (0)
    MORE();
    const size_t first_len = n;                             /* B(r) begins inside the first block (or at a later block's first byte) */
    size_t o = 0; int found = 0;
    /* candidate offsets: every byte of the first block, then (an empty or tiny first block) on into the next ones, up to 1 MB */
    for (; !found; ++o) {
        while (!eof && n < o + (1u << 16)) MORE();
        if (o >= n || o > (1u << 20)) break;
        size_t q = o; int32_t pt = 0, pp = 0; int chain = 0, ok = 1;
        while (chain < 8 && ok) {
            int32_t t, ps; uint32_t bs;
            int rc = bam_record_plausible(buf + q, buf + n, n_ref, &t, &ps, &bs);
            while (rc < 0 && !eof) { MORE(); rc = bam_record_plausible(buf + q, buf + n, n_ref, &t, &ps, &bs); }
            if (rc < 0) { ok = (q == n && chain > 0); break; }                 /* the file ends: a shorter chain that ends exactly there */
            if (rc == 0 || (chain && !bam_key_le(pt, pp, t, ps))) { ok = 0; break; }
            pt = t; pp = ps; ++chain;
            while (!eof && n < q + 4 + (size_t)bs + 36) MORE();
            q += 4 + (size_t)bs;
            if (q > n) { ok = 0; break; }
        }
        if (ok && chain > 0) found = 1;
        if (found) break;
    }
    (void)first_len;
    if (!found) { free(buf); free(blk_off); free(blk_out); return 0; }
    /* from B(r) on until the chromosome changes */
    size_t q = o, kq = 0; int32_t tid0 = 0, walk_t = 0, walk_p = 0; int have = 0;
    for (;;) {
        while (!eof && n < q + 36) MORE();
        if (q + 8 > n) { S->coff = fsz; S->uoff = 0; break; }                  /* no other chromosome behind B(r) */
        uint32_t bs = le32(buf + q);
        int32_t t = (int32_t)le32(buf + q + 4);
        {   /* every record of the walk has to look like one and keep the coordinate order: a start that only LOOKED like a chain of
             * records (eight were checked) would otherwise send the walk after block sizes read from sequence bytes -- buffers of the
             * file's inflated size before dist.py's cross-check rejects the range.  Not plausible: no start, the caller
             * falls back to the whole-file cut. */
            int32_t ps = 0; uint32_t bs2 = 0;
            int rc = bam_record_plausible(buf + q, buf + n, n_ref, &t, &ps, &bs2);
            while (rc < 0 && !eof) { MORE(); rc = bam_record_plausible(buf + q, buf + n, n_ref, &t, &ps, &bs2); }
            if (rc == 0 || (rc > 0 && have && !bam_key_le(walk_t, walk_p, t, ps))) { free(buf); free(blk_off); free(blk_out); return 0; }
            if (rc > 0) { bs = bs2; walk_t = t; walk_p = ps; }
        }
        while (kq + 1 < n_blk && blk_out[kq + 1] <= q) ++kq;                    /* (the block the record begins in) */
        if ((!have || t == tid0) && blk_off[kq] >= c1) *next_begins_before = 1;
        if (!have) { tid0 = t; have = 1; }
        else if (t != tid0) {
            size_t k = n_blk - 1;
            while (k > 0 && blk_out[k] > q) --k;
            S->coff = blk_off[k]; S->uoff = (int64_t)(q - blk_out[k]);
            break;
        }
        while (!eof && n < q + 4 + (size_t)bs) MORE();
        q += 4 + (size_t)bs;
        if (q > n) { free(buf); free(blk_off); free(blk_out); return 0; }     /* truncated */
        if (q > ((size_t)32 << 20)) {
            /* (a chromosome can be long: what lies in front of the current record's block is dropped) */
            size_t k = n_blk - 1;
            while (k > 0 && blk_out[k] > q) --k;
            const size_t shift = blk_out[k];
            if (shift) {
                memmove(buf, buf + shift, n - shift);
                n -= shift; q -= shift;
                for (size_t i = k; i < n_blk; ++i) { blk_off[i - k] = blk_off[i]; blk_out[i - k] = blk_out[i] - shift; }
                n_blk -= k; kq = 0;
            }
        }
    }
#undef MORE
    free(buf); free(blk_off); free(blk_out);
    return 1;
}

int h_read_alignments_blocks(const char *fn, h_chroms *chr, h_reads *out, int skip_unmapped, const char *who, int rank, int world, int64_t info[8])
{
    memset(out, 0, sizeof *out);
    memset(info, 0, 8 * sizeof info[0]);
    FILE *f = fopen(fn, "rb");
    if (!f) h_fatal(who, "Can not open \"%s\"\n", fn);
    if (!file_is_bgzf(f)) { fclose(f); return 0; }
    fseek(f, 0, SEEK_END);
    const int64_t fsz = (int64_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    vpos first;
    if (!bam_header_light(f, chr, &first, who)) { fclose(f); return 0; }
    const int32_t n_ref = (int32_t)chr->n_hdr;
    int64_t inflated = 0;
    vpos S = first;
    const int64_t c1 = rank + 1 < world ? bgzf_block_from(f, fsz, fsz / world * (rank + 1)) : fsz;
    int empty = 0;
    if (rank > 0 && !bam_find_start(f, fsz, bgzf_block_from(f, fsz, fsz / world * rank), c1, n_ref, &S, &empty, &inflated, who)) { fclose(f); return 0; }
    info[0] = S.coff; info[1] = S.uoff; info[5] = fsz;
    reads_reserve(out, 1, 1);
    out->cig_off[0] = 0;
    /* the rank's records: windows of whole blocks from S on (inflated block-parallel), until the first record at or behind block c1
     * has been followed by a record of another chromosome */
    vpos E; E.coff = fsz; E.uoff = 0;
    if (empty) E = S;                                       /* (the next rank begins where this one does) */
    if (S.coff < fsz && !empty) {
        h_aln_stream st; memset(&st, 0, sizeof st);
        st.f = f; st.who = who; st.chr = chr; st.skip_unmapped = skip_unmapped; st.header_done = 1;
        size_t big = (size_t)64 << 20;
        const char *e = getenv("L2R_READ_WINDOW");
        if (e && atoll(e) >= 65536 + 64) big = (size_t)atoll(e);
        st.keep_blk = 1;
        fseek(f, (long)S.coff, SEEK_SET);
        size_t skip = (size_t)S.uoff;
        int phase = 0; int32_t tid_b = 0; int done = 0;
        vpos carry_v = S;                                   /* where the carried bytes begin in the file */
        while (!done) {
            const int64_t win_off = (int64_t)ftell(f);
            const size_t carried = st.carry_n;
            size_t n = 0;
            /* (large windows up to the next rank's target, small ones behind it: what is read past the rank's last record is wasted) */
            st.window = win_off < c1 ? (size_t)(c1 - win_off) + 65536 + 64 : (size_t)2 * (65536 + 64);
            if (st.window > big) st.window = big;
            uint8_t *w = stream_window(&st, &n);
            if (!w) { if (st.carry_n) h_fatal(who, "truncated BAM record at the end of the file"); break; }
            info[4] += (int64_t)ftell(f) - win_off;
            const uint8_t *p = w + skip, *end = w + n;
            skip = 0;
            size_t n_rec = 0, cap_rec = 1 << 14, kb = 0;
            size_t *starts = (size_t *)h_malloc(cap_rec * sizeof *starts);
            while (p + 4 <= end) {
                const uint32_t bs = le32(p);
                if (bs < 32) h_fatal(who, "truncated BAM record");
                if ((size_t)(end - p) < 4 + (size_t)bs) break;
                /* where the record begins in the file */
                const size_t at = (size_t)(p - w);
                vpos v;
                if (at < carried) { v = carry_v; }
                else {
                    while (kb + 1 < st.n_blk && st.blk[kb + 1].out_off <= at) ++kb;
                    v.coff = win_off + (int64_t)st.blk[kb].in_off - 18; v.uoff = (int64_t)(at - st.blk[kb].out_off);
                }
                const int32_t t = (int32_t)le32(p + 4);
                if (phase == 0 && v.coff >= c1) { phase = 1; tid_b = t; }
                else if (phase == 1 && t != tid_b) { E = v; done = 1; break; }
                if (n_rec == cap_rec) { cap_rec *= 2; starts = (size_t *)h_realloc(starts, cap_rec * sizeof *starts); }
                starts[n_rec++] = at;
                p += 4 + bs;
            }
            if (!done && p < end) {
                const size_t at = (size_t)(p - w);
                if (at >= carried) {
                    size_t k2 = 0;
                    while (k2 + 1 < st.n_blk && st.blk[k2 + 1].out_off <= at) ++k2;
                    carry_v.coff = win_off + (int64_t)st.blk[k2].in_off - 18; carry_v.uoff = (int64_t)(at - st.blk[k2].out_off);
                }
                st.carry_n = (size_t)(end - p); st.carry = (uint8_t *)h_malloc(st.carry_n); memcpy(st.carry, p, st.carry_n);
            }
            if (n_rec) { const blob b = {w, n}; bam_extract(&b, starts, n_rec, chr, out, skip_unmapped, who); }
            free(starts); free(w);
        }
        free(st.carry); free(st.blk);
    }
    fclose(f);
    info[2] = E.coff; info[3] = E.uoff; info[4] += inflated; info[6] = out->n;
    return 1;
}

static void read_any(const char *fn, h_chroms *chr, h_reads *out, int skip_unmapped, int header_only, const char *who)
{
    if (out && !header_only && !(g_shard && g_shard->world > 1)) {
        /* a BGZF-compressed BAM: window by window (bounded memory); anything else, and one rank's shard of a file, as a whole */
        h_aln_stream *st = h_aln_stream_open(fn, chr, skip_unmapped, who);
        if (st) {
            reads_reserve(out, 1, 1);
            out->cig_off[0] = 0;
            while (h_aln_stream_next(st, out) >= 0) { }
            h_aln_stream_close(st);
            h_stage_time("  alignments: windows of BGZF blocks (read, inflate, extract)");
            return;
        }
    }
    blob b = h_slurp(fn, who);
    if (!header_only) h_stage_time("  alignments: file read + inflate");
    h_reads tmp; memset(&tmp, 0, sizeof tmp);
    h_reads *dst = out ? out : &tmp;
    reads_reserve(dst, 1, 1);
    dst->cig_off[0] = 0;
    if (b.n >= 4 && memcmp(b.p, "BAM\1", 4) == 0) parse_bam(&b, chr, dst, skip_unmapped, header_only, who);
    else parse_sam(&b, chr, dst, skip_unmapped, header_only, who);
    free(b.p);
    if (!out) h_reads_free(&tmp);
}

void h_read_alignments(const char *fn, h_chroms *chr, h_reads *out, int skip_unmapped, const char *who)
{
    memset(out, 0, sizeof *out);
    read_any(fn, chr, out, skip_unmapped, 0, who);
}

/* h_read_alignments for one rank of `world` (see shard_req): *lo, *hi = the rank's record range, *n_total = records in the file;
 * returns 1 when only that range was loaded, 0 when everything was (input that is not coordinate-sorted BAM). */
int h_read_alignments_shard(const char *fn, h_chroms *chr, h_reads *out, int skip_unmapped, const char *who, int rank, int world,
                            int64_t *lo, int64_t *hi, int64_t *n_total)
{
    shard_req sq = { rank, world, 0, 0, 0, 0 };
    memset(out, 0, sizeof *out);
    g_shard = &sq;
    read_any(fn, chr, out, skip_unmapped, 0, who);
    g_shard = NULL;
    if (!sq.done) { sq.lo = 0; sq.hi = out->n; sq.n_total = out->n; }
    *lo = sq.lo; *hi = sq.hi; *n_total = sq.n_total;
    return sq.done;
}

void h_read_header_only(const char *fn, h_chroms *chr, const char *who)
{
    /* a BGZF-compressed BAM: the blocks the header lies in and nothing else (the whole file used to be read and inflated
     * here, in front of the reader that streams it again); anything else -- SAM text, gzip -- as a whole */
    FILE *f = fopen(fn, "rb");
    if (!f) h_fatal(who, "Can not open \"%s\"\n", fn);
    if (file_is_bgzf(f)) {
        fseek(f, 0, SEEK_SET);
        vpos first;
        if (bam_header_light(f, chr, &first, who)) { fclose(f); return; }
    }
    fclose(f);
    read_any(fn, chr, NULL, 1, 1, who);
}
