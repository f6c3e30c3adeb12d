/* filter.c -- `lr2rmats filter` (reference src/bam_filter.c:88-164): keep, per read, the best alignment that passes the
 * coverage / identity / -r tests, if it is clearly better than the second best; BAM to stdout.
 *
 *   records      SAM text, gzip/BGZF SAM or BAM -> BAM-encoded records in memory (a BAM input is referenced in place, a SAM
 *                line is encoded the way htslib's sam_parse1 does, SAMv1 4.2) + the few fields the tests read
 *   test, score  the engine: l2r_filter_score()   (gtf_filter :61-86, remove_overlap :48-59)
 *   groups       runs of consecutive KEPT records with one read name (the loop of :128-154 never sees a dropped record)
 *   choice       the engine: l2r_filter_select()  (best / second best / intron count, :131-148)
 *   output       BGZF-compressed BAM on stdout, blocks deflated on several threads (bam_writer below)
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <errno.h>
#include <getopt.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#include "l2r_host.h"

#define COV_RATIO 0.67          /* src/bam_filter.c:10-12 */
#define MAP_QUAL  0.75
#define SEC_RATIO 0.98
#define MIN_INTRON_NUM 0        /* src/gtf.h:123 */

static int filter_usage(void)
{
    /* src/bam_filter.c:16-30 */
    fprintf(stderr, "\n");
    fprintf(stderr, "Usage:   %s filter [option] <in.bam/sam> | samtools sort > out.sort.bam\n\n", "lr2rmats");
    fprintf(stderr, "Options:\n");
    fprintf(stderr, "         -v --coverage   [FLOAT]    minimum fraction of aligned bases. [%.2f]\n", COV_RATIO);
    fprintf(stderr, "         -q --map-qual   [FLOAT]    minimum fraction of identically aligned bases. [%.2f]\n", MAP_QUAL);
    fprintf(stderr, "         -s --sec-rat    [FLOAT]    maximum ratio of second best and best score to retain the best\n");
    fprintf(stderr, "                                    alignment, or no alignments will be retained. [%.2f]\n", SEC_RATIO);
    fprintf(stderr, "         -i --intron     [INT]      minimum number of intron indicated by the alignment. [%d]\n", MIN_INTRON_NUM);
    fprintf(stderr, "         -r --remove-gtf [STR]      remove all the alignment record that overlap with transcript in this GTF file. [NONE]\n");
    fprintf(stderr, "\n");
    return 1;
}

/* ------------------------------------------------------------------ records */

static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static inline void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

static void rec_reserve(h_records *r, int64_t more_cig)
{
    if (r->n + 2 > r->cap) {
        int64_t c = r->cap ? r->cap * 2 : 1 << 16;
        r->rec_off = (int64_t *)h_realloc(r->rec_off, (size_t)(c + 1) * 8); r->cig_off = (int64_t *)h_realloc(r->cig_off, (size_t)(c + 1) * 8);
        r->flag = (uint16_t *)h_realloc(r->flag, (size_t)c * 2); r->tid = (int32_t *)h_realloc(r->tid, (size_t)c * 4);
        r->pos = (int32_t *)h_realloc(r->pos, (size_t)c * 4); r->l_qseq = (int32_t *)h_realloc(r->l_qseq, (size_t)c * 4);
        r->nm = (int32_t *)h_realloc(r->nm, (size_t)c * 4); r->nm_seen = (uint8_t *)h_realloc(r->nm_seen, (size_t)c);
        r->cap = c;
    }
    if (r->n_cig + more_cig + 1 > r->cap_cig) {
        int64_t c = r->cap_cig ? r->cap_cig * 2 : 1 << 20;
        while (c < r->n_cig + more_cig + 1) c *= 2;
        r->cig = (uint32_t *)h_realloc(r->cig, (size_t)c * 4); r->cap_cig = c;
    }
}

void h_records_free(h_records *r)
{
    free(r->hdr); free(r->buf); free(r->rec_off); free(r->cig_off); free(r->flag); free(r->tid); free(r->pos); free(r->l_qseq);
    free(r->nm); free(r->nm_seen); free(r->cig);
    memset(r, 0, sizeof *r);
}

static size_t aux_size(uint8_t type, const uint8_t *p, const uint8_t *end)
{
    switch (type) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    case 'Z': case 'H': { const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(end - p)); return z ? (size_t)(z - p) + 1 : 0; }
    case 'B': {
        if (end - p < 5) return 0;
        size_t w;
        switch (p[0]) { case 'c': case 'C': w = 1; break; case 's': case 'S': w = 2; break; case 'i': case 'I': case 'f': w = 4; break; default: return 0; }
        return 5 + w * (size_t)le32(p + 1);
    }
    default: return 0;
    }
}

/* The fields the tests read, from the BAM-encoded record at r->buf + off (block_size word first).  The CIGAR is the real one:
 * beyond 65535 operations BAM keeps it in the CG:B,I tag behind a <l_seq>S<ref len>N placeholder (htslib swaps it back in
 * when it reads the record).  NM: bam_aux2i() of the first NM tag -- its value for the integer types, 0 for any other. */
static void index_record(h_records *r, int64_t off, const char *who)
{
    const uint8_t *p = r->buf + off;
    const uint32_t bs = le32(p);
    const uint8_t *rec = p + 4, *rend = rec + bs;
    const uint32_t l_read_name = rec[8], n_cig = le16(rec + 12), l_seq = le32(rec + 16);
    const uint8_t *cig = rec + 32 + l_read_name;
    const uint8_t *aux = cig + 4 * (size_t)n_cig + (l_seq + 1) / 2 + l_seq;
    if (bs < 32 || aux > rend || l_read_name == 0) h_fatal(who, "corrupt BAM record");
    const uint8_t *cg = NULL; uint32_t cg_n = 0;
    int32_t nm = 0; uint8_t nm_seen = 0;
    for (const uint8_t *a = aux; a + 3 <= rend;) {
        const uint8_t t = a[2];
        const size_t sz = aux_size(t, a + 3, rend);
        if (sz == 0 || a + 3 + sz > rend) h_fatal(who, "corrupt BAM aux field");
        if (!nm_seen && a[0] == 'N' && a[1] == 'M') {
            nm_seen = 1;
            switch (t) {
            case 'c': nm = (int8_t)a[3]; break; case 'C': nm = a[3]; break;
            case 's': nm = (int16_t)le16(a + 3); break; case 'S': nm = le16(a + 3); break;
            case 'i': case 'I': nm = (int32_t)le32(a + 3); break;
            default: nm = 0;
            }
        }
        if (a[0] == 'C' && a[1] == 'G' && t == 'B' && a[3] == 'I') { cg_n = le32(a + 4); cg = a + 8; }
        a += 3 + sz;
    }
    const uint8_t *cp = cig; uint32_t cn = n_cig;
    if (cg && n_cig == 2 && (le32(cig) & 15u) == 4 && (le32(cig) >> 4) == l_seq && (le32(cig + 4) & 15u) == 3) { cp = cg; cn = cg_n; }
    rec_reserve(r, cn);
    const int64_t i = r->n;
    r->rec_off[i] = off; r->rec_off[i + 1] = off + 4 + bs;
    r->tid[i] = (int32_t)le32(rec); r->pos[i] = (int32_t)le32(rec + 4); r->flag[i] = le16(rec + 14); r->l_qseq[i] = (int32_t)l_seq;
    r->nm[i] = nm; r->nm_seen[i] = nm_seen;
    r->cig_off[i] = r->n_cig;
    for (uint32_t k = 0; k < cn; ++k) r->cig[r->n_cig++] = le32(cp + 4 * (size_t)k);
    r->cig_off[i + 1] = r->n_cig;
    r->n = i + 1;
}

static void records_from_bam(h_blob *b, h_chroms *chr, h_records *r, const char *who)
{
    const uint8_t *p = b->p, *end = p + b->n;
    if (end - p < 12) h_fatal(who, "truncated BAM header");
    const uint32_t l_text = le32(p + 4);
    p += 8 + l_text;
    if (p + 4 > end) h_fatal(who, "truncated BAM header");
    const uint32_t n_ref = le32(p); p += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (p + 4 > end) h_fatal(who, "truncated BAM header");
        const uint32_t l_name = le32(p); p += 4;
        if (p + l_name + 4 > end) h_fatal(who, "truncated BAM header");
        h_chrom_intern(chr, (const char *)p);
        p += l_name + 4;
    }
    chr->n_hdr = chr->n;
    r->hdr_len = (size_t)(p - b->p);
    r->hdr = (uint8_t *)h_malloc(r->hdr_len + 1);
    memcpy(r->hdr, b->p, r->hdr_len);
    r->buf = b->p; r->buf_len = b->n; b->p = NULL;           /* the records stay where they were inflated */
    int64_t off = (int64_t)r->hdr_len;
    while (off + 4 <= (int64_t)r->buf_len) {
        const uint32_t bs = le32(r->buf + off);
        if (bs < 32 || off + 4 + (int64_t)bs > (int64_t)r->buf_len) h_fatal(who, "truncated BAM record");
        index_record(r, off, who);
        off += 4 + bs;
    }
}

/* ---- SAM line -> BAM record (what htslib's sam_parse1 + bam_write1 leave on disk; SAMv1 1.4 and 4.2) */

static int reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

static uint8_t nt16(char c)
{
    switch (c) {
    case '=': return 0; case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'M': case 'm': return 3;
    case 'G': case 'g': return 4; case 'R': case 'r': return 5; case 'S': case 's': return 6; case 'V': case 'v': return 7;
    case 'T': case 't': return 8; case 'W': case 'w': return 9; case 'Y': case 'y': return 10; case 'H': case 'h': return 11;
    case 'K': case 'k': return 12; case 'D': case 'd': return 13; case 'B': case 'b': return 14;
    default: return 15;
    }
}

typedef struct { uint8_t *p; size_t n, cap; } bytes;
static void by_need(bytes *b, size_t more) { if (b->n + more > b->cap) { size_t c = b->cap ? b->cap * 2 : 1 << 20; while (c < b->n + more) c *= 2; b->p = (uint8_t *)h_realloc(b->p, c); b->cap = c; } }
static void by_put(bytes *b, const void *s, size_t n) { by_need(b, n); memcpy(b->p + b->n, s, n); b->n += n; }
static void by_u8(bytes *b, uint32_t v) { by_need(b, 1); b->p[b->n++] = (uint8_t)v; }
static void by_u16(bytes *b, uint32_t v) { by_need(b, 2); put16(b->p + b->n, v); b->n += 2; }
static void by_u32(bytes *b, uint32_t v) { by_need(b, 4); put32(b->p + b->n, v); b->n += 4; }

static void encode_aux_int(bytes *o, long long x)
{
    /* the smallest type that holds the value (sam_parse1) */
    if (x < 0) {
        if (x >= -128) { by_u8(o, 'c'); by_u8(o, (uint32_t)(int8_t)x); }
        else if (x >= -32768) { by_u8(o, 's'); by_u16(o, (uint32_t)(int16_t)x); }
        else { by_u8(o, 'i'); by_u32(o, (uint32_t)(int32_t)x); }
    } else {
        if (x <= 255) { by_u8(o, 'C'); by_u8(o, (uint32_t)x); }
        else if (x <= 65535) { by_u8(o, 'S'); by_u16(o, (uint32_t)x); }
        else { by_u8(o, 'I'); by_u32(o, (uint32_t)x); }
    }
}

static void encode_aux(bytes *o, const char *a, const char *ae, const char *who)
{
    if (ae - a < 5 || a[2] != ':' || a[4] != ':') h_fatal(who, "malformed SAM aux field \"%.*s\"", (int)(ae - a), a);
    by_put(o, a, 2);
    const char t = a[3], *v = a + 5;
    switch (t) {
    case 'A': case 'a': case 'c': case 'C': by_u8(o, 'A'); by_u8(o, (uint8_t)*v); break;
    case 'i': case 'I': encode_aux_int(o, strtoll(v, NULL, 10)); break;
    case 'f': { float f = (float)strtod(v, NULL); uint32_t w; memcpy(&w, &f, 4); by_u8(o, 'f'); by_u32(o, w); break; }
    case 'd': { double d = strtod(v, NULL); uint64_t w; memcpy(&w, &d, 8); by_u8(o, 'd'); by_u32(o, (uint32_t)w); by_u32(o, (uint32_t)(w >> 32)); break; }
    case 'Z': case 'H': by_u8(o, (uint8_t)t); by_put(o, v, (size_t)(ae - v)); by_u8(o, 0); break;
    case 'B': {
        if (ae - v < 1) h_fatal(who, "malformed SAM aux array");
        const char st = *v;
        int w = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
        if (!w) h_fatal(who, "unknown SAM aux array type '%c'", st);
        uint32_t n = 0;
        for (const char *q = v + 1; q < ae; ++q) n += *q == ',';
        by_u8(o, 'B'); by_u8(o, (uint8_t)st); by_u32(o, n);
        const char *q = v + 1;
        for (uint32_t k = 0; k < n; ++k) {
            ++q;                                             /* the comma */
            char *e2;
            if (st == 'f') { float f = (float)strtod(q, &e2); uint32_t x; memcpy(&x, &f, 4); by_u32(o, x); }
            else { long long x = strtoll(q, &e2, 10); if (w == 1) by_u8(o, (uint32_t)x); else if (w == 2) by_u16(o, (uint32_t)x); else by_u32(o, (uint32_t)x); }
            q = e2;
        }
        break;
    }
    default: h_fatal(who, "unknown SAM aux type '%c'", t);
    }
}

/* The lines of [p, end) (whole lines) -> BAM-encoded records + their index, into a piece of its own (several pieces are made
 * on several threads and joined by records_join) */
static void encode_sam_lines(const char *p, const char *end, const h_chroms *chr, h_records *r, const char *who)
{
    memset(r, 0, sizeof *r);
    rec_reserve(r, 1);
    r->cig_off[0] = 0; r->rec_off[0] = 0;
    bytes out = {NULL, 0, 0};
    int64_t *offs = NULL; int64_t n_off = 0, cap_off = 0;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *e = nl ? nl : end;
        const char *le = (e > p && e[-1] == '\r') ? e - 1 : e;
        if (le == p || *p == '@') { p = nl ? nl + 1 : end; continue; }
        const char *f[12], *fe[11]; int nf = 0; const char *q = p;
        while (nf < 11) {
            f[nf] = q;
            const char *t = (const char *)memchr(q, '\t', (size_t)(le - q));
            fe[nf++] = t ? t : le;
            if (!t) { q = le; break; }
            q = t + 1;
        }
        if (nf < 11) h_fatal(who, "truncated SAM record");
        f[11] = q;                                           /* first aux field, or le */
#define FLEN(k) ((size_t)(fe[k] - f[k]))
        const size_t l_name = FLEN(0);
        if (l_name == 0 || l_name > 254) h_fatal(who, "read name of %zu characters", l_name);
        uint32_t flag = (uint32_t)strtoul(f[1], NULL, 0);
        char name[H_NAME_MAX];
        int tid = -1, mtid = -1;
        {   size_t len = FLEN(2);
            if (!(len == 1 && f[2][0] == '*')) {
                if (len >= H_NAME_MAX) h_fatal(who, "reference name too long");
                memcpy(name, f[2], len); name[len] = 0;
                tid = h_chrom_find(chr, name, chr->n_hdr);
                if (tid < 0) h_fatal(who, "record \"%.*s\": reference \"%s\" is not in the header", (int)l_name, f[0], name);
            }
            len = FLEN(6);
            if (len == 1 && f[6][0] == '=') mtid = tid;
            else if (!(len == 1 && f[6][0] == '*')) {
                if (len >= H_NAME_MAX) h_fatal(who, "reference name too long");
                memcpy(name, f[6], len); name[len] = 0;
                mtid = h_chrom_find(chr, name, chr->n_hdr);
            }
        }
        const int32_t pos = (int32_t)strtol(f[3], NULL, 10) - 1, mpos = (int32_t)strtol(f[7], NULL, 10) - 1, tlen = (int32_t)strtol(f[8], NULL, 10);
        const uint32_t mapq = (uint32_t)strtoul(f[4], NULL, 10);
        /* CIGAR */
        const size_t rec_at = out.n;
        by_need(&out, 4 + 32 + l_name + 1);
        out.n += 4 + 32;
        by_put(&out, f[0], l_name); by_u8(&out, 0);
        uint32_t n_cig = 0; int64_t rlen = 0;
        const size_t cig_at = out.n;
        {   const char *c = f[5], *ce = f[5] + FLEN(5);
            if (ce - c == 1 && *c == '*') { flag |= 4u; rlen = 1; }            /* "mapped query must have a CIGAR; treated as unmapped" */
            else {
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
                    by_u32(&out, (len << 4) | op);
                    if ((0x18du >> op) & 1u) rlen += len;
                    ++n_cig; ++c;
                }
            }
        }
        /* SEQ, QUAL */
        uint32_t l_seq = 0;
        {   const size_t ls = FLEN(9), lq = FLEN(10);
            if (!(ls == 1 && f[9][0] == '*')) {
                l_seq = (uint32_t)ls;
                by_need(&out, (ls + 1) / 2 + ls);
                uint8_t *s = out.p + out.n;
                memset(s, 0, (ls + 1) / 2);
                for (size_t k = 0; k < ls; ++k) s[k >> 1] |= (uint8_t)(nt16(f[9][k]) << ((~k & 1) << 2));
                out.n += (ls + 1) / 2;
                uint8_t *ql = out.p + out.n;
                if (lq == 1 && f[10][0] == '*') memset(ql, 0xff, ls);
                else {
                    if (lq != ls) h_fatal(who, "SEQ and QUAL of record \"%.*s\" differ in length", (int)l_name, f[0]);
                    for (size_t k = 0; k < ls; ++k) ql[k] = (uint8_t)(f[10][k] - 33);
                }
                out.n += ls;
            }
        }
        /* a CIGAR beyond 65535 operations goes into CG:B,I behind a <l_seq>S<rlen>N placeholder (bam_write1) */
        uint32_t *long_cig = NULL; uint32_t long_n = 0;
        if (n_cig > 65535) {
            long_n = n_cig;
            long_cig = (uint32_t *)h_malloc((size_t)n_cig * 4);
            memcpy(long_cig, out.p + cig_at, (size_t)n_cig * 4);
            memmove(out.p + cig_at + 8, out.p + cig_at + (size_t)n_cig * 4, out.n - (cig_at + (size_t)n_cig * 4));
            out.n -= (size_t)n_cig * 4 - 8;
            put32(out.p + cig_at, (l_seq << 4) | 4u); put32(out.p + cig_at + 4, ((uint32_t)rlen << 4) | 3u);
            n_cig = 2;
        }
        for (const char *a = f[11]; a < le;) {
            const char *t = (const char *)memchr(a, '\t', (size_t)(le - a));
            const char *ae = t ? t : le;
            if (ae > a) encode_aux(&out, a, ae, who);
            if (!t) break;
            a = t + 1;
        }
        if (long_cig) {
            by_put(&out, "CGBI", 4); by_u32(&out, long_n);
            by_put(&out, long_cig, (size_t)long_n * 4);     /* (little-endian host) */
            free(long_cig);
        }
        uint8_t *rec = out.p + rec_at;
        put32(rec, (uint32_t)(out.n - rec_at - 4));
        put32(rec + 4, (uint32_t)tid); put32(rec + 8, (uint32_t)pos);
        rec[12] = (uint8_t)(l_name + 1); rec[13] = (uint8_t)mapq; put16(rec + 14, (uint32_t)reg2bin(pos, pos + rlen));
        put16(rec + 16, n_cig); put16(rec + 18, flag); put32(rec + 20, l_seq);
        put32(rec + 24, (uint32_t)mtid); put32(rec + 28, (uint32_t)mpos); put32(rec + 32, (uint32_t)tlen);
        if (n_off == cap_off) { cap_off = cap_off ? cap_off * 2 : 1 << 16; offs = (int64_t *)h_realloc(offs, (size_t)cap_off * 8); }
        offs[n_off++] = (int64_t)rec_at;
        p = nl ? nl + 1 : end;
#undef FLEN
    }
    r->buf = out.p; r->buf_len = out.n;
    for (int64_t k = 0; k < n_off; ++k) index_record(r, offs[k], who);
    free(offs);
}

typedef struct { const char *p, *end; const h_chroms *chr; h_records piece; const char *who; } sam_piece;
static void *sam_piece_main(void *arg) { sam_piece *q = (sam_piece *)arg; encode_sam_lines(q->p, q->end, q->chr, &q->piece, q->who); return NULL; }

/* pieces, in order -> one h_records (header fields of `r` are kept) */
static void records_join(h_records *r, sam_piece *pc, int n_pc)
{
    int64_t n = 0, n_cig = 0; size_t bytes_total = 0;
    for (int k = 0; k < n_pc; ++k) { n += pc[k].piece.n; n_cig += pc[k].piece.n_cig; bytes_total += pc[k].piece.buf_len; }
    r->n = 0; r->cap = 0; r->n_cig = 0; r->cap_cig = 0;
    r->rec_off = (int64_t *)h_malloc((size_t)(n + 2) * 8); r->cig_off = (int64_t *)h_malloc((size_t)(n + 2) * 8);
    r->flag = (uint16_t *)h_malloc((size_t)(n + 1) * 2); r->tid = (int32_t *)h_malloc((size_t)(n + 1) * 4); r->pos = (int32_t *)h_malloc((size_t)(n + 1) * 4);
    r->l_qseq = (int32_t *)h_malloc((size_t)(n + 1) * 4); r->nm = (int32_t *)h_malloc((size_t)(n + 1) * 4); r->nm_seen = (uint8_t *)h_malloc((size_t)n + 1);
    r->cig = (uint32_t *)h_malloc((size_t)(n_cig + 1) * 4);
    r->buf = (uint8_t *)h_malloc(bytes_total + 1); r->buf_len = bytes_total;
    r->cap = n + 1; r->cap_cig = n_cig + 1;
    int64_t at = 0, cat = 0; size_t bat = 0;
    for (int k = 0; k < n_pc; ++k) {
        h_records *q = &pc[k].piece;
        memcpy(r->buf + bat, q->buf, q->buf_len);
        for (int64_t i = 0; i < q->n; ++i) { r->rec_off[at + i] = q->rec_off[i] + (int64_t)bat; r->cig_off[at + i] = q->cig_off[i] + cat; }
        memcpy(r->flag + at, q->flag, (size_t)q->n * 2); memcpy(r->tid + at, q->tid, (size_t)q->n * 4); memcpy(r->pos + at, q->pos, (size_t)q->n * 4);
        memcpy(r->l_qseq + at, q->l_qseq, (size_t)q->n * 4); memcpy(r->nm + at, q->nm, (size_t)q->n * 4); memcpy(r->nm_seen + at, q->nm_seen, (size_t)q->n);
        memcpy(r->cig + cat, q->cig, (size_t)q->n_cig * 4);
        at += q->n; cat += q->n_cig; bat += q->buf_len;
        uint8_t *hdr_keep = q->hdr; q->hdr = NULL; (void)hdr_keep;
        h_records_free(q);
    }
    r->rec_off[at] = (int64_t)bat; r->cig_off[at] = cat;
    r->n = at; r->n_cig = cat;
}

static void records_from_sam(const h_blob *b, h_chroms *chr, h_records *r, const char *who)
{
    const char *p = (const char *)b->p, *end = p + b->n;
    /* header: the '@' lines verbatim are the BAM header text; the references come from the @SQ lines */
    bytes text = {NULL, 0, 0}, refs = {NULL, 0, 0};
    uint32_t n_ref = 0;
    while (p < end && *p == '@') {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *e = nl ? nl : end;
        by_put(&text, p, (size_t)(e - p)); by_u8(&text, '\n');
        const char *le = (e > p && e[-1] == '\r') ? e - 1 : e;
        if (le - p > 3 && memcmp(p, "@SQ", 3) == 0) {
            const char *sn = NULL, *sne = NULL; long long ln = 0;
            for (const char *q = p; q < le;) {
                const char *t = (const char *)memchr(q, '\t', (size_t)(le - q));
                const char *fe = t ? t : le;
                if (fe - q > 3 && memcmp(q, "SN:", 3) == 0) { sn = q + 3; sne = fe; }
                if (fe - q > 3 && memcmp(q, "LN:", 3) == 0) ln = strtoll(q + 3, NULL, 10);
                if (!t) break;
                q = t + 1;
            }
            if (sn) {
                char name[H_NAME_MAX];
                if (sne - sn >= H_NAME_MAX) h_fatal(who, "reference name of 100 or more characters");
                memcpy(name, sn, (size_t)(sne - sn)); name[sne - sn] = 0;
                h_chrom_intern(chr, name);
                by_u32(&refs, (uint32_t)(sne - sn) + 1); by_put(&refs, name, (size_t)(sne - sn) + 1); by_u32(&refs, (uint32_t)ln);
                ++n_ref;
            }
        }
        p = nl ? nl + 1 : end;
    }
    chr->n_hdr = chr->n;
    r->hdr_len = 4 + 4 + text.n + 4 + refs.n;
    r->hdr = (uint8_t *)h_malloc(r->hdr_len + 1);
    memcpy(r->hdr, "BAM\1", 4); put32(r->hdr + 4, (uint32_t)text.n); memcpy(r->hdr + 8, text.p, text.n);
    put32(r->hdr + 8 + text.n, n_ref); memcpy(r->hdr + 12 + text.n, refs.p, refs.n);
    free(text.p); free(refs.p);

    /* the records: the text is cut at line ends into one piece per thread (L2R_THREADS, default = online CPUs up to 32; small
     * inputs: one), every piece is encoded on its own, the pieces are joined in order */
    {
        const char *e = getenv("L2R_THREADS");
        long n_thr = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
        if (n_thr > 32) n_thr = 32;
        const size_t body = (size_t)(end - p);
        if (n_thr < 1 || (!e && body < ((size_t)8 << 20))) n_thr = 1;
        sam_piece pc[32];
        int n_pc = 0;
        const char *q = p;
        for (long k = 0; k < n_thr && q < end; ++k) {
            const char *stop = (k == n_thr - 1) ? end : p + body * (size_t)(k + 1) / (size_t)n_thr;
            if (stop < q) stop = q;
            if (stop < end) { const char *nl = (const char *)memchr(stop, '\n', (size_t)(end - stop)); stop = nl ? nl + 1 : end; }
            pc[n_pc].p = q; pc[n_pc].end = stop; pc[n_pc].chr = chr; pc[n_pc].who = who; ++n_pc;
            q = stop;
        }
        pthread_t th[32];
        for (int k = 1; k < n_pc; ++k) if (pthread_create(&th[k], NULL, sam_piece_main, &pc[k])) h_fatal(who, "pthread_create failed");
        if (n_pc) sam_piece_main(&pc[0]);
        for (int k = 1; k < n_pc; ++k) pthread_join(th[k], NULL);
        uint8_t *hdr = r->hdr; const size_t hdr_len = r->hdr_len;
        free(r->rec_off); free(r->cig_off); free(r->flag); free(r->tid); free(r->pos); free(r->l_qseq); free(r->nm); free(r->nm_seen); free(r->cig);
        records_join(r, pc, n_pc);
        r->hdr = hdr; r->hdr_len = hdr_len;
    }
}

void h_read_records(const char *fn, h_chroms *chr, h_records *r, const char *who)
{
    memset(r, 0, sizeof *r);
    h_blob b = h_slurp(fn, who);
    rec_reserve(r, 1);
    r->cig_off[0] = 0; r->rec_off[0] = 0;
    if (b.n >= 4 && memcmp(b.p, "BAM\1", 4) == 0) records_from_bam(&b, chr, r, who);
    else records_from_sam(&b, chr, r, who);
    free(b.p);
}

/* ------------------------------------------------------------------ BAM writer: BGZF blocks deflated on several threads */

#define BGZF_PAYLOAD 0xff00      /* bytes of input per block (htslib BGZF_BLOCK_SIZE) */

typedef struct { size_t in_off, in_len; uint8_t *out; size_t out_len; } wblock;
typedef struct { const uint8_t *in; wblock *blk; size_t n_blk, next; pthread_mutex_t mu; int level, failed; } wjob;

static void *deflate_worker(void *arg)
{
    wjob *jb = (wjob *)arg;
    z_stream z; memset(&z, 0, sizeof z);
    if (deflateInit2(&z, jb->level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { jb->failed = 1; return NULL; }
    for (;;) {
        pthread_mutex_lock(&jb->mu);
        size_t lo = jb->next, hi = lo + 16 < jb->n_blk ? lo + 16 : jb->n_blk;
        jb->next = hi;
        pthread_mutex_unlock(&jb->mu);
        if (lo >= hi) break;
        for (size_t k = lo; k < hi; ++k) {
            wblock *b = &jb->blk[k];
            const size_t cap = 18 + deflateBound(&z, (uLong)b->in_len) + 8 + 64;
            b->out = (uint8_t *)h_malloc(cap);
            deflateReset(&z);
            z.next_in = (Bytef *)(jb->in + b->in_off); z.avail_in = (uInt)b->in_len;
            z.next_out = b->out + 18; z.avail_out = (uInt)(cap - 18 - 8);
            if (deflate(&z, Z_FINISH) != Z_STREAM_END) { jb->failed = 1; break; }
            size_t clen = (cap - 18 - 8) - z.avail_out, total = 18 + clen + 8;
            if (total > 65536) {                             /* does not happen for 0xff00 bytes of input; stored block as the way out */
                deflateEnd(&z);
                if (deflateInit2(&z, 0, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { jb->failed = 1; break; }
                z.next_in = (Bytef *)(jb->in + b->in_off); z.avail_in = (uInt)b->in_len;
                z.next_out = b->out + 18; z.avail_out = (uInt)(cap - 18 - 8);
                if (deflate(&z, Z_FINISH) != Z_STREAM_END) { jb->failed = 1; break; }
                clen = (cap - 18 - 8) - z.avail_out; total = 18 + clen + 8;
                deflateEnd(&z);
                if (deflateInit2(&z, jb->level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { jb->failed = 1; break; }
            }
            static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            memcpy(b->out, head, 16);
            put16(b->out + 16, (uint32_t)(total - 1));
            put32(b->out + 18 + clen, (uint32_t)crc32(crc32(0L, Z_NULL, 0), jb->in + b->in_off, (uInt)b->in_len));
            put32(b->out + 18 + clen + 4, (uint32_t)b->in_len);
            b->out_len = total;
        }
    }
    deflateEnd(&z);
    return NULL;
}

/* The stream `data` as BGZF blocks to `fp`; `cut[k]` are offsets no block may straddle unless a single piece is larger than
 * a block (htslib flushes before a record that would not fit: bam_write1 -> bgzf_flush_try), ascending, last = len. */
static int bgzf_write_stream(FILE *fp, const uint8_t *data, const int64_t *cut, int64_t n_cut, int level)
{
    wblock *blk = NULL; size_t n_blk = 0, cap = 0;
    size_t start = 0;
    int64_t k = 0;
    const size_t len = n_cut ? (size_t)cut[n_cut - 1] : 0;
    while (start < len) {
        /* the farthest cut that keeps the block within its payload; a piece larger than a block is split */
        size_t endb = start;
        while (k < n_cut && (size_t)cut[k] - start <= BGZF_PAYLOAD) { endb = (size_t)cut[k]; ++k; }
        if (endb == start) { endb = start + BGZF_PAYLOAD; if (endb > len) endb = len; if (k < n_cut && endb >= (size_t)cut[k]) { endb = (size_t)cut[k]; ++k; } }
        if (n_blk == cap) { cap = cap ? cap * 2 : 1024; blk = (wblock *)h_realloc(blk, cap * sizeof *blk); }
        blk[n_blk].in_off = start; blk[n_blk].in_len = endb - start; blk[n_blk].out = NULL; blk[n_blk].out_len = 0;
        ++n_blk; start = endb;
    }
    wjob jb; memset(&jb, 0, sizeof jb);
    jb.in = data; jb.blk = blk; jb.n_blk = n_blk; jb.level = level; pthread_mutex_init(&jb.mu, NULL);
    const char *e = getenv("L2R_THREADS");
    long n_thr = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    if (n_thr > 32) n_thr = 32;
    if (n_thr < 1 || n_blk < 8) n_thr = 1;
    pthread_t th[32];
    for (long t = 1; t < n_thr; ++t) if (pthread_create(&th[t], NULL, deflate_worker, &jb)) h_fatal("bam_filter", "pthread_create failed");
    deflate_worker(&jb);
    for (long t = 1; t < n_thr; ++t) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&jb.mu);
    int ok = !jb.failed;
    for (size_t b = 0; b < n_blk; ++b) {
        if (ok && fwrite(blk[b].out, 1, blk[b].out_len, fp) != blk[b].out_len) ok = 0;
        free(blk[b].out);
    }
    free(blk);
    return ok ? 0 : -1;
}

int h_write_bam(FILE *fp, const h_records *r, const int64_t *keep, int64_t n_keep)
{
    const char *lv = getenv("L2R_BAM_LEVEL");
    const int level = lv ? atoi(lv) : Z_DEFAULT_COMPRESSION;          /* sam_open_format("-", "wb"): zlib's default level */
    /* header: blocks of its own (bam_hdr_write ends with bgzf_flush) */
    {   int64_t cut = (int64_t)r->hdr_len;
        if (bgzf_write_stream(fp, r->hdr, &cut, 1, level)) return -1; }
    /* records, gathered */
    size_t total = 0;
    for (int64_t k = 0; k < n_keep; ++k) total += (size_t)(r->rec_off[keep[k] + 1] - r->rec_off[keep[k]]);
    uint8_t *data = (uint8_t *)h_malloc(total + 1);
    int64_t *cut = (int64_t *)h_malloc((size_t)(n_keep + 1) * 8);
    size_t at = 0;
    for (int64_t k = 0; k < n_keep; ++k) {
        const size_t len = (size_t)(r->rec_off[keep[k] + 1] - r->rec_off[keep[k]]);
        memcpy(data + at, r->buf + r->rec_off[keep[k]], len);
        at += len; cut[k] = (int64_t)at;
    }
    int rc = n_keep ? bgzf_write_stream(fp, data, cut, n_keep, level) : 0;
    free(data); free(cut);
    /* the end-of-file marker block */
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!rc && fwrite(eof, 1, sizeof eof, fp) != sizeof eof) rc = -1;
    if (fflush(fp) != 0) rc = -1;
    return rc;
}

/* every record of `in_fn` (SAM text, gzip / BGZF SAM, BAM) as a BAM file: the reader + encoder + writer of `filter` without
 * its tests (no GPU involved; the tests compare the stream with an independent encoder) */
int h_records_to_bam(const char *in_fn, const char *out_fn)
{
    h_chroms chr; memset(&chr, 0, sizeof chr);
    h_records r;
    h_read_records(in_fn, &chr, &r, "bam_filter");
    FILE *f = fopen(out_fn, "wb");
    if (!f) h_fatal("bam_filter", "Can not open \"%s\" for writing\n", out_fn);
    int64_t *all = (int64_t *)h_malloc((size_t)(r.n + 1) * 8);
    for (int64_t i = 0; i < r.n; ++i) all[i] = i;
    int rc = h_write_bam(f, &r, all, r.n);
    if (fclose(f) != 0) rc = -1;
    free(all); h_records_free(&r); h_chroms_free(&chr);
    return rc;
}

/* ------------------------------------------------------------------ the sub-command */

int h_filter_run(const char *in_fn, const char *remove_fn, const l2r_filter_params *prm, FILE *out, int64_t *n_written)
{
    h_chroms chr; memset(&chr, 0, sizeof chr);
    h_records r;
    h_stage_time("start");
    h_read_records(in_fn, &chr, &r, "bam_filter");
    h_stage_time("read + encode records");
    h_gtf g; memset(&g, 0, sizeof g);
    l2r_filter_spans spans = {0, NULL, NULL, NULL};
    if (remove_fn && remove_fn[0]) {
        fprintf(stderr, "[read_anno_trans] reading transcript annotation from %s ...\n", remove_fn);
        h_read_gtf(remove_fn, &chr, &g, 0);
        fprintf(stderr, "[read_anno_trans] reading transcript annotation from %s done.\n", remove_fn);
        spans.n = g.n_tx; spans.tid = g.tid; spans.start = g.start; spans.end = g.end;
    }
    /* the reference dereferences the NM tag without a test (bam_aux2i(bam_aux_get(b, "NM")), src/bam_filter.c:78-80):
     * a record without it that gets there is the end of that run; here it is an error message */
    /* ... but only for a record that REACHES that read: gtf_filter() returns on an unmapped record and on one that fails the
     * coverage test (src/bam_filter.c:63,77) before it looks for NM, so such a record is dropped without one */
    for (int64_t i = 0; i < r.n; ++i) if (!(r.flag[i] & 4) && !r.nm_seen[i]) {
        const int64_t c0 = r.cig_off[i], c1 = r.cig_off[i + 1];
        int32_t qlen = r.l_qseq[i];
        if (c1 > c0) {
            const uint32_t w0 = r.cig[c0], w1 = r.cig[c1 - 1];
            if ((w0 & 0xfu) == 4u || (w0 & 0xfu) == 5u) qlen -= (int32_t)(w0 >> 4);
            if (c1 - c0 > 1 && ((w1 & 0xfu) == 4u || (w1 & 0xfu) == 5u)) qlen -= (int32_t)(w1 >> 4);
        }
        if (!(((double)qlen + 0.0) / (double)r.l_qseq[i] < (double)prm->cov_rate))
            h_fatal("bam_filter", "alignment record %lld passes the coverage test and has no NM tag (the reference reads it there without a test)", (long long)i);
    }
    h_stage_time("read -r annotation");
    l2r_ctx *ctx = l2r_create(0);
    if (!ctx) h_fatal("bam_filter", "%s", l2r_last_error());
    h_stage_time("engine: create");
    uint8_t *drop = (uint8_t *)h_malloc((size_t)r.n + 1);
    int32_t *score = (int32_t *)h_malloc((size_t)(r.n + 1) * 4), *intron = (int32_t *)h_malloc((size_t)(r.n + 1) * 4);
    l2r_filter_records fr = { r.n, r.n_cig, r.flag, r.tid, r.pos, r.l_qseq, r.nm, r.cig_off, r.cig };
    if (l2r_filter_score(ctx, &fr, prm, spans.n ? &spans : NULL, drop, score, intron)) h_fatal("bam_filter", "%s", l2r_last_error());
    h_stage_time("engine: score (upload, kernel, download)");
    /* kept records, and the runs of one read name among them */
    int64_t n_kept = 0;
    for (int64_t i = 0; i < r.n; ++i) n_kept += !drop[i];
    int64_t *kept = (int64_t *)h_malloc((size_t)(n_kept + 1) * 8), *goff = (int64_t *)h_malloc((size_t)(n_kept + 2) * 8);
    int32_t *k_score = (int32_t *)h_malloc((size_t)(n_kept + 1) * 4), *k_intron = (int32_t *)h_malloc((size_t)(n_kept + 1) * 4);
    int64_t m = 0, n_groups = 0;
    const char *last = NULL;
    for (int64_t i = 0; i < r.n; ++i) if (!drop[i]) {
        const char *name = (const char *)(r.buf + r.rec_off[i] + 4 + 32);
        if (!last || strcmp(name, last) != 0) goff[n_groups++] = m;
        kept[m] = i; k_score[m] = score[i]; k_intron[m] = intron[i]; ++m;
        last = name;
    }
    goff[n_groups] = m;
    h_stage_time("groups of one read name");
    int64_t *winner = (int64_t *)h_malloc((size_t)(n_groups + 1) * 8);
    if (l2r_filter_select(ctx, n_groups, goff, k_score, k_intron, prm, winner)) h_fatal("bam_filter", "%s", l2r_last_error());
    h_stage_time("engine: select (upload, kernel, download)");
    l2r_destroy(ctx);
    int64_t n_out = 0;
    for (int64_t gi = 0; gi < n_groups; ++gi) {
        /* (a read name that is the empty string is never written: strcmp(lqname, "\0") != 0, :141,:149) */
        const char *name = (const char *)(r.buf + r.rec_off[kept[goff[gi]]] + 4 + 32);
        if (winner[gi] >= 0 && name[0]) kept[n_out++] = kept[winner[gi]];      /* (in place: winner[gi] >= goff[gi] >= n_out) */
    }
    int rc = h_write_bam(out, &r, kept, n_out);
    if (rc) h_fatal("bam_filter", "Error in writing SAM record\n");
    h_stage_time("write BAM (BGZF)");
    if (n_written) *n_written = n_out;
    free(drop); free(score); free(intron); free(kept); free(goff); free(k_score); free(k_intron); free(winner);
    h_gtf_free(&g); h_records_free(&r); h_chroms_free(&chr);
    return 0;
}

int h_cmd_filter(int argc, char **argv)
{
    static const struct option long_opt[] = {
        { "coverage", 1, NULL, 'v' }, { "map-quality", 1, NULL, 'q' }, { "sec-rat", 1, NULL, 's' }, { "intron", 1, NULL, 'i' },
        { "remove-gtf", 1, NULL, 'r' }, { 0, 0, 0, 0 }
    };
    l2r_filter_params prm = { (float)COV_RATIO, (float)MAP_QUAL, (float)SEC_RATIO, MIN_INTRON_NUM };
    char remove_fn[1024] = "";
    int c;
    optind = 1;
    while ((c = getopt_long(argc, argv, "v:q:s:i:r:", long_opt, NULL)) >= 0) {
        switch (c) {
        case 'v': prm.cov_rate = (float)atof(optarg); break;
        case 'q': prm.map_qual = (float)atof(optarg); break;
        case 's': prm.sec_rat = (float)atof(optarg); break;
        case 'i': prm.min_intron_n = atoi(optarg); break;
        case 'r': snprintf(remove_fn, sizeof remove_fn, "%s", optarg); break;
        default: return filter_usage();
        }
    }
    if (argc - optind != 1) return filter_usage();
    int64_t cnt = 0;
    const int rc = h_filter_run(argv[optind], remove_fn, &prm, stdout, &cnt);
    fprintf(stderr, "[%s] Filtered alignments: %d\n", "bam_filter", (int)cnt);
    return rc;
}
