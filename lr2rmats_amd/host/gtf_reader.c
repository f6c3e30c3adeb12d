/* gtf_reader.c -- GTF -> structure-of-arrays transcripts, and STAR SJ.out.tab.
 *
 * Behaviour of the reference readers that decides output bytes is kept:
 *   read_anno_trans  src/gtf.c:468-521   (annotation; as_reads = 0)
 *   read_gtf_trans   src/gtf.c:524-595   (`-m g` input;  as_reads = 1)
 *   gtf_add_info     src/gtf.c:317-326   (first-substring tag lookup, value at tag+2)
 *   read_sj_group    src/gtf.c:431-449
 * i.e. 1023-byte line pieces (fgets(line,1024): Q10), whitespace-separated
 * sscanf fields whose previous values survive a short line, only "exon" rows,
 * transcripts = runs of equal transcript_id (Q12).
 *
 * The file is parsed from memory.  A well-formed line piece takes a hand-written
 * scanner; anything else (a piece the fast scanner cannot prove equivalent) goes
 * through sscanf with the reference's own format string on the same persistent
 * state, so both routes give what the reference's parser gives.
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include "l2r_host.h"

#define PIECE 1023   /* fgets(line, 1024, fp) */

typedef struct {            /* sscanf targets that persist across lines (src/gtf.c:472) */
    char ref[1024], type[1024], attrs[1024];
    int start, end; char strand;
    char gid[1024], gname[1024], tid_s[1024], tname[1024];
} scan_state;

typedef struct { int32_t start, end, tid; uint8_t rev; } gx;

static int is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

/* the hand scanner: returns 1 and fills the state when the piece [p,e) (no '\n' inside, e excludes it)
 * is nine or more TAB separated columns, columns 1-8 free of other white space and non-empty,
 * columns 4 and 5 plain decimal integers, column 9 non-empty after leading white space. */
static int fast_fields(const char *p, const char *e, scan_state *st)
{
    const char *col[9]; int n = 0; const char *q = p;
    if (memchr(p, 0, (size_t)(e - p))) return 0;
    col[n++] = q;
    while (n < 9) {
        const char *t = memchr(q, '\t', (size_t)(e - q));
        if (!t) return 0;
        q = t + 1; col[n++] = q;
    }
    for (int i = 0; i < 8; ++i) {
        const char *a = col[i], *b = col[i + 1] - 1;
        if (a == b) return 0;
        for (const char *c = a; c < b; ++c) if (is_ws(*c)) return 0;
    }
    int v[2];
    for (int k = 0; k < 2; ++k) {
        const char *a = col[3 + k], *b = col[4 + k] - 1;
        if (b - a > 9) return 0;
        int x = 0;
        for (const char *c = a; c < b; ++c) { if (*c < '0' || *c > '9') return 0; x = x * 10 + (*c - '0'); }
        v[k] = x;
    }
    const char *a9 = col[8];
    while (a9 < e && is_ws(*a9)) ++a9;
    if (a9 == e) return 0;
    size_t l1 = (size_t)(col[1] - 1 - col[0]), l3 = (size_t)(col[3] - 1 - col[2]), l9 = (size_t)(e - a9);
    memcpy(st->ref, col[0], l1); st->ref[l1] = 0;
    memcpy(st->type, col[2], l3); st->type[l3] = 0;
    st->start = v[0]; st->end = v[1]; st->strand = *col[6];
    memcpy(st->attrs, a9, l9); st->attrs[l9] = 0;
    return 1;
}

static void tag_value(const char *attrs, const char *tag, char *out)
{
    /* src/gtf.c:317-326: first occurrence of the tag text anywhere in the attribute string; the value is
     * whatever follows two bytes later up to the next double quote (sscanf "%[^\"]": at least one byte) */
    const char *h = strstr(attrs, tag);
    if (!h) return;
    size_t k = strlen(tag), step = 0;
    h += k;
    while (step < 2 && *h) { ++h; ++step; }
    if (step < 2) return;
    size_t n = strcspn(h, "\"");
    if (n == 0) return;
    memcpy(out, h, n); out[n] = 0;
}

typedef struct { h_gtf *g; gx *ex; int n, cap; uint32_t gid, gname, tids, tname; int open; } builder;

static int gx_cmp(const void *pa, const void *pb)
{
    /* src/gtf.c:37-45 trans_exon_comp */
    const gx *a = (const gx *)pa, *b = (const gx *)pb;
    if (a->rev != b->rev) h_fatal("trans_exon_comp", "Strands of exons do NOT match.\n");
    if (a->start != b->start) return a->start - b->start;
    return a->end - b->end;
}

static void flush_tx(builder *b)
{
    h_gtf *g = b->g;
    if (b->n == 0) return;
    qsort(b->ex, (size_t)b->n, sizeof(gx), gx_cmp);          /* src/gtf.c:94-100 set_trans_name */
    if (g->n_tx + 1 >= g->cap_tx) {
        g->cap_tx = g->cap_tx ? g->cap_tx * 2 : 1 << 12;
        g->tid = (int32_t *)h_realloc(g->tid, (size_t)g->cap_tx * 4); g->start = (int32_t *)h_realloc(g->start, (size_t)g->cap_tx * 4);
        g->end = (int32_t *)h_realloc(g->end, (size_t)g->cap_tx * 4); g->rev = (uint8_t *)h_realloc(g->rev, (size_t)g->cap_tx);
        g->ex_off = (int64_t *)h_realloc(g->ex_off, (size_t)(g->cap_tx + 1) * 8);
        g->gid = (uint32_t *)h_realloc(g->gid, (size_t)g->cap_tx * 4); g->gname = (uint32_t *)h_realloc(g->gname, (size_t)g->cap_tx * 4);
        g->tids = (uint32_t *)h_realloc(g->tids, (size_t)g->cap_tx * 4); g->tname = (uint32_t *)h_realloc(g->tname, (size_t)g->cap_tx * 4);
    }
    if (g->n_ex + b->n > g->cap_ex) {
        while (g->n_ex + b->n > g->cap_ex) g->cap_ex = g->cap_ex ? g->cap_ex * 2 : 1 << 14;
        g->ex_start = (int32_t *)h_realloc(g->ex_start, (size_t)g->cap_ex * 4); g->ex_end = (int32_t *)h_realloc(g->ex_end, (size_t)g->cap_ex * 4);
    }
    const int64_t t = g->n_tx;
    g->tid[t] = b->ex[0].tid; g->rev[t] = b->ex[0].rev;
    g->start[t] = b->ex[0].start; g->end[t] = b->ex[b->n - 1].end;      /* end of the LAST exon in (start,end) order */
    g->gid[t] = b->gid; g->gname[t] = b->gname; g->tids[t] = b->tids; g->tname[t] = b->tname;
    g->ex_off[t] = g->n_ex;
    for (int k = 0; k < b->n; ++k) { g->ex_start[g->n_ex] = b->ex[k].start; g->ex_end[g->n_ex] = b->ex[k].end; g->n_ex++; }
    g->n_tx++;
    g->ex_off[g->n_tx] = g->n_ex;
    b->n = 0;
}

static uint32_t add_name(h_gtf *g, const char *s, const char *who)
{
    if (strlen(s) >= H_NAME_MAX) h_fatal(who, "name \"%s\" has 100 or more characters; the reference stores names in char[100]", s);
    return h_str_add(&g->names, s);
}

static void parse_gtf(const char *fn, const h_chroms *chr, h_gtf *g, int as_reads)
{
    const char *who = as_reads ? "read_gtf_trans" : "read_anno_trans";
    memset(g, 0, sizeof *g);
    int fd = open(fn, O_RDONLY);
    if (fd < 0) h_fatal_core(who, "fail to open file '%s'", fn);        /* err_xopen_core */
    struct stat sb;
    if (fstat(fd, &sb) != 0) h_fatal_core(who, "fail to stat '%s'", fn);
    size_t len = (size_t)sb.st_size;
    char *buf = NULL;
    if (len) {
        buf = (char *)mmap(NULL, len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (buf == MAP_FAILED) h_fatal_core(who, "fail to map '%s'", fn);
    }
    close(fd);
    g->ex_off = (int64_t *)h_malloc(8 * 2); g->ex_off[0] = 0;

    scan_state *st = (scan_state *)calloc(1, sizeof *st);
    char last_tid[1024] = "", last_gid[1024] = "", piece[1024];
    builder b; memset(&b, 0, sizeof b); b.g = g;
    const char *p = buf, *fe = buf + len;
    while (p < fe) {
        /* one fgets(line, 1024) piece: through '\n' or PIECE bytes, whichever comes first */
        size_t room = (size_t)(fe - p) < PIECE ? (size_t)(fe - p) : PIECE;
        const char *nl = memchr(p, '\n', room);
        const char *pe = nl ? nl + 1 : p + room;          /* piece = [p, pe) */
        const char *ce = nl ? nl : pe;                     /* content without the newline */
        const int hash_first = (*p == '#');
        int fast = 0;
        if (!(hash_first && !as_reads)) {
            fast = fast_fields(p, ce, st);
            if (!fast) {
                size_t n = (size_t)(pe - p);
                memcpy(piece, p, n); piece[n] = 0;
                sscanf(piece, "%s\t%*s\t%s\t%d\t%d\t%*s\t%c\t%*s\t%[^\n]", st->ref, st->type, &st->start, &st->end, &st->strand, st->attrs);
            }
        }
        p = pe;
        if (hash_first) continue;                          /* src/gtf.c:477 (and :535 after the first sscanf) */
        if (strcmp(st->type, "exon") != 0) continue;
        const uint8_t rev = (st->strand == '-');
        const int tid = h_chrom_find(chr, st->ref, chr->n_hdr);      /* bam_name2id: header names only, else -1 */
        memset(st->gid, 0, strlen(st->gid));     tag_value(st->attrs, "gene_id", st->gid);
        memset(st->gname, 0, strlen(st->gname)); tag_value(st->attrs, "gene_name", st->gname);
        if (!st->gid[0] && !st->gname[0]) h_fatal_core(who, "GTF format error in %s. (No gene id or gene name found.\n", fn);
        if (!st->gid[0]) strcpy(st->gid, st->gname); else if (!st->gname[0]) strcpy(st->gname, st->gid);
        memset(st->tid_s, 0, strlen(st->tid_s)); tag_value(st->attrs, "transcript_id", st->tid_s);
        memset(st->tname, 0, strlen(st->tname)); tag_value(st->attrs, "transcript_name", st->tname);
        if (!st->tid_s[0] && !st->tname[0]) h_fatal_core(who, "GTF format error in %s. (No transcript id or transcript name found.\n", fn);
        if (!st->tid_s[0]) strcpy(st->tid_s, st->tname); else if (!st->tname[0]) strcpy(st->tname, st->tid_s);

        const char *gene_key = as_reads ? st->gname : st->gid;        /* :495 vs :553 */
        g->gene_n += strcmp(gene_key, last_gid) != 0;
        if (strcmp(st->tid_s, last_tid) != 0) {
            flush_tx(&b);
            b.tname = add_name(g, st->tname, who); b.tids = add_name(g, st->tid_s, who);
            b.gname = add_name(g, st->gname, who); b.gid = add_name(g, st->gid, who);
            strcpy(last_tid, st->tid_s); strcpy(last_gid, gene_key);
        }
        if (b.n == b.cap) { b.cap = b.cap ? b.cap * 2 : 16; b.ex = (gx *)h_realloc(b.ex, (size_t)b.cap * sizeof(gx)); }
        b.ex[b.n].start = st->start; b.ex[b.n].end = st->end; b.ex[b.n].tid = tid; b.ex[b.n].rev = rev; b.n++;
    }
    flush_tx(&b);
    free(b.ex); free(st);
    if (buf) munmap(buf, len);
}

/* ---- the parsed GTF on disk (L2R_ANNO_CACHE=<directory>): the transcript arrays and the name table of one GTF as read
 * against one BAM header, keyed by the file's path, size and modification time, the header's chromosome names and the
 * reading mode.  The reference parses the GTF anew on every invocation (src/gtf.c:468-521), the pipeline runs update-gtf
 * twice per sample on the same file.  A file that does not match in any respect is ignored and rewritten. */
static uint64_t fnv64(uint64_t h, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ULL; }
    return h;
}

/* (8 bytes a step: the payload checksum runs over ~30 MB for a GENCODE-size GTF) */
static uint64_t mix64(uint64_t h, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 0x9E3779B97F4A7C15ULL; h ^= h >> 29; }
    uint64_t w = 0;
    if (i < n) { memcpy(&w, b + i, n - i); h = (h ^ w) * 0x9E3779B97F4A7C15ULL; h ^= h >> 29; }
    return (h ^ n) * 0xD6E8FEB86659FD93ULL;
}

typedef struct { char magic[8]; uint64_t key; int64_t n_tx, n_ex, names_len; int32_t gene_n, pad; uint64_t sum; } gtf_cache_head;

static uint64_t gtf_payload_sum(const h_gtf *g)
{
    const size_t T = (size_t)g->n_tx, X = (size_t)g->n_ex;
    uint64_t h = 0x51ed270b;
    h = mix64(h, g->tid, T * 4); h = mix64(h, g->start, T * 4); h = mix64(h, g->end, T * 4); h = mix64(h, g->rev, T);
    h = mix64(h, g->ex_off, (T + 1) * 8); h = mix64(h, g->ex_start, X * 4); h = mix64(h, g->ex_end, X * 4);
    h = mix64(h, g->gid, T * 4); h = mix64(h, g->gname, T * 4); h = mix64(h, g->tids, T * 4); h = mix64(h, g->tname, T * 4);
    return mix64(h, g->names.buf, g->names.len);
}

static char *gtf_cache_path(const char *fn, const h_chroms *chr, int as_reads, uint64_t *key_out)
{
    const char *dir = getenv("L2R_ANNO_CACHE");
    if (!dir || !*dir) return NULL;
    struct stat sb;
    if (stat(fn, &sb) != 0) return NULL;
    char *real = realpath(fn, NULL);
    uint64_t h = 0xcbf29ce484222325ULL;
    h = fnv64(h, real ? real : fn, strlen(real ? real : fn));
    free(real);
    const int64_t meta[4] = {(int64_t)sb.st_size, (int64_t)sb.st_mtim.tv_sec, (int64_t)sb.st_mtim.tv_nsec, as_reads};
    h = fnv64(h, meta, sizeof meta);
    for (int i = 0; i < chr->n_hdr; ++i) h = fnv64(h, chr->name[i], strlen(chr->name[i]) + 1);
    *key_out = h;
    char *path = (char *)h_malloc(strlen(dir) + 64);
    sprintf(path, "%s/l2r_gtf_%016llx.parsed", dir, (unsigned long long)h);
    return path;
}

static int gtf_cache_load(const char *path, uint64_t key, h_gtf *g)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    gtf_cache_head hd;
    int ok = fread(&hd, sizeof hd, 1, f) == 1 && memcmp(hd.magic, "L2RGTF01", 8) == 0 && hd.key == key && hd.n_tx >= 0 && hd.n_ex >= 0 &&
             hd.names_len >= 0 && hd.n_tx < (1LL << 31) && hd.n_ex < (1LL << 31) && hd.names_len < (1LL << 40);
    if (ok) {
        const int64_t want = (int64_t)sizeof hd + hd.n_tx * (4 * 3 + 1 + 4 * 4) + (hd.n_tx + 1) * 8 + hd.n_ex * 8 + hd.names_len;
        fseek(f, 0, SEEK_END);
        ok = ftell(f) == want;
        fseek(f, (long)sizeof hd, SEEK_SET);
    }
    if (!ok) { fclose(f); return 0; }
    memset(g, 0, sizeof *g);
    const size_t T = (size_t)hd.n_tx, X = (size_t)hd.n_ex;
    g->n_tx = g->cap_tx = hd.n_tx; g->n_ex = g->cap_ex = hd.n_ex; g->gene_n = hd.gene_n;
    g->tid = (int32_t *)h_malloc(T * 4 + 4); g->start = (int32_t *)h_malloc(T * 4 + 4); g->end = (int32_t *)h_malloc(T * 4 + 4);
    g->rev = (uint8_t *)h_malloc(T + 1); g->ex_off = (int64_t *)h_malloc((T + 1) * 8);
    g->ex_start = (int32_t *)h_malloc(X * 4 + 4); g->ex_end = (int32_t *)h_malloc(X * 4 + 4);
    g->gid = (uint32_t *)h_malloc(T * 4 + 4); g->gname = (uint32_t *)h_malloc(T * 4 + 4); g->tids = (uint32_t *)h_malloc(T * 4 + 4); g->tname = (uint32_t *)h_malloc(T * 4 + 4);
    g->names.buf = (char *)h_malloc((size_t)hd.names_len + 1); g->names.len = g->names.cap = (size_t)hd.names_len;
#define RD(p, sz, n) (ok = ok && ((n) == 0 || fread((p), (sz), (n), f) == (n)))
    RD(g->tid, 4, T); RD(g->start, 4, T); RD(g->end, 4, T); RD(g->rev, 1, T); RD(g->ex_off, 8, T + 1);
    RD(g->ex_start, 4, X); RD(g->ex_end, 4, X); RD(g->gid, 4, T); RD(g->gname, 4, T); RD(g->tids, 4, T); RD(g->tname, 4, T);
    RD(g->names.buf, 1, (size_t)hd.names_len);
#undef RD
    fclose(f);
    /* the payload is what was written; offsets and name ids stay inside their arrays */
    ok = ok && gtf_payload_sum(g) == hd.sum;
    ok = ok && g->ex_off[0] == 0 && g->ex_off[T] == (int64_t)X && (hd.names_len == 0 || g->names.buf[hd.names_len - 1] == 0);
    for (size_t i = 0; ok && i < T; ++i)
        ok = g->ex_off[i] <= g->ex_off[i + 1] && g->gid[i] < (uint64_t)hd.names_len && g->gname[i] < (uint64_t)hd.names_len &&
             g->tids[i] < (uint64_t)hd.names_len && g->tname[i] < (uint64_t)hd.names_len;
    if (!ok) { h_gtf_free(g); return 0; }
    return 1;
}

static void gtf_cache_store(const char *path, uint64_t key, const h_gtf *g)
{
    const char *dir = getenv("L2R_ANNO_CACHE");
    (void)mkdir(dir, 0777);
    char *tmp = (char *)h_malloc(strlen(path) + 32);
    sprintf(tmp, "%s.tmp.%ld", path, (long)getpid());
    FILE *f = fopen(tmp, "wb");
    if (!f) { free(tmp); return; }                         /* a cache that cannot be written is no error */
    gtf_cache_head hd; memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, "L2RGTF01", 8); hd.key = key; hd.n_tx = g->n_tx; hd.n_ex = g->n_ex; hd.names_len = (int64_t)g->names.len; hd.gene_n = g->gene_n;
    hd.sum = gtf_payload_sum(g);
    const size_t T = (size_t)g->n_tx, X = (size_t)g->n_ex;
    int ok = fwrite(&hd, sizeof hd, 1, f) == 1;
#define WR(p, sz, n) (ok = ok && ((n) == 0 || fwrite((p), (sz), (n), f) == (n)))
    WR(g->tid, 4, T); WR(g->start, 4, T); WR(g->end, 4, T); WR(g->rev, 1, T); WR(g->ex_off, 8, T + 1);
    WR(g->ex_start, 4, X); WR(g->ex_end, 4, X); WR(g->gid, 4, T); WR(g->gname, 4, T); WR(g->tids, 4, T); WR(g->tname, 4, T);
    WR(g->names.buf, 1, g->names.len);
#undef WR
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp, path) != 0) remove(tmp);
    free(tmp);
}

void h_read_gtf(const char *fn, const h_chroms *chr, h_gtf *g, int as_reads)
{
    uint64_t key = 0;
    char *path = gtf_cache_path(fn, chr, as_reads, &key);
    if (path && gtf_cache_load(path, key, g)) { free(path); return; }
    parse_gtf(fn, chr, g, as_reads);
    if (path) { gtf_cache_store(path, key, g); free(path); }
}

void h_gtf_free(h_gtf *g)
{
    free(g->tid); free(g->start); free(g->end); free(g->rev); free(g->ex_off); free(g->ex_start); free(g->ex_end);
    free(g->gid); free(g->gname); free(g->tids); free(g->tname); free(g->names.buf);
    memset(g, 0, sizeof *g);
}

/* ------------------------------------------------------------------ SJ.out.tab */

typedef struct { int32_t tid, don, acc, uniq, multi; int64_t seq; } sjrow;

static int sj_cmp(const void *pa, const void *pb)
{
    /* src/gtf.c:414-420 sj_group_comp; ties keep file order (glibc qsort = merge sort) */
    const sjrow *a = (const sjrow *)pa, *b = (const sjrow *)pb;
    if (a->tid != b->tid) return a->tid - b->tid;
    if (a->don != b->don) return a->don - b->don;
    if (a->acc != b->acc) return a->acc - b->acc;
    return a->seq < b->seq ? -1 : (a->seq > b->seq);
}

void h_read_sj(FILE *fp, h_chroms *chr, h_sj *out)
{
    memset(out, 0, sizeof *out);
    if (!fp) return;
    char line[1024], ref[1024] = "";
    int strand = 0, motif = 0, anno = 0, over = 0;
    int64_t n = 0, cap = 0; sjrow *rows = NULL;
    while (fgets(line, 1024, fp)) {
        if (n == cap) { cap = cap ? cap * 2 : 10000; rows = (sjrow *)h_realloc(rows, (size_t)cap * sizeof(sjrow)); }
        sjrow *r = &rows[n]; memset(r, 0, sizeof *r);
        sscanf(line, "%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d", ref, &r->don, &r->acc, &strand, &motif, &anno, &r->uniq, &r->multi, &over);
        r->tid = h_chrom_intern(chr, ref);
        r->seq = n++;
    }
    qsort(rows, (size_t)n, sizeof(sjrow), sj_cmp);
    out->n = n;
    out->tid = (int32_t *)h_malloc((size_t)n * 4); out->don = (int32_t *)h_malloc((size_t)n * 4); out->acc = (int32_t *)h_malloc((size_t)n * 4);
    out->uniq = (int32_t *)h_malloc((size_t)n * 4); out->multi = (int32_t *)h_malloc((size_t)n * 4);
    for (int64_t i = 0; i < n; ++i) { out->tid[i] = rows[i].tid; out->don[i] = rows[i].don; out->acc[i] = rows[i].acc; out->uniq[i] = rows[i].uniq; out->multi[i] = rows[i].multi; }
    free(rows);
}

void h_sj_free(h_sj *s)
{
    free(s->tid); free(s->don); free(s->acc); free(s->uniq); free(s->multi); memset(s, 0, sizeof *s);
}
