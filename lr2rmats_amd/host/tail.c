/* tail.c -- the order-dependent end of update-gtf on the host.
 *
 * Consumes the per-read arrays the engine returns (exons, flag bytes, class bits,
 * reference transcript) and does, in read order, what the reference does after
 * check_with_anno_trans()/check_with_short_sj():
 *   routing                src/update_gtf.c:943-964
 *   split_trans            src/update_gtf.c:837-913
 *   merge_trans(1,2)       src/update_gtf.c:98-163,  check_iden src/gtf.c:54-92
 *   print_read_trans       src/gtf.c:607-632
 *   print_bam_detail_trans src/update_gtf.c:297-419
 *   print_trans_summary    src/update_gtf.c:421-587 (+ add_simp_* :176-295)
 * Reads are never copied: list members point back into the result arrays; only the
 * merged list owns (mutable) exon copies.
 */
#include <stdlib.h>
#include <string.h>
#include "l2r_host.h"

/* ------------------------------------------------------------------ buffered text output */

typedef struct { FILE *fp; char *b; size_t n, cap; } obuf;

static void ob_init(obuf *o, FILE *fp) { o->fp = fp; o->cap = 1 << 20; o->b = (char *)h_malloc(o->cap + 4096); o->n = 0; }
static void ob_flush(obuf *o) { if (o->n) fwrite(o->b, 1, o->n, o->fp); o->n = 0; }
static void ob_done(obuf *o) { ob_flush(o); free(o->b); o->b = NULL; }
static inline void ob_room(obuf *o, size_t k) { if (o->n + k > o->cap) { ob_flush(o); if (k > o->cap) { o->cap = k; o->b = (char *)h_realloc(o->b, o->cap + 4096); } } }
static inline void ob_c(obuf *o, char c) { ob_room(o, 1); o->b[o->n++] = c; }
static inline void ob_s(obuf *o, const char *s) { size_t k = strlen(s); ob_room(o, k); memcpy(o->b + o->n, s, k); o->n += k; }
static inline void ob_i(obuf *o, long long v)
{
    char t[24]; int k = 0; unsigned long long u = v < 0 ? (unsigned long long)(-v) : (unsigned long long)v;
    ob_room(o, 24);
    do { t[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) o->b[o->n++] = '-';
    while (k) o->b[o->n++] = t[--k];
}

/* ------------------------------------------------------------------ merged / unique lists */

typedef struct {
    int32_t tid, start, end, cov, n;
    uint8_t rev, ex_rev, partial;
    int32_t ex_tid, gene_tx, piece;     /* piece < 0: a whole read */
    int64_t read, ex;                   /* ex = offset of the exon copy in the arena */
} m_ent;

typedef struct {
    m_ent *e; int64_t n, cap;
    int32_t *xs, *xe; uint8_t *xf; int64_t nx, capx;
} m_list;

typedef struct {                        /* a transcript offered to merge_trans */
    int32_t tid, start, end, n;
    uint8_t rev;
    const int32_t *xs, *xe; const uint8_t *xf;
} m_cand;

static float frac_of(int s1, int e1, int s2, int e2)
{
    /* src/update_gtf.c:80-89 exon_overlap_frac (Q6: double divide, returned as float) */
    if (s1 > e2 || s2 > e1) return 0.0;
    int hi = e1 < e2 ? e1 : e2, lo = s1 > s2 ? s1 : s2;
    int l1 = e1 - s1 + 1, l2 = e2 - s2 + 1, mn = l1 < l2 ? l1 : l2;
    return ((hi - lo + 1) / (mn + 0.0));
}

static int chains_match(const int32_t *as, const int32_t *ae, int an, const int32_t *bs, const int32_t *be, int bn, int ss, int ed)
{
    /* src/gtf.c:54-92 check_iden(t1 = a, t2 = b): 0 identical, 2 contained, -1 otherwise (never 1: Q8) */
    if (an == bn) {
        if (abs(as[0] - bs[0]) > ed) return -1;
        for (int i = 0; i + 1 < an; ++i) {
            if (abs(ae[i] - be[i]) > ss) return -1;
            if (abs(as[i + 1] - bs[i + 1]) > ss) return -1;
        }
        if (abs(ae[an - 1] - be[bn - 1]) > ed) return -1;
        return 0;
    }
    const int32_t *ls, *le, *ss_, *se; int ln, sn;
    if (an > bn) { ls = as; le = ae; ln = an; ss_ = bs; se = be; sn = bn; }
    else { ls = bs; le = be; ln = bn; ss_ = as; se = ae; sn = an; }
    int verdict = -1;
    if (abs(ls[0] - ss_[0]) > ed) return -1;
    for (int i = 0; i + 1 < ln; ++i) {
        if (abs(le[i] - se[0]) <= ss && abs(ls[i + 1] - ss_[1]) <= ss) {
            verdict = 2;
            int j = 1;
            for (i = i + 1; i + 1 < ln && j + 1 < sn; ++i, ++j) {
                if (abs(le[i] - se[j]) > ss) return -1;
                if (abs(ls[i + 1] - ss_[j + 1]) > ss) return -1;
            }
            break;
        }
    }
    if (abs(le[ln - 1] - se[sn - 1]) > ed) return -1;
    return verdict;
}

static int m_merge(const m_cand *t, m_list *U, const l2r_params *p)
{
    /* src/update_gtf.c:144-163 merge_trans */
    for (int64_t i = U->n - 1; i >= 0; --i) {
        m_ent *T = &U->e[i];
        if (t->tid > T->tid || t->start > T->end) return 0;
        if (p->force_strand && t->rev != T->rev) continue;
        int32_t *Ts = U->xs + T->ex, *Te = U->xe + T->ex;
        if (t->n == 1 && T->n == 1) {
            /* :122-140 merge_trans2 */
            if (abs(t->xs[0] - Ts[0]) > p->end_dis) continue;
            if (abs(t->xe[0] - Te[0]) > p->end_dis) continue;
            if (frac_of(t->xs[0], t->xe[0], Ts[0], Te[0]) >= p->single_exon_ovlp_frac) {
                T->cov++;
                if (t->xs[0] < Ts[0]) { Ts[0] = t->xs[0]; T->start = t->xs[0]; }
                if (t->xe[0] > Te[0]) { Te[0] = t->xe[0]; T->end = t->xe[0]; }
                return 1;
            }
        } else if (t->n > 1 && T->n > 1) {
            /* :98-119 merge_trans1 */
            int v = chains_match(t->xs, t->xe, t->n, Ts, Te, T->n, p->ss_dis, p->end_dis);
            if (v == 0) {
                T->cov++;
                if (t->xs[0] < Ts[0]) { Ts[0] = t->xs[0]; T->start = t->xs[0]; }
                if (t->xe[t->n - 1] > Te[T->n - 1]) { Te[T->n - 1] = t->xe[t->n - 1]; T->end = t->xe[t->n - 1]; }
                return 1;
            }
            if (v == 2) return 1;
        }
    }
    return 0;
}

static m_ent *m_push(m_list *U, const m_cand *t)
{
    if (U->n == U->cap) { U->cap = U->cap ? U->cap * 2 : 1024; U->e = (m_ent *)h_realloc(U->e, (size_t)U->cap * sizeof(m_ent)); }
    if (U->nx + t->n > U->capx) {
        while (U->nx + t->n > U->capx) U->capx = U->capx ? U->capx * 2 : 8192;
        U->xs = (int32_t *)h_realloc(U->xs, (size_t)U->capx * 4); U->xe = (int32_t *)h_realloc(U->xe, (size_t)U->capx * 4);
        U->xf = (uint8_t *)h_realloc(U->xf, (size_t)U->capx);
    }
    m_ent *e = &U->e[U->n++];
    memset(e, 0, sizeof *e);
    e->tid = t->tid; e->start = t->start; e->end = t->end; e->n = t->n; e->rev = t->rev; e->cov = 1; e->ex = U->nx; e->piece = -1;
    memcpy(U->xs + U->nx, t->xs, (size_t)t->n * 4); memcpy(U->xe + U->nx, t->xe, (size_t)t->n * 4);
    if (t->xf) memcpy(U->xf + U->nx, t->xf, (size_t)t->n); else memset(U->xf + U->nx, 0, (size_t)t->n);
    U->nx += t->n;
    return e;
}

static void m_free(m_list *U) { free(U->e); free(U->xs); free(U->xe); free(U->xf); memset(U, 0, sizeof *U); }

/* ------------------------------------------------------------------ members of the printable lists */

typedef struct { int64_t read; int32_t first, last, piece; } l_ref;      /* piece < 0: whole read */
typedef struct { l_ref *v; int64_t n, cap; } l_list;

static void l_push(l_list *L, int64_t read, int first, int last, int piece)
{
    if (L->n == L->cap) { L->cap = L->cap ? L->cap * 2 : 4096; L->v = (l_ref *)h_realloc(L->v, (size_t)L->cap * sizeof(l_ref)); }
    l_ref *x = &L->v[L->n++]; x->read = read; x->first = first; x->last = last; x->piece = piece;
}

typedef struct {
    const h_update_opts *o; const h_chroms *chr; const h_reads *reads; const h_gtf *anno; const h_result *res;
} tail_ctx;

static const char *gene_id_of(const tail_ctx *c, int32_t ref) { return ref >= 0 ? h_str(&c->anno->names, c->anno->gid[ref]) : "NA"; }
static const char *gene_name_of(const tail_ctx *c, int32_t ref) { return ref >= 0 ? h_str(&c->anno->names, c->anno->gname[ref]) : "NA"; }

/* one GTF block (transcript line + exon lines), src/gtf.c:612-628 */
static void emit_gtf_named(obuf *o, const char *src, const char *tchr, int start, int end, int rev, int cov,
                           const char *gid, const char *tid_s, const char *gname, const char *tname,
                           const char *xchr, int ex_rev, const int32_t *xs, const int32_t *xe, int n)
{
    char attr[1024]; size_t k = 0;
    attr[0] = 0;
    if (gid[0])   k += (size_t)snprintf(attr + k, sizeof attr - k, " gene_id \"%s\";", gid);
    if (tid_s[0]) k += (size_t)snprintf(attr + k, sizeof attr - k, " transcript_id \"%s\";", tid_s);
    if (gname[0]) k += (size_t)snprintf(attr + k, sizeof attr - k, " gene_name \"%s\";", gname);
    if (tname[0]) k += (size_t)snprintf(attr + k, sizeof attr - k, " transcript_name \"%s\";", tname);
    const char *a = attr[0] ? attr + 1 : attr;
    ob_s(o, tchr); ob_c(o, '\t'); ob_s(o, src); ob_s(o, "\ttranscript\t"); ob_i(o, start); ob_c(o, '\t'); ob_i(o, end);
    ob_s(o, "\t.\t"); ob_c(o, "+-"[rev]); ob_s(o, "\t.\t"); ob_s(o, a); ob_s(o, " transcript_cov \""); ob_i(o, cov); ob_s(o, "\";\n");
    for (int q = 0; q < n; ++q) {
        const int j = rev ? n - 1 - q : q;                      /* descending for '-' (:622-628) */
        ob_s(o, xchr); ob_c(o, '\t'); ob_s(o, src); ob_s(o, "\texon\t"); ob_i(o, xs[j]); ob_c(o, '\t'); ob_i(o, xe[j]);
        ob_s(o, "\t.\t"); ob_c(o, "+-"[ex_rev]); ob_s(o, "\t.\t"); ob_s(o, a); ob_c(o, '\n');
    }
}

/* trans_name and trans_id of read i: both the QNAME for alignments, two GTF attributes for `-m g` input */
static const char *read_name(const tail_ctx *c, int64_t i) { return h_str(&c->reads->names, c->reads->qname[i]); }
static const char *read_tid_name(const tail_ctx *c, int64_t i)
{
    return h_str(&c->reads->names, c->reads->tid_name ? c->reads->tid_name[i] : c->reads->qname[i]);
}

static void emit_gtf(obuf *o, const tail_ctx *c, const char *tchr, int start, int end, int rev, int cov,
                     const char *gid, const char *gname, int64_t read, int piece,
                     const char *xchr, int ex_rev, const int32_t *xs, const int32_t *xe, int n)
{
    /* src/update_gtf.c:871-872,904-905: a split piece is "<trans_id>.split.<k>" / "<trans_name>.split.<k>" */
    const char *qname = read_name(c, read), *tidn = read_tid_name(c, read);
    char nm[H_NAME_MAX + 32], ni[H_NAME_MAX + 32];
    if (piece >= 0) { snprintf(nm, sizeof nm, "%s.split.%d", qname, piece); snprintf(ni, sizeof ni, "%s.split.%d", tidn, piece); }
    else { strncpy(nm, qname, sizeof nm - 1); nm[sizeof nm - 1] = 0; strncpy(ni, tidn, sizeof ni - 1); ni[sizeof ni - 1] = 0; }
    if (piece >= 0 && (strlen(nm) >= H_NAME_MAX || strlen(ni) >= H_NAME_MAX)) h_fatal("split_trans", "piece name \"%s\" has 100 or more characters", nm);
    emit_gtf_named(o, c->o->source, tchr, start, end, rev, cov, gid, ni, gname, nm, xchr, ex_rev, xs, xe, n);
}

static void print_ref_list(FILE *fp, const tail_ctx *c, const l_list *L)
{
    obuf o; ob_init(&o, fp);
    const h_result *r = c->res;
    for (int64_t i = 0; i < L->n; ++i) {
        const l_ref *x = &L->v[i];
        const int64_t off = r->ex_off[x->read];
        const uint32_t info = r->info[x->read];
        const int rev = (info & L2R_INFO_REV) != 0;
        const int32_t ref = r->ref_tx[x->read];
        const int64_t q = x->read;
        const char *xc = c->chr->name[c->reads->tid[x->read]];
        if (x->piece < 0) {
            const int n = (int)L2R_INFO_NEXON(info);
            emit_gtf(&o, c, xc, r->ex_start[off], r->ex_end[off + n - 1], rev, 1, gene_id_of(c, ref), gene_name_of(c, ref), q, -1,
                     xc, rev, r->ex_start + off, r->ex_end + off, n);
        } else {
            /* Q2: a split piece keeps tid = start = end = 0 and strand '+' on its transcript line */
            emit_gtf(&o, c, c->chr->name[0], 0, 0, 0, 1, gene_id_of(c, ref), gene_name_of(c, ref), q, x->piece,
                     xc, rev, r->ex_start + off + x->first, r->ex_end + off + x->first, x->last - x->first + 1);
        }
    }
    ob_done(&o);
}

static void print_merged(FILE *fp, const tail_ctx *c, const m_list *U)
{
    obuf o; ob_init(&o, fp);
    for (int64_t i = 0; i < U->n; ++i) {
        const m_ent *e = &U->e[i];
        const int64_t q = e->read;
        emit_gtf(&o, c, c->chr->name[e->tid], e->start, e->end, e->rev, e->cov, gene_id_of(c, e->gene_tx), gene_name_of(c, e->gene_tx),
                 q, e->piece, c->chr->name[e->ex_tid], e->ex_rev, U->xs + e->ex, U->xe + e->ex, e->n);
    }
    ob_done(&o);
}

static void print_all_reads(FILE *fp, const tail_ctx *c)
{
    /* `-a`: bam_T after classification (strand possibly flipped, gene names set) */
    obuf o; ob_init(&o, fp);
    const h_result *r = c->res;
    for (int64_t i = 0; i < r->n; ++i) {
        const int64_t off = r->ex_off[i];
        const uint32_t info = r->info[i];
        const int n = (int)L2R_INFO_NEXON(info), rev = (info & L2R_INFO_REV) != 0;
        const char *xc = c->chr->name[c->reads->tid[i]];
        emit_gtf(&o, c, xc, r->ex_start[off], r->ex_end[off + n - 1], rev, 1, gene_id_of(c, r->ref_tx[i]), gene_name_of(c, r->ref_tx[i]),
                 i, -1, xc, rev, r->ex_start + off, r->ex_end + off, n);
    }
    ob_done(&o);
}

static void index_list(obuf *o, const uint8_t *f, int n, uint8_t bit, int two_per_exon, int trailing_tab)
{
    /* counts + comma separated 0-based indices, "NA" when empty (src/update_gtf.c:338-414) */
    int cnt = 0, first = 1;
    if (!two_per_exon) { for (int j = 0; j < n; ++j) cnt += (f[j] & bit) != 0; }
    else { for (int j = 0; j < n; ++j) cnt += ((f[j] & L2R_EXF_NOVEL_DON) != 0) + ((f[j] & L2R_EXF_NOVEL_ACC) != 0); }
    ob_i(o, cnt); ob_c(o, '\t');
    if (cnt == 0) { ob_s(o, "NA\t"); return; }
    for (int j = 0; j < n; ++j) {
        if (!two_per_exon) { if (f[j] & bit) { if (!first) ob_c(o, ','); first = 0; ob_i(o, j); } }
        else {
            if (f[j] & L2R_EXF_NOVEL_DON) { if (!first) ob_c(o, ','); first = 0; ob_i(o, 2 * j); }
            if (f[j] & L2R_EXF_NOVEL_ACC) { if (!first) ob_c(o, ','); first = 0; ob_i(o, 2 * j + 1); }
        }
    }
    if (trailing_tab) ob_c(o, '\t');
}

static void print_detail(FILE *fp, const tail_ctx *c)
{
    /* src/update_gtf.c:297-419 print_bam_detail_trans */
    obuf o; ob_init(&o, fp);
    const h_result *r = c->res;
    if (!c->o->no_detail_header) {
        ob_s(&o, "ReadName\tchr\tstrand\tNovel\tGeneID\tGeneName\tExonCount\tExonStart\tExonEnd\tNovelExonCount\tNovelExonIndex\tNovelSiteCount\tNovelSiteIndex\tNovelJunctionCount\tNovelJunctionIndex\tUnreliableJunctionCount\tUnreliableJunctionIndex\n");
    }
    for (int64_t i = 0; i < r->n; ++i) {
        const int64_t off = r->ex_off[i];
        const uint32_t info = r->info[i];
        const int n = (int)L2R_INFO_NEXON(info);
        const int cls = (info & L2R_INFO_KNOWN) ? 0 : ((info & L2R_INFO_KNOWN_SITE) ? 1 : 2);
        const uint8_t *f = r->ex_flag + off;
        ob_s(&o, h_str(&c->reads->names, c->reads->qname[i])); ob_c(&o, '\t');
        ob_s(&o, c->chr->name[c->reads->tid[i]]); ob_c(&o, '\t');
        ob_c(&o, "+-"[(info & L2R_INFO_REV) != 0]); ob_c(&o, '\t');
        ob_i(&o, cls); ob_c(&o, '\t');
        ob_s(&o, gene_id_of(c, r->ref_tx[i])); ob_c(&o, '\t'); ob_s(&o, gene_name_of(c, r->ref_tx[i])); ob_c(&o, '\t');
        ob_i(&o, n); ob_c(&o, '\t');
        for (int j = 0; j < n; ++j) { if (j) ob_c(&o, ','); ob_i(&o, r->ex_start[off + j]); }
        ob_c(&o, '\t');
        for (int j = 0; j < n; ++j) { if (j) ob_c(&o, ','); ob_i(&o, r->ex_end[off + j]); }
        ob_c(&o, '\t');
        index_list(&o, f, n, L2R_EXF_NOVEL_EXON, 0, 1);
        index_list(&o, f, n - 1, 0, 1, 1);
        index_list(&o, f, n - 1, L2R_EXF_NOVEL_JUNC, 0, 1);
        index_list(&o, f, n - 1, L2R_EXF_UNREL_JUNC, 0, 0);
        ob_c(&o, '\n');
    }
    ob_done(&o);
}

/* ------------------------------------------------------------------ summary + novel_exon.bed */

typedef struct { int32_t tid; const char *gid; } s_gene;
typedef struct { int32_t tid, start, end, score; uint8_t type, rev; } s_exon;
typedef struct { int32_t tid, site; } s_site;
typedef struct { int32_t tid, don, acc; } s_junc;
#define S_GROW(a, n, cap, T) do { if ((n) == (cap)) { (cap) = (cap) ? (cap) * 2 : 64; (a) = (T *)h_realloc((a), (size_t)(cap) * sizeof(T)); } } while (0)

static void add_gene(s_gene **G, int *n, int *cap, int32_t tid, const char *gid)
{
    /* src/update_gtf.c:181-203: backward scan, equal gene_id first, then stop at a smaller tid */
    for (int k = *n - 1; k >= 0; --k) { if (strcmp(gid, (*G)[k].gid) == 0) return; if (tid > (*G)[k].tid) break; }
    S_GROW(*G, *n, *cap, s_gene); (*G)[*n].tid = tid; (*G)[*n].gid = gid; ++*n;
}

void h_part_genes_free(h_part_genes *g)
{
    for (int q = 0; q < 2; ++q) {
        free(g->last_gid[q]);
        for (int k = 0; k < g->n_first[q]; ++k) free(g->first_gids[q][k]);
        free(g->first_gids[q]);
    }
    memset(g, 0, sizeof *g);
}

int h_part_genes_has_first(const h_part_genes *g, int list, const char *gid)
{
    if (!gid || list < 0 || list > 1) return 0;
    for (int k = 0; k < g->n_first[list]; ++k) if (strcmp(g->first_gids[list][k], gid) == 0) return 1;
    return 0;
}

static void part_genes_take(h_part_genes *g, int list, const s_gene *G, int n)
{
    g->last_gid[list] = NULL; g->first_gids[list] = NULL; g->n_first[list] = 0;
    if (n <= 0) return;
    g->last_gid[list] = strdup(G[n - 1].gid);
    int m = 0;
    while (m < n && G[m].tid == G[0].tid) ++m;
    g->first_gids[list] = (char **)h_malloc((size_t)m * sizeof(char *));
    for (int k = 0; k < m; ++k) g->first_gids[list][k] = strdup(G[k].gid);
    g->n_first[list] = m;
}

void h_write_summary_text(FILE *s, int anno_genes, int anno_tx, const int64_t *k)
{
    /* src/update_gtf.c:530-569; k = the counters in the order summary_and_bed fills them */
    fprintf(s, "==== Annotaion ====\n");
    fprintf(s, "Genes_of_annotation_GTF\t%d\n", anno_genes);
    fprintf(s, "Transcripts_of_annotation_GTF\t%d\n", anno_tx);
    fprintf(s, "\n===================\n\n==== Updated information ====\n");
    fprintf(s, "Updated_Genes\t%d\n", (int)k[0]);
    fprintf(s, "Added_Novel_Transcripts\t%d\n", (int)k[1]);
    fprintf(s, "Added_Novel_Full-read_Transcripts\t%d\n", (int)(k[1] - k[2]));
    fprintf(s, "Added_Novel_Partial-read_Transcripts\t%d\n", (int)k[2]);
    fprintf(s, "Added_Novel_Exons\t%d\n", (int)k[3]);
    fprintf(s, "Added_Novel_Sites\t%d\n", (int)k[4]);
    fprintf(s, "Added_Novel_Splice_Junctions\t%d\n", (int)k[5]);
    fprintf(s, "\n=============================\n\n==== Known information ====\n");
    fprintf(s, "Known_Transcripts_from_BAM\t%d\n", (int)k[6]);
    fprintf(s, "Genes_of_Known_Transcripts_from_BAM\t%d\n", (int)k[7]);
    fprintf(s, "Uniq_Known_Transcripts_from_BAM\t%d\n", (int)k[8]);
    fprintf(s, "\n===========================\n\n==== Novel information ====\n");
    fprintf(s, "Novel_Transcript_from_BAM\t%d\n", (int)(k[9] + k[10]));
    fprintf(s, "Novel_Transcript_from_BAM_with_All_Reliable_Junction\t%d\n", (int)k[9]);
    fprintf(s, "Uniq_Novel_Transcript_from_BAM_with_All_Reliable_Junction\t%d\n", (int)k[11]);
    fprintf(s, "Novel_Transcript_from_BAM_with_Unreliable_Junction\t%d\n", (int)k[10]);
    fprintf(s, "Uniq_Novel_Transcript_from_BAM_with_Unreliable_Junction\t%d\n", (int)k[12]);
    fprintf(s, "\n===========================\n\n==== Unrecognized information ====\n");
    fprintf(s, "Unrecognized_Transcript_from_BAM\t%d\n", (int)k[13]);
    fprintf(s, "Uniq_Unrecognized_Transcript_from_BAM\t%d\n", (int)k[14]);
    fprintf(s, "\n==================================\n");
}

static void summary_and_bed(const tail_ctx *c, const m_list *U)
{
    /* src/update_gtf.c:421-587 print_trans_summary */
    const h_result *r = c->res; const l2r_params *p = &c->o->prm;
    s_gene *G = NULL; int g_cap = 0, upd_genes = 0, known_genes = 0;
    s_exon *E = NULL; int e_n = 0, e_cap = 0;
    s_site *D = NULL, *A = NULL; int d_n = 0, d_cap = 0, a_n = 0, a_cap = 0;
    s_junc *J = NULL; int j_n = 0, j_cap = 0;
    int partial = 0;
    for (int64_t i = 0; i < U->n; ++i) {
        const m_ent *t = &U->e[i];
        const int32_t *xs = U->xs + t->ex, *xe = U->xe + t->ex; const uint8_t *xf = U->xf + t->ex;
        add_gene(&G, &upd_genes, &g_cap, t->tid, gene_id_of(c, t->gene_tx));
        partial += t->partial;
        for (int j = 0; j < t->n; ++j) if (xf[j] & L2R_EXF_NOVEL_EXON) {
            const uint8_t type = t->n > 1 ? ((j == 0 || j == t->n - 1) ? 0 : 1) : 2;
            int hit = 0;
            for (int k = e_n - 1; k >= 0; --k) {                    /* :211-222 merge_exon */
                if (E[k].tid == t->ex_tid && E[k].start == xs[j] && E[k].end == xe[j]) { E[k].score += t->cov; hit = 1; break; }
                if (t->ex_tid > E[k].tid) break;
            }
            if (!hit) { S_GROW(E, e_n, e_cap, s_exon); E[e_n].tid = t->ex_tid; E[e_n].start = xs[j]; E[e_n].end = xe[j]; E[e_n].score = t->cov; E[e_n].type = type; E[e_n].rev = t->ex_rev; ++e_n; }
        }
        for (int j = 0; j + 1 < t->n; ++j) if (xf[j] & L2R_EXF_NOVEL_DON) {
            int hit = 0;
            for (int k = d_n - 1; k >= 0; --k) { if (D[k].tid == t->tid && D[k].site == xe[j]) { hit = 1; break; } if (t->tid > D[k].tid) break; }
            if (!hit) { S_GROW(D, d_n, d_cap, s_site); D[d_n].tid = t->tid; D[d_n].site = xe[j]; ++d_n; }
        }
        for (int j = 0; j + 1 < t->n; ++j) if (xf[j] & L2R_EXF_NOVEL_ACC) {
            int hit = 0;
            for (int k = a_n - 1; k >= 0; --k) { if (A[k].tid == t->tid && A[k].site == xs[j + 1]) { hit = 1; break; } if (t->tid > A[k].tid) break; }
            if (!hit) { S_GROW(A, a_n, a_cap, s_site); A[a_n].tid = t->tid; A[a_n].site = xs[j + 1]; ++a_n; }
        }
        for (int j = 0; j + 1 < t->n; ++j) if (xf[j] & L2R_EXF_NOVEL_JUNC) {
            int hit = 0;
            for (int k = j_n - 1; k >= 0; --k) { if (J[k].tid == t->tid && J[k].don == xe[j] && J[k].acc == xs[j + 1]) { hit = 1; break; } if (t->tid > J[k].tid) break; }
            if (!hit) { S_GROW(J, j_n, j_cap, s_junc); J[j_n].tid = t->tid; J[j_n].don = xe[j]; J[j_n].acc = xs[j + 1]; ++j_n; }
        }
    }
    if (c->o->part_genes) { memset(c->o->part_genes, 0, sizeof *c->o->part_genes); part_genes_take(c->o->part_genes, 0, G, upd_genes); }
    if (c->o->summary || c->o->summary_counts) {
        /* :496-528: classes of every input read + unique counts through merge_trans on fresh lists */
        int n_known = 0, n_rel = 0, n_unrel = 0, n_unrec = 0;
        m_list uk, ur, uu, un; memset(&uk, 0, sizeof uk); memset(&ur, 0, sizeof ur); memset(&uu, 0, sizeof uu); memset(&un, 0, sizeof un);
        for (int64_t i = 0; i < r->n; ++i) {
            const uint32_t info = r->info[i]; const int64_t off = r->ex_off[i]; const int n = (int)L2R_INFO_NEXON(info);
            m_cand t = { c->reads->tid[i], r->ex_start[off], r->ex_end[off + n - 1], n, (uint8_t)((info & L2R_INFO_REV) != 0), r->ex_start + off, r->ex_end + off, NULL };
            m_list *dst;
            if (info & L2R_INFO_KNOWN) { ++n_known; add_gene(&G, &known_genes, &g_cap, t.tid, gene_id_of(c, r->ref_tx[i])); dst = &uk; }
            else if (info & L2R_INFO_KNOWN_SITE) { if (info & L2R_INFO_UNREL) { ++n_unrel; dst = &uu; } else { ++n_rel; dst = &ur; } }
            else { ++n_unrec; dst = &un; }
            if (!m_merge(&t, dst, p)) m_push(dst, &t);
        }
        if (c->o->part_genes) part_genes_take(c->o->part_genes, 1, G, known_genes);
        int64_t cnt[H_N_SUMMARY] = { upd_genes, U->n, partial, e_n, (int64_t)d_n + a_n, j_n, n_known, known_genes, uk.n,
                                     n_rel, n_unrel, ur.n, uu.n, n_unrec, un.n, 0 };
        if (c->o->summary_counts) memcpy(c->o->summary_counts, cnt, sizeof cnt);
        if (c->o->summary) h_write_summary_text(c->o->summary, c->anno->gene_n, (int)c->anno->n_tx, cnt);
        m_free(&uk); m_free(&ur); m_free(&uu); m_free(&un);
    }
    if (c->o->exon_bed) {                                          /* :571-576 (BAM header names) */
        obuf o; ob_init(&o, c->o->exon_bed);
        for (int i = 0; i < e_n; ++i) {
            ob_s(&o, c->chr->name[E[i].tid]); ob_c(&o, '\t'); ob_i(&o, E[i].start - 1); ob_c(&o, '\t'); ob_i(&o, E[i].end); ob_c(&o, '\t');
            ob_c(&o, "TIS"[E[i].type]); ob_s(&o, "_exon\t"); ob_i(&o, E[i].score); ob_c(&o, '\t'); ob_c(&o, "+-"[E[i].rev]); ob_c(&o, '\n');
        }
        ob_done(&o);
    }
    free(G); free(E); free(D); free(A); free(J);
}

/* ------------------------------------------------------------------ driver */

void h_update_tail(const h_update_opts *o, const h_chroms *chr, const h_reads *reads, const h_gtf *anno,
                   const h_result *res, int64_t n_sj)
{
    tail_ctx c = {o, chr, reads, anno, res};
    const l2r_params *p = &o->prm;
    m_list U; memset(&U, 0, sizeof U);
    l_list K, N, X; memset(&K, 0, sizeof K); memset(&N, 0, sizeof N); memset(&X, 0, sizeof X);
    const int want_k = o->known_gtf != NULL, want_n = o->novel_gtf != NULL, want_x = o->unrecog_gtf != NULL;

    for (int64_t i = 0; i < res->n; ++i) {                         /* src/update_gtf.c:939-964 */
        const uint32_t info = res->info[i];
        if (!(info & L2R_INFO_FULL)) continue;                     /* Q3 */
        if (info & L2R_INFO_KNOWN) { if (want_k) l_push(&K, i, 0, 0, -1); continue; }
        if (!(info & L2R_INFO_KNOWN_SITE)) { if (want_x) l_push(&X, i, 0, 0, -1); continue; }
        const int64_t off = res->ex_off[i];
        const int n = (int)L2R_INFO_NEXON(info);
        const uint8_t rev = (info & L2R_INFO_REV) != 0;
        const int32_t *xs = res->ex_start + off, *xe = res->ex_end + off; const uint8_t *xf = res->ex_flag + off;
        if (n_sj == 0 || (info & L2R_INFO_SJ_PASS)) {
            if (want_n) l_push(&N, i, 0, n - 1, -1);
            m_cand t = { reads->tid[i], xs[0], xe[n - 1], n, rev, xs, xe, xf };
            if (!m_merge(&t, &U, p)) {
                m_ent *e = m_push(&U, &t);
                e->read = i; e->ex_rev = rev; e->ex_tid = reads->tid[i]; e->gene_tx = res->ref_tx[i];
            }
        } else if (p->split_trans) {
            /* src/update_gtf.c:837-913 split_trans: cut at unreliable junctions; keep pieces with >= 2 exons
             * that saw both a novel and a known junction */
            int first = 0, seen_novel = 0, seen_known = 0, k = 0, j;
            for (j = 0; j < n; ++j) {
                const int at_end = (j == n - 1);
                if (!at_end) { if (xf[j] & L2R_EXF_NOVEL_JUNC) seen_novel = 1; else seen_known = 1; }
                if (at_end || (xf[j] & L2R_EXF_UNREL_JUNC)) {
                    if (seen_novel && seen_known && j - first >= 1) {
                        if (want_n) l_push(&N, i, first, j, k);
                        uint8_t pf[4096], *fl = pf; const int pn = j - first + 1;
                        if (pn > 4096) fl = (uint8_t *)h_malloc((size_t)pn);
                        /* flags of the piece: exon flags copied; site/junction flags of junctions first..j-1 copied,
                         * the piece has no junction after its last exon, unreliable flags start at 0 */
                        for (int q = 0; q < pn; ++q) {
                            uint8_t f = xf[first + q] & L2R_EXF_NOVEL_EXON;
                            if (q + 1 < pn) f |= xf[first + q] & (L2R_EXF_NOVEL_DON | L2R_EXF_NOVEL_ACC | L2R_EXF_NOVEL_JUNC);
                            fl[q] = f;
                        }
                        m_cand t = { 0, 0, 0, pn, 0, xs + first, xe + first, fl };          /* Q2: tid/start/end/strand stay 0 */
                        if (!m_merge(&t, &U, p)) {
                            m_ent *e = m_push(&U, &t);
                            e->read = i; e->piece = k; e->partial = 1; e->ex_rev = rev; e->ex_tid = reads->tid[i]; e->gene_tx = res->ref_tx[i];
                        }
                        if (fl != pf) free(fl);
                        ++k;
                    }
                    first = j + 1; seen_novel = seen_known = 0;
                }
            }
        }
    }

    print_merged(o->out_gtf, &c, &U);                              /* :1087 */
    if (o->bam_gtf) print_all_reads(o->bam_gtf, &c);
    if (o->bam_detail) print_detail(o->bam_detail, &c);
    if (want_k) print_ref_list(o->known_gtf, &c, &K);
    if (want_n) print_ref_list(o->novel_gtf, &c, &N);
    if (want_x) print_ref_list(o->unrecog_gtf, &c, &X);
    if (o->summary || o->summary_counts || o->exon_bed) summary_and_bed(&c, &U);
    m_free(&U); free(K.v); free(N.v); free(X.v);
}

/* ------------------------------------------------------------------ unique-gtf */

/* src/unique_gtf.c:73-84 uniq_trans + :147-148: every input transcript is offered to merge_trans on the
 * growing unique list; the ones that merged form the "shared" list (-I). */
void h_unique_tail(const l2r_params *p, const char *source, FILE *out, int intersect, const h_chroms *chr,
                   int64_t n, const int32_t *tid, const uint8_t *rev, const int64_t *ex_off, const int32_t *xs, const int32_t *xe,
                   const h_strtab *names, const uint32_t *gid, const uint32_t *tids, const uint32_t *gname, const uint32_t *tname)
{
    m_list U; memset(&U, 0, sizeof U);
    obuf o; ob_init(&o, out);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t off = ex_off[i]; const int k = (int)(ex_off[i + 1] - off);
        m_cand t = { tid[i], xs[off], xe[off + k - 1], k, rev[i], xs + off, xe + off, NULL };
        if (!m_merge(&t, &U, p)) { m_ent *e = m_push(&U, &t); e->read = i; }
        else if (intersect) {
            if (tid[i] < 0) h_fatal("unique_gtf", "transcript on a chromosome that is not in the BAM header");
            emit_gtf_named(&o, source, chr->name[tid[i]], t.start, t.end, rev[i], 1, h_str(names, gid[i]), h_str(names, tids[i]),
                           h_str(names, gname[i]), h_str(names, tname[i]), chr->name[tid[i]], rev[i], t.xs, t.xe, k);
        }
    }
    if (!intersect) for (int64_t i = 0; i < U.n; ++i) {
        const m_ent *e = &U.e[i]; const int64_t r = e->read;
        if (e->tid < 0) h_fatal("unique_gtf", "transcript on a chromosome that is not in the BAM header");
        emit_gtf_named(&o, source, chr->name[e->tid], e->start, e->end, e->rev, e->cov, h_str(names, gid[r]), h_str(names, tids[r]),
                       h_str(names, gname[r]), h_str(names, tname[r]), chr->name[e->tid], e->rev, U.xs + e->ex, U.xe + e->ex, e->n);
    }
    ob_done(&o);
    m_free(&U);
}
