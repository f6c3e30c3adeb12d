/* oracle_core.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Sequential, array-of-structs restatement of the reference's comparison
 * algorithm.  Every routine names the reference lines it follows
 * (paths relative to /root/reference/src).  Quirk numbers (Q1..Q14) refer to
 * SURVEY.md Appendix A.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_int.h"

/* ------------------------------------------------------------------ utils */

void orc_die(int core, const char *where, const char *msg)
{
    /* utils.c:91-111 : err_fatal -> exit(1); err_fatal_core -> abort() */
    fprintf(stderr, "[%s] %s%s\n", where, msg, core ? " Abort!" : "");
    if (core) abort();
    exit(EXIT_FAILURE);
}

static void *xmalloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) orc_die(1, "oracle", "Malloc fail!");
    return p;
}

static void *xrealloc(void *q, size_t n)
{
    void *p = realloc(q, n ? n : 1);
    if (!p) orc_die(1, "oracle", "Realloc fail!");
    return p;
}

void orc_set_name(char dst[ORC_NAME_MAX], const char *src, const char *what)
{
    /* gtf.h:44-45: fixed char[100] + strcpy (Q14).  The oracle refuses names the
     * reference would overflow on instead of truncating them. */
    size_t n = strlen(src);
    if (n >= ORC_NAME_MAX) orc_die(0, what, "name of 100 or more characters (reference buffer is char[100])");
    memcpy(dst, src, n + 1);
}

/* ------------------------------------------------------------ containers */

void tr_zero(orc_trans *t)
{
    /* gtf.c:18-24 trans_init: calloc, cov = 1 */
    memset(t, 0, sizeof *t);
    t->cov = 1;
}

void tr_release(orc_trans *t)
{
    /* gtf.c:219-226 */
    free(t->ex); free(t->nov_exon); free(t->nov_site); free(t->nov_junc); free(t->unrel);
    t->ex = NULL; t->nov_exon = t->nov_site = t->nov_junc = t->unrel = NULL;
    t->n = t->cap = 0;
}

void tr_push_exon(orc_trans *t, int tid, int start, int end, uint8_t rev)
{
    /* gtf.c:26-35 add_exon, gtf.c:121-130 exon_realloc (2, then doubling) */
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 2;
        t->ex = (orc_exon *)xrealloc(t->ex, (size_t)t->cap * sizeof(orc_exon));
    }
    orc_exon *e = &t->ex[t->n++];
    e->tid = tid; e->start = start; e->end = end; e->rev = rev; e->score = 0; e->etype = 0;
}

static int exon_order(const void *pa, const void *pb)
{
    /* gtf.c:37-45 trans_exon_comp: strand mismatch is fatal (exit 1) */
    const orc_exon *a = (const orc_exon *)pa, *b = (const orc_exon *)pb;
    if (a->rev != b->rev) orc_die(0, "trans_exon_comp", "Strands of exons do NOT match.\n");
    if (a->start != b->start) return a->start - b->start;
    return a->end - b->end;
}

void tr_finish(orc_trans *t)
{
    /* gtf.c:94-100 set_trans_name head: sort exons, take tid/strand/start from the
     * first exon and end from the LAST exon in (start,end) order. */
    qsort(t->ex, (size_t)t->n, sizeof(orc_exon), exon_order);
    t->tid = t->ex[0].tid;
    t->rev = t->ex[0].rev;
    t->start = t->ex[0].start;
    t->end = t->ex[t->n - 1].end;
}

void tr_alloc_read_flags(orc_trans *t)
{
    /* bam2gtf.c:97-102 (and gtf.c:557-562): per-read state.  With n == 0 the
     * reference asks malloc for (0-1)*2 bytes and aborts (Q9). */
    if (t->n < 1) orc_die(1, "read_bam_trans", "Malloc fail!\nSize: -2\n");
    t->full = t->lfull = t->rfull = 0; t->lnoth = t->rnoth = 1;
    t->known = t->has_known_site = t->has_unrel = t->partial = 0;
    t->nov_exon = (uint8_t *)xmalloc((size_t)t->n);           memset(t->nov_exon, 1, (size_t)t->n);
    t->nov_site = (uint8_t *)xmalloc((size_t)(t->n - 1) * 2); memset(t->nov_site, 1, (size_t)(t->n - 1) * 2);
    t->nov_junc = (uint8_t *)xmalloc((size_t)(t->n - 1));     memset(t->nov_junc, 1, (size_t)(t->n - 1));
    t->unrel    = (uint8_t *)xmalloc((size_t)(t->n - 1));     memset(t->unrel, 0, (size_t)(t->n - 1));
}

orc_list *ls_new(void)
{
    /* gtf.c:134-140 read_trans_init(1) */
    orc_list *l = (orc_list *)xmalloc(sizeof *l);
    l->n = 0; l->cap = 1; l->gene_n = 0;
    l->t = (orc_trans *)xmalloc(sizeof(orc_trans));
    tr_zero(&l->t[0]);
    return l;
}

static orc_trans *ls_slot(orc_list *l)
{
    /* gtf.c:206-217 read_trans_realloc: doubling */
    if (l->n == l->cap) {
        l->cap *= 2;
        l->t = (orc_trans *)xrealloc(l->t, (size_t)l->cap * sizeof(orc_trans));
    }
    orc_trans *d = &l->t[l->n++];
    memset(d, 0, sizeof *d);
    return d;
}

static void copy_core(orc_trans *d, const orc_trans *s)
{
    int i;
    d->cov = s->cov;
    for (i = 0; i < s->n; ++i) tr_push_exon(d, s->ex[i].tid, s->ex[i].start, s->ex[i].end, s->ex[i].rev);
    d->tid = s->tid; d->rev = s->rev; d->start = s->start; d->end = s->end;
    memcpy(d->gid, s->gid, ORC_NAME_MAX); memcpy(d->gname, s->gname, ORC_NAME_MAX);
    memcpy(d->tids, s->tids, ORC_NAME_MAX); memcpy(d->tname, s->tname, ORC_NAME_MAX);
}

void ls_push_read(orc_list *l, const orc_trans *s)
{
    /* gtf.c:142-164 add_read_trans: deep copy incl. state bits and the four flag
     * arrays; exon score/type are NOT copied. */
    orc_trans *d = ls_slot(l);
    copy_core(d, s);
    d->full = s->full; d->lfull = s->lfull; d->lnoth = s->lnoth; d->rfull = s->rfull; d->rnoth = s->rnoth;
    d->known = s->known; d->has_known_site = s->has_known_site; d->has_unrel = s->has_unrel; d->partial = s->partial;
    d->nov_exon = (uint8_t *)xmalloc((size_t)d->n);           memcpy(d->nov_exon, s->nov_exon, (size_t)d->n);
    d->nov_site = (uint8_t *)xmalloc((size_t)(d->n - 1) * 2); memcpy(d->nov_site, s->nov_site, (size_t)(d->n - 1) * 2);
    d->nov_junc = (uint8_t *)xmalloc((size_t)(d->n - 1));     memcpy(d->nov_junc, s->nov_junc, (size_t)(d->n - 1));
    d->unrel    = (uint8_t *)xmalloc((size_t)(d->n - 1));     memcpy(d->unrel, s->unrel, (size_t)(d->n - 1));
}

void ls_push_anno(orc_list *l, const orc_trans *s)
{
    /* gtf.c:188-204 add_anno_trans: cov forced to 1, no flag arrays */
    orc_trans *d = ls_slot(l);
    copy_core(d, s);
    d->cov = 1;
}

void ls_free(orc_list *l)
{
    int i;
    for (i = 0; i < l->n; ++i) tr_release(&l->t[i]);
    free(l->t); free(l);
}

/* ------------------------------------------------------- CIGAR -> exons */

void orc_cigar_to_exons(orc_trans *t, int tid, int pos0, uint8_t rev,
                        const uint32_t *cig, int n_cig, int min_exon, int min_intron, int max_delet)
{
    /* bam2gtf.c:31-78 gen_exon.  1-based closed exons; an N of at least
     * min_intron or a D longer than max_delet closes the running exon, but the
     * closed exon is only kept if it is the first one or at least min_exon long
     * (Q4: a dropped short exon still moves `start`, fusing its two introns). */
    int start = pos0 + 1, end = start - 1, k;
    t->n = 0;
    for (k = 0; k < n_cig; ++k) {
        int len = (int)(cig[k] >> 4);
        unsigned op = cig[k] & 0xfu;
        switch (op) {
        case 3: /* N */
            if (len >= min_intron) {
                if (t->n == 0 || end - start + 1 >= min_exon) tr_push_exon(t, tid, start, end, rev);
                start = end + len + 1;
            }
            end += len;
            break;
        case 2: /* D */
            if (len > max_delet) {
                if (t->n == 0 || end - start + 1 >= min_exon) tr_push_exon(t, tid, start, end, rev);
                start = end + len + 1;
            }
            end += len;
            break;
        case 0: case 7: case 8: /* M = X */
            end += len;
            break;
        case 1: case 4: case 5: case 6: case 9: /* I S H P B */
            break;
        default:
            fprintf(stderr, "Error: unknown cigar type: %d.\n", (int)op);
            break;
        }
    }
    tr_push_exon(t, tid, start, end, rev);
}

/* ----------------------------------------------------- exon predicates */

static float ovlp_frac(const orc_exon *a, const orc_exon *b)
{
    /* update_gtf.c:80-89 exon_overlap_frac: int / double -> returned as float (Q6) */
    if (a->start > b->end || b->start > a->end) return 0.0;
    int hi = a->end < b->end ? a->end : b->end;
    int lo = a->start > b->start ? a->start : b->start;
    int la = a->end - a->start + 1, lb = b->end - b->start + 1;
    int mn = la < lb ? la : lb;
    return ((hi - lo + 1) / (mn + 0.0));
}

static int ovlp(const orc_exon *a, const orc_exon *b)
{
    /* update_gtf.c:91-95 exon_overlap: closed intervals */
    return !(a->start > b->end || b->start > a->end);
}

/* ------------------------------------------------- full-length evidence */

static void full_evidence(orc_trans *r, const orc_trans *a, int level)
{
    /* update_gtf.c:629-681 check_full */
    int last_r = r->n - 1, last_a = a->n - 1, k;
    if (r->lfull && r->rfull) return;
    if (level == 1) {
        if (!r->lfull && r->ex[0].end == a->ex[0].end) r->lfull = 1;
        if (!r->rfull && r->ex[last_r].start == a->ex[last_a].start) r->rfull = 1;
    } else if (level == 2) {
        if (!r->lfull && ovlp(&r->ex[0], &a->ex[0])) r->lfull = 1;
        if (!r->rfull && ovlp(&r->ex[last_r], &a->ex[last_a])) r->rfull = 1;
    } else if (level == 3 || level == 4) {
        if (!r->lfull) {
            if (ovlp(&r->ex[0], &a->ex[0])) r->lfull = 1;
            else for (k = 0; k < a->n; ++k) if (ovlp(&r->ex[0], &a->ex[k])) { r->lnoth = 0; break; }
        }
        if (level == 3 && !r->rfull) {
            if (ovlp(&r->ex[last_r], &a->ex[last_a])) r->rfull = 1;
            else for (k = 0; k < a->n; ++k) if (ovlp(&r->ex[last_r], &a->ex[k])) { r->rnoth = 0; break; }
        }
    }
}

static void full_decide(orc_trans *r, int level)
{
    /* update_gtf.c:683-696 set_full */
    if (level == 5) r->full = 1;
    else if (level == 4) r->full = (r->lfull || r->lnoth);
    else if (level == 3) r->full = ((r->lfull || r->lnoth) && (r->rfull || r->rnoth));
    else r->full = (r->lfull && r->rfull);
}

/* ------------------------------------------------- splice-site matching */

static int site_compare(orc_trans *r, const orc_trans *a, int dis)
{
    /* update_gtf.c:717-779 check_splice_site.  returns 1 known / 2 has known
     * site / 0 neither.  Acceptor loop compares r->ex[j].start (Q1). */
    int r_sites = (r->n - 1) * 2, r_in = 0, same = 0, i, j;
    int lo = r->start > a->start ? r->start : a->start;
    int hi = r->end < a->end ? r->end : a->end;
    for (j = 0; j + 1 < r->n; ++j) {
        if (r->ex[j].end >= lo && r->ex[j].end <= hi) ++r_in;
        if (r->ex[j + 1].start >= lo && r->ex[j + 1].start <= hi) ++r_in;
    }
    for (i = 0; i + 1 < a->n; ++i) {
        int don = a->ex[i].end, acc = a->ex[i + 1].start;
        if (don >= lo && don <= hi)
            for (j = 0; j + 1 < r->n; ++j)
                if (abs(don - r->ex[j].end) <= dis) { ++same; r->nov_site[2 * j] = 0; }
        if (acc >= lo && acc <= hi)
            for (j = 0; j + 1 < r->n; ++j)
                if (abs(acc - r->ex[j].start) <= dis) { ++same; r->nov_site[2 * j + 1] = 0; }
    }
    for (i = 0; i < a->n; ++i)
        for (j = 0; j < r->n; ++j)
            if (abs(a->ex[i].start - r->ex[j].start) <= dis && abs(a->ex[i].end - r->ex[j].end) <= dis)
                r->nov_exon[j] = 0;
    for (i = 0; i + 1 < a->n; ++i)
        for (j = 0; j + 1 < r->n; ++j)
            if (abs(a->ex[i].end - r->ex[j].end) <= dis && abs(a->ex[i + 1].start - r->ex[j + 1].start) <= dis)
                r->nov_junc[j] = 0;
    if (r_sites == r_in && r_in == same) { r->known = 1; return 1; }
    if (same > 0) { r->has_known_site = 1; return 2; }
    return 0;
}

static int span_order(const orc_trans *x, const orc_trans *y)
{
    /* update_gtf.c:786-790 comp_trans: <= makes 1-bp contact "no overlap" (Q5) */
    if (x->tid < y->tid || (x->tid == y->tid && x->end <= y->start)) return -1;
    if (y->tid < x->tid || (y->tid == x->tid && y->end <= x->start)) return 1;
    return 0;
}

int orc_sweep_annotation(orc_trans *r, const orc_list *A, int *cursor, const orc_params *p)
{
    /* update_gtf.c:792-835 check_with_anno_trans; returns ref_anno_i (or -1).
     * The gene-name copy is left to the caller (it needs A's strings). */
    int i, ref = -1, single = (r->n == 1);
    for (i = *cursor; i < A->n; ++i) {
        const orc_trans *a = &A->t[i];
        int c = span_order(r, a);
        if (c < 0) break;
        if (c > 0) { if (*cursor == i) ++*cursor; continue; }
        full_evidence(r, a, p->full_level);
        if (single && a->n == 1) {
            if (ovlp_frac(&r->ex[0], &a->ex[0]) >= p->single_exon_ovlp_frac) { ref = i; r->known = 1; break; }
        } else if (!single && a->n > 1) {
            int v = site_compare(r, a, p->ss_dis);
            if (v == 1) { ref = i; break; }
            if (v == 2) ref = i;
        }
    }
    if (ref != -1) {
        uint8_t arev = A->t[ref].rev;
        if (arev != r->rev) { for (i = 0; i < r->n; ++i) r->ex[i].rev = arev; r->rev = arev; }
    }
    full_decide(r, p->full_level);
    return ref;
}

/* -------------------------------------------- short-read junction check */

static int sj_find(int tid, int don, int acc, const orc_sj *S, int n, int from, const orc_params *p)
{
    /* update_gtf.c:589-603 check_short_sj1 */
    int i;
    for (i = from; i < n; ++i) {
        if (S[i].tid > tid || (S[i].tid == tid && S[i].don >= acc)) return 0;
        if (abs(S[i].don - don) <= p->ss_dis && abs(S[i].acc - acc) <= p->ss_dis) {
            int c = p->use_multi ? S[i].uniq_c + S[i].multi_c : S[i].uniq_c;
            if (c >= p->min_sj_cnt) return 1;
        }
    }
    return 0;
}

int orc_validate_junctions(orc_trans *r, const orc_sj *S, int n, int *cursor, const orc_params *p)
{
    /* update_gtf.c:698-709 check_with_short_sj + :609-627 check_short_sj.
     * Q7: when the table is exhausted or its cursor row lies beyond the read the
     * answer is "unsupported" without any unreliable flag being set. */
    int i = *cursor, j, ok = 0, decided = 0;
    while (i < n) {
        if (S[i].tid < r->tid || (S[i].tid == r->tid && S[i].acc <= r->start)) { ++i; *cursor = i; continue; }
        if (S[i].tid > r->tid || (S[i].tid == r->tid && S[i].don >= r->end)) { ok = 0; decided = 1; break; }
        ok = 1;
        for (j = 0; j + 1 < r->n; ++j)
            if (r->nov_junc[j] && !sj_find(r->tid, r->ex[j].end + 1, r->ex[j + 1].start - 1, S, n, i, p)) {
                r->unrel[j] = 1; ok = 0;
            }
        decided = 1;
        break;
    }
    if (!decided) ok = 0;
    r->has_unrel = (uint8_t)(1 - ok);
    return ok;
}

/* ------------------------------------------------------------ split */

static void emit_piece(orc_list *out, const orc_trans *r, int first, int last, int ordinal)
{
    /* update_gtf.c:850-876 / :883-909.  tid/start/end/rev stay 0 (Q2). */
    orc_trans *t = ls_slot(out);
    int j, w;
    char buf[256];
    t->cov = 1;
    for (j = first; j <= last; ++j) tr_push_exon(t, r->ex[j].tid, r->ex[j].start, r->ex[j].end, r->ex[j].rev);
    t->full = t->lfull = t->rfull = 0; t->lnoth = t->rnoth = 1;
    t->known = t->has_known_site = t->has_unrel = 0; t->partial = 1;
    t->nov_exon = (uint8_t *)xmalloc((size_t)t->n);
    t->nov_site = (uint8_t *)xmalloc((size_t)(t->n - 1) * 2);
    t->nov_junc = (uint8_t *)xmalloc((size_t)(t->n - 1));
    t->unrel    = (uint8_t *)xmalloc((size_t)(t->n - 1));
    memset(t->unrel, 0, (size_t)(t->n - 1));
    for (j = first; j <= last; ++j) t->nov_exon[j - first] = r->nov_exon[j];
    for (j = first; j < last; ++j) {
        t->nov_site[(j - first) * 2] = r->nov_site[j * 2];
        t->nov_site[(j - first) * 2 + 1] = r->nov_site[j * 2 + 1];
        t->nov_junc[j - first] = r->nov_junc[j];
    }
    w = snprintf(buf, sizeof buf, "%s.split.%d", r->tids, ordinal);  orc_set_name(t->tids, buf, "split_trans");
    w = snprintf(buf, sizeof buf, "%s.split.%d", r->tname, ordinal); orc_set_name(t->tname, buf, "split_trans");
    (void)w;
    memcpy(t->gid, r->gid, ORC_NAME_MAX); memcpy(t->gname, r->gname, ORC_NAME_MAX);
}

orc_list *orc_split(const orc_trans *r)
{
    /* update_gtf.c:837-913 split_trans */
    orc_list *out = ls_new();
    int i, first = 0, seen_novel = 0, seen_known = 0, k = 0;
    for (i = 0; i + 1 < r->n; ++i) {
        if (r->nov_junc[i]) seen_novel = 1; else seen_known = 1;
        if (r->unrel[i]) {
            if (seen_novel && seen_known && i - first >= 1) emit_piece(out, r, first, i, k++);
            first = i + 1; seen_novel = seen_known = 0;
        }
    }
    /* here i == r->n - 1 */
    if (seen_novel && seen_known && i - first >= 1) emit_piece(out, r, first, i, k++);
    return out;
}

/* ------------------------------------------------------------ merge */

static int chain_identity(const orc_trans *x, const orc_trans *y, int ss, int ed)
{
    /* gtf.c:54-92 check_iden: 0 identical, 2 contained, -1 different.  The
     * unequal-length branch can never return 1 (Q8), and its inner walk stops
     * when the longer chain runs out, leaving the rest of the shorter unchecked. */
    const orc_trans *l, *s;
    int i, j;
    if (x->n == y->n) {
        l = x; s = y;
        if (abs(l->ex[0].start - s->ex[0].start) > ed) return -1;
        for (i = 0; i + 1 < l->n; ++i) {
            if (abs(l->ex[i].end - s->ex[i].end) > ss) return -1;
            if (abs(l->ex[i + 1].start - s->ex[i + 1].start) > ss) return -1;
        }
        if (abs(l->ex[l->n - 1].end - s->ex[s->n - 1].end) > ed) return -1;
        return 0;
    }
    if (x->n > y->n) { l = x; s = y; } else { l = y; s = x; }
    int verdict = -1;
    if (abs(l->ex[0].start - s->ex[0].start) > ed) return -1;
    for (i = 0; i + 1 < l->n; ++i) {
        if (abs(l->ex[i].end - s->ex[0].end) <= ss && abs(l->ex[i + 1].start - s->ex[1].start) <= ss) {
            verdict = 2;
            for (i = i + 1, j = 1; i + 1 < l->n && j + 1 < s->n; ++i, ++j) {
                if (abs(l->ex[i].end - s->ex[j].end) > ss) return -1;
                if (abs(l->ex[i + 1].start - s->ex[j + 1].start) > ss) return -1;
            }
            break;
        }
    }
    if (abs(l->ex[l->n - 1].end - s->ex[s->n - 1].end) > ed) return -1;
    return verdict;
}

static int fold_multi(const orc_trans *t, orc_trans *T, int ss, int ed)
{
    /* update_gtf.c:98-119 merge_trans1 */
    int v = chain_identity(t, T, ss, ed);
    if (v == 0) {
        int a = t->n - 1, b = T->n - 1;
        T->cov++;
        if (t->ex[0].start < T->ex[0].start) { T->ex[0].start = t->ex[0].start; T->start = t->ex[0].start; }
        if (t->ex[a].end > T->ex[b].end) { T->ex[b].end = t->ex[a].end; T->end = t->ex[a].end; }
        return 1;
    }
    return v == 2;   /* v == 1 is unreachable (Q8) */
}

static int fold_single(const orc_trans *t, orc_trans *T, int ed, float frac)
{
    /* update_gtf.c:122-140 merge_trans2 */
    if (abs(t->ex[0].start - T->ex[0].start) > ed) return 0;
    if (abs(t->ex[0].end - T->ex[0].end) > ed) return 0;
    if (ovlp_frac(&t->ex[0], &T->ex[0]) >= frac) {
        T->cov++;
        if (t->ex[0].start < T->ex[0].start) { T->ex[0].start = t->ex[0].start; T->start = t->ex[0].start; }
        if (t->ex[0].end > T->ex[0].end) { T->ex[0].end = t->ex[0].end; T->end = t->ex[0].end; }
        return 1;
    }
    return 0;
}

int orc_merge(const orc_trans *t, orc_list *U, const orc_params *p)
{
    /* update_gtf.c:144-163 merge_trans: greedy backward scan, early return */
    int i;
    for (i = U->n - 1; i >= 0; --i) {
        orc_trans *T = &U->t[i];
        if (t->tid > T->tid || t->start > T->end) return 0;
        if (p->force_strand && t->rev != T->rev) continue;
        if (t->n == 1 && T->n == 1) { if (fold_single(t, T, p->end_dis, p->single_exon_ovlp_frac)) return 1; }
        else if (t->n > 1 && T->n > 1) { if (fold_multi(t, T, p->ss_dis, p->end_dis)) return 1; }
    }
    return 0;
}

/* ---------------------------------------------------------- driver loop */

void orc_check_all(orc_list *R, const orc_list *A, const orc_sj *S, int n_sj,
                   orc_list *updated, orc_list *known, orc_list *novel, orc_list *unrecog,
                   const orc_params *p)
{
    /* update_gtf.c:936-965 check_trans */
    int i, j, anno_cur = 0, sj_cur = 0;
    for (i = 0; i < R->n; ++i) {
        orc_trans *r = &R->t[i];
        int ref = orc_sweep_annotation(r, A, &anno_cur, p);
        /* update_gtf.c:823-833: gene id/name from the reference transcript or "NA" */
        if (ref != -1) { tr_finish(r); memcpy(r->gid, A->t[ref].gid, ORC_NAME_MAX); memcpy(r->gname, A->t[ref].gname, ORC_NAME_MAX); }
        else { tr_finish(r); strcpy(r->gid, "NA"); strcpy(r->gname, "NA"); }
        if (!r->full) continue;                                   /* Q3 */
        if (r->known) { ls_push_read(known, r); continue; }
        if (r->has_known_site) {
            if (n_sj == 0 || orc_validate_junctions(r, S, n_sj, &sj_cur, p)) {
                ls_push_read(novel, r);
                if (!orc_merge(r, updated, p)) ls_push_read(updated, r);
            } else if (p->split_trans) {
                orc_list *pieces = orc_split(r);
                for (j = 0; j < pieces->n; ++j) {
                    ls_push_read(novel, &pieces->t[j]);
                    if (!orc_merge(&pieces->t[j], updated, p)) ls_push_read(updated, &pieces->t[j]);
                }
                ls_free(pieces);
            }
        } else ls_push_read(unrecog, r);
    }
}

/* ------------------------------------------------ SoA entry (see oracle.h) */

int64_t orc_classify_soa(
    int64_t n_reads, const int32_t *r_tid, const int32_t *r_pos, const uint8_t *r_rev,
    const int64_t *cig_off, const uint32_t *cig,
    int64_t n_tx, const int32_t *tx_tid, const int32_t *tx_start, const int32_t *tx_end,
    const uint8_t *tx_rev, const int64_t *tx_ex_off, const int32_t *ex_start, const int32_t *ex_end,
    int64_t n_sj, const int32_t *sj_tid, const int32_t *sj_don, const int32_t *sj_acc,
    const int32_t *sj_uniq, const int32_t *sj_multi,
    const orc_params *prm,
    int64_t ex_cap, int64_t *out_ex_off, int32_t *out_ex_start, int32_t *out_ex_end,
    uint8_t *out_ex_flag, uint32_t *out_info, int32_t *out_ref_tx)
{
    int64_t i, k, used = 0;
    orc_list *A = ls_new();
    for (i = 0; i < n_tx; ++i) {
        orc_trans *a = ls_slot(A);
        for (k = tx_ex_off[i]; k < tx_ex_off[i + 1]; ++k) tr_push_exon(a, tx_tid[i], ex_start[k], ex_end[k], tx_rev[i]);
        a->tid = tx_tid[i]; a->rev = tx_rev[i]; a->start = tx_start[i]; a->end = tx_end[i]; a->cov = 1;
    }
    orc_sj *S = (orc_sj *)xmalloc((size_t)(n_sj > 0 ? n_sj : 1) * sizeof(orc_sj));
    for (i = 0; i < n_sj; ++i) {
        S[i].tid = sj_tid[i]; S[i].don = sj_don[i]; S[i].acc = sj_acc[i];
        S[i].uniq_c = sj_uniq[i]; S[i].multi_c = sj_multi[i];
    }
    int anno_cur = 0, sj_cur = 0;
    orc_trans r; tr_zero(&r);
    for (i = 0; i < n_reads; ++i) {
        orc_cigar_to_exons(&r, r_tid[i], r_pos[i], r_rev[i], cig + cig_off[i], (int)(cig_off[i + 1] - cig_off[i]),
                           prm->min_exon, prm->min_intron, prm->max_delet);
        free(r.nov_exon); free(r.nov_site); free(r.nov_junc); free(r.unrel);
        tr_alloc_read_flags(&r);
        tr_finish(&r);
        int ref = orc_sweep_annotation(&r, A, &anno_cur, prm);
        uint32_t info = 0;
        if (r.full && !r.known && r.has_known_site && n_sj > 0) {
            info |= ORC_INFO_SJ_CHECKED;
            if (orc_validate_junctions(&r, S, (int)n_sj, &sj_cur, prm)) info |= ORC_INFO_SJ_PASS;
        }
        if (r.known) info |= ORC_INFO_KNOWN;
        if (r.has_known_site) info |= ORC_INFO_KNOWN_SITE;
        if (r.full) info |= ORC_INFO_FULL;
        if (r.rev) info |= ORC_INFO_REV;
        if (r.has_unrel) info |= ORC_INFO_UNREL;
        if (used + r.n > ex_cap) { used = -1; break; }
        out_ex_off[i] = used;
        for (k = 0; k < r.n; ++k) {
            uint8_t f = 0;
            if (r.nov_exon[k]) f |= ORC_EXF_NOVEL_EXON;
            if (k + 1 < r.n) {
                if (r.nov_site[2 * k]) f |= ORC_EXF_NOVEL_DON;
                if (r.nov_site[2 * k + 1]) f |= ORC_EXF_NOVEL_ACC;
                if (r.nov_junc[k]) f |= ORC_EXF_NOVEL_JUNC;
                if (r.unrel[k]) f |= ORC_EXF_UNREL_JUNC;
            }
            out_ex_start[used + k] = r.ex[k].start; out_ex_end[used + k] = r.ex[k].end; out_ex_flag[used + k] = f;
        }
        used += r.n;
        out_info[i] = info; out_ref_tx[i] = ref;
    }
    if (used >= 0) out_ex_off[n_reads] = used;
    tr_release(&r);
    free(S);
    /* annotation entries own only exons */
    ls_free(A);
    return used;
}
