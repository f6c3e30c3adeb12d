/* oracle_io.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Text readers (SAM, GTF, STAR SJ.out.tab), the byte-exact writers and the
 * three sub-command drivers of the restatement.  SAM *text* only: the oracle
 * never needs BAM because parity runs feed both sides the same SAM.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <getopt.h>
#include "oracle_int.h"

/* --------------------------------------------------------- chromosomes */

static orc_chroms *chroms_new(void)
{
    orc_chroms *c = (orc_chroms *)calloc(1, sizeof *c);
    c->cap = 32; c->name = (char **)calloc((size_t)c->cap, sizeof(char *));
    return c;
}

static int chroms_find(const orc_chroms *c, const char *s, int limit)
{
    int i;
    for (i = 0; i < limit; ++i) if (strcmp(c->name[i], s) == 0) return i;
    return -1;
}

static int chroms_intern(orc_chroms *c, const char *s)
{
    /* gtf.c:389-403 get_chr_id: linear search, append when missing */
    int i = chroms_find(c, s, c->n);
    if (i >= 0) return i;
    if (strlen(s) >= ORC_NAME_MAX) orc_die(0, "get_chr_id", "chromosome name of 100 or more characters");
    if (c->n == c->cap) { c->cap *= 2; c->name = (char **)realloc(c->name, (size_t)c->cap * sizeof(char *)); }
    c->name[c->n] = strdup(s);
    return c->n++;
}

static void chroms_free(orc_chroms *c)
{
    int i; for (i = 0; i < c->n; ++i) free(c->name[i]);
    free(c->name); free(c);
}

/* ------------------------------------------------------------ SAM text */

typedef struct { FILE *fp; char *line; size_t cap; int have_line; } sam_in;

static sam_in *sam_open_text(const char *fn, orc_chroms *c, const char *who)
{
    /* replaces sam_open + sam_hdr_read + bam_set_cname (update_gtf.c:1065-1067,
     * gtf.c:405-412): @SQ SN: order defines tid */
    sam_in *s = (sam_in *)calloc(1, sizeof *s);
    s->fp = fopen(fn, "r");
    if (!s->fp) { char m[1200]; snprintf(m, sizeof m, "Can not open \"%s\"\n", fn); orc_die(0, who, m); }
    ssize_t len;
    while ((len = getline(&s->line, &s->cap, s->fp)) >= 0) {
        if (s->line[0] != '@') { s->have_line = 1; break; }
        if (strncmp(s->line, "@SQ", 3) == 0) {
            char *p = strstr(s->line, "\tSN:");
            if (p) {
                p += 4; size_t k = strcspn(p, "\t\r\n");
                char save = p[k]; p[k] = 0; chroms_intern(c, p); p[k] = save;
            }
        }
    }
    c->n_hdr = c->n;
    return s;
}

static void sam_close_text(sam_in *s) { if (s->fp) fclose(s->fp); free(s->line); free(s); }

typedef struct {
    char qname[ORC_NAME_MAX];
    int flag, tid, pos0;
    uint32_t *cig; int n_cig, cap_cig;
    int has_xs; char xs_type, xs_val;
} sam_rec;

static int sam_next(sam_in *s, const orc_chroms *c, sam_rec *r)
{
    /* replaces sam_read1 for the fields the path reads:
     * core.tid/pos/flag/n_cigar, cigar, qname, aux "XS" */
    for (;;) {
        if (!s->have_line) { if (getline(&s->line, &s->cap, s->fp) < 0) return -1; }
        s->have_line = 0;
        if (s->line[0] == '@' || s->line[0] == '\n' || s->line[0] == 0) continue;
        break;
    }
    char *f[12]; int nf = 0; char *p = s->line, *aux = NULL;
    while (nf < 11) {
        f[nf++] = p;
        char *q = strpbrk(p, "\t\r\n");
        if (!q) { p = NULL; break; }
        if (*q != '\t') { *q = 0; p = NULL; break; }
        *q = 0; p = q + 1;
    }
    if (nf < 11) orc_die(0, "sam_read1", "truncated SAM record");
    aux = p;
    orc_set_name(r->qname, f[0], "sam_read1");
    r->flag = atoi(f[1]);
    if (strcmp(f[2], "*") == 0) r->tid = -1;
    else { r->tid = chroms_find(c, f[2], c->n_hdr); if (r->tid < 0) orc_die(0, "sam_read1", "reference name not in header"); }
    r->pos0 = atoi(f[3]) - 1;
    r->n_cig = 0;
    if (strcmp(f[5], "*") != 0) {
        const char *q = f[5];
        while (*q) {
            long len = strtol(q, (char **)&q, 10);
            const char *ops = "MIDNSHP=XB", *o = strchr(ops, *q);
            if (!o || !*q) orc_die(0, "sam_read1", "bad CIGAR");
            if (r->n_cig == r->cap_cig) { r->cap_cig = r->cap_cig ? r->cap_cig * 2 : 16; r->cig = (uint32_t *)realloc(r->cig, (size_t)r->cap_cig * 4); }
            r->cig[r->n_cig++] = ((uint32_t)len << 4) | (uint32_t)(o - ops);
            ++q;
        }
    }
    r->has_xs = 0;
    while (aux && *aux) {
        size_t k = strcspn(aux, "\t\r\n");
        if (!r->has_xs && k >= 5 && aux[0] == 'X' && aux[1] == 'S' && aux[2] == ':') {
            r->has_xs = 1; r->xs_type = aux[3]; r->xs_val = (k >= 6) ? aux[5] : 0;
        }
        if (aux[k] != '\t') break;
        aux += k + 1;
    }
    return 0;
}

static uint8_t sam_strand(const sam_rec *r)
{
    /* bam2gtf.c:35-37: XS present -> (bam_aux2A == '+') ? 0 : 1, where bam_aux2A
     * yields 0 for any non-'A' typed tag; otherwise FLAG & 16 */
    if (r->has_xs) return (r->xs_type == 'A' && r->xs_val == '+') ? 0 : 1;
    return (r->flag & 16) ? 1 : 0;
}

static void load_sam_reads(const char *fn, orc_chroms *c, orc_list *R, const orc_params *p, const char *who)
{
    /* bam2gtf.c:89-110 read_bam_trans (every record, mapped or not: Q9) */
    sam_in *s = sam_open_text(fn, c, who);
    sam_rec rec; memset(&rec, 0, sizeof rec);
    orc_trans t; tr_zero(&t);
    while (sam_next(s, c, &rec) == 0) {
        tr_release(&t); tr_zero(&t);
        if (!(rec.flag & 4))
            orc_cigar_to_exons(&t, rec.tid, rec.pos0, sam_strand(&rec), rec.cig, rec.n_cig, p->min_exon, p->min_intron, p->max_delet);
        tr_alloc_read_flags(&t);
        tr_finish(&t);
        memcpy(t.gid, rec.qname, ORC_NAME_MAX); memcpy(t.gname, rec.qname, ORC_NAME_MAX);
        memcpy(t.tids, rec.qname, ORC_NAME_MAX); memcpy(t.tname, rec.qname, ORC_NAME_MAX);
        ls_push_read(R, &t);
    }
    tr_release(&t); free(rec.cig);
    sam_close_text(s);
}

/* ----------------------------------------------------------------- GTF */

static void attr_value(const char *attrs, const char *tag, char *out)
{
    /* gtf.c:317-326 gtf_add_info: first substring hit, value starts 2 bytes after
     * the tag, runs to the next double quote (Q11) */
    size_t tl = strlen(tag), i;
    for (i = 0; attrs[i]; ++i)
        if (strncmp(attrs + i, tag, tl) == 0) {
            /* the reference reads at attrs+i+tl+2 even if that is past the NUL;
             * only do so while inside the string */
            size_t k = i + tl, step = 0;
            while (step < 2 && attrs[k]) { ++k; ++step; }
            if (step == 2) sscanf(attrs + k, "%[^\"]", out);
            return;
        }
}

static void load_gtf(const char *fn, const orc_chroms *c, orc_list *T, int as_reads)
{
    /* gtf.c:468-521 read_anno_trans (as_reads = 0) and gtf.c:524-595
     * read_gtf_trans (as_reads = 1).  fgets(1024) line splitting (Q10), sscanf
     * whitespace field splitting with values that persist across lines, only
     * "exon" rows, transcripts = runs of equal transcript_id (Q12). */
    FILE *fp = fopen(fn, "r");
    if (!fp) { char m[1200]; snprintf(m, sizeof m, "fail to open file '%s'", fn); orc_die(1, as_reads ? "read_gtf_trans" : "read_anno_trans", m); }
    char line[1024], ref[1024] = "", type[1024] = "", attrs[1024] = "";
    char gid[1024] = "", gname[1024] = "", tid_s[1024] = "", tname[1024] = "";
    char last_tid[1024] = "", last_gid[1024] = "";
    int start = 0, end = 0; char strand = 0;
    orc_trans t; tr_zero(&t);
    const char *who = as_reads ? "read_gtf_trans" : "read_anno_trans";
    while (fgets(line, 1024, fp)) {
        if (as_reads) sscanf(line, "%s\t%*s\t%s\t%d\t%d\t%*s\t%c\t%*s\t%[^\n]", ref, type, &start, &end, &strand, attrs);
        if (line[0] == '#') continue;
        sscanf(line, "%s\t%*s\t%s\t%d\t%d\t%*s\t%c\t%*s\t%[^\n]", ref, type, &start, &end, &strand, attrs);
        if (strcmp(type, "exon") != 0) continue;
        uint8_t rev = (strand == '-');
        int tid = chroms_find(c, ref, c->n_hdr);        /* bam_name2id: header names only */
        memset(gid, 0, strlen(gid));     attr_value(attrs, "gene_id", gid);
        memset(gname, 0, strlen(gname)); attr_value(attrs, "gene_name", gname);
        if (!gid[0] && !gname[0]) orc_die(1, who, "GTF format error. (No gene id or gene name found.");
        if (!gid[0]) strcpy(gid, gname); else if (!gname[0]) strcpy(gname, gid);
        memset(tid_s, 0, strlen(tid_s)); attr_value(attrs, "transcript_id", tid_s);
        memset(tname, 0, strlen(tname)); attr_value(attrs, "transcript_name", tname);
        if (!tid_s[0] && !tname[0]) orc_die(1, who, "GTF format error. (No transcript id or transcript name found.");
        if (!tid_s[0]) strcpy(tid_s, tname); else if (!tname[0]) strcpy(tname, tid_s);

        T->gene_n += strcmp(as_reads ? gname : gid, last_gid) != 0;
        if (strcmp(tid_s, last_tid) != 0) {
            if (t.n >= 1) {
                if (as_reads) { tr_alloc_read_flags(&t); tr_finish(&t); ls_push_read(T, &t); }
                else { tr_finish(&t); ls_push_anno(T, &t); }
                tr_release(&t); tr_zero(&t);
            }
            t.n = 0;
            orc_set_name(t.tname, tname, who); orc_set_name(t.tids, tid_s, who);
            orc_set_name(t.gname, gname, who); orc_set_name(t.gid, gid, who);
            strcpy(last_tid, tid_s); strcpy(last_gid, as_reads ? gname : gid);
        }
        tr_push_exon(&t, tid, start, end, rev);
    }
    if (t.n != 0) {
        if (as_reads) { tr_alloc_read_flags(&t); tr_finish(&t); ls_push_read(T, &t); }
        else { tr_finish(&t); ls_push_anno(T, &t); }
    }
    tr_release(&t);
    fclose(fp);
}

/* -------------------------------------------------------- SJ.out.tab */

typedef struct { orc_sj v; int seq; } sj_row;

static int sj_order(const void *pa, const void *pb)
{
    /* gtf.c:414-420 sj_group_comp; `seq` keeps equal keys in file order, which is
     * what glibc's (merge-sort) qsort gives the reference */
    const sj_row *a = (const sj_row *)pa, *b = (const sj_row *)pb;
    if (a->v.tid != b->v.tid) return a->v.tid - b->v.tid;
    if (a->v.don != b->v.don) return a->v.don - b->v.don;
    if (a->v.acc != b->v.acc) return a->v.acc - b->v.acc;
    return a->seq - b->seq;
}

static int load_sj(FILE *fp, orc_chroms *c, orc_sj **out)
{
    /* gtf.c:431-449 read_sj_group */
    *out = NULL;
    if (!fp) return 0;
    char line[1024], ref[1024] = "";
    int strand = 0, motif = 0, anno = 0, n = 0, cap = 0, i;
    sj_row *rows = NULL;
    while (fgets(line, 1024, fp)) {
        if (n == cap) { cap = cap ? cap * 2 : 10000; rows = (sj_row *)realloc(rows, (size_t)cap * sizeof(sj_row)); }
        sj_row *r = &rows[n]; memset(r, 0, sizeof *r);
        sscanf(line, "%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d", ref, &r->v.don, &r->v.acc, &strand, &motif, &anno,
               &r->v.uniq_c, &r->v.multi_c, &r->v.max_over);
        r->v.rev = (strand == 1 ? 0 : 1);
        r->v.tid = chroms_intern(c, ref);
        r->seq = n++;
    }
    qsort(rows, (size_t)n, sizeof(sj_row), sj_order);
    *out = (orc_sj *)malloc((size_t)(n ? n : 1) * sizeof(orc_sj));
    for (i = 0; i < n; ++i) (*out)[i] = rows[i].v;
    free(rows);
    return n;
}

/* ------------------------------------------------------------- writers */

static void write_trans_list(const orc_list *L, const orc_chroms *c, const char *src, FILE *out)
{
    /* gtf.c:607-632 print_read_trans */
    int i, j; char tmp[1024], ex_attr[1024], tr_attr[1024];
    for (i = 0; i < L->n; ++i) {
        const orc_trans *t = &L->t[i];
        ex_attr[0] = 0;
        if (t->gid[0])   { sprintf(tmp, " gene_id \"%s\";", t->gid); strcat(ex_attr, tmp); }
        if (t->tids[0])  { sprintf(tmp, " transcript_id \"%s\";", t->tids); strcat(ex_attr, tmp); }
        if (t->gname[0]) { sprintf(tmp, " gene_name \"%s\";", t->gname); strcat(ex_attr, tmp); }
        if (t->tname[0]) { sprintf(tmp, " transcript_name \"%s\";", t->tname); strcat(ex_attr, tmp); }
        strcpy(tr_attr, ex_attr); sprintf(tmp, " transcript_cov \"%d\";", t->cov); strcat(tr_attr, tmp);
        fprintf(out, "%s\t%s\ttranscript\t%d\t%d\t.\t%c\t.\t%s\n", c->name[t->tid], src, t->start, t->end, "+-"[t->rev], tr_attr + 1);
        if (t->rev) for (j = t->n - 1; j >= 0; --j)
            fprintf(out, "%s\t%s\texon\t%d\t%d\t.\t%c\t.\t%s\n", c->name[t->ex[j].tid], src, t->ex[j].start, t->ex[j].end, "+-"[t->ex[j].rev], ex_attr + 1);
        else for (j = 0; j < t->n; ++j)
            fprintf(out, "%s\t%s\texon\t%d\t%d\t.\t%c\t.\t%s\n", c->name[t->ex[j].tid], src, t->ex[j].start, t->ex[j].end, "+-"[t->ex[j].rev], ex_attr + 1);
    }
}

static void write_index_list(FILE *fp, const uint8_t *flag, int n, int trailing_tab)
{
    int j, cnt = 0, first = 1;
    for (j = 0; j < n; ++j) cnt += flag[j];
    fprintf(fp, "%d\t", cnt);
    if (cnt == 0) { fprintf(fp, "NA\t"); return; }
    for (j = 0; j < n; ++j) if (flag[j]) { if (!first) fputc(',', fp); first = 0; fprintf(fp, "%d", j); }
    if (trailing_tab) fputc('\t', fp);
}

static void write_detail(const orc_list *R, const orc_chroms *c, FILE *fp)
{
    /* update_gtf.c:297-419 print_bam_detail_trans: every field is followed by a
     * tab except a non-empty last index list */
    int i, j;
    fprintf(fp, "ReadName\tchr\tstrand\tNovel\tGeneID\tGeneName\tExonCount\tExonStart\tExonEnd\tNovelExonCount\tNovelExonIndex\tNovelSiteCount\tNovelSiteIndex\tNovelJunctionCount\tNovelJunctionIndex\tUnreliableJunctionCount\tUnreliableJunctionIndex\n");
    for (i = 0; i < R->n; ++i) {
        const orc_trans *t = &R->t[i];
        int cls = t->known ? 0 : (t->has_known_site ? 1 : 2);
        fprintf(fp, "%s\t%s\t%c\t%d\t%s\t%s\t%d\t", t->tname, c->name[t->tid], "+-"[t->rev], cls, t->gid, t->gname, t->n);
        for (j = 0; j < t->n; ++j) fprintf(fp, j ? ",%d" : "%d", t->ex[j].start);
        fputc('\t', fp);
        for (j = 0; j < t->n; ++j) fprintf(fp, j ? ",%d" : "%d", t->ex[j].end);
        fputc('\t', fp);
        write_index_list(fp, t->nov_exon, t->n, 1);
        write_index_list(fp, t->nov_site, (t->n - 1) * 2, 1);
        write_index_list(fp, t->nov_junc, t->n - 1, 1);
        write_index_list(fp, t->unrel, t->n - 1, 0);
        fputc('\n', fp);
    }
}

/* summary de-dup lists: update_gtf.c:165-295 */
typedef struct { int tid; char gid[ORC_NAME_MAX]; } s_gene;
typedef struct { int tid, site; } s_site;
typedef struct { int tid, don, acc; } s_junc;

#define GROW(arr, n, cap, T) do { if ((n) == (cap)) { (cap) = (cap) ? (cap) * 2 : 1; (arr) = (T *)realloc((arr), (size_t)(cap) * sizeof(T)); } } while (0)

static void write_summary(const orc_chroms *c, const orc_list *A, orc_list *U, const orc_list *R,
                          const orc_params *p, FILE *sum, FILE *bed)
{
    /* update_gtf.c:421-587 print_trans_summary */
    int i, j, k;
    s_gene *G = NULL; int g_cap = 0, upd_genes = 0, known_genes = 0;
    orc_exon *E = NULL; int e_n = 0, e_cap = 0;
    s_site *D = NULL, *Ac = NULL; int d_n = 0, d_cap = 0, a_n = 0, a_cap = 0;
    s_junc *J = NULL; int j_n = 0, j_cap = 0;
    int partial = 0;
    for (i = 0; i < U->n; ++i) {
        orc_trans *t = &U->t[i];
        /* gene: update_gtf.c:181-203, match first, then the tid stop */
        int hit = 0;
        for (k = upd_genes - 1; k >= 0; --k) { if (strcmp(t->gid, G[k].gid) == 0) { hit = 1; break; } if (t->tid > G[k].tid) break; }
        if (!hit) { GROW(G, upd_genes, g_cap, s_gene); G[upd_genes].tid = t->tid; strcpy(G[upd_genes].gid, t->gid); ++upd_genes; }
        partial += t->partial;
        for (j = 0; j < t->n; ++j) if (t->nov_exon[j]) {
            t->ex[j].etype = t->n > 1 ? ((j == 0 || j == t->n - 1) ? 0 : 1) : 2;
            hit = 0;
            for (k = e_n - 1; k >= 0; --k) {           /* update_gtf.c:211-222 */
                if (E[k].tid == t->ex[j].tid && E[k].start == t->ex[j].start && E[k].end == t->ex[j].end) { E[k].score += t->cov; hit = 1; break; }
                if (t->ex[j].tid > E[k].tid) break;
            }
            if (!hit) { GROW(E, e_n, e_cap, orc_exon); E[e_n] = t->ex[j]; E[e_n].score = t->cov; ++e_n; }
        }
        for (j = 0; j + 1 < t->n; ++j) if (t->nov_site[j * 2]) {
            hit = 0;
            for (k = d_n - 1; k >= 0; --k) { if (D[k].tid == t->tid && D[k].site == t->ex[j].end) { hit = 1; break; } if (t->tid > D[k].tid) break; }
            if (!hit) { GROW(D, d_n, d_cap, s_site); D[d_n].tid = t->tid; D[d_n].site = t->ex[j].end; ++d_n; }
        }
        for (j = 0; j + 1 < t->n; ++j) if (t->nov_site[j * 2 + 1]) {
            hit = 0;
            for (k = a_n - 1; k >= 0; --k) { if (Ac[k].tid == t->tid && Ac[k].site == t->ex[j + 1].start) { hit = 1; break; } if (t->tid > Ac[k].tid) break; }
            if (!hit) { GROW(Ac, a_n, a_cap, s_site); Ac[a_n].tid = t->tid; Ac[a_n].site = t->ex[j + 1].start; ++a_n; }
        }
        for (j = 0; j + 1 < t->n; ++j) if (t->nov_junc[j]) {
            hit = 0;
            for (k = j_n - 1; k >= 0; --k) { if (J[k].tid == t->tid && J[k].don == t->ex[j].end && J[k].acc == t->ex[j + 1].start) { hit = 1; break; } if (t->tid > J[k].tid) break; }
            if (!hit) { GROW(J, j_n, j_cap, s_junc); J[j_n].tid = t->tid; J[j_n].don = t->ex[j].end; J[j_n].acc = t->ex[j + 1].start; ++j_n; }
        }
    }
    /* read classes: update_gtf.c:496-528 (the known-gene list restarts at index 0
     * of the same array) */
    int n_known = 0, n_rel = 0, n_unrel = 0, n_unrec = 0;
    orc_list *uk = ls_new(), *ur = ls_new(), *uu = ls_new(), *un = ls_new();
    for (i = 0; i < R->n; ++i) {
        const orc_trans *t = &R->t[i];
        if (t->known) {
            ++n_known;
            int hit = 0;
            for (k = known_genes - 1; k >= 0; --k) { if (strcmp(t->gid, G[k].gid) == 0) { hit = 1; break; } if (t->tid > G[k].tid) break; }
            if (!hit) { GROW(G, known_genes, g_cap, s_gene); G[known_genes].tid = t->tid; strcpy(G[known_genes].gid, t->gid); ++known_genes; }
            if (!orc_merge(t, uk, p)) ls_push_read(uk, t);
        } else if (t->has_known_site) {
            if (t->has_unrel) { ++n_unrel; if (!orc_merge(t, uu, p)) ls_push_read(uu, t); }
            else { ++n_rel; if (!orc_merge(t, ur, p)) ls_push_read(ur, t); }
        } else { ++n_unrec; if (!orc_merge(t, un, p)) ls_push_read(un, t); }
    }
    if (sum) {
        fprintf(sum, "==== Annotaion ====\n");
        fprintf(sum, "Genes_of_annotation_GTF\t%d\n", A->gene_n);
        fprintf(sum, "Transcripts_of_annotation_GTF\t%d\n", A->n);
        fprintf(sum, "\n===================\n");
        fprintf(sum, "\n==== Updated information ====\n");
        fprintf(sum, "Updated_Genes\t%d\n", upd_genes);
        fprintf(sum, "Added_Novel_Transcripts\t%d\n", U->n);
        fprintf(sum, "Added_Novel_Full-read_Transcripts\t%d\n", U->n - partial);
        fprintf(sum, "Added_Novel_Partial-read_Transcripts\t%d\n", partial);
        fprintf(sum, "Added_Novel_Exons\t%d\n", e_n);
        fprintf(sum, "Added_Novel_Sites\t%d\n", d_n + a_n);
        fprintf(sum, "Added_Novel_Splice_Junctions\t%d\n", j_n);
        fprintf(sum, "\n=============================\n");
        fprintf(sum, "\n==== Known information ====\n");
        fprintf(sum, "Known_Transcripts_from_BAM\t%d\n", n_known);
        fprintf(sum, "Genes_of_Known_Transcripts_from_BAM\t%d\n", known_genes);
        fprintf(sum, "Uniq_Known_Transcripts_from_BAM\t%d\n", uk->n);
        fprintf(sum, "\n===========================\n");
        fprintf(sum, "\n==== Novel information ====\n");
        fprintf(sum, "Novel_Transcript_from_BAM\t%d\n", n_rel + n_unrel);
        fprintf(sum, "Novel_Transcript_from_BAM_with_All_Reliable_Junction\t%d\n", n_rel);
        fprintf(sum, "Uniq_Novel_Transcript_from_BAM_with_All_Reliable_Junction\t%d\n", ur->n);
        fprintf(sum, "Novel_Transcript_from_BAM_with_Unreliable_Junction\t%d\n", n_unrel);
        fprintf(sum, "Uniq_Novel_Transcript_from_BAM_with_Unreliable_Junction\t%d\n", uu->n);
        fprintf(sum, "\n===========================\n");
        fprintf(sum, "\n==== Unrecognized information ====\n");
        fprintf(sum, "Unrecognized_Transcript_from_BAM\t%d\n", n_unrec);
        fprintf(sum, "Uniq_Unrecognized_Transcript_from_BAM\t%d\n", un->n);
        fprintf(sum, "\n==================================\n");
    }
    if (bed) for (i = 0; i < e_n; ++i)      /* update_gtf.c:571-576, header names */
        fprintf(bed, "%s\t%d\t%d\t%c_exon\t%d\t%c\n", c->name[E[i].tid], E[i].start - 1, E[i].end, "TIS"[E[i].etype], E[i].score, "+-"[E[i].rev]);
    free(G); free(E); free(D); free(Ac); free(J);
    ls_free(uk); ls_free(ur); ls_free(uu); ls_free(un);
}

/* --------------------------------------------------------- sub-commands */

static void default_params(orc_params *p)
{
    /* update_gtf.c:24-35, gtf.h:118-127 */
    p->min_exon = 3; p->min_intron = 3; p->max_delet = 50; p->ss_dis = 0; p->end_dis = 0x7fffffff;
    p->full_level = 5; p->split_trans = 0; p->use_multi = 0; p->min_sj_cnt = 1; p->force_strand = 0;
    p->single_exon_ovlp_frac = 0.80;
}

static FILE *open_out(const char *fn)
{
    FILE *f = fopen(fn, "w");
    if (!f) { char m[1200]; snprintf(m, sizeof m, "cannot write \"%s\"", fn); orc_die(0, "update_gtf", m); }
    return f;
}

static int cmd_update_gtf(int argc, char **argv)
{
    /* update_gtf.c:995-1117; option table :967-993 and optstring :999 kept as
     * they are, including "M:" taking an argument and --source mapping to 's' (Q13) */
    static const struct option lopt[] = {
        {"input-mode", 1, 0, 'm'}, {"bam", 1, 0, 'b'}, {"sj", 1, 0, 'j'}, {"force-strand", 0, 0, 'c'},
        {"min-exon", 1, 0, 'e'}, {"min-intron", 1, 0, 'i'}, {"distance", 1, 0, 'd'}, {"DISTANCE", 1, 0, 'D'},
        {"frac", 1, 0, 'f'}, {"full-gtf", 1, 0, 'l'}, {"use-multi", 0, 0, 'M'}, {"min_sj_cnt", 1, 0, 'J'},
        {"output", 1, 0, 'o'}, {"bam-gtf", 1, 0, 'a'}, {"known-gtf", 1, 0, 'k'}, {"novel-gtf", 1, 0, 'v'},
        {"unrecog", 1, 0, 'u'}, {"source", 1, 0, 's'}, {0, 0, 0, 0}};
    orc_params p; default_params(&p);
    int mode = 0, c; const char *hdr_sam = NULL; char source[1024] = "lr2rmats";
    FILE *sj_fp = NULL, *out = stdout, *bed = NULL, *rgtf = NULL, *detail = NULL, *kf = NULL, *vf = NULL, *uf = NULL, *sum = NULL;
    optind = 1;
    while ((c = getopt_long(argc, argv, "m:b:j:J:M:e:i:t:sd:D:f:cl:o:nE:a:A:k:v:u:y:S:", lopt, NULL)) >= 0) {
        switch (c) {
        case 'm': if (optarg[0] == 'b') mode = 0; else if (optarg[0] == 'g') mode = 1; else return 1; break;
        case 'b': hdr_sam = optarg; break;
        case 'j': sj_fp = fopen(optarg, "r"); if (!sj_fp) orc_die(0, "update_gtf", "Can not open splice-junction file"); break;
        case 'e': p.min_exon = atoi(optarg); break;
        case 'i': p.min_intron = atoi(optarg); break;
        case 't': p.max_delet = atoi(optarg); break;
        case 'd': p.ss_dis = atoi(optarg); break;
        case 'D': p.end_dis = atoi(optarg); break;
        case 'f': p.single_exon_ovlp_frac = atof(optarg); break;
        case 'c': p.force_strand = 1; break;
        case 's': p.split_trans = 1; break;
        case 'l': p.full_level = atoi(optarg); break;
        case 'M': p.use_multi = 1; break;
        case 'J': p.min_sj_cnt = atoi(optarg); break;
        case 'o': out = open_out(optarg); break;
        case 'n': break;
        case 'E': bed = open_out(optarg); break;
        case 'a': rgtf = open_out(optarg); break;
        case 'A': detail = open_out(optarg); break;
        case 'k': kf = open_out(optarg); break;
        case 'v': vf = open_out(optarg); break;
        case 'u': uf = open_out(optarg); break;
        case 'y': sum = open_out(optarg); break;
        case 'S': strncpy(source, optarg, sizeof source - 1); break;
        default: fprintf(stderr, "Error: unknown option: %s.\n", optarg); return 1;
        }
    }
    if (argc - optind != 2) { fprintf(stderr, "Usage:   lr2rmats update-gtf [option] <in.bam/in.gtf> <old.gtf> > new.gtf\n"); return 1; }

    orc_chroms *chr = chroms_new();
    orc_list *A = ls_new(), *R = ls_new(), *U = ls_new(), *K = ls_new(), *N = ls_new(), *X = ls_new();
    if (mode == 0) load_sam_reads(argv[optind], chr, R, &p, "update_gtf");
    else {
        if (!hdr_sam) orc_die(0, "update_gtf", "Couldn't read header of provided BAM file.\n");
        sam_in *s = sam_open_text(hdr_sam, chr, "update_gtf"); sam_close_text(s);
        load_gtf(argv[optind], chr, R, 1);
    }
    load_gtf(argv[optind + 1], chr, A, 0);
    orc_sj *S = NULL; int n_sj = load_sj(sj_fp, chr, &S);

    orc_check_all(R, A, S, n_sj, U, K, N, X, &p);

    write_trans_list(U, chr, source, out);
    if (rgtf) write_trans_list(R, chr, source, rgtf);
    if (detail) write_detail(R, chr, detail);
    if (kf) write_trans_list(K, chr, source, kf);
    if (vf) write_trans_list(N, chr, source, vf);
    if (uf) write_trans_list(X, chr, source, uf);
    if (sum || bed) write_summary(chr, A, U, R, &p, sum, bed);

    ls_free(A); ls_free(R); ls_free(U); ls_free(K); ls_free(N); ls_free(X); free(S); chroms_free(chr);
    if (out != stdout) fclose(out); else fflush(stdout);
    if (sj_fp) fclose(sj_fp);
    if (bed) fclose(bed);
    if (rgtf) fclose(rgtf);
    if (detail) fclose(detail);
    if (kf) fclose(kf);
    if (vf) fclose(vf);
    if (uf) fclose(uf);
    if (sum) fclose(sum);
    return 0;
}

static int cmd_bam2gtf(int argc, char **argv)
{
    /* bam2gtf.c:112-161; output format gtf.c:597-604 print_trans */
    static const struct option lopt[] = {{"exon-min", 1, 0, 'e'}, {"intron-len", 1, 0, 'i'}, {"source", 1, 0, 's'}, {0, 0, 0, 0}};
    orc_params p; default_params(&p);
    char source[100] = "lr2rmats"; int c, i;
    optind = 1;
    while ((c = getopt_long(argc, argv, "s:e:i:t:", lopt, NULL)) >= 0) {
        switch (c) {
        case 'e': p.min_exon = atoi(optarg); break;
        case 'i': p.min_intron = atoi(optarg); break;
        case 't': p.max_delet = atoi(optarg); break;
        case 's': strncpy(source, optarg, sizeof source - 1); break;
        default: fprintf(stderr, "Error: unknown option: %s.\n", optarg); return 1;
        }
    }
    if (argc - optind != 1) { fprintf(stderr, "Usage:   lr2rmats bam2gtf [option] <in.bam> > out.gtf\n"); return 1; }
    orc_chroms *chr = chroms_new();
    sam_in *s = sam_open_text(argv[optind], chr, "bam2gtf");
    sam_rec rec; memset(&rec, 0, sizeof rec);
    orc_trans t; tr_zero(&t);
    while (sam_next(s, chr, &rec) == 0) {
        if (rec.flag & 4) continue;                        /* bam2gtf.c:82,151 */
        orc_cigar_to_exons(&t, rec.tid, rec.pos0, sam_strand(&rec), rec.cig, rec.n_cig, p.min_exon, p.min_intron, p.max_delet);
        tr_finish(&t);
        printf("%s\t%s\ttranscript\t%d\t%d\t.\t%c\t.\tgene_id \"%s\"; transcript_id \"%s\";\n", chr->name[t.tid], source, t.start, t.end, "+-"[t.rev], rec.qname, rec.qname);
        for (i = 0; i < t.n; ++i)
            printf("%s\t%s\texon\t%d\t%d\t.\t%c\t.\tgene_id \"%s\"; transcript_id \"%s\";\n", chr->name[t.tid], source, t.ex[i].start, t.ex[i].end, "+-"[t.ex[i].rev], rec.qname, rec.qname);
    }
    tr_release(&t); free(rec.cig); sam_close_text(s); chroms_free(chr);
    fflush(stdout);
    return 0;
}

static int cmd_unique_gtf(int argc, char **argv)
{
    /* unique_gtf.c:53-158 (optstring :90 has no 't' entry although :109 handles it) */
    static const struct option lopt[] = {
        {"input-mode", 1, 0, 'm'}, {"bam", 1, 0, 'b'}, {"force-strand", 0, 0, 's'}, {"min-exon", 1, 0, 'e'},
        {"min-intron", 1, 0, 'i'}, {"distance", 1, 0, 'd'}, {"DISTANCE", 1, 0, 'D'}, {"frac", 1, 0, 'f'},
        {"intersect", 0, 0, 'I'}, {"output", 1, 0, 'o'}, {"source", 1, 0, 's'}, {0, 0, 0, 0}};
    orc_params p; default_params(&p);
    int mode = 0, c, i, intersect = 0; const char *hdr_sam = NULL; char source[1024] = "lr2rmats"; FILE *out = stdout;
    optind = 1;
    while ((c = getopt_long(argc, argv, "m:b:se:i:Id:D:f:o:S:", lopt, NULL)) >= 0) {
        switch (c) {
        case 'm': if (optarg[0] == 'b') mode = 0; else if (optarg[0] == 'g') mode = 1; else return 1; break;
        case 'b': hdr_sam = optarg; break;
        case 's': p.force_strand = 1; break;
        case 'e': p.min_exon = atoi(optarg); break;
        case 'i': p.min_intron = atoi(optarg); break;
        case 'd': p.ss_dis = atoi(optarg); break;
        case 'D': p.end_dis = atoi(optarg); break;
        case 'f': p.single_exon_ovlp_frac = atof(optarg); break;
        case 'I': intersect = 1; break;
        case 'o': out = open_out(optarg); break;
        case 'S': strncpy(source, optarg, sizeof source - 1); break;
        default: fprintf(stderr, "Error: unknown option: %s.\n", optarg); return 1;
        }
    }
    if (argc - optind != 1) { fprintf(stderr, "Usage:   lr2rmats unique-gtf [option] <in.sorted.bam/in.sorted.gtf> > unique.gtf\n"); return 1; }
    orc_chroms *chr = chroms_new();
    orc_list *R = ls_new(), *U = ls_new(), *Sh = ls_new();
    if (mode == 0) load_sam_reads(argv[optind], chr, R, &p, "unique_gtf");
    else {
        if (!hdr_sam) orc_die(0, "unique_gtf", "Couldn't read header of provided BAM file.\n");
        sam_in *s = sam_open_text(hdr_sam, chr, "unique_gtf"); sam_close_text(s);
        load_gtf(argv[optind], chr, R, 1);
    }
    for (i = 0; i < R->n; ++i) {               /* unique_gtf.c:73-84 uniq_trans */
        if (!orc_merge(&R->t[i], U, &p)) ls_push_read(U, &R->t[i]);
        else ls_push_read(Sh, &R->t[i]);
    }
    write_trans_list(intersect ? Sh : U, chr, source, out);
    ls_free(R); ls_free(U); ls_free(Sh); chroms_free(chr);
    if (out != stdout) fclose(out); else fflush(stdout);
    return 0;
}

int orc_main(int argc, char **argv)
{
    /* main.c:37-49 dispatch */
    if (argc < 1) return 1;
    if (strcmp(argv[0], "update-gtf") == 0) return cmd_update_gtf(argc, argv);
    if (strcmp(argv[0], "bam2gtf") == 0) return cmd_bam2gtf(argc, argv);
    if (strcmp(argv[0], "unique-gtf") == 0) return cmd_unique_gtf(argc, argv);
    fprintf(stderr, "[main] unrecognized command '%s'\n", argv[0]);
    return 1;
}
