/* cmds.c -- sub-command drivers with the reference's option tables.
 *
 *   update-gtf  src/update_gtf.c:967-1117  (optstring :999, long options :967-993 -- kept verbatim,
 *               including "M:" consuming an argument and --source mapping to 's': Q13)
 *   bam2gtf     src/bam2gtf.c:112-161
 *   unique-gtf  src/unique_gtf.c:53-158
 *   dispatch    src/main.c:37-49
 *
 * The per-read work goes through the C-ABI engine (include/lr2rmats_hip.h); there is no other
 * implementation of it in this program.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "l2r_host.h"

static const char PROG[] = "lr2rmats";

struct h_job {
    h_update_opts o;
    h_chroms chr;
    h_reads reads;
    h_gtf anno;
    h_sj sj;
    FILE *sj_fp;
    int mode;
    l2r_params engine_prm;       /* what the engine gets: o.prm, except that `-m g` input passes its exons through unchanged */
    int sharded; int64_t shard_lo, shard_hi, shard_total;       /* h_job_open_rank: only records [lo, hi) of `total` are loaded */
    int64_t shard_blocks[8];                                     /* sharded == 2: the rank's block range (h_read_alignments_blocks); lo / hi / total are dist.py's to work out */
    char *out_path[8];           /* 0 updated gtf (NULL = stdout), 1 exon bed, 2 bam gtf, 3 detail, 4 known, 5 novel, 6 unrecog, 7 summary */
    h_part_genes part_genes;     /* of the part h_job_finish_part last ran */
};

static int g_open_outputs = 1;

/* L2R_TIMING=1: wall-clock per stage on stderr (diagnostics; the reference prints nothing comparable) */
#include <time.h>
#include <stdint.h>
static double h_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static double g_t_last = 0.0;
void h_stage_time(const char *what)
{
    if (!getenv("L2R_TIMING")) return;
    const double t = h_now();
    if (g_t_last > 0.0) fprintf(stderr, "[timing] %-28s %9.3f s\n", what, t - g_t_last);
    g_t_last = t;
}
static FILE *open_w(const char *fn)
{
    if (!g_open_outputs) return NULL;
    FILE *f = fopen(fn, "w");
    if (!f) h_fatal("update_gtf", "Can not open \"%s\" for writing\n", fn);
    return f;
}

static int update_usage(void)
{
    /* src/update_gtf.c:37-77 */
    fprintf(stderr, "\nUsage:   %s update-gtf [option] <in.bam/in.gtf> <old.gtf> > new.gtf\n\n", PROG);
    fprintf(stderr, "Notice:  the BAM and GTF files should be sorted in advance.\n\n");
    fprintf(stderr, "Input options:\n\n");
    fprintf(stderr, "         -m --input-mode   [STR]    format of input file <in.bam/in.gtf>, BAM file(b) or GTF file(g). [b]\n");
    fprintf(stderr, "         -b --bam          [STR]    for GTF input <in.gtf>, BAM file is needed to obtain BAM header information. [NULL]\n");
    fprintf(stderr, "         -j --sj           [STR]    junction information file output by STAR(*.out.tab). [NULL]\n\n");
    fprintf(stderr, "Function options:\n\n");
    fprintf(stderr, "         -c --force-strand         force to match strand when merging transcripts. [False]\n");
    fprintf(stderr, "         -e --min-exon     [INT]    minimum length of internal exon. [3]\n");
    fprintf(stderr, "         -i --min-intron   [INT]    minimum length of intron. [3]\n");
    fprintf(stderr, "         -t --max-delet    [INT]    maximum length of deletion, longer deletion will be considered as intron. [50]\n");
    fprintf(stderr, "         -d --distance     [INT]    consider same if distance between two splice site is not bigger than d. [0]\n");
    fprintf(stderr, "         -D --DISTANCE     [INT]    consider same if distance between two start/end site is not bigger than D. [%d]\n", 0x7fffffff);
    fprintf(stderr, "         -f --frac         [INT]    consider same if overlapping between two single-exon transcript is bigger than f. [0.80]\n");
    fprintf(stderr, "         -s --split-trans           split read on unreliable junctions. [False]\n");
    fprintf(stderr, "         -M --use-multi             use junction information of multi-mapped read. [False]\n");
    fprintf(stderr, "         -J --min-junc-cnt [INT]    minimum short-read junction count of novel junction. [1]\n");
    fprintf(stderr, "         -l --full-length  [INT]    level of strict criterion for considering full-length transcript. \n");
    fprintf(stderr, "                                    (1->5, most strict->most relaxed) [5]\n\n");
    fprintf(stderr, "Output options:\n\n");
    fprintf(stderr, "         -o --output       [STR]    updated GTF file. [stdout]\n");
    fprintf(stderr, "         -n --min-output            only keep the minimal set of novel transcripts in the updated GTF file. [False]\n");
    fprintf(stderr, "         -E --exon-bed     [STR]    updated novel exon file in bed format. [NULL]\n");
    fprintf(stderr, "         -a --bam-gtf      [STR]    bam-derived transcript GTF file. [NULL]\n");
    fprintf(stderr, "         -A --bam-detial   [STR]    detailed information of each bam-derived transcript. [NULL]\n");
    fprintf(stderr, "         -k --known-gtf    [STR]    bam-derived known transcript GTF file. [NULL]\n");
    fprintf(stderr, "         -v --novel-gtf    [STR]    bam-derived novel transcript GTF file. [NULL]\n");
    fprintf(stderr, "         -u --unrecog      [STR]    bam-derived unrecognized transcript GTF file. [NULL]\n");
    fprintf(stderr, "         -y --summary      [STR]    Staticstic summary of bam-derived transcript. [NULL]\n");
    fprintf(stderr, "         -S --source       [STR]    'source' field in GTF: program, database or project name. [%s]\n\n", PROG);
    return 1;
}

static void default_params(l2r_params *p)
{
    /* src/update_gtf.c:24-35, src/gtf.h:118-127 */
    p->min_exon = 3; p->min_intron = 3; p->max_delet = 50; p->ss_dis = 0; p->end_dis = 0x7fffffff; p->full_level = 5;
    p->split_trans = 0; p->use_multi = 0; p->min_sj_cnt = 1; p->force_strand = 0; p->single_exon_ovlp_frac = 0.80;
}

/* `-m g` (src/update_gtf.c:1071-1075, read_gtf_trans src/gtf.c:524-595): the transcripts of a GTF take the place of the
 * alignment records.  The engine consumes CIGARs, so every transcript becomes <exon length>M <gap>N ...; the engine is
 * then run with thresholds that keep every exon and cut at every N (h_job_open), i.e. the exons arrive unchanged.
 * trans_name -> QNAME, trans_id -> tid_name; gene names come from the annotation, as for alignments (:832-833). */
static void reads_from_gtf(const char *fn, const h_chroms *chr, h_reads *out)
{
    h_gtf g;
    h_read_gtf(fn, chr, &g, 1);
    memset(out, 0, sizeof *out);
    const int64_t T = g.n_tx;
    out->n = T; out->cap = T + 1;
    out->tid = (int32_t *)h_malloc((size_t)(T + 1) * 4); out->pos = (int32_t *)h_malloc((size_t)(T + 1) * 4);
    out->rev = (uint8_t *)h_malloc((size_t)T + 1); out->cig_off = (int64_t *)h_malloc((size_t)(T + 2) * 8);
    out->qname = (uint32_t *)h_malloc((size_t)(T + 1) * 4); out->tid_name = (uint32_t *)h_malloc((size_t)(T + 1) * 4);
    out->cap_cig = 2 * g.n_ex + 1; out->cig = (uint32_t *)h_malloc((size_t)out->cap_cig * 4);
    out->cig_off[0] = 0;
    for (int64_t i = 0; i < T; ++i) {
        if (g.tid[i] < 0)
            h_fatal("read_gtf_trans", "transcript \"%s\" is on a chromosome that is not in the BAM header (the reference indexes its name table with -1 there)",
                    h_str(&g.names, g.tids[i]));
        out->tid[i] = g.tid[i]; out->rev[i] = g.rev[i];
        out->qname[i] = g.tname[i]; out->tid_name[i] = g.tids[i];
        const int64_t lo = g.ex_off[i], hi = g.ex_off[i + 1];
        out->pos[i] = g.ex_start[lo] - 1;
        for (int64_t k = lo; k < hi; ++k) {
            const int64_t len = (int64_t)g.ex_end[k] - g.ex_start[k] + 1;
            if (k > lo) {
                const int64_t gap = (int64_t)g.ex_start[k] - g.ex_end[k - 1] - 1;
                if (gap < 0 || gap >= (1 << 28)) h_fatal("read_gtf_trans", "transcript \"%s\": exons overlap or lie more than 2^28 bp apart", h_str(&g.names, g.tids[i]));
                out->cig[out->n_cig++] = ((uint32_t)gap << 4) | 3u;
            }
            if (len < 1 || len >= (1 << 28)) h_fatal("read_gtf_trans", "transcript \"%s\": exon of length %lld", h_str(&g.names, g.tids[i]), (long long)len);
            out->cig[out->n_cig++] = ((uint32_t)len << 4) | 0u;
        }
        out->cig_off[i + 1] = out->n_cig;
    }
    out->names = g.names; memset(&g.names, 0, sizeof g.names);      /* the string table moves to the reads */
    h_gtf_free(&g);
}

static int g_rank = 0, g_world = 1;
static int g_want_early_engine = 0;        /* set by h_cmd_update_gtf around its h_job_open: see early_engine_start() */
static void early_engine_start(void);
static void early_engine_annotation(const l2r_annotation *a);
typedef struct { const char *fn; const h_chroms *chr; h_gtf *out; } gtf_thread_arg;
static void *gtf_thread_main(void *p)
{
    gtf_thread_arg *a = (gtf_thread_arg *)p;
    fprintf(stderr, "[read_anno_trans] reading transcript annotation from %s ...\n", a->fn);
    h_read_gtf(a->fn, a->chr, a->out, 0);
    fprintf(stderr, "[read_anno_trans] reading transcript annotation from %s done.\n", a->fn);
    if (g_want_early_engine) {         /* (the command's own engine: its tables are built beside the record reader) */
        l2r_annotation v;
        v.n_tx = a->out->n_tx; v.n_exon = a->out->n_ex; v.tx_tid = a->out->tid; v.tx_start = a->out->start; v.tx_end = a->out->end;
        v.tx_rev = a->out->rev; v.tx_ex_off = a->out->ex_off; v.ex_start = a->out->ex_start; v.ex_end = a->out->ex_end;
        early_engine_annotation(&v);
    }
    return NULL;
}

h_job *h_job_open_rank(int argc, char **argv, int *exit_code, int open_outputs, int rank, int world)
{
    g_rank = rank; g_world = world;
    h_job *j = h_job_open2(argc, argv, exit_code, open_outputs);
    g_rank = 0; g_world = 1;
    return j;
}
int h_job_shard(const h_job *j, int64_t *lo, int64_t *hi, int64_t *n_total)
{
    *lo = j->sharded == 1 ? j->shard_lo : 0; *hi = j->sharded == 1 ? j->shard_hi : j->reads.n; *n_total = j->sharded == 1 ? j->shard_total : j->reads.n;
    return j->sharded;
}
void h_job_shard_blocks(const h_job *j, int64_t info[8])
{
    memcpy(info, j->shard_blocks, sizeof j->shard_blocks);
}

h_job *h_job_open2(int argc, char **argv, int *exit_code, int open_outputs)
{
    /* ranks other than the writer of a multi-process run parse the same command line without
     * creating (truncating) the output files */
    g_open_outputs = open_outputs;
    h_job *j = h_job_open(argc, argv, exit_code);
    g_open_outputs = 1;
    if (j && !open_outputs) j->o.out_gtf = NULL;
    return j;
}

h_job *h_job_open(int argc, char **argv, int *exit_code)
{
    static const struct option lopt[] = {
        {"input-mode", 1, 0, 'm'}, {"bam", 1, 0, 'b'}, {"sj", 1, 0, 'j'}, {"force-strand", 0, 0, 'c'},
        {"min-exon", 1, 0, 'e'}, {"min-intron", 1, 0, 'i'}, {"distance", 1, 0, 'd'}, {"DISTANCE", 1, 0, 'D'},
        {"frac", 1, 0, 'f'}, {"full-gtf", 1, 0, 'l'}, {"use-multi", 0, 0, 'M'}, {"min_sj_cnt", 1, 0, 'J'},
        {"output", 1, 0, 'o'}, {"bam-gtf", 1, 0, 'a'}, {"known-gtf", 1, 0, 'k'}, {"novel-gtf", 1, 0, 'v'},
        {"unrecog", 1, 0, 'u'}, {"source", 1, 0, 's'}, {0, 0, 0, 0}};
    h_job *j = (h_job *)calloc(1, sizeof *j);
    default_params(&j->o.prm);
    j->o.out_gtf = stdout; strcpy(j->o.source, PROG);
    const char *hdr_file = NULL;
    int c;
    *exit_code = 0;
    optind = 1;
    while ((c = getopt_long(argc, argv, "m:b:j:J:M:e:i:t:sd:D:f:cl:o:nE:a:A:k:v:u:y:S:", lopt, NULL)) >= 0) {
        switch (c) {
        case 'm': if (optarg[0] == 'b') j->mode = 0; else if (optarg[0] == 'g') j->mode = 1; else { *exit_code = update_usage(); free(j); return NULL; } break;
        case 'b': hdr_file = optarg; { FILE *t = fopen(optarg, "rb"); if (!t) h_fatal("update_gtf", "Cannot open \"%s\"\n", optarg); fclose(t); } break;
        case 'j': if (!(j->sj_fp = fopen(optarg, "r"))) h_fatal("update_gtf", "Can not open splice-junction file \"%s\"\n", optarg); break;
        case 'e': j->o.prm.min_exon = atoi(optarg); break;
        case 'i': j->o.prm.min_intron = atoi(optarg); break;
        case 't': j->o.prm.max_delet = atoi(optarg); break;
        case 'd': j->o.prm.ss_dis = atoi(optarg); break;
        case 'D': j->o.prm.end_dis = atoi(optarg); break;
        case 'f': j->o.prm.single_exon_ovlp_frac = atof(optarg); break;
        case 'c': j->o.prm.force_strand = 1; break;
        case 's': j->o.prm.split_trans = 1; break;
        case 'l': j->o.prm.full_level = atoi(optarg); break;
        case 'M': j->o.prm.use_multi = 1; break;
        case 'J': j->o.prm.min_sj_cnt = atoi(optarg); break;
        case 'o': j->o.out_gtf = open_w(optarg); free(j->out_path[0]); j->out_path[0] = strdup(optarg); break;
        case 'n': break;                                   /* accepted, unused (src/update_gtf.c:1031,1084) */
        case 'E': j->o.exon_bed = open_w(optarg); free(j->out_path[1]); j->out_path[1] = strdup(optarg); break;
        case 'a': j->o.bam_gtf = open_w(optarg); free(j->out_path[2]); j->out_path[2] = strdup(optarg); break;
        case 'A': j->o.bam_detail = open_w(optarg); free(j->out_path[3]); j->out_path[3] = strdup(optarg); break;
        case 'k': j->o.known_gtf = open_w(optarg); free(j->out_path[4]); j->out_path[4] = strdup(optarg); break;
        case 'v': j->o.novel_gtf = open_w(optarg); free(j->out_path[5]); j->out_path[5] = strdup(optarg); break;
        case 'u': j->o.unrecog_gtf = open_w(optarg); free(j->out_path[6]); j->out_path[6] = strdup(optarg); break;
        case 'y': j->o.summary = open_w(optarg); free(j->out_path[7]); j->out_path[7] = strdup(optarg); break;
        case 'S': strncpy(j->o.source, optarg, sizeof j->o.source - 1); break;
        default: fprintf(stderr, "Error: unknown option: %s.\n", optarg); *exit_code = update_usage(); free(j); return NULL;
        }
    }
    if (argc - optind != 2) { *exit_code = update_usage(); free(j); return NULL; }

    h_stage_time("start");
    int anno_read = 0;
    if (g_want_early_engine) early_engine_start();
    if (j->mode == 0) {
        /* a rank of a multi-process run loads its shard only -- unless the run needs the gathered route (split pieces are compared
         * across chromosomes, Q2) or the caller forces it */
        const char *fg = getenv("L2R_DIST_GATHER");
        if (g_world > 1 && !(j->o.prm.split_trans && j->sj_fp) && !(fg && fg[0] == '1')) {
            /* first choice: only the BGZF blocks of this rank's records are inflated (L2R_DIST_BLOCKS=0: every rank inflates the file
             * and cuts it by the records' weights -- what dist.py falls back to when the ranks' block ranges do not meet) */
            const char *fb = getenv("L2R_DIST_BLOCKS");
            if (!(fb && fb[0] == '0') && h_read_alignments_blocks(argv[optind], &j->chr, &j->reads, 0, "update_gtf", g_rank, g_world, j->shard_blocks)) j->sharded = 2;
            else {
                if (j->reads.cig_off) h_reads_free(&j->reads);       /* (the header's names are interned again: same ids) */
                j->sharded = h_read_alignments_shard(argv[optind], &j->chr, &j->reads, 0, "update_gtf", g_rank, g_world, &j->shard_lo, &j->shard_hi, &j->shard_total);
            }
        } else {
            /* one process: the annotation is parsed on a thread of its own while the records are read -- it only LOOKS UP the header's
             * chromosome names (h_read_gtf takes the table const), which are complete once the header has been read, and the record
             * reader interns nothing behind the header (L2R_PARALLEL_READ=0: one after the other, as the reference does it) */
            const char *pe = getenv("L2R_PARALLEL_READ");
            pthread_t gt; int par = 0;
            gtf_thread_arg ga = { argv[optind + 1], &j->chr, &j->anno };
            if (!(pe && pe[0] == '0')) {
                h_read_header_only(argv[optind], &j->chr, "update_gtf");
                par = pthread_create(&gt, NULL, gtf_thread_main, &ga) == 0;
            }
            h_read_alignments(argv[optind], &j->chr, &j->reads, 0, "update_gtf");
            if (par) { pthread_join(gt, NULL); anno_read = 1; }
        }
        h_stage_time("read alignments");
    } else {
        if (!hdr_file) h_fatal("update_gtf", "Couldn't read header of provided BAM file.\n");
        h_read_header_only(hdr_file, &j->chr, "update_gtf");
        reads_from_gtf(argv[optind], &j->chr, &j->reads);
        h_stage_time("read GTF as reads");
    }
    j->engine_prm = j->o.prm;
    if (j->mode == 1) { j->engine_prm.min_exon = INT32_MIN; j->engine_prm.min_intron = 0; j->engine_prm.max_delet = INT32_MAX; }
    if (!anno_read) {
        fprintf(stderr, "[read_anno_trans] reading transcript annotation from %s ...\n", argv[optind + 1]);
        h_read_gtf(argv[optind + 1], &j->chr, &j->anno, 0);
        fprintf(stderr, "[read_anno_trans] reading transcript annotation from %s done.\n", argv[optind + 1]);
    }
    h_stage_time("read annotation");
    h_read_sj(j->sj_fp, &j->chr, &j->sj);
    return j;
}

void h_job_views(h_job *j, l2r_params *prm, l2r_annotation *a, l2r_junctions *s, l2r_reads *r)
{
    *prm = j->engine_prm;
    a->n_tx = j->anno.n_tx; a->n_exon = j->anno.n_ex; a->tx_tid = j->anno.tid; a->tx_start = j->anno.start; a->tx_end = j->anno.end;
    a->tx_rev = j->anno.rev; a->tx_ex_off = j->anno.ex_off; a->ex_start = j->anno.ex_start; a->ex_end = j->anno.ex_end;
    s->n = j->sj.n; s->tid = j->sj.tid; s->don = j->sj.don; s->acc = j->sj.acc; s->uniq_c = j->sj.uniq; s->multi_c = j->sj.multi;
    r->n_reads = j->reads.n; r->n_cigar = j->reads.n_cig; r->tid = j->reads.tid; r->pos = j->reads.pos; r->rev = j->reads.rev;
    r->cig_off = j->reads.cig_off; r->cig = j->reads.cig; r->first_read_index = 0; r->cig_summary = j->reads.cig_sum;
}

static int tail_threads(const h_job *j, const h_reads *reads);
static int finish_threaded(h_job *j, const h_reads *reads, const h_update_opts *files, const l2r_result *res, int n_thr, int first_part,
                           int64_t *counters_out, h_part_genes *genes_out);

int h_job_finish(h_job *j, const l2r_result *res)
{
    h_result hr;
    hr.n = res->n_reads; hr.n_ex = res->n_exons; hr.ex_off = res->ex_off; hr.ex_start = res->ex_start; hr.ex_end = res->ex_end;
    hr.ex_flag = res->ex_flag; hr.info = res->info; hr.ref_tx = res->ref_tx;
    if (hr.n != j->reads.n) h_fatal("update_gtf", "result covers %lld reads, input has %lld", (long long)hr.n, (long long)j->reads.n);
    const int n_thr = tail_threads(j, &j->reads);
    h_stage_time("  tail: input checks");
    if (n_thr > 1) finish_threaded(j, &j->reads, &j->o, res, n_thr, 1, NULL, NULL);
    else h_update_tail(&j->o, &j->chr, &j->reads, &j->anno, &hr, j->sj.n);
    h_stage_time("  tail: parts + writers");
    FILE **fs[] = {&j->o.exon_bed, &j->o.bam_gtf, &j->o.bam_detail, &j->o.known_gtf, &j->o.novel_gtf, &j->o.unrecog_gtf, &j->o.summary};
    for (size_t k = 0; k < sizeof fs / sizeof fs[0]; ++k) if (*fs[k]) { fclose(*fs[k]); *fs[k] = NULL; }
    if (j->o.out_gtf && j->o.out_gtf != stdout) { fclose(j->o.out_gtf); j->o.out_gtf = NULL; } else fflush(stdout);
    h_stage_time("  tail: files closed");
    return 0;
}

void h_job_open_outputs(h_job *j)
{
    /* a job opened without output files (h_job_open2(..., 0)) that turns out to be the writer */
    FILE **fs[8] = {&j->o.out_gtf, &j->o.exon_bed, &j->o.bam_gtf, &j->o.bam_detail, &j->o.known_gtf, &j->o.novel_gtf, &j->o.unrecog_gtf, &j->o.summary};
    for (int k = 0; k < 8; ++k) if (j->out_path[k] && (!*fs[k] || *fs[k] == stdout)) {
        *fs[k] = fopen(j->out_path[k], "w");
        if (!*fs[k]) h_fatal("update_gtf", "Can not open \"%s\" for writing\n", j->out_path[k]);
    }
    if (!j->o.out_gtf) j->o.out_gtf = stdout;
}

void h_job_set_out_path(h_job *j, int which, const char *path)
{
    if (which < 0 || which >= 8) return;
    free(j->out_path[which]); j->out_path[which] = path ? strdup(path) : NULL;
}

const char *h_job_out_path(const h_job *j, int which) { return (j && which >= 0 && which < 8) ? j->out_path[which] : NULL; }

static FILE *open_part(const char *path, const char *suffix)
{
    if (!path) return NULL;
    char *fn = (char *)h_malloc(strlen(path) + strlen(suffix) + 1);
    strcpy(fn, path); strcat(fn, suffix);
    FILE *f = fopen(fn, "w");
    if (!f) h_fatal("update_gtf", "Can not open \"%s\" for writing\n", fn);
    free(fn);
    return f;
}

int h_job_finish_part(h_job *j, int64_t lo, int64_t hi, const l2r_result *res, const char *suffix, const char *stdout_base,
                      int first_part, int64_t counters[H_N_SUMMARY])
{
    if (lo < 0 || hi < lo || hi > j->reads.n) h_fatal("update_gtf", "bad read range [%lld, %lld)", (long long)lo, (long long)hi);
    if (res->n_reads != hi - lo) h_fatal("update_gtf", "result covers %lld reads, the part has %lld", (long long)res->n_reads, (long long)(hi - lo));
    h_update_opts o = j->o;
    o.out_gtf = open_part(j->out_path[0] ? j->out_path[0] : stdout_base, suffix);
    o.exon_bed = open_part(j->out_path[1], suffix); o.bam_gtf = open_part(j->out_path[2], suffix); o.bam_detail = open_part(j->out_path[3], suffix);
    o.known_gtf = open_part(j->out_path[4], suffix); o.novel_gtf = open_part(j->out_path[5], suffix); o.unrecog_gtf = open_part(j->out_path[6], suffix);
    o.summary = NULL; o.summary_counts = counters; o.no_detail_header = !first_part;
    h_part_genes_free(&j->part_genes); o.part_genes = &j->part_genes;
    if (!o.out_gtf) h_fatal("update_gtf", "a partitioned run needs -o or a base path for the updated GTF");
    h_reads part = j->reads;                                   /* a view: per-read arrays shifted, string table shared */
    part.n = hi - lo; part.tid += lo; part.pos += lo; part.rev += lo; part.qname += lo; part.cig_off += lo;
    if (part.tid_name) part.tid_name += lo;
    h_result hr;
    hr.n = res->n_reads; hr.n_ex = res->n_exons; hr.ex_off = res->ex_off; hr.ex_start = res->ex_start; hr.ex_end = res->ex_end;
    hr.ex_flag = res->ex_flag; hr.info = res->info; hr.ref_tx = res->ref_tx;
    memset(counters, 0, H_N_SUMMARY * sizeof counters[0]);
    const int n_thr = tail_threads(j, &part);
    if (n_thr > 1) { o.summary_counts = NULL; o.part_genes = NULL; finish_threaded(j, &part, &o, res, n_thr, first_part, counters, &j->part_genes); }
    else h_update_tail(&o, &j->chr, &part, &j->anno, &hr, j->sj.n);
    FILE *fs[] = {o.out_gtf, o.exon_bed, o.bam_gtf, o.bam_detail, o.known_gtf, o.novel_gtf, o.unrecog_gtf};
    for (size_t k = 0; k < sizeof fs / sizeof fs[0]; ++k) if (fs[k]) fclose(fs[k]);
    return 0;
}

/* ---- the tail on several host threads ------------------------------------------------------------------
 * Same partition argument as h_job_finish_part: chromosome-aligned parts of the read array do not interact in the
 * order-dependent tail, so the parts run on their own threads into memory streams and are written out in order. */
#include <pthread.h>
#include <unistd.h>
typedef struct {
    h_job *j; const h_reads *reads; const h_update_opts *files; const l2r_result *res; int64_t lo, hi; int first;
    char *buf[7]; size_t len[7]; int64_t cnt[H_N_SUMMARY];
    h_part_genes genes;
    int done;
} tail_part;

static void *tail_part_main(void *arg)
{
    tail_part *t = (tail_part *)arg;
    h_job *j = t->j;
    h_update_opts o = *t->files;
    FILE *want[7] = {o.out_gtf, o.exon_bed, o.bam_gtf, o.bam_detail, o.known_gtf, o.novel_gtf, o.unrecog_gtf};
    FILE *fs[7];
    for (int k = 0; k < 7; ++k) {
        t->buf[k] = NULL; t->len[k] = 0;
        fs[k] = want[k] ? h_open_growbuf(&t->buf[k], &t->len[k]) : NULL;
        if (want[k] && !fs[k]) h_fatal("update_gtf", "fopencookie failed");
    }
    o.out_gtf = fs[0]; o.exon_bed = fs[1]; o.bam_gtf = fs[2]; o.bam_detail = fs[3]; o.known_gtf = fs[4]; o.novel_gtf = fs[5]; o.unrecog_gtf = fs[6];
    o.summary = NULL; o.summary_counts = t->cnt; o.no_detail_header = !t->first; o.part_genes = &t->genes;
    h_reads part = *t->reads;
    part.n = t->hi - t->lo; part.tid += t->lo; part.pos += t->lo; part.rev += t->lo; part.qname += t->lo; part.cig_off += t->lo;
    if (part.tid_name) part.tid_name += t->lo;
    const l2r_result *res = t->res;
    const int64_t x0 = res->ex_off[t->lo], x1 = res->ex_off[t->hi];
    int64_t *off = (int64_t *)h_malloc((size_t)(part.n + 1) * sizeof(int64_t));
    for (int64_t i = 0; i <= part.n; ++i) off[i] = res->ex_off[t->lo + i] - x0;
    h_result hr;
    hr.n = part.n; hr.n_ex = x1 - x0; hr.ex_off = off; hr.ex_start = res->ex_start + x0; hr.ex_end = res->ex_end + x0;
    hr.ex_flag = res->ex_flag + x0; hr.info = res->info + t->lo; hr.ref_tx = res->ref_tx + t->lo;
    memset(t->cnt, 0, sizeof t->cnt);
    h_update_tail(&o, &j->chr, &part, &j->anno, &hr, j->sj.n);
    for (int k = 0; k < 7; ++k) if (fs[k]) fclose(fs[k]);
    free(off);
    return NULL;
}

/* number of tail threads: L2R_THREADS, else the online CPUs (at most 64); 1 when the partition argument does not hold */
static int tail_threads(const h_job *j, const h_reads *reads)
{
    const char *e = getenv("L2R_THREADS");
    long n = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    if (n > 64) n = 64;
    if (n < 2 || (!e && reads->n < 20000)) return 1;                         /* small inputs: not worth the threads (unless asked for) */
    if (j->o.prm.split_trans && j->sj.n > 0) return 1;                      /* Q2: split pieces are compared across chromosomes */
    for (int64_t i = 1; i < reads->n; ++i) if (reads->tid[i] < reads->tid[i - 1]) return 1;     /* not grouped by chromosome */
    return (int)n;
}

/* The parts of the threaded tail.  The order-dependent lists of the tail look backwards from their end and stop at the first entry
 * that lies before the new one: merge_trans at `t->start > T->t[i].end` or a smaller tid (src/update_gtf.c:147), the novel exon /
 * site / junction lists at a smaller tid, comparing coordinates (:181-222).  So the reads can be cut wherever every read from there on
 * starts behind the END OF EVERY READ in front of the cut on its chromosome (a gap in the coverage; chromosome boundaries are such places): no list entry made
 * from the reads in front of the cut can stop short of, merge with or equal an entry made behind it.  Every part lies inside one
 * chromosome; their outputs are concatenated and their counters added.  The one list that compares across a cut is the gene list
 * (merge_gene, :181-189: equal gene_id anywhere on the chromosome, and with the list's last entry of a smaller tid): the parts report
 * the ids they added and the join below counts an id once, as the sequential list would (gene_fix). */
typedef struct {
    int32_t cur_tid; const char **cur; int n_cur, cap_cur;       /* ids of the sequential list's entries under its last tid */
    const char *lower;                                           /* id of its last entry with a smaller tid, NULL: none */
    const char *last;                                            /* id of its last entry, NULL: empty */
    const char **first; int n_first, first_done;                 /* ids of the entries under the list's FIRST tid (what a caller that joins whole shards needs) */
} gene_fix;

static void gene_fix_snapshot(gene_fix *f)
{
    if (f->first_done) return;
    f->first = (const char **)h_malloc((size_t)(f->n_cur ? f->n_cur : 1) * sizeof(char *));
    memcpy((void *)f->first, f->cur, (size_t)f->n_cur * sizeof(char *));
    f->n_first = f->n_cur; f->first_done = 1;
}

/* ids `gids[0 .. n)` (in the order the part added them, all under `tid`) join the sequential list: returns how many of them it would
 * NOT have taken (the part counted them, the total must not) */
static int gene_fix_join(gene_fix *f, int32_t tid, char **gids, int n)
{
    if (n <= 0) return 0;
    if (f->last != NULL && tid > f->cur_tid) gene_fix_snapshot(f);
    if (f->last == NULL || tid > f->cur_tid) { f->lower = f->last; f->n_cur = 0; f->cur_tid = tid; }
    int dup = 0;
    for (int k = 0; k < n; ++k) {
        int hit = f->lower && strcmp(f->lower, gids[k]) == 0;
        for (int q = f->n_cur - 1; !hit && q >= 0; --q) hit = strcmp(f->cur[q], gids[k]) == 0;
        if (hit) { ++dup; continue; }
        if (f->n_cur == f->cap_cur) { f->cap_cur = f->cap_cur ? f->cap_cur * 2 : 256; f->cur = (const char **)h_realloc((void *)f->cur, (size_t)f->cap_cur * sizeof(char *)); }
        f->cur[f->n_cur++] = gids[k]; f->last = gids[k];
    }
    return dup;
}

typedef struct { tail_part *parts; int n_parts; int next; pthread_mutex_t mu; } tail_queue;
/* one writer per output file: the parts' streams of that file in order, each as soon as its part is done (the files are different
 * inodes: the writers do not contend; pwrite of one file by many threads did, see below) */
typedef struct { tail_part *parts; int n_parts; int which; FILE *out; } tail_writer;
static void *tail_writer_main(void *arg)
{
    tail_writer *w = (tail_writer *)arg;
    for (int k = 0; k < w->n_parts; ++k) {
        while (!__atomic_load_n(&w->parts[k].done, __ATOMIC_ACQUIRE)) usleep(200);
        if (w->parts[k].len[w->which]) fwrite(w->parts[k].buf[w->which], 1, w->parts[k].len[w->which], w->out);
        free(w->parts[k].buf[w->which]); w->parts[k].buf[w->which] = NULL;
    }
    return NULL;
}
static void *tail_queue_main(void *arg)
{
    tail_queue *q = (tail_queue *)arg;
    for (;;) {
        pthread_mutex_lock(&q->mu);
        const int k = q->next < q->n_parts ? q->next++ : -1;
        pthread_mutex_unlock(&q->mu);
        if (k < 0) break;
        tail_part_main(&q->parts[k]);
        __atomic_store_n(&q->parts[k].done, 1, __ATOMIC_RELEASE);
    }
    return NULL;
}

static int finish_threaded(h_job *j, const h_reads *reads, const h_update_opts *files, const l2r_result *res, int n_thr, int first_part,
                           int64_t *counters_out, h_part_genes *genes_out)
{
    const int64_t N = reads->n;
    /* cut points: coverage gaps (see above), a part of about N / (4 n_thr) reads ends at the next one; always at a new chromosome */
    int64_t target = N / ((int64_t)n_thr * 4);
    if (target < 5000) target = 5000;
    {   const char *e = getenv("L2R_TAIL_PART_READS"); if (e && atoll(e) > 0) target = atoll(e); }      /* (tests force small parts) */
    int64_t *cut = NULL; int n_cut = 0, cap_cut = 0;
#define PUSH_CUT(v) do { if (n_cut == cap_cut) { cap_cut = cap_cut ? cap_cut * 2 : 256; cut = (int64_t *)h_realloc(cut, (size_t)cap_cut * sizeof(int64_t)); } cut[n_cut++] = (v); } while (0)
    PUSH_CUT(0);
    {
        /* a cut in front of read i is safe iff every read from i on (of its chromosome) starts behind the end of every read in front
         * of i: suffix minimum of the starts against the running maximum of the ends (records need not be sorted inside a chromosome) */
        int32_t *smin = (int32_t *)h_malloc((size_t)(N + 1) * sizeof(int32_t));
        for (int64_t i = N - 1; i >= 0; --i) {
            const int64_t x0 = res->ex_off[i], x1 = res->ex_off[i + 1];
            const int32_t start = x1 > x0 ? res->ex_start[x0] : reads->pos[i] + 1;
            smin[i] = (i + 1 < N && reads->tid[i + 1] == reads->tid[i] && smin[i + 1] < start) ? smin[i + 1] : start;
        }
        int32_t run_end = INT32_MIN;
        for (int64_t i = 0; i < N; ++i) {
            const int64_t x0 = res->ex_off[i], x1 = res->ex_off[i + 1];
            const int32_t end = x1 > x0 ? res->ex_end[x1 - 1] : reads->pos[i];
            if (i > 0) {
                const int new_chrom = reads->tid[i] != reads->tid[i - 1];
                if (new_chrom || (smin[i] > run_end && i - cut[n_cut - 1] >= target)) { PUSH_CUT(i); }
                if (new_chrom) run_end = INT32_MIN;
            }
            if (end > run_end) run_end = end;
        }
        free(smin);
    }
    PUSH_CUT(N);
#undef PUSH_CUT
    h_stage_time("  tail: cut points");
    const int n_parts = n_cut - 1;
    tail_part *parts = (tail_part *)calloc((size_t)n_parts, sizeof *parts);
    for (int k = 0; k < n_parts; ++k) { parts[k].j = j; parts[k].reads = reads; parts[k].files = files; parts[k].res = res; parts[k].lo = cut[k]; parts[k].hi = cut[k + 1]; parts[k].first = first_part && k == 0; }
    tail_queue tq; tq.parts = parts; tq.n_parts = n_parts; tq.next = 0; pthread_mutex_init(&tq.mu, NULL);
    if (n_thr > n_parts) n_thr = n_parts;
    pthread_t *th = (pthread_t *)calloc((size_t)n_thr, sizeof *th);
    for (int k = 0; k < n_thr; ++k) if (pthread_create(&th[k], NULL, tail_queue_main, &tq)) h_fatal("update_gtf", "pthread_create failed");
    int64_t total[H_N_SUMMARY]; memset(total, 0, sizeof total);
    FILE *outs[7] = {files->out_gtf, files->exon_bed, files->bam_gtf, files->bam_detail, files->known_gtf, files->novel_gtf, files->unrecog_gtf};
    gene_fix fix[2]; memset(fix, 0, sizeof fix);
    static const int gene_cnt[2] = {H_CNT_UPDATED_GENES, H_CNT_KNOWN_GENES};
    const double t_w0 = h_now();
    tail_writer wr[7]; pthread_t wth[7]; int n_wr = 0;
    for (int q = 0; q < 7; ++q) if (outs[q]) {
        wr[n_wr].parts = parts; wr[n_wr].n_parts = n_parts; wr[n_wr].which = q; wr[n_wr].out = outs[q];
        if (pthread_create(&wth[n_wr], NULL, tail_writer_main, &wr[n_wr])) h_fatal("update_gtf", "pthread_create failed");
        ++n_wr;
    }
    for (int k = 0; k < n_parts; ++k) {
        while (!__atomic_load_n(&parts[k].done, __ATOMIC_ACQUIRE)) usleep(200);
        for (int q = 0; q < H_N_SUMMARY; ++q) total[q] += parts[k].cnt[q];
        const int32_t tid = parts[k].hi > parts[k].lo ? reads->tid[parts[k].lo] : 0;
        for (int q = 0; q < 2; ++q) total[gene_cnt[q]] -= gene_fix_join(&fix[q], tid, parts[k].genes.first_gids[q], parts[k].genes.n_first[q]);
    }
    const double t_join = h_now() - t_w0;
    for (int q = 0; q < n_wr; ++q) pthread_join(wth[q], NULL);
    const double t_write = h_now() - t_w0;
    for (int k = 0; k < n_thr; ++k) pthread_join(th[k], NULL);
    pthread_mutex_destroy(&tq.mu);
    if (genes_out) {                                        /* the whole range as ONE part of a caller that joins shards (h_part_genes) */
        h_part_genes_free(genes_out);
        for (int q = 0; q < 2; ++q) {
            gene_fix_snapshot(&fix[q]);
            genes_out->last_gid[q] = fix[q].last ? strdup(fix[q].last) : NULL;
            genes_out->n_first[q] = fix[q].n_first;
            genes_out->first_gids[q] = (char **)h_malloc((size_t)(fix[q].n_first ? fix[q].n_first : 1) * sizeof(char *));
            for (int k = 0; k < fix[q].n_first; ++k) genes_out->first_gids[q][k] = strdup(fix[q].first[k]);
        }
    }
    for (int q = 0; q < 2; ++q) { free((void *)fix[q].cur); free((void *)fix[q].first); }
    for (int k = 0; k < n_parts; ++k) h_part_genes_free(&parts[k].genes);
    if (counters_out) memcpy(counters_out, total, sizeof total);
    else if (files->summary) h_write_summary_text(files->summary, j->anno.gene_n, (int)j->anno.n_tx, total);
    /* (tried: detail.txt -- more than half of the bytes, rows independent -- in 64 parts of its own beside the chromosome-aligned
     *  ones: slower, 2.2 -> 2.35 s on the 256-core GPU box; the tail is bound by page faults / stream growth of ~4 GB of fresh
     *  memory in one process, not by formatting; the three writer groups of a part -- lists | per-read files | summary -- side by side
     *  on threads of their own: slower as well, 2.1 -> 2.46 s; the parts' streams written by 16 threads with pwrite at their offsets:
     *  0.40 -> 0.55 s, writes to one file serialise on its inode lock) */
    if (getenv("L2R_TIMING")) fprintf(stderr, "[timing]   tail: %d parts on %d threads done after %.3f s, their streams written (one writer per file) after %.3f s\n", n_parts, n_thr, t_join, t_write);
    free(parts); free(th); free(cut);
    return 0;
}

int h_job_write_summary(h_job *j, const int64_t counters[H_N_SUMMARY], const char *path)
{
    FILE *f = fopen(path, "w");
    if (!f) h_fatal("update_gtf", "Can not open \"%s\" for writing\n", path);
    h_write_summary_text(f, j->anno.gene_n, (int)j->anno.n_tx, counters);
    fclose(f);
    return 0;
}

const char *h_job_part_last_gene(const h_job *j, int list) { return (j && list >= 0 && list < 2) ? j->part_genes.last_gid[list] : NULL; }
int h_job_part_has_first_gene(const h_job *j, int list, const char *gid) { return j ? h_part_genes_has_first(&j->part_genes, list, gid) : 0; }

void h_job_free(h_job *j)
{
    if (!j) return;
    h_part_genes_free(&j->part_genes);
    for (int k = 0; k < 8; ++k) free(j->out_path[k]);
    if (j->sj_fp) fclose(j->sj_fp);
    h_reads_free(&j->reads); h_gtf_free(&j->anno); h_sj_free(&j->sj); h_chroms_free(&j->chr);
    free(j);
}

/* engine helpers ----------------------------------------------------------- */

static void engine_fail(const char *who) { h_fatal(who, "%s", l2r_last_error()); }

/* One engine shard holds fewer than 2^32 reads + CIGAR ops (32-bit exon offsets on the device); an input beyond that is
 * classified shard by shard -- the reference has no such limit (src/update_gtf.c:1063-1083 reads whatever memory
 * holds).  Results do not depend on where the cuts fall: every read owns its result, and for unsorted input the
 * engine carries the two sequential cursors from upload to upload (l2r_upload_reads with consecutive
 * first_read_index).  L2R_CHUNK_READS = reads per shard (tests force small shards with it). */
static int64_t shard_end(const l2r_reads *r, int64_t lo)
{
    const char *e = getenv("L2R_CHUNK_READS");
    const int64_t max_reads = (e && atoll(e) > 0) ? atoll(e) : INT64_MAX;
    const int64_t max_units = 0xf0000000LL;                 /* reads + ops of a shard */
    int64_t a = lo, b = r->n_reads;                         /* largest hi with units(lo, hi) <= max_units */
    while (a < b) {
        const int64_t mid = a + (b - a + 1) / 2;
        if ((r->cig_off[mid] - r->cig_off[lo]) + (mid - lo) <= max_units) a = mid; else b = mid - 1;
    }
    if (a == lo && lo < r->n_reads) h_fatal("update_gtf", "record %lld alone exceeds the engine's shard limit", (long long)lo);
    if (a - lo > max_reads) a = lo + max_reads;
    return a;
}

static void result_reserve(h_result *r, int64_t reads_cap, int64_t ex_cap)
{
    r->ex_off = (int64_t *)h_realloc(r->ex_off, (size_t)(reads_cap + 1) * 8);
    r->info = (uint32_t *)h_realloc(r->info, (size_t)(reads_cap ? reads_cap : 1) * 4);
    r->ref_tx = (int32_t *)h_realloc(r->ref_tx, (size_t)(reads_cap ? reads_cap : 1) * 4);
    r->ex_start = (int32_t *)h_realloc(r->ex_start, (size_t)(ex_cap ? ex_cap : 1) * 4);
    r->ex_end = (int32_t *)h_realloc(r->ex_end, (size_t)(ex_cap ? ex_cap : 1) * 4);
    r->ex_flag = (uint8_t *)h_realloc(r->ex_flag, (size_t)(ex_cap ? ex_cap : 1));
}

/* Runs the engine over the whole input.
 *   acc_read == NULL: `out` = the per-read results of every record, in input order (l2r_download);
 *   acc_read != NULL: `out` = only the reads check_trans() hands to novel_T / merge_trans (update_gtf.c:946-960), in
 *                     input order, through the device-side compaction (l2r_download_accepted: about a third of the bytes
 *                     over PCIe and no host pass over the rest); *acc_read[k] = input index of row k. */
static int g_device = 0;                                    /* the HIP device of this process (a child of the multi-GPU run: its own) */

/* The engine's context (HIP runtime + device context: 0.1 - 0.2 s) can be made while the input files are read: a thread started
 * by the command before it opens anything (never in front of the fork of the multi-GPU mode: nothing there may touch HIP).
 * run_engine() takes the context over; a failure is reported there, where the one-thread order would have reported it. */
static struct { pthread_t th; int started; l2r_ctx *ctx; char err[512];
                pthread_mutex_t mu; pthread_cond_t cv; int anno_state;      /* 0: not parsed yet, 1: offered, 2: none will come */
                l2r_annotation anno; int anno_set; } g_early = { .mu = PTHREAD_MUTEX_INITIALIZER, .cv = PTHREAD_COND_INITIALIZER };
static void *early_engine_main(void *arg)
{
    (void)arg;
    g_early.ctx = l2r_create(g_device);
    if (!g_early.ctx) { snprintf(g_early.err, sizeof g_early.err, "%s", l2r_last_error()); }
    else (void)l2r_hint_single_run(g_early.ctx, 1);            /* (this program classifies every upload once) */
    /* ... and its annotation tables (0.15 s for a GENCODE-size GTF) as soon as the GTF has been parsed, beside the record reader */
    pthread_mutex_lock(&g_early.mu);
    while (g_early.anno_state == 0) pthread_cond_wait(&g_early.cv, &g_early.mu);
    const int have = g_early.anno_state == 1;
    pthread_mutex_unlock(&g_early.mu);
    if (have && g_early.ctx) {
        if (l2r_set_annotation(g_early.ctx, &g_early.anno)) { snprintf(g_early.err, sizeof g_early.err, "%s", l2r_last_error()); l2r_destroy(g_early.ctx); g_early.ctx = NULL; }
        else g_early.anno_set = 1;
    }
    return NULL;
}
/* h_fatal on another thread (a missing input, a malformed GTF) while this one may be inside hipInit / l2r_create: exit() must not
 * tear the runtime down under it -- tell the thread that no annotation will come and wait for it (not from the thread itself) */
static void early_engine_before_exit(void)
{
    if (!g_early.started || pthread_equal(pthread_self(), g_early.th)) return;
    early_engine_annotation(NULL);
    pthread_join(g_early.th, NULL);
    g_early.started = 0;
}
static void early_engine_start(void)
{
    const char *off = getenv("L2R_EARLY_ENGINE");
    if (g_early.started || (off && off[0] == '0')) return;
    g_early.ctx = NULL; g_early.err[0] = 0; g_early.anno_state = 0; g_early.anno_set = 0;
    if (pthread_create(&g_early.th, NULL, early_engine_main, NULL) == 0) { g_early.started = 1; h_before_exit = early_engine_before_exit; }
}
/* the parsed annotation for the early thread (a: arrays that stay where they are until the engine has run), or NULL: none will come */
static void early_engine_annotation(const l2r_annotation *a)
{
    if (!g_early.started) return;
    pthread_mutex_lock(&g_early.mu);
    if (g_early.anno_state == 0) { if (a) { g_early.anno = *a; g_early.anno_state = 1; } else g_early.anno_state = 2; }
    pthread_cond_signal(&g_early.cv);
    pthread_mutex_unlock(&g_early.mu);
}
/* the context of the early thread (its message when it failed), or a fresh one; *anno_set: its annotation tables are in place */
static l2r_ctx *engine_take(const char *who, int *anno_set)
{
    *anno_set = 0;
    if (g_early.started) {
        early_engine_annotation(NULL);
        pthread_join(g_early.th, NULL);
        g_early.started = 0;
        if (!g_early.ctx) h_fatal(who, "%s", g_early.err);
        l2r_ctx *c = g_early.ctx; g_early.ctx = NULL;
        *anno_set = g_early.anno_set;
        return c;
    }
    l2r_ctx *c = l2r_create(g_device);
    if (!c) engine_fail(who);
    (void)l2r_hint_single_run(c, 1);
    return c;
}
static void early_engine_drop(void)
{
    if (!g_early.started) return;
    early_engine_annotation(NULL);
    pthread_join(g_early.th, NULL);
    g_early.started = 0;
    if (g_early.ctx) { l2r_destroy(g_early.ctx); g_early.ctx = NULL; }
}

static void run_engine(const char *who, const l2r_params *prm, const l2r_annotation *a, const l2r_junctions *s,
                       const l2r_reads *r, h_result *out, int64_t **acc_read)
{
    int anno_set = 0;
    l2r_ctx *ctx = engine_take(who, &anno_set);
    h_stage_time("engine: create");
    if (l2r_set_params(ctx, prm) || l2r_set_outputs(ctx, acc_read ? L2R_WANT_ACCEPTED : L2R_WANT_RESULTS) ||
        (!anno_set && l2r_set_annotation(ctx, a)) || l2r_set_junctions(ctx, s->n ? s : NULL)) engine_fail(who);
    h_stage_time("engine: annotation tables");
    memset(out, 0, sizeof *out);
    int64_t rows = 0, exons = 0, rows_cap = 0, ex_cap = 0, *idx = NULL, *off_tmp = NULL;
    l2r_accepted_read *rec = NULL; int64_t rec_cap = 0;
    int64_t lo = 0;
    do {
        const int64_t hi = shard_end(r, lo), n = hi - lo;
        l2r_reads sub = *r;
        sub.n_reads = n; sub.n_cigar = r->cig_off[hi] - r->cig_off[lo];
        sub.tid = r->tid + lo; sub.pos = r->pos + lo; sub.rev = r->rev + lo; sub.cig = r->cig + r->cig_off[lo];
        sub.cig_summary = r->cig_summary ? r->cig_summary + 3 * (size_t)lo : NULL;
        sub.first_read_index = r->first_read_index + lo;
        if (lo > 0) {                                        /* the engine wants offsets that start at 0 */
            off_tmp = (int64_t *)h_realloc(off_tmp, (size_t)(n + 1) * 8);
            for (int64_t i = 0; i <= n; ++i) off_tmp[i] = r->cig_off[lo + i] - r->cig_off[lo];
            sub.cig_off = off_tmp;
        } else sub.cig_off = r->cig_off;
        if (l2r_upload_reads(ctx, &sub)) engine_fail(who);
        if (l2r_run(ctx) || l2r_sync(ctx)) engine_fail(who);
        int64_t nr = 0, nx = 0, na = 0, nax = 0;
        if (l2r_result_sizes(ctx, &nr, &nx, &na, &nax)) engine_fail(who);
        const int64_t add_rows = acc_read ? na : nr, add_ex = acc_read ? nax : nx;
        if (rows + add_rows > rows_cap || exons + add_ex > ex_cap || !out->ex_off) {
            /* a single shard (the usual case) is allocated exactly; later shards grow geometrically */
            rows_cap = lo == 0 && hi == r->n_reads ? add_rows : (rows + add_rows) * 2;
            ex_cap = lo == 0 && hi == r->n_reads ? add_ex : (exons + add_ex) * 2;
            result_reserve(out, rows_cap, ex_cap);
            if (acc_read) idx = (int64_t *)h_realloc(idx, (size_t)(rows_cap ? rows_cap : 1) * 8);
        }
        if (!acc_read) {
            l2r_result res = { add_rows, add_ex, 0, out->ex_off + rows, out->ex_start + exons, out->ex_end + exons, out->ex_flag + exons,
                               out->info + rows, out->ref_tx + rows };
            if (l2r_download(ctx, &res)) engine_fail(who);
        } else {
            if (add_rows > rec_cap) { rec_cap = add_rows; rec = (l2r_accepted_read *)h_realloc(rec, (size_t)(rec_cap ? rec_cap : 1) * sizeof *rec); }
            l2r_accepted acc = { add_rows, add_ex, 0, rec, out->ex_off + rows, out->ex_start + exons, out->ex_end + exons, out->ex_flag + exons };
            if (l2r_download_accepted(ctx, &acc)) engine_fail(who);
            for (int64_t k = 0; k < add_rows; ++k) {
                idx[rows + k] = (int64_t)(((uint64_t)rec[k].read_hi << 32) | rec[k].read_lo);     /* global index: first_read_index + local */
                out->info[rows + k] = rec[k].info; out->ref_tx[rows + k] = rec[k].ref_tx;
            }
        }
        if (exons) for (int64_t k = 0; k <= add_rows; ++k) out->ex_off[rows + k] += exons;          /* shard-local -> global offsets */
        rows += add_rows; exons += add_ex;
        lo = hi;
    } while (lo < r->n_reads);
    out->n = rows; out->n_ex = exons;
    if (!out->ex_off) result_reserve(out, 0, 0);
    out->ex_off[rows] = exons;
    if (acc_read) *acc_read = idx;
    free(off_tmp); free(rec);
    h_stage_time("engine: upload, kernels, download");
    l2r_destroy(ctx);
}

/* which outputs need every read (detail.txt, the all / known / unrecognised lists, summary.txt) -- without them only
 * the accepted reads matter: that is the pipeline's first pass, `update-gtf -l N in.bam old.gtf > new.gtf` (Snakefile:93) */
static int needs_all_reads(const h_job *j)
{
    return j->o.bam_gtf || j->o.bam_detail || j->o.known_gtf || j->o.unrecog_gtf || j->o.summary ||
           j->out_path[2] || j->out_path[3] || j->out_path[4] || j->out_path[6] || j->out_path[7];
}
int h_job_needs_all_reads(const h_job *j) { return needs_all_reads(j); }

/* The tail over the accepted reads alone: `res` = their rows in input order, read_idx[k] = input index of row k.  The
 * tail sees a view of the input records that keeps just them (routing, split, merge and the writers of the updated GTF /
 * novel GTF / exon bed only ever look at the reads check_trans() accepted, update_gtf.c:946-960). */
int h_job_finish_accepted(h_job *j, const l2r_result *res, const int64_t *read_idx)
{
    if (needs_all_reads(j)) h_fatal("update_gtf", "an output of this run needs every read: the accepted reads alone do not do");
    h_reads *rd = &j->reads;
    const int64_t m = res->n_reads;
    int32_t *tid = (int32_t *)h_malloc((size_t)(m + 1) * 4), *pos = (int32_t *)h_malloc((size_t)(m + 1) * 4);
    uint8_t *rev = (uint8_t *)h_malloc((size_t)m + 1);
    uint32_t *qn = (uint32_t *)h_malloc((size_t)(m + 1) * 4), *tn = rd->tid_name ? (uint32_t *)h_malloc((size_t)(m + 1) * 4) : NULL;
    for (int64_t k = 0; k < m; ++k) {
        const int64_t i = read_idx[k];
        if (i < 0 || i >= rd->n) h_fatal("update_gtf", "accepted record %lld points at read %lld of %lld", (long long)k, (long long)i, (long long)rd->n);
        if (k && i <= read_idx[k - 1]) h_fatal("update_gtf", "accepted records are not in input order at row %lld", (long long)k);
        tid[k] = rd->tid[i]; pos[k] = rd->pos[i]; rev[k] = rd->rev[i]; qn[k] = rd->qname[i];
        if (tn) tn[k] = rd->tid_name[i];
    }
    int32_t *o_tid = rd->tid, *o_pos = rd->pos; uint8_t *o_rev = rd->rev; uint32_t *o_qn = rd->qname, *o_tn = rd->tid_name; const int64_t o_n = rd->n;
    rd->tid = tid; rd->pos = pos; rd->rev = rev; rd->qname = qn; rd->tid_name = tn; rd->n = m;      /* (cig_off is not read by the tail) */
    const int rc = h_job_finish(j, res);
    rd->tid = o_tid; rd->pos = o_pos; rd->rev = o_rev; rd->qname = o_qn; rd->tid_name = o_tn; rd->n = o_n;
    free(tid); free(pos); free(rev); free(qn); free(tn);
    return rc;
}

/* ---- several GPUs of one node, in C (L2R_GPUS=N) -------------------------------------------------------------------------
 * The parent parses the inputs ONCE (the BAM is read and inflated once per node, the GTF parsed once) and, before anything has
 * touched HIP, forks one child per device.  The children inherit the record and annotation arrays (copy on write: nothing is
 * copied or re-read), take one chromosome-aligned shard each -- cut by bytes per read like workload.aligned_shard_bounds --,
 * classify it on their GPU and run the order-dependent tail on it (h_job_finish_part: the partition argument of the threaded
 * tail, no shard looks across a chromosome boundary) into part files; the parent concatenates the parts, adds the counters and
 * counts a gene id that continues across a cut once.  No data-path collective: nothing of size O(reads) crosses xGMI.  The route
 * that needs one (-s with a junction table: split pieces compare across chromosomes) and records that are not grouped by
 * chromosome run on one GPU; `python -m lr2rmats_amd.dist` has the RCCL all-gatherv for them.
 * L2R_GPU_MAP="0,0,0": device of every child (tests put several children on one GPU). */
#include <sys/wait.h>
/* 1: the partitioned route (every child merges and writes its own chromosome-aligned shard); 2: the gathered route (-s with a
 * junction table: split pieces are compared across chromosomes, Q2 -- the children classify, their results are gathered on child 0,
 * which runs the tail once); 0: one stream of records is needed (records not coordinate sorted, -m g): one GPU */
static int multi_gpu_ok(const h_job *j)
{
    if (j->mode != 0) return 0;
    for (int64_t i = 1; i < j->reads.n; ++i)
        if (j->reads.tid[i] < j->reads.tid[i - 1] || (j->reads.tid[i] == j->reads.tid[i - 1] && j->reads.pos[i] < j->reads.pos[i - 1])) return 0;
    const char *force = getenv("L2R_MULTI_ROUTE");          /* diagnostics / tests: "gathered" */
    if ((j->o.prm.split_trans && j->sj.n > 0) || (force && !strcmp(force, "gathered"))) return 2;
    return 1;
}

/* ---- the gathered route's exchange (one node): the children's per-read results -> child 0.
 * rccl: from the engines' HBM over RCCL / xGMI (l2r_xchg_*: ncclSend / ncclRecv inside one group), one GPU per child;
 * shm:  every child downloads into memory shared since before the fork -- the same interface where RCCL cannot run (several
 *       children on one GPU: the tests' L2R_GPU_MAP="0,0,0").
 * What the children share (mapped before the fork; the parent never touches HIP): */
#include <sys/mman.h>
#include <signal.h>
typedef struct {
    pthread_barrier_t bar;
    volatile int id_ready;
    char id[256];
    int64_t counts[256][2];                                  /* {reads, exons} per child */
    /* shm transport: the result arrays of the whole input (upper bounds; pages are touched by whoever writes them) */
    int64_t *ex_off; uint32_t *info; int32_t *ref_tx, *ex_start, *ex_end; uint8_t *ex_flag;
    /* ... or, when no output wants every read, of the accepted reads alone (+ their records: global read index, info, ref_tx) */
    l2r_accepted_read *rec;
} gather_shared;

static void *shared_pages(size_t bytes)
{
    void *p = mmap(NULL, bytes ? bytes : 1, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) h_fatal("update_gtf", "mmap of %zu shared bytes failed", bytes);
    return p;
}

/* One child's shard classified, its results left in HBM: the engine is returned alive */
static l2r_ctx *classify_keep(const char *who, const l2r_params *prm, const l2r_annotation *a, const l2r_junctions *s, const l2r_reads *r, unsigned want)
{
    int anno_set = 0;
    l2r_ctx *ctx = engine_take(who, &anno_set);
    if (l2r_set_params(ctx, prm) || l2r_set_outputs(ctx, want) || (!anno_set && l2r_set_annotation(ctx, a)) ||
        l2r_set_junctions(ctx, s->n ? s : NULL)) engine_fail(who);
    if (shard_end(r, 0) != r->n_reads) h_fatal(who, "a child's shard is too large for one upload (%lld reads): use more GPUs", (long long)r->n_reads);
    if (l2r_upload_reads(ctx, r) || l2r_run(ctx) || l2r_sync(ctx)) engine_fail(who);
    return ctx;
}

static void meta_write(const char *path, const int64_t *cnt, const h_part_genes *g)
{
    FILE *f = fopen(path, "w");
    if (!f) h_fatal("update_gtf", "Can not open \"%s\" for writing\n", path);
    for (int k = 0; k < H_N_SUMMARY; ++k) fprintf(f, "%lld\n", (long long)cnt[k]);
    for (int q = 0; q < 2; ++q) {
        fprintf(f, "%d %d\n", g->last_gid[q] ? 1 : 0, g->n_first[q]);
        if (g->last_gid[q]) fprintf(f, "%s\n", g->last_gid[q]);
        for (int k = 0; k < g->n_first[q]; ++k) fprintf(f, "%s\n", g->first_gids[q][k]);
    }
    if (fclose(f) != 0) h_fatal("update_gtf", "write error on \"%s\"", path);
}

static void meta_read(const char *path, int64_t *cnt, h_part_genes *g)
{
    FILE *f = fopen(path, "r");
    if (!f) h_fatal("update_gtf", "a child of the multi-GPU run left no \"%s\"", path);
    char line[2 * H_NAME_MAX + 64];
    memset(g, 0, sizeof *g);
    for (int k = 0; k < H_N_SUMMARY; ++k) { long long v = 0; if (!fgets(line, sizeof line, f) || sscanf(line, "%lld", &v) != 1) h_fatal("update_gtf", "bad \"%s\"", path); cnt[k] = v; }
    for (int q = 0; q < 2; ++q) {
        int has_last = 0, n = 0;
        if (!fgets(line, sizeof line, f) || sscanf(line, "%d %d", &has_last, &n) != 2) h_fatal("update_gtf", "bad \"%s\"", path);
        if (has_last) { if (!fgets(line, sizeof line, f)) h_fatal("update_gtf", "bad \"%s\"", path); line[strcspn(line, "\n")] = 0; g->last_gid[q] = strdup(line); }
        g->first_gids[q] = (char **)h_malloc((size_t)(n ? n : 1) * sizeof(char *)); g->n_first[q] = n;
        for (int k = 0; k < n; ++k) { if (!fgets(line, sizeof line, f)) h_fatal("update_gtf", "bad \"%s\"", path); line[strcspn(line, "\n")] = 0; g->first_gids[q][k] = strdup(line); }
    }
    fclose(f);
}

#define H_MULTI_DECLINED (-77)                               /* update_gtf_multi: nothing was started, the caller takes the one-GPU path */
static int update_gtf_multi(h_job *j, int n_gpus, int gathered)
{
    const int64_t N = j->reads.n;
    if (n_gpus > 256) h_fatal("update_gtf", "L2R_GPUS=%d: at most 256", n_gpus);
    /* chromosome-aligned cuts by bytes per read (4 per CIGAR op + 64): the nearest chromosome boundary to every ideal cut */
    int64_t *cut = (int64_t *)h_malloc((size_t)(n_gpus + 1) * sizeof(int64_t));
    {
        double total = 0.0;
        for (int64_t i = 0; i < N; ++i) total += 4.0 * (double)(j->reads.cig_off[i + 1] - j->reads.cig_off[i]) + 64.0;
        double run = 0.0; int k = 1;
        cut[0] = 0;
        for (int64_t i = 0; i < N && k < n_gpus; ++i) {
            while (k < n_gpus && run >= total * (double)k / (double)n_gpus) {
                int64_t lo = i, hi = i;
                /* (the gathered route cuts anywhere: its tail runs once, over everything) */
                while (!gathered && lo > cut[k - 1] && j->reads.tid[lo] == j->reads.tid[lo - 1]) --lo;
                while (!gathered && hi < N && hi > 0 && j->reads.tid[hi] == j->reads.tid[hi - 1]) ++hi;
                int64_t c = (i - lo <= hi - i) ? lo : hi;
                if (c < cut[k - 1]) c = cut[k - 1];
                cut[k++] = c;
            }
            run += 4.0 * (double)(j->reads.cig_off[i + 1] - j->reads.cig_off[i]) + 64.0;
        }
        while (k <= n_gpus) cut[k++] = N;
    }
    /* the part files sit next to the outputs; an updated GTF that goes to stdout is collected in a temporary file */
    char tmp_base[1024] = "";
    if (!j->out_path[0] && !gathered) {
        const char *td = getenv("TMPDIR");
        snprintf(tmp_base, sizeof tmp_base, "%s/l2r_gtf_XXXXXX", td && td[0] ? td : "/tmp");
        const int fd = mkstemp(tmp_base);
        if (fd < 0) h_fatal("update_gtf", "mkstemp failed");
        close(fd);
    }
    if (gathered) {
        /* a child of the gathered route classifies its shard in ONE upload (its results stay in HBM for the exchange): a shard beyond the
         * engine's limit -- or beyond L2R_CHUNK_READS, which the tests set -- is the one-GPU path's (upload by upload), decided before
         * anything is forked */
        l2r_params prm; l2r_annotation a; l2r_junctions s; l2r_reads r;
        h_job_views(j, &prm, &a, &s, &r);
        for (int k = 0; k < n_gpus; ++k)
            if (cut[k + 1] > cut[k] && shard_end(&r, cut[k]) < cut[k + 1]) { free(cut); return H_MULTI_DECLINED; }
    }
    const char *map = getenv("L2R_GPU_MAP");
    pid_t *pid = (pid_t *)calloc((size_t)n_gpus, sizeof *pid);
    int *dev_of = (int *)calloc((size_t)n_gpus, sizeof *dev_of);
    for (int k = 0; k < n_gpus; ++k) {
        int dev = k;
        if (map) { const char *p = map; for (int q = 0; q < k && p; ++q) { p = strchr(p, ','); if (p) ++p; } if (p) dev = atoi(p); }
        dev_of[k] = dev;
    }
    fflush(NULL);
    {   /* how many devices are there?  Asked in a short-lived child: this process must not touch HIP in front of its forks.  A child per
         * device that does not exist would fail in l2r_create only after the whole input has been parsed and cut. */
        const pid_t pp = fork();
        if (pp < 0) h_fatal("update_gtf", "fork failed");
        if (pp == 0) { const int n = l2r_device_count(); _exit(n < 0 ? 0 : (n > 250 ? 250 : n)); }
        int st = 0;
        if (waitpid(pp, &st, 0) < 0 || !WIFEXITED(st)) h_fatal("update_gtf", "the device query of the multi-GPU run failed");
        const int n_dev = WEXITSTATUS(st);
        for (int k = 0; k < n_gpus; ++k)
            if (dev_of[k] < 0 || dev_of[k] >= n_dev) {
                if (tmp_base[0]) remove(tmp_base);
                h_fatal("update_gtf", "L2R_GPUS=%d%s%s: child %d would run on device %d, this node has %d", n_gpus, map ? " with L2R_GPU_MAP=" : "", map ? map : "", k, dev_of[k], n_dev);
            }
    }
    /* ---- the gathered route: the exchange's shared block, and which transport */
    gather_shared *sh = NULL;
    int use_rccl = 0;
    /* what travels: the per-read results of every read (a detail table, the known / unrecognised lists, the summary want them), or --
     * SURVEY 8(e)'s message -- the accepted-novel records alone, 16 + 9 n bytes each, which is all the order-dependent merge reads */
    const int acc_only = gathered && !needs_all_reads(j) && !getenv("L2R_GATHER_ALL");
    if (gathered) {
        sh = (gather_shared *)shared_pages(sizeof *sh);
        memset(sh, 0, sizeof *sh);
        pthread_barrierattr_t ba; pthread_barrierattr_init(&ba); pthread_barrierattr_setpshared(&ba, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&sh->bar, &ba, (unsigned)n_gpus);
        use_rccl = 1;
        for (int k = 0; k < n_gpus; ++k) for (int q = 0; q < k; ++q) if (dev_of[q] == dev_of[k]) use_rccl = 0;      /* RCCL wants a GPU per rank */
        const char *xe = getenv("L2R_XCHG");
        if (xe && !strcmp(xe, "shm")) use_rccl = 0;
        if (xe && !strcmp(xe, "rccl")) use_rccl = 1;
        if (!use_rccl) {
            const size_t xb = (size_t)(j->reads.cig_off[N] + N) + 1;         /* n_exon(read) <= ops(read) + 1 */
            if (acc_only) sh->rec = (l2r_accepted_read *)shared_pages((size_t)(N + 1) * sizeof(l2r_accepted_read));
            sh->ex_off = (int64_t *)shared_pages((size_t)(N + n_gpus + 1) * 8); sh->info = (uint32_t *)shared_pages((size_t)(N + 1) * 4);
            sh->ref_tx = (int32_t *)shared_pages((size_t)(N + 1) * 4);
            sh->ex_start = (int32_t *)shared_pages(xb * 4); sh->ex_end = (int32_t *)shared_pages(xb * 4); sh->ex_flag = (uint8_t *)shared_pages(xb);
        }
        fprintf(stderr, "[update_gtf] L2R_GPUS=%d: gathered route (-s with a junction table): %d children classify, child 0 merges and writes; exchange: %s, %s\n",
                n_gpus, n_gpus, use_rccl ? "RCCL (ncclSend / ncclRecv to rank 0)" : "shared memory",
                acc_only ? "the accepted reads alone (no output of this run wants every read)" : "the per-read results");
    }
    if (!gathered)
    {   /* shards are whole chromosomes: say so when that leaves children without work or far out of balance */
        int64_t mx = 0; int empty = 0;
        for (int k = 0; k < n_gpus; ++k) { const int64_t n = cut[k + 1] - cut[k]; if (n == 0) ++empty; if (n > mx) mx = n; }
        if (empty || (N > 0 && (double)mx * n_gpus > 2.0 * (double)N))
            fprintf(stderr, "[update_gtf] L2R_GPUS=%d: shards are cut at chromosome boundaries -- %d of %d without records, the largest holds %lld of %lld\n",
                    n_gpus, empty, n_gpus, (long long)mx, (long long)N);
    }
    for (int k = 0; k < n_gpus; ++k) {
        const int dev = dev_of[k];
        pid[k] = fork();
        if (pid[k] < 0) {
            /* (the children started so far wait for the others at a barrier: nobody else will end them) */
            for (int q = 0; q < k; ++q) kill(pid[q], SIGKILL);
            for (int q = 0; q < k; ++q) (void)waitpid(pid[q], NULL, 0);
            h_fatal("update_gtf", "fork failed (child %d of %d)", k, n_gpus);
        }
        if (pid[k] == 0) {
            /* ---- a child: its shard on its GPU (the first HIP call of this process is in here) */
            g_device = dev;
            const int64_t lo = cut[k], hi = cut[k + 1];
            if (gathered) {
                l2r_params prm; l2r_annotation a; l2r_junctions s; l2r_reads r;
                h_job_views(j, &prm, &a, &s, &r);
                l2r_reads sub = r;
                sub.n_reads = hi - lo; sub.n_cigar = r.cig_off[hi] - r.cig_off[lo];
                sub.tid = r.tid + lo; sub.pos = r.pos + lo; sub.rev = r.rev + lo; sub.cig = r.cig + r.cig_off[lo];
                sub.cig_summary = r.cig_summary ? r.cig_summary + 3 * (size_t)lo : NULL;
                int64_t *off = (int64_t *)h_malloc((size_t)(hi - lo + 1) * 8);
                for (int64_t i = 0; i <= hi - lo; ++i) off[i] = r.cig_off[lo + i] - r.cig_off[lo];
                sub.cig_off = off; sub.first_read_index = lo;
                l2r_ctx *ctx = classify_keep("update_gtf", &prm, &a, &s, &sub, acc_only ? L2R_WANT_ACCEPTED : L2R_WANT_RESULTS);
                int rc_c = 0;
                h_result out; memset(&out, 0, sizeof out);
                int64_t *acc_idx = NULL;                                  /* acc_only, child 0: input index of every gathered record */
                if (acc_only) {
                    /* ---- the accepted reads alone */
                    l2r_accepted_read *rec = NULL;
                    int64_t m_all = 0, x_all = 0;
                    if (use_rccl) {
                        if (k == 0) {
                            if (l2r_xchg_id_bytes() > (int)sizeof sh->id || l2r_xchg_unique_id(sh->id)) engine_fail("update_gtf");
                            __sync_synchronize(); sh->id_ready = 1;
                        }
                        pthread_barrier_wait(&sh->bar);
                        if (!sh->id_ready) h_fatal("update_gtf", "child 0 left no RCCL id");
                        l2r_xchg *x = l2r_xchg_create(ctx, k, n_gpus, sh->id);
                        if (!x) engine_fail("update_gtf");
                        l2r_accepted acc; memset(&acc, 0, sizeof acc);
                        if (k == 0) {
                            const int64_t xb = j->reads.cig_off[N] + N;
                            result_reserve(&out, N, xb);
                            rec = (l2r_accepted_read *)h_malloc((size_t)(N + 1) * sizeof *rec);
                            acc.n_reads = N; acc.ex_cap = xb; acc.rec = rec; acc.ex_off = out.ex_off; acc.ex_start = out.ex_start; acc.ex_end = out.ex_end; acc.ex_flag = out.ex_flag;
                        }
                        if (l2r_xchg_gather_accepted(x, k == 0 ? &acc : NULL, NULL)) engine_fail("update_gtf");
                        l2r_xchg_destroy(x);
                        if (k == 0) { m_all = acc.n_reads; x_all = acc.n_exons; }
                    } else {
                        int64_t na = 0, nax = 0;
                        if (l2r_result_sizes(ctx, NULL, NULL, &na, &nax)) engine_fail("update_gtf");
                        sh->counts[k][0] = na; sh->counts[k][1] = nax;
                        pthread_barrier_wait(&sh->bar);
                        int64_t m_at = 0, x_at = 0;
                        for (int q = 0; q < k; ++q) { m_at += sh->counts[q][0]; x_at += sh->counts[q][1]; }
                        /* (offsets: one entry more than records, as above -- child q writes m_at + q .. ; child 0 closes the gaps) */
                        l2r_accepted acc = { na, nax, 0, sh->rec + m_at, sh->ex_off + m_at + k, sh->ex_start + x_at, sh->ex_end + x_at, sh->ex_flag + x_at };
                        if (l2r_download_accepted(ctx, &acc)) engine_fail("update_gtf");
                        pthread_barrier_wait(&sh->bar);
                        if (k == 0) {
                            int64_t ms = 0, xs = 0;
                            for (int q = 0; q < n_gpus; ++q) {
                                for (int64_t i = 0; i < sh->counts[q][0]; ++i) sh->ex_off[ms + i] = sh->ex_off[ms + i + q] + xs;
                                ms += sh->counts[q][0]; xs += sh->counts[q][1];
                            }
                            sh->ex_off[ms] = xs;
                            m_all = ms; x_all = xs; rec = sh->rec;
                            out.ex_off = sh->ex_off; out.ex_start = sh->ex_start; out.ex_end = sh->ex_end; out.ex_flag = sh->ex_flag;
                            out.info = (uint32_t *)h_malloc((size_t)(ms + 1) * 4); out.ref_tx = (int32_t *)h_malloc((size_t)(ms + 1) * 4);
                        }
                    }
                    if (k == 0) {
                        acc_idx = (int64_t *)h_malloc((size_t)(m_all + 1) * 8);
                        for (int64_t i = 0; i < m_all; ++i) {
                            acc_idx[i] = (int64_t)(((uint64_t)rec[i].read_hi << 32) | rec[i].read_lo);
                            out.info[i] = rec[i].info; out.ref_tx[i] = rec[i].ref_tx;
                        }
                        out.n = m_all; out.n_ex = x_all;
                    }
                } else
                if (use_rccl) {
                    /* rank 0 makes the id BEHIND the fork and hands it on through the shared block */
                    if (k == 0) {
                        if (l2r_xchg_id_bytes() > (int)sizeof sh->id || l2r_xchg_unique_id(sh->id)) engine_fail("update_gtf");
                        __sync_synchronize(); sh->id_ready = 1;
                    }
                    pthread_barrier_wait(&sh->bar);
                    if (!sh->id_ready) h_fatal("update_gtf", "child 0 left no RCCL id");
                    l2r_xchg *x = l2r_xchg_create(ctx, k, n_gpus, sh->id);
                    if (!x) engine_fail("update_gtf");
                    l2r_result res; memset(&res, 0, sizeof res);
                    if (k == 0) {
                        const int64_t xb = j->reads.cig_off[N] + N;
                        result_reserve(&out, N, xb);
                        res.n_reads = N; res.ex_cap = xb; res.ex_off = out.ex_off; res.ex_start = out.ex_start; res.ex_end = out.ex_end;
                        res.ex_flag = out.ex_flag; res.info = out.info; res.ref_tx = out.ref_tx;
                    }
                    if (l2r_xchg_gather_results(x, k == 0 ? &res : NULL, NULL)) engine_fail("update_gtf");
                    l2r_xchg_destroy(x);
                    if (k == 0) { out.n = res.n_reads; out.n_ex = res.n_exons; }
                } else {
                    int64_t nr = 0, nx = 0;
                    if (l2r_result_sizes(ctx, &nr, &nx, NULL, NULL)) engine_fail("update_gtf");
                    sh->counts[k][0] = nr; sh->counts[k][1] = nx;
                    pthread_barrier_wait(&sh->bar);
                    int64_t x_at = 0;
                    for (int q = 0; q < k; ++q) x_at += sh->counts[q][1];
                    /* (a shard's offsets have one entry more than it has reads: every child writes lo + k .. hi + k, child 0 closes the gaps) */
                    l2r_result res = { nr, nx, 0, sh->ex_off + lo + k, sh->ex_start + x_at, sh->ex_end + x_at, sh->ex_flag + x_at, sh->info + lo, sh->ref_tx + lo };
                    if (l2r_download(ctx, &res)) engine_fail("update_gtf");
                    pthread_barrier_wait(&sh->bar);
                    if (k == 0) {
                        int64_t xs = 0;
                        for (int q = 0; q < n_gpus; ++q) {
                            for (int64_t i = cut[q]; i < cut[q + 1]; ++i) sh->ex_off[i] = sh->ex_off[i + q] + xs;      /* (q = 0: in place; later ones move down) */
                            xs += sh->counts[q][1];
                        }
                        sh->ex_off[N] = xs;
                        out.n = N; out.n_ex = xs; out.ex_off = sh->ex_off; out.ex_start = sh->ex_start; out.ex_end = sh->ex_end;
                        out.ex_flag = sh->ex_flag; out.info = sh->info; out.ref_tx = sh->ref_tx;
                    }
                }
                l2r_destroy(ctx);
                if (k == 0) {
                    h_stage_time("children: engines, exchange");
                    l2r_result res = { out.n, out.n_ex, out.n_ex, out.ex_off, out.ex_start, out.ex_end, out.ex_flag, out.info, out.ref_tx };
                    rc_c = acc_only ? h_job_finish_accepted(j, &res, acc_idx) : h_job_finish(j, &res);
                    h_stage_time("child 0: merge + writers");
                }
                fflush(NULL);
                _exit(rc_c ? 1 : 0);
            }
            char suffix[32]; snprintf(suffix, sizeof suffix, ".part%03d", k);
            int64_t cnt[H_N_SUMMARY]; memset(cnt, 0, sizeof cnt);
            l2r_params prm; l2r_annotation a; l2r_junctions s; l2r_reads r;
            h_job_views(j, &prm, &a, &s, &r);
            h_result out; memset(&out, 0, sizeof out);
            int64_t *off = NULL;
            if (hi > lo) {
                l2r_reads sub = r;
                sub.n_reads = hi - lo; sub.n_cigar = r.cig_off[hi] - r.cig_off[lo];
                sub.tid = r.tid + lo; sub.pos = r.pos + lo; sub.rev = r.rev + lo; sub.cig = r.cig + r.cig_off[lo];
                sub.cig_summary = r.cig_summary ? r.cig_summary + 3 * (size_t)lo : NULL;
                off = (int64_t *)h_malloc((size_t)(hi - lo + 1) * 8);
                for (int64_t i = 0; i <= hi - lo; ++i) off[i] = r.cig_off[lo + i] - r.cig_off[lo];
                sub.cig_off = off; sub.first_read_index = lo;
                run_engine("update_gtf", &prm, &a, &s, &sub, &out, NULL);
            } else result_reserve(&out, 0, 0);
            l2r_result res = { out.n, out.n_ex, out.n_ex, out.ex_off, out.ex_start, out.ex_end, out.ex_flag, out.info, out.ref_tx };
            h_job_finish_part(j, lo, hi, &res, suffix, tmp_base, k == 0, cnt);
            char meta[1200]; snprintf(meta, sizeof meta, "%s%s.meta", j->out_path[0] ? j->out_path[0] : tmp_base, suffix);
            meta_write(meta, cnt, &j->part_genes);
            fflush(NULL);
            _exit(0);
        }
    }
    int failed = 0;
    if (gathered) {
        /* (the children wait for each other at barriers: one that dies takes the others with it) */
        for (int left = n_gpus; left > 0; --left) {
            int st = 0;
            const pid_t w = waitpid(-1, &st, 0);
            if (w < 0) {
                failed = 1;
                for (int k = 0; k < n_gpus; ++k) kill(pid[k], SIGKILL);      /* (whoever is left sits at a barrier) */
                for (int k = 0; k < n_gpus; ++k) (void)waitpid(pid[k], NULL, 0);
                break;
            }
            if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
                failed = 1;
                for (int k = 0; k < n_gpus; ++k) if (pid[k] != w) kill(pid[k], SIGKILL);
                for (int k = 0; k < n_gpus; ++k) if (pid[k] != w) (void)waitpid(pid[k], NULL, 0);
                break;
            }
        }
        if (failed) h_fatal("update_gtf", "a child of the multi-GPU run failed");
        /* child 0 has written every output through the streams it inherited (flushed before the fork: nothing of them is left here) */
        FILE **fs[] = {&j->o.exon_bed, &j->o.bam_gtf, &j->o.bam_detail, &j->o.known_gtf, &j->o.novel_gtf, &j->o.unrecog_gtf, &j->o.summary};
        for (size_t k = 0; k < sizeof fs / sizeof fs[0]; ++k) if (*fs[k]) { fclose(*fs[k]); *fs[k] = NULL; }
        if (j->o.out_gtf && j->o.out_gtf != stdout) { fclose(j->o.out_gtf); j->o.out_gtf = NULL; }
        free(pid); free(cut); free(dev_of);
        h_stage_time("children: engines, exchange, child 0's tail");
        return 0;
    }
    for (int k = 0; k < n_gpus; ++k) { int st = 0; if (waitpid(pid[k], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) failed = 1; }
    if (failed) {
        /* nothing of a failed run stays behind: part files, their counters, the temporary GTF */
        const char *b0 = j->out_path[0] ? j->out_path[0] : tmp_base;
        for (int k = 0; k < n_gpus; ++k) {
            char nm[1300];
            snprintf(nm, sizeof nm, "%s.part%03d.meta", b0, k); remove(nm);
            for (int w = 0; w < 8; ++w) {
                const char *path = w == 0 ? b0 : j->out_path[w];
                if (!path) continue;
                snprintf(nm, sizeof nm, "%s.part%03d", path, k); remove(nm);
            }
        }
        if (tmp_base[0]) remove(tmp_base);
        h_fatal("update_gtf", "a child of the multi-GPU run failed");
    }
    h_stage_time("children: engine + tail on every shard");
    /* ---- join: counters, gene lists (an id equal to the last entry of the parts before is not counted again, h_part_genes), files */
    int64_t total[H_N_SUMMARY]; memset(total, 0, sizeof total);
    static const int gene_cnt[2] = {H_CNT_UPDATED_GENES, H_CNT_KNOWN_GENES};
    char *last_gene[2] = {NULL, NULL};
    const char *base0 = j->out_path[0] ? j->out_path[0] : tmp_base;
    for (int k = 0; k < n_gpus; ++k) {
        char meta[1200]; snprintf(meta, sizeof meta, "%s.part%03d.meta", base0, k);
        int64_t cnt[H_N_SUMMARY]; h_part_genes g;
        meta_read(meta, cnt, &g);
        remove(meta);
        for (int q = 0; q < H_N_SUMMARY; ++q) total[q] += cnt[q];
        for (int q = 0; q < 2; ++q) {
            if (h_part_genes_has_first(&g, q, last_gene[q])) total[gene_cnt[q]] -= 1;
            if (g.last_gid[q]) { free(last_gene[q]); last_gene[q] = strdup(g.last_gid[q]); }
        }
        h_part_genes_free(&g);
    }
    free(last_gene[0]); free(last_gene[1]);
    FILE *outs[7] = {j->o.out_gtf, j->o.exon_bed, j->o.bam_gtf, j->o.bam_detail, j->o.known_gtf, j->o.novel_gtf, j->o.unrecog_gtf};
    char *buf = (char *)h_malloc(1 << 24);
    for (int w = 0; w < 7; ++w) {
        const char *path = w == 0 ? base0 : j->out_path[w];
        if (!path || !outs[w]) continue;
        for (int k = 0; k < n_gpus; ++k) {
            char part[1200]; snprintf(part, sizeof part, "%s.part%03d", path, k);
            FILE *f = fopen(part, "rb");
            if (!f) h_fatal("update_gtf", "a child of the multi-GPU run left no \"%s\"", part);
            size_t got;
            while ((got = fread(buf, 1, 1 << 24, f)) > 0) fwrite(buf, 1, got, outs[w]);
            fclose(f); remove(part);
        }
    }
    free(buf);
    if (tmp_base[0]) remove(tmp_base);
    if (j->o.summary) h_write_summary_text(j->o.summary, j->anno.gene_n, (int)j->anno.n_tx, total);
    FILE **fs[] = {&j->o.exon_bed, &j->o.bam_gtf, &j->o.bam_detail, &j->o.known_gtf, &j->o.novel_gtf, &j->o.unrecog_gtf, &j->o.summary};
    for (size_t k = 0; k < sizeof fs / sizeof fs[0]; ++k) if (*fs[k]) { fclose(*fs[k]); *fs[k] = NULL; }
    if (j->o.out_gtf && j->o.out_gtf != stdout) { fclose(j->o.out_gtf); j->o.out_gtf = NULL; } else fflush(stdout);
    free(pid); free(cut); free(dev_of);
    h_stage_time("join: parts -> files");
    return 0;
}

int h_cmd_update_gtf(int argc, char **argv)
{
    int rc = 0;
    {   /* one process, one GPU: its context is made while the files are read (the multi-GPU mode forks first); h_job_open starts
         * the thread once the command line has been accepted */
        const char *eg = getenv("L2R_GPUS");
        g_want_early_engine = (eg ? atoi(eg) : 1) <= 1 && !getenv("L2R_MULTI_ROUTE");       /* (the multi-GPU mode forks first: nothing may touch HIP in front of that) */
    }
    h_job *j = h_job_open(argc, argv, &rc);
    g_want_early_engine = 0;
    if (!j) { early_engine_drop(); return rc; }
    {   /* L2R_GPUS=N: one child per GPU (update_gtf_multi) when the partition argument holds, else this process and one GPU */
        const char *eg = getenv("L2R_GPUS");
        const int n_gpus = eg ? atoi(eg) : 1;
        if (n_gpus > 1 || (n_gpus == 1 && getenv("L2R_MULTI_ROUTE"))) {      /* (one child: diagnostics / tests -- RCCL with a world of one) */
            const int route = multi_gpu_ok(j);
            if (route) {
                rc = update_gtf_multi(j, n_gpus, route == 2);
                if (rc != H_MULTI_DECLINED) { h_job_free(j); return rc; }
                fprintf(stderr, "[update_gtf] L2R_GPUS=%d: a child's shard would not fit one upload: running on one GPU, upload by upload\n", n_gpus);
            } else
            fprintf(stderr, "[update_gtf] L2R_GPUS=%d: this input needs one stream of records (records not coordinate sorted, or -m g): running on one GPU\n", n_gpus);
        }
    }
    l2r_params prm; l2r_annotation a; l2r_junctions s; l2r_reads r;
    h_job_views(j, &prm, &a, &s, &r);
    h_result out;
    const char *force = getenv("L2R_ROUTE");                 /* diagnostics / tests: "full" or "accepted" */
    const int accepted_only = force ? !strcmp(force, "accepted") && !needs_all_reads(j) : !needs_all_reads(j);
    if (!accepted_only) {
        run_engine("update_gtf", &prm, &a, &s, &r, &out, NULL);
        l2r_result res = { out.n, out.n_ex, out.n_ex, out.ex_off, out.ex_start, out.ex_end, out.ex_flag, out.info, out.ref_tx };
        rc = h_job_finish(j, &res);
    } else {
        int64_t *idx = NULL;
        run_engine("update_gtf", &prm, &a, &s, &r, &out, &idx);
        l2r_result res = { out.n, out.n_ex, out.n_ex, out.ex_off, out.ex_start, out.ex_end, out.ex_flag, out.info, out.ref_tx };
        rc = h_job_finish_accepted(j, &res, idx);
        free(idx);
    }
    h_stage_time("merge + writers");
    h_result_free(&out);
    h_job_free(j);
    return rc;
}

static int bam2gtf_usage(void)
{
    fprintf(stderr, "\nUsage:   %s bam2gtf [option] <in.bam> > out.gtf\n\nOptions:\n\n", PROG);
    fprintf(stderr, "         -e --min-exon    [INT]    minimum length of internal exon. [3]\n");
    fprintf(stderr, "         -i --min-intron  [INT]    minimum length of intron. [3]\n");
    fprintf(stderr, "         -t --max-delet   [INT]    maximum length of deletion, longer deletion will be considered as intron. [50]\n");
    fprintf(stderr, "         -s --source      [STR]    source field in GTF, program, database or project name. [%s]\n\n", PROG);
    return 1;
}

/* exon chains of every mapped record through the engine (annotation empty) */
static void exons_only(const char *who, const char *fn, const l2r_params *prm, h_chroms *chr, h_reads *reads, h_result *out, int skip_unmapped)
{
    h_read_alignments(fn, chr, reads, skip_unmapped, who);
    l2r_annotation a; memset(&a, 0, sizeof a);
    int64_t zero_off = 0; a.tx_ex_off = &zero_off;
    l2r_junctions s; memset(&s, 0, sizeof s);
    l2r_reads r = { reads->n, reads->n_cig, reads->tid, reads->pos, reads->rev, reads->cig_off, reads->cig, 0, reads->cig_sum };
    run_engine(who, prm, &a, &s, &r, out, NULL);
}

/* src/gtf.c:597-604 print_trans: gene_id + transcript_id only, exons always ascending */
static void bam2gtf_print(const h_reads *reads, const h_result *out, const h_chroms *chr, const char *src)
{
    char line[512];
    for (int64_t i = 0; i < out->n; ++i) {
        const int64_t off = out->ex_off[i]; const int n = (int)L2R_INFO_NEXON(out->info[i]);
        const char *q = h_str(&reads->names, reads->qname[i]), *cn = chr->name[reads->tid[i]];
        const char st = "+-"[reads->rev[i] != 0];
        snprintf(line, sizeof line, "%s\t%s\ttranscript\t%d\t%d\t.\t%c\t.\tgene_id \"%s\"; transcript_id \"%s\";\n", cn, src, out->ex_start[off], out->ex_end[off + n - 1], st, q, q);
        fputs(line, stdout);
        for (int k = 0; k < n; ++k) {
            snprintf(line, sizeof line, "%s\t%s\texon\t%d\t%d\t.\t%c\t.\tgene_id \"%s\"; transcript_id \"%s\";\n", cn, src, out->ex_start[off + k], out->ex_end[off + k], st, q, q);
            fputs(line, stdout);
        }
    }
}

int h_cmd_bam2gtf(int argc, char **argv)
{
    static const struct option lopt[] = {{"exon-min", 1, 0, 'e'}, {"intron-len", 1, 0, 'i'}, {"source", 1, 0, 's'}, {0, 0, 0, 0}};
    l2r_params p; default_params(&p);
    char src[1024]; strcpy(src, PROG);
    int c;
    optind = 1;
    while ((c = getopt_long(argc, argv, "s:e:i:t:", lopt, NULL)) >= 0) {
        switch (c) {
        case 'e': p.min_exon = atoi(optarg); break;
        case 'i': p.min_intron = atoi(optarg); break;
        case 't': p.max_delet = atoi(optarg); break;
        case 's': strncpy(src, optarg, sizeof src - 1); break;
        default: fprintf(stderr, "Error: unknown option: %s.\n", optarg); return bam2gtf_usage();
        }
    }
    if (argc - optind != 1) return bam2gtf_usage();
    h_chroms chr; memset(&chr, 0, sizeof chr);
    /* A BGZF-compressed BAM is converted window by window, as the reference converts record by record (src/bam2gtf.c:150): one
     * engine, a batch of records uploaded, walked, downloaded and printed at a time; memory is bounded by the window, not by the file. */
    h_aln_stream *st = h_aln_stream_open(argv[optind], &chr, 1, "bam2gtf");
    if (st) {
        l2r_ctx *ctx = l2r_create(0);
        if (!ctx) engine_fail("bam2gtf");
        (void)l2r_hint_single_run(ctx, 1);
        l2r_annotation a; memset(&a, 0, sizeof a);
        int64_t zero_off = 0; a.tx_ex_off = &zero_off;
        if (l2r_set_params(ctx, &p) || l2r_set_outputs(ctx, L2R_WANT_RESULTS) || l2r_set_annotation(ctx, &a) || l2r_set_junctions(ctx, NULL)) engine_fail("bam2gtf");
        for (;;) {
            h_reads reads; memset(&reads, 0, sizeof reads);
            const int64_t got = h_aln_stream_next(st, &reads);
            if (got > 0) {
                l2r_reads r = { reads.n, reads.n_cig, reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, 0, reads.cig_sum };
                if (l2r_upload_reads(ctx, &r) || l2r_run(ctx) || l2r_sync(ctx)) engine_fail("bam2gtf");
                int64_t nr = 0, nx = 0;
                if (l2r_result_sizes(ctx, &nr, &nx, NULL, NULL)) engine_fail("bam2gtf");
                h_result out; memset(&out, 0, sizeof out);
                result_reserve(&out, nr, nx);
                l2r_result res = { nr, nx, 0, out.ex_off, out.ex_start, out.ex_end, out.ex_flag, out.info, out.ref_tx };
                if (l2r_download(ctx, &res)) engine_fail("bam2gtf");
                out.n = nr; out.n_ex = nx;
                bam2gtf_print(&reads, &out, &chr, src);
                h_result_free(&out);
            }
            h_reads_free(&reads);
            if (got < 0) break;
        }
        h_aln_stream_close(st);
        l2r_destroy(ctx);
        fflush(stdout);
        h_chroms_free(&chr);
        return 0;
    }
    h_reads reads; h_result out;
    exons_only("bam2gtf", argv[optind], &p, &chr, &reads, &out, 1);
    bam2gtf_print(&reads, &out, &chr, src);
    fflush(stdout);
    h_result_free(&out); h_reads_free(&reads); h_chroms_free(&chr);
    return 0;
}

static int unique_usage(void)
{
    fprintf(stderr, "\nUsage:   %s unique-gtf [option] <in.sorted.bam/in.sorted.gtf> > unique.gtf\n\n", PROG);
    fprintf(stderr, "Notice:  the BAM and GTF files should be sorted in advance.\n\n");
    fprintf(stderr, "         -m --input-mode  [STR]    BAM file(b) or GTF file(g). [b]\n");
    fprintf(stderr, "         -b --bam         [STR]    for GTF input, BAM file to obtain BAM header information. [NULL]\n");
    fprintf(stderr, "         -s --force-strand         force to match strand when merging transcripts. [False]\n");
    fprintf(stderr, "         -e -i -d -D -f            as update-gtf\n");
    fprintf(stderr, "         -I --intersect            output intersected transcript. [False]\n");
    fprintf(stderr, "         -o --output      [STR]    unique GTF file. [stdout]\n");
    fprintf(stderr, "         -S --source      [STR]    'source' field in GTF. [%s]\n\n", PROG);
    return 1;
}

int h_cmd_unique_gtf(int argc, char **argv)
{
    /* src/unique_gtf.c:53-158; optstring :90 (no 't': -t is not accepted although :109 would handle it) */
    static const struct option lopt[] = {
        {"input-mode", 1, 0, 'm'}, {"bam", 1, 0, 'b'}, {"force-strand", 0, 0, 's'}, {"min-exon", 1, 0, 'e'},
        {"min-intron", 1, 0, 'i'}, {"distance", 1, 0, 'd'}, {"DISTANCE", 1, 0, 'D'}, {"frac", 1, 0, 'f'},
        {"intersect", 0, 0, 'I'}, {"output", 1, 0, 'o'}, {"source", 1, 0, 's'}, {0, 0, 0, 0}};
    l2r_params p; default_params(&p);
    int mode = 0, c, intersect = 0; const char *hdr_file = NULL; char src[1024]; strcpy(src, PROG); FILE *out = stdout;
    optind = 1;
    while ((c = getopt_long(argc, argv, "m:b:se:i:Id:D:f:o:S:", lopt, NULL)) >= 0) {
        switch (c) {
        case 'm': if (optarg[0] == 'b') mode = 0; else if (optarg[0] == 'g') mode = 1; else return unique_usage(); break;
        case 'b': hdr_file = optarg; break;
        case 's': p.force_strand = 1; break;
        case 'e': p.min_exon = atoi(optarg); break;
        case 'i': p.min_intron = atoi(optarg); break;
        case 'd': p.ss_dis = atoi(optarg); break;
        case 'D': p.end_dis = atoi(optarg); break;
        case 'f': p.single_exon_ovlp_frac = atof(optarg); break;
        case 'I': intersect = 1; break;
        case 'o': out = open_w(optarg); break;
        case 'S': strncpy(src, optarg, sizeof src - 1); break;
        default: fprintf(stderr, "Error: unknown option: %s.\n", optarg); return unique_usage();
        }
    }
    if (argc - optind != 1) return unique_usage();
    h_chroms chr; memset(&chr, 0, sizeof chr);
    if (mode == 0) {
        h_reads reads; h_result res;
        exons_only("unique_gtf", argv[optind], &p, &chr, &reads, &res, 0);
        /* strand as gen_exon gives it; all four names = QNAME (src/bam2gtf.c:104) */
        h_unique_tail(&p, src, out, intersect, &chr, res.n, reads.tid, reads.rev, res.ex_off, res.ex_start, res.ex_end,
                      &reads.names, reads.qname, reads.qname, reads.qname, reads.qname);
        h_result_free(&res); h_reads_free(&reads);
    } else {
        if (!hdr_file) h_fatal("unique_gtf", "Couldn't read header of provided BAM file.\n");
        h_read_header_only(hdr_file, &chr, "unique_gtf");
        h_gtf g;
        h_read_gtf(argv[optind], &chr, &g, 1);
        h_unique_tail(&p, src, out, intersect, &chr, g.n_tx, g.tid, g.rev, g.ex_off, g.ex_start, g.ex_end,
                      &g.names, g.gid, g.tids, g.gname, g.tname);
        h_gtf_free(&g);
    }
    if (out != stdout) fclose(out); else fflush(stdout);
    h_chroms_free(&chr);
    return 0;
}

int h_main(int argc, char **argv)
{
    /* src/main.c:37-49 */
    if (argc < 1) return 1;
    if (strcmp(argv[0], "update-gtf") == 0) return h_cmd_update_gtf(argc, argv);
    if (strcmp(argv[0], "bam2gtf") == 0) return h_cmd_bam2gtf(argc, argv);
    if (strcmp(argv[0], "unique-gtf") == 0) return h_cmd_unique_gtf(argc, argv);
    if (strcmp(argv[0], "filter") == 0) return h_cmd_filter(argc, argv);
    /* (diagnostics, no GPU: every record of a SAM / BAM file written out as BAM -- reader, encoder and BGZF writer of `filter`) */
    if (strcmp(argv[0], "records2bam") == 0 && argc == 3) return h_records_to_bam(argv[1], argv[2]) ? 1 : 0;
    /* (diagnostics, no GPU: the block ranges `world` ranks of a multi-process run would inflate of a BAM file, one line per rank:
     *  rank, start and end (block offset : offset inside), compressed bytes inflated, records; last line: whether the ranges meet) */
    if (strcmp(argv[0], "bam-shards") == 0 && argc == 3) {
        const int world = atoi(argv[2]);
        int64_t prev_end[2] = {0, 0}, total = 0; int ok = world >= 1;
        for (int r = 0; r < world && ok; ++r) {
            h_chroms chr; memset(&chr, 0, sizeof chr);
            h_reads rd; int64_t info[8];
            if (!h_read_alignments_blocks(argv[1], &chr, &rd, 0, "bam-shards", r, world, info)) { printf("rank %d: not readable by block ranges\n", r); ok = 0; break; }
            printf("rank %d: %lld:%lld .. %lld:%lld, %lld of %lld bytes inflated, %lld records\n", r, (long long)info[0], (long long)info[1], (long long)info[2],
                   (long long)info[3], (long long)info[4], (long long)info[5], (long long)info[6]);
            if (r && (info[0] != prev_end[0] || info[1] != prev_end[1])) ok = 0;
            prev_end[0] = info[2]; prev_end[1] = info[3]; total += info[6];
            h_reads_free(&rd);
        }
        printf("%s, %lld records\n", ok ? "ranges meet" : "RANGES DO NOT MEET", (long long)total);
        return ok ? 0 : 1;
    }
    if (!strcmp(argv[0], "fusion") || !strcmp(argv[0], "bam2sj")) {
        fprintf(stderr, "[main] command '%s' is outside the MI355X build (see DESIGN.md, scope)\n", argv[0]);
        return 1;
    }
    fprintf(stderr, "[main] unrecognized command '%s'\n", argv[0]);
    return 1;
}
