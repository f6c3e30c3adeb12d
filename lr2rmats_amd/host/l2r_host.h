/* l2r_host.h -- host side (C) of the MI355X lr2rmats build.
 *
 * Mirrors the reference's sub-command surface (src/main.c:41-47:
 * update-gtf, unique-gtf, bam2gtf) around the C-ABI engine of
 * include/lr2rmats_hip.h.  Data lives in structure-of-arrays containers that
 * are handed to the engine unchanged; the order-dependent tail of the
 * reference (list routing, split_trans, merge_trans, writers) runs here.
 */
#ifndef L2R_HOST_H
#define L2R_HOST_H
#include <stdint.h>
#include <stdio.h>
#include "../../include/lr2rmats_hip.h"

#define H_NAME_MAX 100          /* reference name buffers are char[100] (src/gtf.h:44-45) */

/* ---- errors: reference exit behaviour (src/utils.c:91-111) */
extern void (*h_before_exit)(void);            /* called by h_fatal in front of exit(): a thread that is starting the HIP runtime is waited for */
void h_fatal(const char *where, const char *fmt, ...);        /* "[where] msg" + exit(1) */
void h_fatal_core(const char *where, const char *fmt, ...);   /* "[where] msg Abort!" + abort() */
void h_stage_time(const char *what);                          /* L2R_TIMING=1: wall clock since the last call, on stderr */
void *h_malloc(size_t n);
void *h_realloc(void *p, size_t n);
FILE *h_open_growbuf(char **buf, size_t *len);                /* like open_memstream, on h_realloc (huge pages); *buf, *len valid after fclose */

/* ---- append-only string table: ids are byte offsets */
typedef struct { char *buf; size_t len, cap; } h_strtab;
uint32_t h_str_add(h_strtab *t, const char *s);
static inline const char *h_str(const h_strtab *t, uint32_t id) { return t->buf + id; }

/* ---- chromosome names: BAM header first, then first-seen in the SJ file (src/gtf.c:389-412) */
typedef struct { char **name; int n, cap, n_hdr; } h_chroms;
int  h_chrom_find(const h_chroms *c, const char *s, int limit);
int  h_chrom_intern(h_chroms *c, const char *s);
void h_chroms_free(h_chroms *c);

/* ---- alignment records (what l2r_reads wants) + names */
typedef struct {
    int64_t n, cap;
    int32_t *tid, *pos;
    uint8_t *rev;
    int64_t *cig_off;
    uint32_t *cig; int64_t n_cig, cap_cig;
    uint32_t *cig_sum;           /* alignments: three words per record, made while its CIGAR is converted (l2r_reads::cig_summary); NULL for `-m g` input */
    uint32_t *qname;             /* ids into names: QNAME = trans_name (and trans_id, unless tid_name is set) */
    uint32_t *tid_name;          /* `-m g` input only: transcript_id of the read-like transcript (NULL for alignments) */
    h_strtab names;
} h_reads;

/* Reads every record of a SAM (text), gzip/BGZF-compressed SAM or BAM file.
 * skip_unmapped = 0: an unmapped record is the reference's abort (Q9). */
void h_read_alignments(const char *fn, h_chroms *chr, h_reads *out, int skip_unmapped, const char *who);
int  h_read_alignments_shard(const char *fn, h_chroms *chr, h_reads *out, int skip_unmapped, const char *who, int rank, int world,
                             int64_t *lo, int64_t *hi, int64_t *n_total);
void h_read_header_only(const char *fn, h_chroms *chr, const char *who);
/* a BGZF-compressed BAM window by window (aln_reader.c); open returns NULL for any other file */
/* one rank's block range of a coordinate-sorted BGZF BAM (aln_reader.c); info[8]: start / end positions, bytes inflated, file size, records */
int  h_read_alignments_blocks(const char *fn, h_chroms *chr, h_reads *out, int skip_unmapped, const char *who, int rank, int world, int64_t info[8]);
typedef struct h_aln_stream h_aln_stream;
h_aln_stream *h_aln_stream_open(const char *fn, h_chroms *chr, int skip_unmapped, const char *who);
int64_t h_aln_stream_next(h_aln_stream *s, h_reads *out);     /* appends a batch to *out; records appended, -1 at the end */
void h_aln_stream_close(h_aln_stream *s);
void h_reads_free(h_reads *r);
/* per-record CIGAR summaries (l2r_reads::cig_summary, three words each) of records that are in memory already */
void h_cigar_summaries(int64_t n, const int64_t *cig_off, const uint32_t *cig, uint32_t *out);

/* ---- whole alignment records, BAM-encoded (`filter`: the kept ones are written out again) */
typedef struct { uint8_t *p; size_t n; } h_blob;
h_blob h_slurp(const char *fn, const char *who);             /* whole file, gzip / BGZF inflated (BGZF: on several threads) */
typedef struct {
    uint8_t *hdr; size_t hdr_len;          /* the BAM header block: magic, text, references */
    uint8_t *buf; size_t buf_len;          /* records back to back, each behind its block_size word */
    int64_t n, cap;
    int64_t *rec_off;                      /* n + 1: record i = buf[rec_off[i], rec_off[i + 1]) */
    /* what the tests of `filter` read (l2r_filter_records) */
    uint16_t *flag; int32_t *tid, *pos, *l_qseq, *nm; uint8_t *nm_seen;
    int64_t *cig_off; uint32_t *cig; int64_t n_cig, cap_cig;
} h_records;
void h_read_records(const char *fn, h_chroms *chr, h_records *out, const char *who);
void h_records_free(h_records *r);
/* header + the records keep[0..n_keep) (indices, ascending) as a BGZF-compressed BAM stream */
int  h_write_bam(FILE *fp, const h_records *r, const int64_t *keep, int64_t n_keep);
int  h_records_to_bam(const char *in_fn, const char *out_fn);
int  h_filter_run(const char *in_fn, const char *remove_fn, const l2r_filter_params *prm, FILE *out, int64_t *n_written);

/* ---- transcripts from a GTF (annotation, or read-like input of `-m g`) */
typedef struct {
    int64_t n_tx, cap_tx, n_ex, cap_ex;
    int32_t *tid, *start, *end;
    uint8_t *rev;
    int64_t *ex_off;             /* n_tx + 1 */
    int32_t *ex_start, *ex_end;
    uint32_t *gid, *gname, *tids, *tname;   /* ids into names */
    h_strtab names;
    int gene_n;
} h_gtf;
void h_read_gtf(const char *fn, const h_chroms *chr, h_gtf *out, int as_reads);
void h_gtf_free(h_gtf *g);

/* ---- STAR SJ.out.tab */
typedef struct { int64_t n; int32_t *tid, *don, *acc, *uniq, *multi; } h_sj;
void h_read_sj(FILE *fp, h_chroms *chr, h_sj *out);
void h_sj_free(h_sj *s);

/* ---- per-read results as the engine returns them */
typedef struct {
    int64_t n, n_ex;
    int64_t *ex_off;
    int32_t *ex_start, *ex_end;
    uint8_t *ex_flag;
    uint32_t *info;
    int32_t *ref_tx;
} h_result;
void h_result_alloc(h_result *r, int64_t n_reads, int64_t ex_cap);
void h_result_free(h_result *r);

/* ---- the sequential tail + writers */
#define H_N_SUMMARY 16            /* counters of summary.txt, see tail.c summary_and_bed */
#define H_CNT_UPDATED_GENES 0
#define H_CNT_KNOWN_GENES   7
/* What a chromosome-aligned PART of the reads leaves behind for the one place where the reference's lists do look across
 * a chromosome boundary: merge_gene (src/update_gtf.c:181-189) compares the gene_id with the list's last entry BEFORE
 * it tests the tid break, so a gene whose id equals the last entry of the previous chromosome is not counted again.
 * List 0: genes of the updated transcripts, list 1: genes of the known reads.  A part starts with empty lists; with
 * L = gene_id of the last entry of the parts before it, the part counted one gene too many iff L is among the ids it
 * added under the tid of its own first entry (h_part_genes_fix does the bookkeeping over the parts in order). */
typedef struct {
    char *last_gid[2];            /* gene_id of the list's last entry, NULL: the part added nothing */
    char **first_gids[2]; int n_first[2];      /* ids added under the tid of the list's first entry */
} h_part_genes;
void h_part_genes_free(h_part_genes *g);
int  h_part_genes_has_first(const h_part_genes *g, int list, const char *gid);
typedef struct {
    l2r_params prm;
    FILE *out_gtf, *exon_bed, *bam_gtf, *bam_detail, *known_gtf, *novel_gtf, *unrecog_gtf, *summary;
    char source[1024];
    int no_detail_header;        /* parts after the first of a partitioned run: detail.txt without its column line */
    int64_t *summary_counts;     /* non-NULL: the summary counters go here (H_N_SUMMARY values) instead of being printed */
    h_part_genes *part_genes;    /* non-NULL (parts of a partitioned run): filled for the caller's fix-up of the gene counters */
} h_update_opts;

/* Everything update_gtf() does after check_with_anno_trans/check_with_short_sj:
 * routing + split + merge (src/update_gtf.c:943-964) and all writers (:1087-1095).
 * read names / gene names come from `reads` and `anno`. */
void h_update_tail(const h_update_opts *o, const h_chroms *chr, const h_reads *reads, const h_gtf *anno,
                   const h_result *res, int64_t n_sj);

void h_write_summary_text(FILE *s, int anno_genes, int anno_tx, const int64_t *counters);

void h_unique_tail(const l2r_params *p, const char *source, FILE *out, int intersect, const h_chroms *chr,
                   int64_t n, const int32_t *tid, const uint8_t *rev, const int64_t *ex_off, const int32_t *xs, const int32_t *xe,
                   const h_strtab *names, const uint32_t *gid, const uint32_t *tids, const uint32_t *gname, const uint32_t *tname);

/* ---- sub-commands (argv[0] = sub-command name) */
int h_cmd_update_gtf(int argc, char **argv);
int h_cmd_bam2gtf(int argc, char **argv);
int h_cmd_unique_gtf(int argc, char **argv);
int h_cmd_filter(int argc, char **argv);
int h_main(int argc, char **argv);

/* ---- staged form of update-gtf, used by the CLI itself and by the one-process-per-GPU driver
 * (lr2rmats_amd/dist.py): open = parse options + read all inputs; views = the structure-of-arrays
 * the engine consumes; finish = sequential tail + writers with the per-read results. */
typedef struct h_job h_job;
h_job *h_job_open(int argc, char **argv, int *exit_code);
h_job *h_job_open2(int argc, char **argv, int *exit_code, int open_outputs);   /* 0: do not create output files */
/* One rank of a one-process-per-GPU run: when the alignments are a coordinate-sorted BAM and the run can take the partitioned
 * route (no -s with a junction table), only the rank's chromosome-aligned shard of the records is loaded -- h_job_shard() then
 * gives its place: records [*lo, *hi) of *n_total, and returns 1; the job's read arrays hold just those (index 0 = record *lo). */
h_job *h_job_open_rank(int argc, char **argv, int *exit_code, int open_outputs, int rank, int world);
int    h_job_shard(const h_job *j, int64_t *lo, int64_t *hi, int64_t *n_total);
/* h_job_shard() == 2: the rank loaded the BGZF blocks of its own records only (h_read_alignments_blocks); info as there */
void   h_job_shard_blocks(const h_job *j, int64_t info[8]);
void   h_job_views(h_job *j, l2r_params *prm, l2r_annotation *anno, l2r_junctions *sj, l2r_reads *reads);
int    h_job_finish(h_job *j, const l2r_result *res);
void   h_job_free(h_job *j);
/* 1: an output of this run lists every read (detail.txt, -a / -k / -u, summary.txt); 0: the accepted reads are all the
 * tail needs (`update-gtf ... > new.gtf`, -v, -E) and h_job_finish_accepted() may be used */
int    h_job_needs_all_reads(const h_job *j);
/* finish with the rows of the accepted reads only (l2r_download_accepted, or the records gathered from several GPUs):
 * res->n_reads rows in input order, read_idx[k] = input index of row k */
int    h_job_finish_accepted(h_job *j, const l2r_result *res, const int64_t *read_idx);

/* Partitioned form (one-process-per-GPU runs whose read shards are cut at chromosome boundaries: the order-dependent
 * tail never looks across chromosomes -- merge_trans stops at a smaller tid, src/update_gtf.c:147, the novel-exon
 * and gene lists likewise, :181-222 -- so every rank can run it on its own shard).  Writes every requested output
 * of the reads [lo, hi) to "<output path><suffix>" (the updated GTF of a stdout run to "<stdout_base><suffix>") and
 * returns the summary counters; the caller concatenates the parts in shard order and adds the counters up.
 * Not valid with -s and a junction table (split pieces carry tid 0 and are compared across chromosomes, Q2). */
int    h_job_finish_part(h_job *j, int64_t lo, int64_t hi, const l2r_result *res, const char *suffix, const char *stdout_base,
                         int first_part, int64_t counters[H_N_SUMMARY]);
/* which: 0 updated gtf (NULL = stdout), 1 exon bed, 2 bam gtf, 3 detail, 4 known, 5 novel, 6 unrecog, 7 summary */
/* the gene lists of the part h_job_finish_part() last ran (see h_part_genes): gene_id of list `list`'s last entry (NULL: none
 * added), and whether `gid` is among the ids the part added under the tid of its first entry */
const char *h_job_part_last_gene(const h_job *j, int list);
int         h_job_part_has_first_gene(const h_job *j, int list, const char *gid);
const char *h_job_out_path(const h_job *j, int which);
void   h_job_set_out_path(h_job *j, int which, const char *path);
void   h_job_open_outputs(h_job *j);                       /* open the output files of a job that was opened without them */
int    h_job_write_summary(h_job *j, const int64_t counters[H_N_SUMMARY], const char *path);

#endif
