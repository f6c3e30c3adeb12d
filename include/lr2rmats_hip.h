/* lr2rmats_hip.h -- C-ABI of the MI355X (gfx950) read-vs-annotation engine.
 *
 * This is the drop-in boundary for the one hot path of lr2rmats `update-gtf`
 * (and the CIGAR half of `bam2gtf` / `unique-gtf`): everything between
 * "alignment records and annotation are in memory" and "every read carries its
 * exons, class, flags and reference transcript".  In the reference that seam is
 *
 *     read_bam_trans()   src/bam2gtf.c:89-110   (gen_exon :31-78 per record)
 *     check_trans()      src/update_gtf.c:936-965
 *         check_with_anno_trans()  :792-835   check_full() :629-681
 *         check_splice_site()      :717-779   set_full()   :683-696
 *         check_with_short_sj()    :698-709   check_short_sj{,1}() :589-627
 *
 * called from update_gtf() src/update_gtf.c:1069 and :1083.  The list routing,
 * split_trans(), merge_trans() and the writers stay on the host (they are
 * order dependent); they consume the arrays this library returns.
 *
 * Plain C types only: caller-owned host buffers, sizes, int return codes
 * (0 = ok, negative = error, text from l2r_last_error()).  One context drives
 * one GPU (one process per GPU); all device work of a context is issued on the
 * context's own HIP stream.  There is no CPU fallback: l2r_create() fails when
 * no gfx950 device is usable.
 */
#ifndef LR2RMATS_HIP_H
#define LR2RMATS_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L2R_ABI_VERSION 3

typedef struct l2r_ctx l2r_ctx;

/* The fields of update_gtf_para (src/update_gtf.h:8-15) that the path reads;
 * defaults in src/update_gtf.c:24-35 and src/gtf.h:118-127. */
typedef struct {
    int32_t min_exon;               /* -e  INTER_EXON_MIN_LEN 3  */
    int32_t min_intron;             /* -i  INTRON_MIN_LEN 3      */
    int32_t max_delet;              /* -t  DELETION_MAX_LEN 50   */
    int32_t ss_dis;                 /* -d  SPLICE_DISTANCE 0     */
    int32_t end_dis;                /* -D  END_DISTANCE 0x7fffffff (host merge only) */
    int32_t full_level;             /* -l  1..5, default 5       */
    int32_t split_trans;            /* -s  */
    int32_t use_multi;              /* -M  */
    int32_t min_sj_cnt;             /* -J  MIN_SJ_CNT 1          */
    int32_t force_strand;           /* -c  (host merge only)     */
    float   single_exon_ovlp_frac;  /* -f  SING_OVLP_FRAC 0.80   */
} l2r_params;

/* Annotation transcripts in GTF FILE ORDER, as read_anno_trans() leaves them
 * (src/gtf.c:468-521): exons of a transcript sorted by (start,end), tx_start =
 * first exon start, tx_end = end of the LAST exon in that order, tid = index in
 * the BAM header or -1.  1-based closed coordinates. */
typedef struct {
    int64_t n_tx, n_exon;
    const int32_t *tx_tid, *tx_start, *tx_end;
    const uint8_t *tx_rev;
    const int64_t *tx_ex_off;       /* n_tx + 1 */
    const int32_t *ex_start, *ex_end;
} l2r_annotation;

/* STAR SJ.out.tab rows as read_sj_group() leaves them (src/gtf.c:431-449):
 * sorted by (tid, don, acc); don/acc = first/last intron base. */
typedef struct {
    int64_t n;
    const int32_t *tid, *don, *acc, *uniq_c, *multi_c;
} l2r_junctions;

/* Alignment records in input order: what gen_exon() reads from a bam1_t
 * (src/bam2gtf.c:34-41): core.tid, core.pos (0-based), the strand resolved as
 * "XS aux present ? (XS:A == '+' ? 0 : 1) : FLAG & 16", and the raw CIGAR words
 * (len << 4 | op).  Unmapped records must be rejected by the caller (the
 * reference aborts on them in update-gtf, skips them in bam2gtf). */
typedef struct {
    int64_t n_reads, n_cigar;
    const int32_t *tid, *pos;
    const uint8_t *rev;
    const int64_t *cig_off;         /* n_reads + 1 */
    const uint32_t *cig;
    int64_t first_read_index;       /* global index of record 0 (read shards of a multi-GPU run) */
    /* Optional (NULL: the engine walks every CIGAR once at upload to make it, k_tile_index): what a reader knows of a record's CIGAR
     * while it converts it (host/aln_reader.c does for every input format; lr2rmats_amd/synth.py for synthetic reads), whatever the
     * thresholds of a run -- three words per record:
     *   [0] reference bases of the CIGAR (ops M D N = X): the record ends at pos + [0] (src/bam2gtf.c:41-74: `end` only grows by these);
     *   [1] N operations | the shortest of them << 16;     [2] the longest D operation | the shortest stretch of reference bases
     *       between two N operations << 16 -- each of the four saturated at 65535 (the shortest ones 65535 where there is none).
     * With it a run knows a tile's exon count without a CIGAR walk wherever no threshold is borderline (exons = records + N operations
     * while -i <= shortest N, -t >= longest D, -e <= shortest stretch), and the upload's index is a scan over 15 bytes per record. */
    const uint32_t *cig_summary;
} l2r_reads;

/* bits of info[] : one word per read; bits 8..31 = exon count */
#define L2R_INFO_KNOWN      0x01u   /* trans_t.known                       */
#define L2R_INFO_KNOWN_SITE 0x02u   /* trans_t.has_known_site              */
#define L2R_INFO_FULL       0x04u   /* trans_t.full                        */
#define L2R_INFO_REV        0x08u   /* is_rev after update_gtf.c:825-831   */
#define L2R_INFO_UNREL      0x10u   /* has_unreliable_junction             */
#define L2R_INFO_SJ_CHECKED 0x20u   /* check_with_short_sj() was reached   */
#define L2R_INFO_SJ_PASS    0x40u   /* ... and returned 1                  */
#define L2R_INFO_ACCEPTED   0x80u   /* goes to novel_T / updated_T (whole or, with -s, as split pieces) */
#define L2R_INFO_NEXON(i)   ((i) >> 8)

/* bits of ex_flag[] : one byte per exon j; junction j (exon j -> j+1) lives with exon j */
#define L2R_EXF_NOVEL_EXON  0x01u   /* novel_exon_flag[j]          */
#define L2R_EXF_NOVEL_DON   0x02u   /* novel_site_flag[2j]         */
#define L2R_EXF_NOVEL_ACC   0x04u   /* novel_site_flag[2j+1]       */
#define L2R_EXF_NOVEL_JUNC  0x08u   /* novel_junction_flag[j]      */
#define L2R_EXF_UNREL_JUNC  0x10u   /* unreliable_junction_flag[j] */

/* Per-read results in read order (what check_trans() leaves in bam_T). */
typedef struct {
    int64_t n_reads;                /* in: capacity of the per-read arrays; out: reads written  */
    int64_t ex_cap;                 /* in: capacity of the per-exon arrays                       */
    int64_t n_exons;                /* out */
    int64_t *ex_off;                /* n_reads + 1 */
    int32_t *ex_start, *ex_end;
    uint8_t *ex_flag;
    uint32_t *info;
    int32_t *ref_tx;                /* ref_anno_i or -1 */
} l2r_result;

/* Accepted-novel records, compacted in read order: the reads check_trans()
 * hands to novel_T/merge_trans (update_gtf.c:946-960).  This is the message of
 * the multi-GPU all-gatherv. */
typedef struct {
    uint32_t read_lo, read_hi;      /* global read index (64 bit, split) */
    uint32_t info;
    int32_t  ref_tx;
} l2r_accepted_read;                /* 16 bytes */

typedef struct {
    int64_t n_reads, ex_cap;        /* in: capacities; out n_reads = records written */
    int64_t n_exons;                /* out */
    l2r_accepted_read *rec;
    int64_t *ex_off;                /* n_reads + 1, into the arrays below */
    int32_t *ex_start, *ex_end;
    uint8_t *ex_flag;
} l2r_accepted;

/* Raw device views (for a caller that keeps data in HBM: bench, RCCL gather). */
typedef struct {
    int64_t n_reads, n_exons, n_accepted, n_accepted_exons;
    const uint32_t *ex_off;         /* device, n_reads (exclusive offsets) */
    const int32_t *ex_start, *ex_end;
    const uint8_t *ex_flag;
    const uint32_t *info;
    const int32_t *ref_tx;
    /* Accepted list in HBM.  acc_rec / acc_ex_off: one entry per accepted read; record i has info >> 8 exons starting
     * at acc_ex_off[i] in the three exon arrays.  All five arrays are dense (n_accepted / n_accepted_exons entries, no
     * gaps) but made of one CHUNK PER TILE of reads: inside a chunk the reads are in read order, the chunks follow each
     * other in the order the kernels handed them out.  A consumer that needs read order sorts by the 64-bit read index
     * of the records (l2r_download_accepted() does, and lays the exons out record by record); always address the exon
     * arrays through acc_ex_off. */
    const l2r_accepted_read *acc_rec;
    const uint32_t *acc_ex_off;     /* device, n_accepted */
    const int32_t *acc_ex_start, *acc_ex_end;
    const uint8_t *acc_ex_flag;
} l2r_device_view;

/* Average device time per kernel of the last l2r_run_timed(), milliseconds.  Stages 0..2 are the three kernels of the
 * pipeline the engine chose for the uploaded records (l2r_stage_kernel() names them):
 *     slab    (coordinate-sorted records; default)  0 k_walk_slab (the tile's reads by CIGAR length, CIGAR -> exons; long CIGARs:
 *             k_walk_slab_long, one wave per read as a scan over the op stream,
 *             read-order places)  1 k_describe_scan (the tiles' descriptors and windows, one wave per tile; the tiles' exon counts -> their
 *             first result slots; the tile lists of the wide / chunked kernels)  2 k_probe_slab
 *             (annotation window, site probes, verdicts, read-order results) + k_probe_slab_wide (tiles whose window holds 33 .. 63
 *             transcripts) + k_probe_slab_chunked (tiles beyond that, or with a dictionary key in several entries: the window 63
 *             members at a time).  Every run launches all of them: nothing is kept from an earlier run of the same records.
 *     tile    (coordinate-sorted records with short CIGARs, -e >= 1, an upload that carries the tile index -- i.e. not a single-run one,
 *             l2r_hint_single_run / l2r_classify; default where it applies)  0 k_describe_scan (the tiles' descriptors
 *             and windows from the spans the upload recorded; first kernel of the run)  1 k_tile (CIGAR -> exons, window, probes, verdicts,
 *             junction check, read-order results: one workgroup per tile, nothing handed over through HBM)  2 k_probe_slab for the few
 *             tiles k_tile left in slab form + k_tile's WIDE instance (windows of 33 .. 63 transcripts) + k_tile_chunk (windows beyond that) -- both launched beside
 *             k_tile on streams of their own, so with per-stage events they count here and in stage 1 one behind the other -- + k_probe_slab_wide + k_probe_slab_chunked
 *             (each not launched once a run has shown its list empty)
 *     classic (unsorted records, long CIGARs with -e < 1, L2R_PIPELINE=classic)  0 k_pass_a  1 k_scan_u32 (tile sums)  2 k_classify_fast */
#define L2R_N_STAGES 8
typedef struct {
    float stage_ms[L2R_N_STAGES];   /* 0..2 see above  3 classify_generic (redo list) 4 validate_junctions (+ recount)
                                       5 scan of accepted counts 6 gather_accepted (records; exons of the tiles
                                       the classification did not compact itself) 7 reserved */
    float total_ms;                 /* first launch -> last completion, per iteration */
    int32_t iters;
} l2r_timing;

int          l2r_abi_version(void);
const char  *l2r_last_error(void);
int          l2r_device_count(void);

l2r_ctx     *l2r_create(int device);
void         l2r_destroy(l2r_ctx *ctx);

int          l2r_set_params(l2r_ctx *ctx, const l2r_params *prm);
/* Which outputs a run produces (default: both).  L2R_WANT_RESULTS = the per-read arrays of l2r_result (what
 * check_trans() leaves in bam_T: detail.txt, -a/-k/-u, summary.txt need them); L2R_WANT_ACCEPTED = the compacted
 * accepted-novel records of l2r_accepted (what check_trans() hands to novel_T / merge_trans, update_gtf.c:946-960:
 * all that `update-gtf ... > new.gtf`, -v and -E need; the message of the multi-GPU all-gatherv). */
#define L2R_WANT_RESULTS  1u
#define L2R_WANT_ACCEPTED 2u
int          l2r_set_outputs(l2r_ctx *ctx, unsigned want);
int          l2r_set_annotation(l2r_ctx *ctx, const l2r_annotation *anno);
/* The tables l2r_set_annotation derives from the annotation (transcript headers, cursor keys, the two site dictionaries and
 * their bucket directories: two sorts of ~1.5 M rows each and the dictionary build for a GENCODE-size GTF) can be kept
 * on disk between runs: `dir` (or the environment variable L2R_ANNO_CACHE when no call is made; NULL / "" = off, the
 * default) holds one file per annotation, named by a hash of the arrays passed in.  A hit replaces the build by one
 * read; a file that does not match in any respect is ignored and rewritten.  l2r_annotation_cache_state(): what the last
 * l2r_set_annotation did -- 0 no cache, 1 built and stored, 2 read from the cache.  (The reference re-reads and re-sorts
 * the GTF on every invocation, src/gtf.c:468-521; the pipeline calls update-gtf twice per sample on one GTF.) */
int          l2r_set_annotation_cache(l2r_ctx *ctx, const char *dir);
int          l2r_annotation_cache_state(l2r_ctx *ctx);
int          l2r_set_junctions(l2r_ctx *ctx, const l2r_junctions *sj);   /* NULL or n == 0: no -j file */

/* host -> HBM; detects whether the records are coordinate sorted and, if not,
 * prepares the history-dependent cursor values on the host. */
int          l2r_upload_reads(l2r_ctx *ctx, const l2r_reads *reads);
/* GPU time of the last upload's tile index (k_tile_index; 0 where the upload made none), by HIP events on the context stream: the part of
 * an upload that is kernel work on the records -- bench.py adds it to a first run for the cost of ONE classification of fresh input. */
float        l2r_upload_index_ms(l2r_ctx *ctx);
/* on != 0: every following upload will be classified ONCE (what the CLI does: one l2r_run per l2r_upload_reads).  Such an upload makes
 * no tile index and its run takes the two-kernel pipeline -- by total GPU time the cheaper way for a single run (l2r_classify does the
 * same by itself); the one-kernel tile path is for uploads that are run again (parameter sweeps, the benchmark's resident steps). */
int          l2r_hint_single_run(l2r_ctx *ctx, int on);

/* The hot path on resident inputs; asynchronous on the context stream. */
int          l2r_run(l2r_ctx *ctx);          /* one pass of the path over the resident upload; results in read order in HBM.  (Launches that a
                                              * completed run of the same upload and parameters has shown to be empty -- list kernels, the generic
                                              * kernel -- are dropped from later runs; the one-kernel tile path reads the index the upload made.) */
int          l2r_sync(l2r_ctx *ctx);
/* `iters` back-to-back runs bracketed by HIP events on the context stream. */
int          l2r_run_timed(l2r_ctx *ctx, int iters, l2r_timing *out);
const char  *l2r_stage_kernel(l2r_ctx *ctx, int stage);                 /* kernel behind stage_ms[stage] ("" = none) */

int          l2r_result_sizes(l2r_ctx *ctx, int64_t *n_reads, int64_t *n_exons,
                              int64_t *n_accepted, int64_t *n_accepted_exons);
int          l2r_download(l2r_ctx *ctx, l2r_result *res);               /* HBM -> host */
int          l2r_download_accepted(l2r_ctx *ctx, l2r_accepted *acc);
int          l2r_device_view_get(l2r_ctx *ctx, l2r_device_view *view);
void        *l2r_stream(l2r_ctx *ctx);                                  /* hipStream_t */

/* ---- several GPUs of one node, one process (rank) per GPU: the gathered route's exchange over RCCL (xGMI).
 * Inputs that cannot be cut into independent shards (-s with a junction table: split pieces are compared across chromosomes,
 * src/update_gtf.c:837-913,946-960) are classified shard by shard and their per-read results gathered on rank 0, which runs the
 * order-dependent tail once over the whole read-order set.  Rank 0 makes the id (l2r_xchg_unique_id: l2r_xchg_id_bytes() bytes) and
 * the caller carries it to the other ranks; every rank then creates its end (collective) and, behind l2r_run + l2r_sync, calls
 * l2r_xchg_gather_results (collective): `res` (rank 0 only; capacities as for l2r_download, for the reads / exons of ALL ranks)
 * receives the results of every rank as one set with global exon offsets; counts_out (may be NULL) gets {reads, exons} per rank.
 * l2r_xchg_gather_accepted (collective; L2R_WANT_ACCEPTED): the accepted-novel records alone -- SURVEY 8(e)'s all-gatherv message -- of
 * every rank to rank 0 in read order (`acc`, rank 0 only, capacities for all ranks; read indices are the global ones the uploads'
 * first_read_index gave): all the order-dependent merge needs when no output wants every read; 16 + 9 n bytes per accepted read. */
typedef struct l2r_xchg l2r_xchg;
int          l2r_xchg_id_bytes(void);
int          l2r_xchg_unique_id(void *id_out);
l2r_xchg    *l2r_xchg_create(l2r_ctx *ctx, int rank, int world, const void *id);
int          l2r_xchg_gather_results(l2r_xchg *x, l2r_result *res, int64_t *counts_out);
int          l2r_xchg_gather_accepted(l2r_xchg *x, l2r_accepted *acc, int64_t *counts_out);
void         l2r_xchg_destroy(l2r_xchg *x);

/* ---- diagnostics (tools/, bench.py and the tests read them; no product path does).
 * l2r_debug_counters: out[0] reads the last run left to the generic kernel (the redo list), [1] dictionary entries whose key has
 *   several entries, [2] annotation transcripts the mask kernels take, [3] tiles, [4..11] tiles by the reason their descriptor
 *   is not on the 32-bit masks (0 = it is), [12] tiles of the 64-bit-mask kernel, [13] runs that l2r_sync did again on the slab pipeline
 *   because a tile of k_tile had waited in vain for the exon counts in front of it, [14] entries of the chunked kernel's list that k_tile_chunk
 *   declined in the last run (its staging caps), [15] tiles handed to the chunked kernel late (a key in several entries); n = words of out
 *   (4, 12, 13, 14 or 16).
 * l2r_debug_stamps: with L2R_STAMPS=1 in the environment at l2r_upload_reads, per-phase cycle sums of the classification kernel
 *   (and clears them); zeros otherwise.
 * l2r_debug_tile_times: with L2R_STAMPS=1, one-kernel tile path: four words per tile -- the chip's 100 MHz clock at the tile's start
 *   << 3 | its XCD, at the publication of its exon count, at the begin and the end of its wait for the counts in front. */
int          l2r_debug_counters(l2r_ctx *ctx, long long *out, int n);
int          l2r_debug_stamps(l2r_ctx *ctx, unsigned long long *out, int n);
int          l2r_debug_tile_times(l2r_ctx *ctx, uint32_t *out, int64_t n_tiles);

/* upload + run + sync + download in one call (the host CLI uses this). */
int          l2r_classify(l2r_ctx *ctx, const l2r_reads *reads, l2r_result *res);

/* ---- `filter` (src/bam_filter.c): the per-record test + score and the per-read choice of the best alignment.
 * Replaces gtf_filter() :61-86 with remove_overlap() :48-59, and the selection loop of bam_filter() :128-154; reading the
 * records and writing the BAM stay with the caller (host/filter.c). */
typedef struct { float cov_rate, map_qual, sec_rat; int32_t min_intron_n; } l2r_filter_params;   /* -v 0.67  -q 0.75  -s 0.98  -i 0 */
typedef struct {
    int64_t n, n_cigar;
    const uint16_t *flag;           /* FLAG */
    const int32_t *tid, *pos;       /* core.tid, core.pos (0-based) */
    const int32_t *l_qseq;          /* core.l_qseq */
    const int32_t *nm;              /* bam_aux2i() of the NM tag: its value for the integer types, 0 for any other type */
    const int64_t *cig_off;         /* n + 1 */
    const uint32_t *cig;            /* len << 4 | op */
} l2r_filter_records;
/* transcripts of the -r GTF in FILE order, as read_anno_trans() leaves them (tid -1: chromosome not in the header) */
typedef struct { int64_t n; const int32_t *tid, *start, *end; } l2r_filter_spans;
/* drop[i] = gtf_filter() != 0; score[i] / intron_n[i] = its two outputs (score 0 for dropped records) */
int          l2r_filter_score(l2r_ctx *ctx, const l2r_filter_records *recs, const l2r_filter_params *prm,
                              const l2r_filter_spans *remove /* NULL: no -r */, uint8_t *drop, int32_t *score, int32_t *intron_n);
/* Groups = runs of consecutive KEPT records with one read name (the dropped ones are invisible to bam_filter()'s loop);
 * group g covers rows [group_off[g], group_off[g+1]) of score / intron_n (the kept records, in input order).
 * winner[g] = row of the alignment that is written, or -1 when the read is not retained. */
int          l2r_filter_select(l2r_ctx *ctx, int64_t n_groups, const int64_t *group_off, const int32_t *score,
                               const int32_t *intron_n, const l2r_filter_params *prm, int64_t *winner);

#ifdef __cplusplus
}
#endif
#endif
