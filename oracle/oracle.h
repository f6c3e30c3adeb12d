/* oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the lr2rmats `update-gtf` / `bam2gtf` / `unique-gtf`
 * comparison path, used as the parity checker for the MI355X build.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load or execute anything in this directory; nothing under lr2rmats_amd/
 * links, imports or calls it.
 *
 * PARITY PINNING: the reference ships no tests and no golden outputs, and its
 * htslib dependency is an empty un-vendored submodule, so the reference cannot
 * be compiled in this image without writing a stand-in for htslib (not
 * allowed).  The oracle is therefore pinned only by the known-answer vectors
 * recorded from the reference in SURVEY.md Appendix D.2 (tests/golden/toy).
 * Beyond those vectors: "parity unpinned" -- see DESIGN.md.
 *
 * The algorithm is restated from the reference sources (file:line cited on
 * every function); the sequential, cursor-carrying, array-of-structs form is
 * kept on purpose so that it shares no structure with the device kernels.
 */
#ifndef LR2RMATS_ORACLE_H
#define LR2RMATS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t min_exon;      /* -e, gtf.h:119  */
    int32_t min_intron;    /* -i, gtf.h:118  */
    int32_t max_delet;     /* -t, gtf.h:120  */
    int32_t ss_dis;        /* -d, gtf.h:121  */
    int32_t end_dis;       /* -D, gtf.h:122  */
    int32_t full_level;    /* -l, update_gtf.c:28 */
    int32_t split_trans;   /* -s */
    int32_t use_multi;     /* -M */
    int32_t min_sj_cnt;    /* -J, update_gtf.h:6 */
    int32_t force_strand;  /* -c */
    float   single_exon_ovlp_frac; /* -f, gtf.h:127 */
} orc_params;

/* bits of out_info[] (one word per read) */
#define ORC_INFO_KNOWN      0x01u
#define ORC_INFO_KNOWN_SITE 0x02u
#define ORC_INFO_FULL       0x04u
#define ORC_INFO_REV        0x08u   /* strand after the annotation flip   */
#define ORC_INFO_UNREL      0x10u   /* has_unreliable_junction            */
#define ORC_INFO_SJ_CHECKED 0x20u   /* check_with_short_sj was run        */
#define ORC_INFO_SJ_PASS    0x40u   /* ... and returned 1                 */

/* bits of out_ex_flag[] (one byte per exon j of a read; junction j joins
 * exon j and j+1 and is stored with exon j) */
#define ORC_EXF_NOVEL_EXON  0x01u
#define ORC_EXF_NOVEL_DON   0x02u   /* novel_site_flag[2j]   */
#define ORC_EXF_NOVEL_ACC   0x04u   /* novel_site_flag[2j+1] */
#define ORC_EXF_NOVEL_JUNC  0x08u
#define ORC_EXF_UNREL_JUNC  0x10u

/* Structure-of-arrays entry point (kernel-level parity + cpu_baseline).
 * Runs gen_exon + check_with_anno_trans + check_with_short_sj for every read in
 * input order with the reference's sequential cursors.  Returns the total
 * number of exons written, or <0 on error (-1: exon capacity, -2: bad read). */
int64_t orc_classify_soa(
    int64_t n_reads, const int32_t *r_tid, const int32_t *r_pos, const uint8_t *r_rev,
    const int64_t *cig_off, const uint32_t *cig,
    int64_t n_tx, const int32_t *tx_tid, const int32_t *tx_start, const int32_t *tx_end,
    const uint8_t *tx_rev, const int64_t *tx_ex_off, const int32_t *ex_start, const int32_t *ex_end,
    int64_t n_sj, const int32_t *sj_tid, const int32_t *sj_don, const int32_t *sj_acc,
    const int32_t *sj_uniq, const int32_t *sj_multi,
    const orc_params *prm,
    int64_t ex_cap, int64_t *out_ex_off /* n_reads+1 */, int32_t *out_ex_start, int32_t *out_ex_end,
    uint8_t *out_ex_flag, uint32_t *out_info, int32_t *out_ref_tx);

/* Whole sub-commands, argv as after the program name (argv[0] = sub-command). */
int orc_main(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif
