/* oracle_int.h -- TEST INFRASTRUCTURE ONLY (see oracle.h). Internal types. */
#ifndef LR2RMATS_ORACLE_INT_H
#define LR2RMATS_ORACLE_INT_H
#include <stdint.h>
#include <stdio.h>
#include "oracle.h"

#define ORC_NAME_MAX 100   /* gtf.h:44-45 */

typedef struct {           /* gtf.h:16-22 exon_t */
    int tid, start, end;
    uint8_t rev;
    int score; uint8_t etype;   /* 0 first/last, 1 internal, 2 single */
} orc_exon;

typedef struct {           /* gtf.h:24-28 sj_t (fields the path uses) */
    int tid, don, acc;
    uint8_t rev;
    int uniq_c, multi_c, max_over, score;
} orc_sj;

typedef struct {           /* gtf.h:39-53 trans_t */
    orc_exon *ex; int n, cap;
    int tid; uint8_t rev;
    int start, end;
    char tname[ORC_NAME_MAX], tids[ORC_NAME_MAX], gname[ORC_NAME_MAX], gid[ORC_NAME_MAX];
    int cov;
    uint8_t full, lfull, lnoth, rfull, rnoth;
    uint8_t known, has_known_site, has_unrel, partial;
    uint8_t *nov_exon, *nov_site, *nov_junc, *unrel;
} orc_trans;

typedef struct {           /* gtf.h:55-58 read_trans_t */
    orc_trans *t; int n, cap;
    int gene_n;
} orc_list;

typedef struct {           /* gtf.h:71-74 chr_name_t + bam_hdr_t names */
    char **name; int n, cap;
    int n_hdr;             /* the first n_hdr names come from the SAM header */
} orc_chroms;

/* oracle_core.c */
void orc_die(int core, const char *where, const char *msg);
void orc_set_name(char dst[ORC_NAME_MAX], const char *src, const char *what);
void tr_zero(orc_trans *t);
void tr_release(orc_trans *t);
void tr_push_exon(orc_trans *t, int tid, int start, int end, uint8_t rev);
void tr_finish(orc_trans *t);
void tr_alloc_read_flags(orc_trans *t);
orc_list *ls_new(void);
void ls_push_read(orc_list *l, const orc_trans *s);
void ls_push_anno(orc_list *l, const orc_trans *s);
void ls_free(orc_list *l);
void orc_cigar_to_exons(orc_trans *t, int tid, int pos0, uint8_t rev, const uint32_t *cig, int n_cig,
                        int min_exon, int min_intron, int max_delet);
int orc_sweep_annotation(orc_trans *r, const orc_list *A, int *cursor, const orc_params *p);
int orc_validate_junctions(orc_trans *r, const orc_sj *S, int n, int *cursor, const orc_params *p);
orc_list *orc_split(const orc_trans *r);
int orc_merge(const orc_trans *t, orc_list *U, const orc_params *p);
void orc_check_all(orc_list *R, const orc_list *A, const orc_sj *S, int n_sj,
                   orc_list *updated, orc_list *known, orc_list *novel, orc_list *unrecog, const orc_params *p);

#endif
