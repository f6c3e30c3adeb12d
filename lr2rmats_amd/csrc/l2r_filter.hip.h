// l2r_filter.hip.h -- the two kernels of `filter` (reference src/bam_filter.c:61-164): the per-record test + score
// (gtf_filter :61-86, remove_overlap :48-59) and the per-read choice of the best alignment (bam_filter :128-154).
//
//   k_filter_score    one thread per alignment record: one pass over its CIGAR words (introns, deleted bases, clipped
//                     ends, reference length), the coverage / identity tests in the reference's own arithmetic (double
//                     for the coverage ratio, float for the identity product), the overlap test against the -r GTF
//   k_filter_select   one thread per group of consecutive kept records with one read name: best and second-best score
//                     in record order (the first of equal scores wins), then the reference's two conditions
//
// HBM-bound integer work: the CIGAR words are read once (4 bytes per operation), 15 bytes per record come in and 9 go out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace l2r {

struct FilterPrm { float cov_rate, map_qual, sec_rat; int32_t min_intron_n; };

// Transcripts of the -r GTF a record of chromosome `tid` can meet: remove_overlap() walks the transcripts in FILE order
// and stops at the first one of a larger tid, so they are the transcripts of that tid in front of that stop.  Per tid they
// are kept sorted by start with the running maximum of their ends: "some transcript with start <= hi and end >= lo" is one
// binary search (the reference compares the 0-based position with 1-based transcript coordinates: kept as is).
struct FilterSpans { const int64_t *off; const int32_t *start, *pmax_end; int32_t n_tid; };

__global__ __launch_bounds__(256)
void k_filter_score(int64_t n, const uint16_t *__restrict__ flag, const int32_t *__restrict__ tid, const int32_t *__restrict__ pos,
                    const int32_t *__restrict__ l_qseq, const int32_t *__restrict__ nm, const int64_t *__restrict__ cig_off,
                    const uint32_t *__restrict__ cig, FilterPrm p, FilterSpans sp,
                    uint8_t *__restrict__ drop, int32_t *__restrict__ score, int32_t *__restrict__ intron_n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t d = 0; int32_t sc = 0, in = 0;
    if (flag[i] & 4u) d = 1;                                       // bam_unmap
    else {
        const int64_t c0 = cig_off[i], c1 = cig_off[i + 1];
        int32_t del = 0, rlen = 0;
        for (int64_t k = c0; k < c1; ++k) {
            const uint32_t w = cig[k], op = w & 0xfu; const int32_t len = (int32_t)(w >> 4);
            in += op == 3u;                                        // BAM_CREF_SKIP
            del += op == 2u ? len : 0;                             // BAM_CDEL
            rlen += ((0x18du >> op) & 1u) ? len : 0;               // M D N = X consume the reference (bam_cigar2rlen)
        }
        const int32_t lq = l_qseq[i];
        int32_t qlen = lq;
        if (c1 > c0) {
            const uint32_t w0 = cig[c0], w1 = cig[c1 - 1];
            if ((w0 & 0xfu) == 4u || (w0 & 0xfu) == 5u) qlen -= (int32_t)(w0 >> 4);
            if (c1 - c0 > 1 && ((w1 & 0xfu) == 4u || (w1 & 0xfu) == 5u)) qlen -= (int32_t)(w1 >> 4);
        }
        if (((double)qlen + 0.0) / (double)lq < (double)p.cov_rate) d = 1;
        else {
            const int32_t s = qlen - nm[i] + del;
            if ((float)s < p.map_qual * (float)qlen) d = 1;
            else {
                const int32_t t = tid[i];
                if (sp.n_tid > 0 && t >= 0 && t < sp.n_tid) {
                    // some transcript with end >= pos and start <= pos + rlen - 1
                    const int64_t a = sp.off[t], b = sp.off[t + 1];
                    const int32_t lo = pos[i], hi = pos[i] + rlen - 1;
                    int64_t l = a, r = b;                          // first index with start > hi
                    while (l < r) { const int64_t m = (l + r) >> 1; if (sp.start[m] <= hi) l = m + 1; else r = m; }
                    if (l > a && sp.pmax_end[l - 1] >= lo) d = 1;
                }
                sc = s;
            }
        }
    }
    drop[i] = d; score[i] = d ? 0 : sc; intron_n[i] = in;
}

__global__ __launch_bounds__(256)
void k_filter_select(int64_t n_groups, const int64_t *__restrict__ group_off, const int32_t *__restrict__ score,
                     const int32_t *__restrict__ intron_n, FilterPrm p, int64_t *__restrict__ winner)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int64_t a = group_off[g], b = group_off[g + 1];
    int64_t best = a;
    int32_t b_score = score[a], s_score = 0;
    for (int64_t k = a + 1; k < b; ++k) {
        const int32_t s = score[k];
        if (s > b_score) { s_score = b_score; b_score = s; best = k; }
        else if (s > s_score) s_score = s;
    }
    const bool keep = (float)s_score < p.sec_rat * (float)b_score && intron_n[best] >= p.min_intron_n;
    winner[g] = keep ? best : -1;
}

}  // namespace l2r
