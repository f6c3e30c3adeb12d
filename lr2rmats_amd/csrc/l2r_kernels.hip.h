// l2r_kernels.hip.h -- gfx950 device code of the read-vs-annotation path.
//
// Work decomposition (all int32 interval arithmetic, no contraction -> no MFMA):
//   * a TILE is up to 256 consecutive alignment records handled by one 256-thread
//     workgroup (4 wave64), one thread per record;
//   * k_count_exons   : CIGAR -> exon count per record + per-tile sums
//   * k_scan_tiles    : exclusive scan of the per-tile sums (one workgroup)
//   * k_fill_classify : CIGAR -> exons into an LDS tile, annotation sweep with the
//                       reference's early-exit rules, flag bytes, coalesced write-out
//   * k_validate_sj   : short-read junction support for accepted candidates
//   * k_count_accepted / k_gather_accepted : wave-ballot + prefix-sum compaction of
//                       the accepted-novel records in read order
//
// Semantics follow the reference line by line where bytes of the graded outputs
// depend on it; each device function cites the lines it restates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace l2r {

constexpr int TILE_THREADS = 256;
constexpr int WAVE = 64;
constexpr int LDS_EXON_CAP = 4096;      // exons of one tile staged in LDS (9 B each)

struct TxHdr {                          // 32 B, one annotation transcript (file order)
    int32_t tid, start, end, ex_off;
    int32_t n, rev, mono, pad;
};

struct DevParams {
    int32_t min_exon, min_intron, max_delet, ss_dis;
    int32_t full_level, use_multi, min_sj_cnt, split_trans;
    float   frac;
    int32_t n_tx, n_sj, reads_per_tile;
};

// info / exon-flag bit layout: keep in sync with include/lr2rmats_hip.h
constexpr uint32_t I_KNOWN = 1u, I_KSITE = 2u, I_FULL = 4u, I_REV = 8u, I_UNREL = 16u,
                   I_SJCHK = 32u, I_SJPASS = 64u, I_ACCEPT = 128u;
constexpr uint8_t F_EXON = 1, F_DON = 2, F_ACC = 4, F_JUNC = 8, F_UNREL = 16;

__device__ __forceinline__ int64_t pack_key(int32_t tid, int32_t x)
{
    return ((int64_t)(tid + 1) << 32) | (uint32_t)x;
}

// ------------------------------------------------------------------ block primitives

// Exclusive scan over the 256 threads of a workgroup; returns the exclusive prefix,
// `total` receives the workgroup sum.  Wave-level shuffles + 4 wave totals in LDS.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_wave /*[4]*/, uint32_t &total)
{
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, WAVE);
        if (lane >= d) inc += t;
    }
    if (lane == WAVE - 1) s_wave[w] = inc;
    __syncthreads();
    uint32_t w0 = s_wave[0], w1 = s_wave[1], w2 = s_wave[2], w3 = s_wave[3];
    uint32_t base = (w > 0 ? w0 : 0) + (w > 1 ? w1 : 0) + (w > 2 ? w2 : 0);
    total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + inc - v;
}

// ------------------------------------------------------------------ CIGAR -> exons

// src/bam2gtf.c:31-78 gen_exon.  EMIT(start,end) is called for every exon kept.
template <typename Emit>
__device__ __forceinline__ int walk_cigar(const uint32_t *cig, int n_cig, int pos0, const DevParams &p, Emit emit)
{
    int start = pos0 + 1, end = start - 1, n = 0;
    for (int k = 0; k < n_cig; ++k) {
        const uint32_t c = cig[k];
        const int len = (int)(c >> 4);
        const uint32_t op = c & 0xfu;
        // N (3) cuts at len >= min_intron, D (2) at len > max_delet; M,=,X,N,D advance the reference
        const bool cut = (op == 3u && len >= p.min_intron) || (op == 2u && len > p.max_delet);
        if (cut) {
            if (n == 0 || end - start + 1 >= p.min_exon) { emit(n, start, end); ++n; }
            start = end + len + 1;
        }
        if (op == 0u || op == 2u || op == 3u || op == 7u || op == 8u) end += len;
    }
    emit(n, start, end);
    return n + 1;
}

__global__ __launch_bounds__(TILE_THREADS)
void k_count_exons(int64_t n_reads, const int32_t *__restrict__ r_pos, const int64_t *__restrict__ cig_off,
                   const uint32_t *__restrict__ cig, DevParams p,
                   uint32_t *__restrict__ n_ex, uint32_t *__restrict__ tile_sum)
{
    __shared__ uint32_t s_wave[4];
    const int64_t r = (int64_t)blockIdx.x * p.reads_per_tile + threadIdx.x;
    uint32_t n = 0;
    if (threadIdx.x < p.reads_per_tile && r < n_reads) {
        const int64_t a = cig_off[r], b = cig_off[r + 1];
        n = (uint32_t)walk_cigar(cig + a, (int)(b - a), r_pos[r], p, [](int, int, int) {});
        n_ex[r] = n;
    }
    uint32_t total;
    block_exclusive_scan(n, s_wave, total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

// In-place exclusive scan of `n` uint32 by ONE workgroup of 1024 threads; *total = sum.
__global__ __launch_bounds__(1024)
void k_scan_tiles(uint32_t *__restrict__ v, int64_t n, uint32_t *__restrict__ total)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const uint32_t x = i < n ? v[i] : 0u;
        uint32_t inc = x;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            uint32_t t = __shfl_up(inc, d, WAVE);
            if (lane >= d) inc += t;
        }
        if (lane == WAVE - 1) s_wave[w] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint32_t t = s_wave[k]; if (k < w) wbase += t; tot += t; }
        const uint32_t carry = s_carry;
        if (i < n) v[i] = carry + wbase + inc - x;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

// ------------------------------------------------------------------ comparison rules

__device__ __forceinline__ bool near_eq(int a, int b, int dis) { return __builtin_abs(a - b) <= dis; }
__device__ __forceinline__ bool closed_overlap(int s1, int e1, int s2, int e2)
{   // src/update_gtf.c:91-95 exon_overlap
    return !(s1 > e2 || s2 > e1);
}

// src/update_gtf.c:80-89 exon_overlap_frac: int / double, rounded to float, compared as float (Q6)
__device__ __forceinline__ float overlap_frac(int s1, int e1, int s2, int e2)
{
    if (s1 > e2 || s2 > e1) return 0.0f;
    const int ov = min(e1, e2) - max(s1, s2) + 1;
    const int ml = min(e1 - s1 + 1, e2 - s2 + 1);
    return (float)((double)ov / ((double)ml + 0.0));
}

struct ReadState {
    bool lfull, rfull, lnoth, rnoth, known, ksite;
};

// src/update_gtf.c:629-681 check_full for one overlapping annotation transcript
__device__ __forceinline__ void full_evidence(ReadState &st, int level, const int *S, const int *E, int n,
                                              const int2 *__restrict__ ax, int m)
{
    if (st.lfull && st.rfull) return;
    const int2 a0 = ax[0], al = ax[m - 1];
    if (level == 1) {
        if (!st.lfull && E[0] == a0.y) st.lfull = true;
        if (!st.rfull && S[n - 1] == al.x) st.rfull = true;
    } else if (level == 2) {
        if (!st.lfull && closed_overlap(S[0], E[0], a0.x, a0.y)) st.lfull = true;
        if (!st.rfull && closed_overlap(S[n - 1], E[n - 1], al.x, al.y)) st.rfull = true;
    } else if (level == 3 || level == 4) {
        if (!st.lfull) {
            const int s = S[0], e = E[0];
            if (closed_overlap(s, e, a0.x, a0.y)) st.lfull = true;
            else if (st.lnoth)
                for (int k = 0; k < m; ++k) { const int2 a = ax[k]; if (closed_overlap(s, e, a.x, a.y)) { st.lnoth = false; break; } }
        }
        if (level == 3 && !st.rfull) {
            const int s = S[n - 1], e = E[n - 1];
            if (closed_overlap(s, e, al.x, al.y)) st.rfull = true;
            else if (st.rnoth)
                for (int k = 0; k < m; ++k) { const int2 a = ax[k]; if (closed_overlap(s, e, a.x, a.y)) { st.rnoth = false; break; } }
        }
    }
}

// src/update_gtf.c:717-779 check_splice_site, all four double loops folded into one
// (i over annotation exons, j over read exons); clears are idempotent and the
// counters are plain sums, so the visiting order does not matter.  The acceptor
// comparison uses the read exon j start, j < n-1 (Q1, :746).
// returns 1 known, 2 has known site, 0 neither.
__device__ __forceinline__ int site_compare(const int *S, const int *E, uint8_t *F, int n, int r_start, int r_end,
                                            const TxHdr &a, const int2 *__restrict__ ax, int dis)
{
    const int lo = max(r_start, a.start), hi = min(r_end, a.end);
    int r_in = 0, same = 0;
    for (int j = 0; j + 1 < n; ++j) {
        const int e = E[j], s2 = S[j + 1];
        r_in += (e >= lo && e <= hi) + (s2 >= lo && s2 <= hi);
    }
    const int m = a.n;
    int2 cur = ax[0];
    for (int i = 0; i < m; ++i) {
        const bool has_next = i + 1 < m;
        const int2 nxt = has_next ? ax[i + 1] : cur;
        const bool don_in = has_next && cur.y >= lo && cur.y <= hi;
        const bool acc_in = has_next && nxt.x >= lo && nxt.x <= hi;
        for (int j = 0; j < n; ++j) {
            const int sj = S[j], ej = E[j];
            const bool end_eq = near_eq(cur.y, ej, dis);
            uint8_t clr = 0;
            if (end_eq && near_eq(cur.x, sj, dis)) clr |= F_EXON;
            if (has_next && j + 1 < n) {
                if (don_in && end_eq) { ++same; clr |= F_DON; }
                if (acc_in && near_eq(nxt.x, sj, dis)) { ++same; clr |= F_ACC; }
                if (end_eq && near_eq(nxt.x, S[j + 1], dis)) clr |= F_JUNC;
            }
            if (clr) F[j] &= (uint8_t)~clr;
        }
        cur = nxt;
    }
    if (2 * (n - 1) == r_in && r_in == same) return 1;
    return same > 0 ? 2 : 0;
}

// src/update_gtf.c:792-835 check_with_anno_trans for one read whose exons (S,E) and
// flag bytes F are addressable.  j0 = cursor value the sequential code would have
// (SURVEY.md 3.3).  Returns info bits (without exon count), ref in `ref`.
__device__ __forceinline__ uint32_t sweep_annotation(const int *S, const int *E, uint8_t *F, int n, int tid, bool rev,
                                                     int j0, const TxHdr *__restrict__ hdr, const int2 *__restrict__ anno_ex,
                                                     const DevParams &p, int &ref)
{
    const int r_start = S[0], r_end = E[n - 1];
    ReadState st{false, false, true, true, false, false};
    ref = -1;
    for (int j = j0; j < p.n_tx; ++j) {
        const int4 h0 = reinterpret_cast<const int4 *>(hdr + j)[0];
        TxHdr a; a.tid = h0.x; a.start = h0.y; a.end = h0.z; a.ex_off = h0.w;
        // src/update_gtf.c:786-790 comp_trans: <= (Q5)
        if (tid < a.tid || (tid == a.tid && r_end <= a.start)) break;
        if (a.tid < tid || (a.tid == tid && a.end <= r_start)) continue;
        const int4 h1 = reinterpret_cast<const int4 *>(hdr + j)[1];
        a.n = h1.x; a.rev = h1.y; a.mono = h1.z;
        const int2 *ax = anno_ex + a.ex_off;
        full_evidence(st, p.full_level, S, E, n, ax, a.n);
        if (n == 1 && a.n == 1) {
            const int2 a0 = ax[0];
            if (overlap_frac(S[0], E[0], a0.x, a0.y) >= p.frac) { ref = j; st.known = true; break; }
        } else if (n > 1 && a.n > 1) {
            const int v = site_compare(S, E, F, n, r_start, r_end, a, ax, p.ss_dis);
            if (v == 1) { st.known = true; ref = j; break; }
            if (v == 2) { st.ksite = true; ref = j; }
        }
    }
    bool out_rev = rev;
    if (ref >= 0) out_rev = hdr[ref].rev != 0;          // :825-831 strand taken from the reference transcript
    bool full;                                           // :683-696 set_full
    if (p.full_level == 5) full = true;
    else if (p.full_level == 4) full = st.lfull || st.lnoth;
    else if (p.full_level == 3) full = (st.lfull || st.lnoth) && (st.rfull || st.rnoth);
    else full = st.lfull && st.rfull;
    uint32_t info = 0;
    if (st.known) info |= I_KNOWN;
    if (st.ksite) info |= I_KSITE;
    if (full) info |= I_FULL;
    if (out_rev) info |= I_REV;
    return info;
}

__device__ __forceinline__ int first_key_above(const int64_t *__restrict__ key, int n, int64_t q)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key[mid] > q) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// One read: exons -> (S,E,F) at the given pointers, then the annotation sweep.
__device__ __forceinline__ uint32_t process_read(int *S, int *E, uint8_t *F, int n, int32_t tid, int32_t pos0, bool rev,
                                                 const uint32_t *cig, int n_cig, int64_t r,
                                                 const int64_t *__restrict__ anno_key, const int32_t *__restrict__ win_start,
                                                 const TxHdr *__restrict__ hdr, const int2 *__restrict__ anno_ex,
                                                 const DevParams &p, int &ref)
{
    walk_cigar(cig, n_cig, pos0, p, [&](int k, int s, int e) {
        S[k] = s; E[k] = e;
    });
    for (int k = 0; k < n; ++k) F[k] = (k + 1 < n) ? (uint8_t)(F_EXON | F_DON | F_ACC | F_JUNC) : F_EXON;
    const int j0 = win_start ? win_start[r] : first_key_above(anno_key, p.n_tx, pack_key(tid, S[0]));
    uint32_t info = sweep_annotation(S, E, F, n, tid, rev, j0, hdr, anno_ex, p, ref);
    // routing of update_gtf.c:943-950 when there is no junction table
    if (p.n_sj == 0 && (info & (I_FULL | I_KNOWN | I_KSITE)) == (I_FULL | I_KSITE)) info |= I_ACCEPT;
    return info | ((uint32_t)n << 8);
}

__global__ __launch_bounds__(TILE_THREADS)
void k_fill_classify(int64_t n_reads, const int32_t *__restrict__ r_tid, const int32_t *__restrict__ r_pos,
                     const uint8_t *__restrict__ r_rev, const int64_t *__restrict__ cig_off, const uint32_t *__restrict__ cig,
                     const uint32_t *__restrict__ n_ex, const uint32_t *__restrict__ tile_base,
                     const int64_t *__restrict__ anno_key, const int32_t *__restrict__ win_start,
                     const TxHdr *__restrict__ hdr, const int2 *__restrict__ anno_ex, DevParams p,
                     uint32_t *__restrict__ ex_off, int32_t *__restrict__ ex_start, int32_t *__restrict__ ex_end,
                     uint8_t *__restrict__ ex_flag, uint32_t *__restrict__ info_out, int32_t *__restrict__ ref_out)
{
    __shared__ uint32_t s_wave[4];
    __shared__ int s_start[LDS_EXON_CAP];
    __shared__ int s_end[LDS_EXON_CAP];
    __shared__ uint8_t s_flag[LDS_EXON_CAP];

    const int64_t r = (int64_t)blockIdx.x * p.reads_per_tile + threadIdx.x;
    const bool active = threadIdx.x < p.reads_per_tile && r < n_reads;
    const uint32_t n = active ? n_ex[r] : 0u;
    uint32_t tile_total;
    const uint32_t local = block_exclusive_scan(n, s_wave, tile_total);
    const uint32_t base = tile_base[blockIdx.x];
    uint32_t info = 0; int ref = -1;
    if (tile_total <= (uint32_t)LDS_EXON_CAP) {
        if (active) {
            const int64_t a = cig_off[r], b = cig_off[r + 1];
            info = process_read(s_start + local, s_end + local, s_flag + local, (int)n, r_tid[r], r_pos[r], r_rev[r] != 0,
                                cig + a, (int)(b - a), r, anno_key, win_start, hdr, anno_ex, p, ref);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < tile_total; i += TILE_THREADS) {      // coalesced write-out
            ex_start[base + i] = s_start[i];
            ex_end[base + i] = s_end[i];
            ex_flag[base + i] = s_flag[i];
        }
    } else if (active) {
        // oversize tile (very long exon chains): work directly on the output arrays in HBM
        const int64_t a = cig_off[r], b = cig_off[r + 1];
        info = process_read(ex_start + base + local, ex_end + base + local, ex_flag + base + local, (int)n, r_tid[r], r_pos[r],
                            r_rev[r] != 0, cig + a, (int)(b - a), r, anno_key, win_start, hdr, anno_ex, p, ref);
    }
    if (active) {
        ex_off[r] = base + local;
        info_out[r] = info;
        ref_out[r] = ref;
    }
}

// ------------------------------------------------------------------ short-read junction support

// src/update_gtf.c:589-603 check_short_sj1 with the linear scan from the cursor row
// replaced by a lower-bound on (tid, don): rows below don-dis cannot match, and the
// reference stops at the first row with don >= acc (intron end).
__device__ __forceinline__ bool junction_supported(int tid, int don, int acc, int from, const int32_t *__restrict__ sj_tid,
                                                   const int32_t *__restrict__ sj_don, const int32_t *__restrict__ sj_acc,
                                                   const int32_t *__restrict__ sj_uniq, const int32_t *__restrict__ sj_multi,
                                                   const DevParams &p)
{
    int lo = from, hi = p.n_sj;
    const int want = don - p.ss_dis;
    // (a degenerate intron with acc < don, or a negative -d, keeps the literal linear scan:
    //  only then could a skipped row have triggered the reference's early "don >= acc" stop)
    if (acc < don || p.ss_dis < 0) hi = lo;
    while (lo < hi) {                       // first row >= (tid, want) at or after `from`
        const int mid = (lo + hi) >> 1;
        const int t = sj_tid[mid];
        if (t < tid || (t == tid && sj_don[mid] < want)) lo = mid + 1; else hi = mid;
    }
    for (int i = lo; i < p.n_sj; ++i) {
        const int t = sj_tid[i], d = sj_don[i];
        if (t > tid || (t == tid && d >= acc)) return false;
        if (p.ss_dis >= 0 && d - don > p.ss_dis) return false;   // sorted by don: nothing further can match
        if (near_eq(d, don, p.ss_dis) && near_eq(sj_acc[i], acc, p.ss_dis)) {
            const int c = p.use_multi ? sj_uniq[i] + sj_multi[i] : sj_uniq[i];
            if (c >= p.min_sj_cnt) return true;
        }
    }
    return false;
}

// src/update_gtf.c:698-709 check_with_short_sj + :609-627 check_short_sj for every
// read that reaches it (full, not known, has a known site), one thread per read.
__global__ __launch_bounds__(TILE_THREADS)
void k_validate_sj(int64_t n_reads, const int32_t *__restrict__ r_tid, const uint32_t *__restrict__ ex_off,
                   const int32_t *__restrict__ ex_start, const int32_t *__restrict__ ex_end, uint8_t *__restrict__ ex_flag,
                   const int64_t *__restrict__ sj_key, const int32_t *__restrict__ sj_cursor,
                   const int32_t *__restrict__ sj_tid, const int32_t *__restrict__ sj_don, const int32_t *__restrict__ sj_acc,
                   const int32_t *__restrict__ sj_uniq, const int32_t *__restrict__ sj_multi, DevParams p,
                   uint32_t *__restrict__ info_io)
{
    const int64_t r = (int64_t)blockIdx.x * TILE_THREADS + threadIdx.x;
    if (r >= n_reads) return;
    uint32_t info = info_io[r];
    if ((info & (I_FULL | I_KNOWN | I_KSITE)) != (I_FULL | I_KSITE)) return;
    const int n = (int)(info >> 8), tid = r_tid[r];
    const uint32_t off = ex_off[r];
    const int r_start = ex_start[off], r_end = ex_end[off + n - 1];
    const int from = sj_cursor ? sj_cursor[r] : first_key_above(sj_key, p.n_sj, pack_key(tid, r_start));
    bool ok = false;
    if (from < p.n_sj) {
        const int t = sj_tid[from];
        // Q7: cursor row beyond the read -> unsupported, no unreliable flag
        if (!(t > tid || (t == tid && sj_don[from] >= r_end))) {
            ok = true;
            for (int j = 0; j + 1 < n; ++j) {
                const uint8_t f = ex_flag[off + j];
                if ((f & F_JUNC) &&
                    !junction_supported(tid, ex_end[off + j] + 1, ex_start[off + j + 1] - 1, from, sj_tid, sj_don, sj_acc, sj_uniq, sj_multi, p)) {
                    ex_flag[off + j] = f | F_UNREL;
                    ok = false;
                }
            }
        }
    }
    info |= I_SJCHK;
    if (ok) info |= I_SJPASS; else info |= I_UNREL;
    if (ok || p.split_trans) info |= I_ACCEPT;
    info_io[r] = info;
}

// ------------------------------------------------------------------ compaction of accepted reads

__global__ __launch_bounds__(TILE_THREADS)
void k_count_accepted(int64_t n_reads, const uint32_t *__restrict__ info, uint32_t *__restrict__ tile_reads,
                      uint32_t *__restrict__ tile_exons)
{
    __shared__ uint32_t s_cnt[4], s_ex[4];
    const int64_t r = (int64_t)blockIdx.x * TILE_THREADS + threadIdx.x;
    const uint32_t w = r < n_reads ? info[r] : 0u;
    const bool acc = (w & I_ACCEPT) != 0;
    const unsigned long long m = __ballot(acc);
    uint32_t ex = acc ? (w >> 8) : 0u;
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) ex += __shfl_down(ex, d, WAVE);
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    if (lane == 0) { s_cnt[wv] = (uint32_t)__popcll(m); s_ex[wv] = ex; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_reads[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        tile_exons[blockIdx.x] = s_ex[0] + s_ex[1] + s_ex[2] + s_ex[3];
    }
}

struct AccRec { uint32_t read_lo, read_hi, info; int32_t ref_tx; };

__global__ __launch_bounds__(TILE_THREADS)
void k_gather_accepted(int64_t n_reads, int64_t first_read, const uint32_t *__restrict__ info, const int32_t *__restrict__ ref_tx,
                       const uint32_t *__restrict__ ex_off, const int32_t *__restrict__ ex_start, const int32_t *__restrict__ ex_end,
                       const uint8_t *__restrict__ ex_flag, const uint32_t *__restrict__ tile_reads, const uint32_t *__restrict__ tile_exons,
                       AccRec *__restrict__ rec, uint32_t *__restrict__ acc_ex_off, int32_t *__restrict__ acc_start,
                       int32_t *__restrict__ acc_end, uint8_t *__restrict__ acc_flag)
{
    __shared__ uint32_t s_wcnt[4], s_wex[4];
    const int64_t r = (int64_t)blockIdx.x * TILE_THREADS + threadIdx.x;
    const uint32_t w = r < n_reads ? info[r] : 0u;
    const bool acc = (w & I_ACCEPT) != 0;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    // rank inside the wave: ballot + popcount of the lanes below
    const unsigned long long m = __ballot(acc);
    const uint32_t rank_w = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    // exon prefix inside the wave: shuffle scan
    const uint32_t nex = acc ? (w >> 8) : 0u;
    uint32_t inc = nex;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) { const uint32_t t = __shfl_up(inc, d, WAVE); if (lane >= d) inc += t; }
    if (lane == WAVE - 1) { s_wcnt[wv] = (uint32_t)__popcll(m); s_wex[wv] = inc; }
    __syncthreads();
    uint32_t cbase = tile_reads[blockIdx.x], ebase = tile_exons[blockIdx.x];
    for (int k = 0; k < wv; ++k) { cbase += s_wcnt[k]; ebase += s_wex[k]; }
    if (!acc) return;
    const uint32_t slot = cbase + rank_w, eo = ebase + inc - nex;
    const uint64_t gidx = (uint64_t)(first_read + r);
    AccRec a; a.read_lo = (uint32_t)gidx; a.read_hi = (uint32_t)(gidx >> 32); a.info = w; a.ref_tx = ref_tx[r];
    rec[slot] = a;
    acc_ex_off[slot] = eo;
    const uint32_t src = ex_off[r];
    for (uint32_t k = 0; k < nex; ++k) {
        acc_start[eo + k] = ex_start[src + k];
        acc_end[eo + k] = ex_end[src + k];
        acc_flag[eo + k] = ex_flag[src + k];
    }
}

}  // namespace l2r
