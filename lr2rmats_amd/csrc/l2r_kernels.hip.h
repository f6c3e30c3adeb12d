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
constexpr int LDS_EXON_CAP = 3072;      // exons of one tile staged in LDS (9 B each)

// One annotation transcript (file order), 96 B = six 16-byte loads; the sweep reads h0 for every
// transcript it passes, h1..h2 for the ones that overlap, h3..h5 only on the dictionary path.
struct TxHdr {
    int32_t tid, start, end, ex_off;          // h0
    int32_t n, rev, flags, pad;               // h1  flags: TX_MONO | TX_COMPACT
    int32_t s0, e0, sl, el;                   // h2  first and last exon
    int32_t gb_d, gb_a, gb_x, gb_j;           // h3  rank of the transcript's first donor / acceptor / exon / junction
    uint32_t md[2], ma[2];                    // h4  site masks relative to those ranks (bit k = rank gb+k is in the transcript)
    uint32_t mx[2], mj[2];                    // h5
};
constexpr int TX_MONO = 1;      // exon starts and ends strictly increasing
constexpr int TX_COMPACT = 2;   // TX_MONO, start <= end for every exon, header span == exon span, all four masks fit 64 bits

// Site dictionaries (built once per annotation on the host).  Every distinct annotation site of a
// kind (donor, acceptor, exon, junction) has a RANK = its index among the sorted distinct sites of
// that kind on all chromosomes.  Two probe structures serve the four kinds:
//   START dictionary: the distinct exons sorted by (tid, start, end); entry {start, end, acceptor rank of
//                     `start` or -1, 0}; the entry's index is the exon rank;
//   END dictionary:   the distinct junctions sorted by (tid, end, next start); entry {end, next start,
//                     donor rank of `end`, 0}; the entry's index is the junction rank;
// each with a directory over 512-bp coordinate buckets: dir[tid_base[tid] + (k1 >> 9)] = first entry of the
// bucket.  Neighbouring coordinates hit neighbouring directory words and entries, so the reads of a tile
// (one locus) keep re-using a handful of lines, which the tile stages in LDS.
constexpr int SITE_SHIFT = 9;
struct SiteDict {
    const int4 *ent;           // entries by rank
    const uint32_t *dir;       // bucket -> first entry; one extra word closes the last bucket
};
struct SiteTabs {
    SiteDict st, en;           // START and END dictionaries
    const int32_t *tid_base;   // [n_tid + 1] first bucket of every tid (one bucket grid for both)
    int32_t n_tid;
};

// Cursor directory: the prefix-max keys of the annotation with the same kind of
// 512-bp bucket directory, so the cursor value of a read costs two directory words and a search inside
// one bucket instead of a 17-step binary search.  dir[kb_base[tid] + c] = first j with key_j >= (tid, c << 9).
struct CursorDir {
    const int64_t *key;        // [n_tx] non-decreasing
    const uint32_t *dir;
    const int32_t *kb_base;    // [n_tid + 1]
    int32_t n_tid, n_tx;
};

struct DevParams {
    int32_t min_exon, min_intron, max_delet, ss_dis;
    int32_t full_level, use_multi, min_sj_cnt, split_trans;
    float   frac;
    int32_t n_tx, n_sj, reads_per_tile;
    int32_t ablate;          // diagnostics (env L2R_ABLATE): 1 no sweep, 2 no site match, 4 no full-length test, 16 no dictionary path
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

// src/bam2gtf.c:31-78 gen_exon.  emit(k, start, end) is called for every exon kept.
// CIGAR words are fetched four at a time so that the loads of one read are in flight together.
template <typename Emit>
__device__ __forceinline__ int walk_cigar(const uint32_t *__restrict__ cig, int n_cig, int pos0, const DevParams &p, Emit emit)
{
    int start = pos0 + 1, end = start - 1, n = 0;
    auto step = [&](uint32_t c) {
        const int len = (int)(c >> 4);
        const uint32_t op = c & 0xfu;
        // N (3) cuts at len >= min_intron, D (2) at len > max_delet; M,=,X,N,D advance the reference
        const bool cut = (op == 3u && len >= p.min_intron) || (op == 2u && len > p.max_delet);
        if (cut) {
            if (n == 0 || end - start + 1 >= p.min_exon) { emit(n, start, end); ++n; }
            start = end + len + 1;
        }
        if (op == 0u || op == 2u || op == 3u || op == 7u || op == 8u) end += len;
    };
    int k = 0;
    for (; k + 4 <= n_cig; k += 4) {
        const uint32_t c0 = cig[k], c1 = cig[k + 1], c2 = cig[k + 2], c3 = cig[k + 3];
        step(c0); step(c1); step(c2); step(c3);
    }
    for (; k < n_cig; ++k) step(cig[k]);
    emit(n, start, end);
    return n + 1;
}

// first j with key_j > (tid, start): the value the reference's annotation cursor has for this read
__device__ __forceinline__ int cursor_value(const CursorDir &cd, int32_t tid, int32_t start)
{
    if (tid >= cd.n_tid) return cd.n_tx;                       // beyond every annotated chromosome
    const int32_t kb = cd.kb_base[tid], nb = cd.kb_base[tid + 1] - kb;
    if (nb <= 0) return (int)cd.dir[kb];
    const int c = min(max(start, 0) >> SITE_SHIFT, nb - 1);
    int lo = (int)cd.dir[kb + c], hi = (int)cd.dir[kb + c + 1];
    if (c == nb - 1) hi = (int)cd.dir[kb + nb];               // last bucket of the chromosome is open ended
    const int64_t q = pack_key(tid, start);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cd.key[mid] > q) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ __launch_bounds__(TILE_THREADS)
void k_count_exons(int64_t n_reads, const int32_t *__restrict__ r_tid, const int32_t *__restrict__ r_pos,
                   const int64_t *__restrict__ cig_off, const uint32_t *__restrict__ cig, CursorDir cd, DevParams p,
                   uint32_t *__restrict__ n_ex, int32_t *__restrict__ j0_out, uint32_t *__restrict__ tile_sum)
{
    __shared__ uint32_t s_wave[4];
    const int64_t r = (int64_t)blockIdx.x * p.reads_per_tile + threadIdx.x;
    uint32_t n = 0;
    if (threadIdx.x < p.reads_per_tile && r < n_reads) {
        const int64_t a = cig_off[r], b = cig_off[r + 1];
        const int32_t pos = r_pos[r];
        if (j0_out) j0_out[r] = cursor_value(cd, r_tid[r], pos + 1);       // first exon always starts at pos + 1
        n = (uint32_t)walk_cigar(cig + a, (int)(b - a), pos, p, [](int, int, int) {});
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

// src/update_gtf.c:629-681 check_full for one overlapping annotation transcript.  The read's and the
// transcript's terminal exons come in registers / from the header; only the "overlaps some other exon"
// scans of levels 3 and 4 touch the transcript's exon array.
struct ReadEnds { int s0, e0, sl, el; };     // first and last exon of the read

__device__ __forceinline__ void full_evidence(ReadState &st, int level, const ReadEnds &r, const TxHdr &a, const int2 *__restrict__ ax)
{
    if (st.lfull && st.rfull) return;
    if (level == 1) {
        if (!st.lfull && r.e0 == a.e0) st.lfull = true;
        if (!st.rfull && r.sl == a.sl) st.rfull = true;
    } else if (level == 2) {
        if (!st.lfull && closed_overlap(r.s0, r.e0, a.s0, a.e0)) st.lfull = true;
        if (!st.rfull && closed_overlap(r.sl, r.el, a.sl, a.el)) st.rfull = true;
    } else if (level == 3 || level == 4) {
        if (!st.lfull) {
            if (closed_overlap(r.s0, r.e0, a.s0, a.e0)) st.lfull = true;
            else if (st.lnoth)
                for (int k = 0; k < a.n; ++k) { const int2 x = ax[k]; if (closed_overlap(r.s0, r.e0, x.x, x.y)) { st.lnoth = false; break; } }
        }
        if (level == 3 && !st.rfull) {
            if (closed_overlap(r.sl, r.el, a.sl, a.el)) st.rfull = true;
            else if (st.rnoth)
                for (int k = 0; k < a.n; ++k) { const int2 x = ax[k]; if (closed_overlap(r.sl, r.el, x.x, x.y)) { st.rnoth = false; break; } }
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

// ------------------------------------------------------------------ site dictionaries (-d 0)
//
// With -d 0, check_splice_site only asks "is this read coordinate (pair) also a site of the transcript".
// Every distinct annotation site has a rank (host, once per annotation); a transcript carries a 64-bit
// mask of its sites relative to its first rank, a read maps its sites to ranks once (hash probes) and
// keeps a 64-bit mask relative to its first hit.  A candidate is then shift + AND + popcount.
// Preconditions, checked per read and per transcript (anything else takes the literal loops):
// strictly increasing exon starts and ends and start <= end on both sides -- then a value can pair with
// at most one value of the other chain (pair count == common values) and equality implies that the
// site lies inside both spans, i.e. inside the overlap window of check_splice_site.

// Probe one bucket for key (k1, k2): `single` = third word of any entry whose first word is k1 (acceptor
// rank of a start / donor rank of an end), `pair` = index of the entry equal to (k1, k2).
struct Probe { int single, pair; };

__device__ __forceinline__ Probe site_probe(const SiteDict &t, int bucket, int32_t k1, int32_t k2)
{
    Probe pr{-1, -1};
    if (bucket < 0) return pr;
    const uint32_t lo = t.dir[bucket], hi = t.dir[bucket + 1];
    for (uint32_t r = lo; r < hi; ++r) {
        const int4 k = t.ent[r];
        if (k.x == k1) { pr.single = k.z; if (k.y == k2) pr.pair = (int)r; }
    }
    return pr;
}

// bucket of coordinate x on `tid`, or -1 when the annotation has no site that far
__device__ __forceinline__ int site_bucket(int32_t tb, int32_t nb, int32_t x)
{
    const int b = x >> SITE_SHIFT;
    return (x >= 0 && b < nb) ? tb + b : -1;
}

struct ReadSites {          // the read's sites in rank space
    unsigned long long rd, ra, rx, rj;      // masks relative to bd/ba/bx/bj
    int bd, ba, bx, bj;                     // rank of the first hit (or -1)
    uint32_t hd, ha, hx, hj;                // bit j: exon/junction j of the read had a hit
    bool ok;                                // representable (every hit within 64 ranks of the first)
};

__device__ __forceinline__ void add_hit(unsigned long long &m, int &base, uint32_t &h, bool &ok, int g, int j)
{
    if (g < 0) return;
    if (base < 0) base = g;
    const int off = g - base;
    if (off >= 64) { ok = false; return; }
    m |= 1ull << off;
    h |= 1u << j;
}

__device__ __forceinline__ unsigned long long align_mask(const uint32_t w[2], int tx_base, int read_base)
{
    const unsigned long long m = ((unsigned long long)w[1] << 32) | w[0];
    const int d = tx_base - read_base;               // transcript bit k is rank tx_base + k
    if (d >= 0) return d < 64 ? m << d : 0ull;
    return -d < 64 ? m >> (-d) : 0ull;
}

// src/update_gtf.c:792-835 check_with_anno_trans for one read.  (S,E,F) address the read's exons and
// flag bytes (LDS or HBM).  j0 = cursor value the sequential code would have.
// Returns info bits (without exon count), ref in `ref`.
// src/update_gtf.c:792-835 check_with_anno_trans, WAVE-UNIFORM form.  The 64 reads of a wave are
// neighbours in a coordinate-sorted input, so they sweep (almost) the same annotation transcripts.  The
// whole wave therefore walks ONE transcript index j upwards from the smallest cursor value of its reads;
// j is wave-uniform, so the transcript header (and, for the exon scans of check_full, its exons) come in
// through scalar loads once per wave, and every lane applies its own read's predicates: not started yet
// (j < its cursor), finished (the read lies before transcript j, or it was found known -- the reference's
// two `break`s), transcript before the read (`continue`), or overlap.  Per read the transcripts are still
// visited in file order with the reference's early exits; only the interleaving across reads changes.
// Every lane of the wave must call this (act = lane owns a read).
__device__ __forceinline__ uint32_t sweep_annotation(bool act, const int *S, const int *E, uint8_t *F, int n, int tid, bool rev, bool read_ok,
                                                     const ReadEnds &re, int e_pen, int s_2nd, const ReadSites &rs,
                                                     int j0, const TxHdr *__restrict__ hdr, const int2 *__restrict__ anno_ex,
                                                     const DevParams &p, int &ref)
{
    const int r_start = re.s0, r_end = re.el;
    ReadState st{false, false, true, true, false, false};
    unsigned long long md = 0, ma = 0, mx = 0, mj = 0;      // sites matched by some visited transcript
    ref = -1;
    int ref_rev = 0;
    bool done = !act || (p.ablate & 1);
    int jm = done ? INT32_MAX : j0;
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) jm = min(jm, __shfl_xor(jm, d, WAVE));
    const int level = p.full_level;
    for (int j = __builtin_amdgcn_readfirstlane(jm); j < p.n_tx && __any(!done); ++j) {
        const int4 *hp = reinterpret_cast<const int4 *>(hdr + j);           // wave-uniform address
        const int4 h0 = hp[0];
        bool ov = false;
        if (!done && j >= j0) {
            // src/update_gtf.c:786-790 comp_trans: <= (Q5)
            if (tid < h0.x || (tid == h0.x && r_end <= h0.y)) done = true;                  // :799-800 break
            else ov = !(h0.x < tid || (h0.x == tid && h0.z <= r_start));                    // :801 skip when before
        }
        if (!__any(ov)) continue;
        const int4 h1 = hp[1], h2 = hp[2];
        TxHdr a; a.tid = h0.x; a.start = h0.y; a.end = h0.z; a.ex_off = h0.w;
        a.n = h1.x; a.rev = h1.y; a.flags = h1.z; a.s0 = h2.x; a.e0 = h2.y; a.sl = h2.z; a.el = h2.w;
        const int2 *ax = anno_ex + a.ex_off;
        // ---- check_full :629-681
        if (!(p.ablate & 4) && ov && !(st.lfull && st.rfull)) {
            if (level == 1) {
                if (!st.lfull && re.e0 == a.e0) st.lfull = true;
                if (!st.rfull && re.sl == a.sl) st.rfull = true;
            } else if (level == 2) {
                if (!st.lfull && closed_overlap(re.s0, re.e0, a.s0, a.e0)) st.lfull = true;
                if (!st.rfull && closed_overlap(re.sl, re.el, a.sl, a.el)) st.rfull = true;
            }
        }
        if (!(p.ablate & 4) && (level == 3 || level == 4)) {
            bool need_l = false, need_r = false;
            if (ov && !(st.lfull && st.rfull)) {
                if (!st.lfull) { if (closed_overlap(re.s0, re.e0, a.s0, a.e0)) st.lfull = true; else need_l = st.lnoth; }
                if (level == 3 && !st.rfull) { if (closed_overlap(re.sl, re.el, a.sl, a.el)) st.rfull = true; else need_r = st.rnoth; }
            }
            if (__any(need_l || need_r)) {
                for (int k = 0; k < a.n; ++k) {                      // exon k of the transcript: scalar load
                    const int2 x = ax[k];
                    if (need_l && closed_overlap(re.s0, re.e0, x.x, x.y)) { st.lnoth = false; need_l = false; }
                    if (need_r && closed_overlap(re.sl, re.el, x.x, x.y)) { st.rnoth = false; need_r = false; }
                }
            }
        }
        // ---- :806-820
        int v = 0;
        if (a.n == 1) {
            if (ov && n == 1 && overlap_frac(re.s0, re.e0, a.s0, a.e0) >= p.frac) { st.known = true; v = 1; }
        } else if (!(p.ablate & 2)) {
            const bool multi = ov && n > 1;
            const bool use_dict = multi && read_ok && (a.flags & TX_COMPACT);
            if (__any(use_dict)) {
                const int4 h3 = hp[3], h4 = hp[4], h5 = hp[5];
                if (use_dict) {
                    const uint32_t wd[2] = {(uint32_t)h4.x, (uint32_t)h4.y}, wa[2] = {(uint32_t)h4.z, (uint32_t)h4.w};
                    const uint32_t wx[2] = {(uint32_t)h5.x, (uint32_t)h5.y}, wj[2] = {(uint32_t)h5.z, (uint32_t)h5.w};
                    const unsigned long long cd = rs.rd & align_mask(wd, h3.x, rs.bd), ca = rs.ra & align_mask(wa, h3.y, rs.ba);
                    md |= cd; ma |= ca;
                    mx |= rs.rx & align_mask(wx, h3.z, rs.bx);
                    mj |= rs.rj & align_mask(wj, h3.w, rs.bj);
                    const int same = __popcll(cd) + __popcll(ca);
                    const int lo = max(r_start, a.start), hi = min(r_end, a.end);
                    // every one of the 2(n-1) read sites inside [lo,hi]: donors e_0..e_{n-2}, acceptors s_1..s_{n-1} increase
                    const bool all_in = re.e0 >= lo && e_pen <= hi && s_2nd >= lo && re.sl <= hi;
                    v = (all_in && same == 2 * (n - 1)) ? 1 : (same > 0 ? 2 : 0);
                }
            }
            if (multi && !use_dict) v = site_compare(S, E, F, n, r_start, r_end, a, ax, p.ss_dis);      // literal loops
            if (v == 1) st.known = true;
            if (v == 2) st.ksite = true;
        }
        if (v) { ref = j; ref_rev = a.rev; }
        if (v == 1) done = true;                                                            // :810,816 break
    }
    if (md | ma | mx | mj) {
        // hits were recorded in exon order and ranks increase with the exon index, so the k-th hit of a
        // kind is the k-th set bit of its mask
        unsigned long long qd = rs.rd, qa = rs.ra, qx = rs.rx, qj = rs.rj;
        for (int k = 0; k < n; ++k) {
            uint8_t clr = 0;
            if ((rs.hx >> k) & 1u) { const unsigned long long b = qx & (0ull - qx); qx ^= b; if (mx & b) clr |= F_EXON; }
            if ((rs.hd >> k) & 1u) { const unsigned long long b = qd & (0ull - qd); qd ^= b; if (md & b) clr |= F_DON; }
            if ((rs.ha >> k) & 1u) { const unsigned long long b = qa & (0ull - qa); qa ^= b; if (ma & b) clr |= F_ACC; }
            if ((rs.hj >> k) & 1u) { const unsigned long long b = qj & (0ull - qj); qj ^= b; if (mj & b) clr |= F_JUNC; }
            if (clr) F[k] &= (uint8_t)~clr;
        }
    }
    bool out_rev = rev;
    if (ref >= 0) out_rev = ref_rev != 0;               // :825-831 strand taken from the reference transcript
    bool full;                                           // :683-696 set_full
    if (level == 5) full = true;
    else if (level == 4) full = st.lfull || st.lnoth;
    else if (level == 3) full = (st.lfull || st.lnoth) && (st.rfull || st.rnoth);
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

// Exons of one read into (S,E,F); terminal-exon registers; `sane` = strictly increasing starts and ends
// and start <= end (the precondition of the dictionary path on the read side).
struct ReadShape { ReadEnds re; int s_2nd, e_pen; bool sane; };

__device__ __forceinline__ ReadShape exons_from_cigar(int *S, int *E, uint8_t *F, int n, int32_t pos0,
                                                      const uint32_t *__restrict__ cig, int n_cig, const DevParams &p)
{
    ReadShape sh{{0, 0, 0, 0}, 0, 0, true};
    int ps = INT32_MIN, pe = INT32_MIN;
    walk_cigar(cig, n_cig, pos0, p, [&](int k, int s, int e) {
        S[k] = s; E[k] = e;
        sh.sane = sh.sane && s > ps && e > pe && s <= e;
        if (k == 0) { sh.re.s0 = s; sh.re.e0 = e; }
        if (k == 1) sh.s_2nd = s;
        sh.e_pen = pe; ps = s; pe = e;
    });
    sh.re.sl = ps; sh.re.el = pe;
    for (int k = 0; k < n; ++k) F[k] = (k + 1 < n) ? (uint8_t)(F_EXON | F_DON | F_ACC | F_JUNC) : F_EXON;
    return sh;
}

// The slice of one dictionary that covers a tile, staged in LDS: directory words of the buckets
// [b0, b0 + nb] and the entries [r0, r0 + nk).
struct DictSlice { const uint32_t *dir; const int4 *ent; int b0; uint32_t r0; };

__device__ __forceinline__ Probe slice_probe(const DictSlice &t, int bucket, int32_t k1, int32_t k2)
{
    Probe pr{-1, -1};
    if (bucket < 0) return pr;
    const uint32_t lo = t.dir[bucket - t.b0] - t.r0, hi = t.dir[bucket - t.b0 + 1] - t.r0;
    for (uint32_t r = lo; r < hi; ++r) {
        const int4 k = t.ent[r];
        if (k.x == k1) { pr.single = k.z; if (k.y == k2) pr.pair = (int)(r + t.r0); }
    }
    return pr;
}

// ranks of the read's sites -> ReadSites; PROBE(which 0 = START / 1 = END, bucket, k1, k2)
template <typename ProbeFn>
__device__ __forceinline__ ReadSites map_read_sites(const int *S, const int *E, int n, int32_t tb, int32_t nb, ProbeFn probe)
{
    ReadSites rs{0, 0, 0, 0, -1, -1, -1, -1, 0, 0, 0, 0, true};
    int s = S[0], e = E[0];
    for (int k = 0; k < n; ++k) {
        const Probe ps = probe(0, site_bucket(tb, nb, s), s, e);           // exon (s,e); acceptor rank of s
        add_hit(rs.rx, rs.bx, rs.hx, rs.ok, ps.pair, k);
        if (k + 1 < n) {
            const int s2 = S[k + 1], e2 = E[k + 1];
            const Probe pe = probe(1, site_bucket(tb, nb, e), e, s2);      // junction (e,s2); donor rank of e
            add_hit(rs.ra, rs.ba, rs.ha, rs.ok, ps.single, k);             // Q1: start of exon k itself, k < n-1
            add_hit(rs.rd, rs.bd, rs.hd, rs.ok, pe.single, k);
            add_hit(rs.rj, rs.bj, rs.hj, rs.ok, pe.pair, k);
            s = s2; e = e2;
        }
    }
    return rs;
}

__device__ __forceinline__ uint32_t finish_info(uint32_t info, int n, const DevParams &p)
{
    // routing of update_gtf.c:943-950 when there is no junction table
    if (p.n_sj == 0 && (info & (I_FULL | I_KNOWN | I_KSITE)) == (I_FULL | I_KSITE)) info |= I_ACCEPT;
    return info | ((uint32_t)n << 8);
}

constexpr int DIR_CAP = 384;        // directory words staged per kind (tile span up to ~196 kb of buckets)
constexpr int KEY_CAP = 256;        // dictionary entries staged per dictionary


__global__ __launch_bounds__(TILE_THREADS)
void k_fill_classify(int64_t n_reads, const int32_t *__restrict__ r_tid, const int32_t *__restrict__ r_pos,
                     const uint8_t *__restrict__ r_rev, const int64_t *__restrict__ cig_off, const uint32_t *__restrict__ cig,
                     const uint32_t *__restrict__ n_ex, const uint32_t *__restrict__ tile_base,
                     const int32_t *__restrict__ j0_arr,
                     const TxHdr *__restrict__ hdr, const int2 *__restrict__ anno_ex, SiteTabs tabs, DevParams p,
                     uint32_t *__restrict__ ex_off, int32_t *__restrict__ ex_start, int32_t *__restrict__ ex_end,
                     uint8_t *__restrict__ ex_flag, uint32_t *__restrict__ info_out, int32_t *__restrict__ ref_out)
{
    __shared__ uint32_t s_wave[4];
    __shared__ int s_bmin, s_bmax;
    __shared__ int s_start[LDS_EXON_CAP];
    __shared__ int s_end[LDS_EXON_CAP];
    __shared__ uint8_t s_flag[LDS_EXON_CAP];
    __shared__ uint32_t s_dir[2][DIR_CAP + 1];
    __shared__ __attribute__((aligned(16))) int4 s_keys[2][KEY_CAP];

    const int64_t r = (int64_t)blockIdx.x * p.reads_per_tile + threadIdx.x;
    const bool active = threadIdx.x < p.reads_per_tile && r < n_reads;
    const uint32_t n = active ? n_ex[r] : 0u;
    if (threadIdx.x == 0) { s_bmin = INT32_MAX; s_bmax = -1; }
    uint32_t tile_total;
    const uint32_t local = block_exclusive_scan(n, s_wave, tile_total);
    const uint32_t base = tile_base[blockIdx.x];
    const bool in_lds = tile_total <= (uint32_t)LDS_EXON_CAP;       // else: very long exon chains, work in HBM

    // ---- phase 1: CIGAR -> exons, cursor value, bucket span of the read
    ReadShape sh{{0, 0, 0, 0}, 0, 0, true};
    int32_t tid = 0, tb = 0, nb = 0; int j0 = 0;
    bool want_dict = false;
    if (active) {
        const int64_t a = cig_off[r], b = cig_off[r + 1];
        tid = r_tid[r];
        if (in_lds) sh = exons_from_cigar(s_start + local, s_end + local, s_flag + local, (int)n, r_pos[r], cig + a, (int)(b - a), p);
        else sh = exons_from_cigar(ex_start + base + local, ex_end + base + local, ex_flag + base + local, (int)n, r_pos[r], cig + a, (int)(b - a), p);
        j0 = j0_arr[r];
        want_dict = p.ss_dis == 0 && sh.sane && n > 1 && n <= 32 && tid < tabs.n_tid && !(p.ablate & (16 | 1));
        if (want_dict) { tb = tabs.tid_base[tid]; nb = tabs.tid_base[tid + 1] - tb; want_dict = nb > 0; }
    }
    {   // bucket span of the tile's dictionary reads: wave reduction, one LDS atomic per wave
        int lo = INT32_MAX, hi = -1;
        if (want_dict) {
            lo = tb + min(max(sh.re.s0, 0) >> SITE_SHIFT, nb - 1);
            hi = tb + min(max(sh.re.el, 0) >> SITE_SHIFT, nb - 1);
        }
#pragma unroll
        for (int d = WAVE / 2; d > 0; d >>= 1) { lo = min(lo, __shfl_xor(lo, d, WAVE)); hi = max(hi, __shfl_xor(hi, d, WAVE)); }
        if ((threadIdx.x & (WAVE - 1)) == 0 && hi >= 0) { atomicMin(&s_bmin, lo); atomicMax(&s_bmax, hi); }
    }
    __syncthreads();
    // ---- phase 2: stage the dictionary slices of the tile in LDS (directory words, then keys)
    const int b0 = s_bmin, nbk = s_bmax - s_bmin + 1;                // buckets b0 .. b0+nbk-1
    bool staged = in_lds && s_bmax >= 0 && nbk <= DIR_CAP && !(p.ablate & 64);
    if (staged) {
        for (int i = threadIdx.x; i < 2 * (nbk + 1); i += TILE_THREADS) {
            const int kind = i >= nbk + 1, w = i - kind * (nbk + 1);
            s_dir[kind][w] = (kind ? tabs.en.dir : tabs.st.dir)[b0 + w];
        }
    }
    __syncthreads();
    if (staged) {
#pragma unroll
        for (int kind = 0; kind < 2; ++kind) staged = staged && (s_dir[kind][nbk] - s_dir[kind][0]) <= (uint32_t)KEY_CAP;
    }
    if (staged) {
#pragma unroll
        for (int kind = 0; kind < 2; ++kind) {
            const int4 *src = kind ? tabs.en.ent : tabs.st.ent;
            const uint32_t r0 = s_dir[kind][0], nk = s_dir[kind][nbk] - r0;
            for (uint32_t i = threadIdx.x; i < nk; i += TILE_THREADS) s_keys[kind][i] = src[r0 + i];
        }
    }
    __syncthreads();
    // ---- phase 3: ranks of the read's sites, then the sweep
    uint32_t info = 0; int ref = -1;
    auto phase3 = [&](const int *S, const int *E, uint8_t *F) {
        ReadSites rs{0, 0, 0, 0, -1, -1, -1, -1, 0, 0, 0, 0, true};
        bool read_ok = want_dict;
        if (want_dict && !(p.ablate & 32)) {
            if (staged) {
                const DictSlice ss{s_dir[0], s_keys[0], b0, s_dir[0][0]}, se{s_dir[1], s_keys[1], b0, s_dir[1][0]};
                rs = map_read_sites(S, E, (int)n, tb, nb, [&](int which, int bucket, int k1, int k2) {
                    return which ? slice_probe(se, bucket, k1, k2) : slice_probe(ss, bucket, k1, k2);
                });
            } else {
                rs = map_read_sites(S, E, (int)n, tb, nb, [&](int which, int bucket, int k1, int k2) {
                    return which ? site_probe(tabs.en, bucket, k1, k2) : site_probe(tabs.st, bucket, k1, k2);
                });
            }
            read_ok = rs.ok;
        }
        // the sweep is wave-uniform: every lane of the wave takes part
        info = sweep_annotation(active, S, E, F, (int)n, tid, active && r_rev[r] != 0, read_ok, sh.re, sh.e_pen, sh.s_2nd, rs, j0,
                                hdr, anno_ex, p, ref);
        info = finish_info(info, (int)n, p);
    };
    if (in_lds) phase3(s_start + local, s_end + local, s_flag + local);
    else phase3(ex_start + base + local, ex_end + base + local, ex_flag + base + local);
    // ---- phase 4: coalesced write-out of the tile
    if (in_lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < tile_total; i += TILE_THREADS) {
            ex_start[base + i] = s_start[i];
            ex_end[base + i] = s_end[i];
            ex_flag[base + i] = s_flag[i];
        }
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
