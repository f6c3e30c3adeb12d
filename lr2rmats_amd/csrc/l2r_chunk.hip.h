// l2r_chunk.hip.h -- the slab pipeline's probe kernel for tiles whose window does not fit ANY mask width: k_probe_slab_chunked.
//
// A tile whose reads can meet more than 63 annotation transcripts (a locus with very many isoforms) has no window record.  This
// kernel takes such a tile's window 63 members at a time, in annotation order -- CHUNKS -- and carries the sweep's state from chunk
// to chunk per read, exactly as the sequential sweep of check_with_anno_trans (src/update_gtf.c:792-835) meets the transcripts:
//
//   per chunk   one wave scans the transcript headers on from where the last chunk ended and collects the next 63 members that
//               overlap the tile's span (the scan of make_descriptor, resumable); the tile's dictionary slices are staged with
//               their masks re-based to the chunk; visit_chunk64 / map_exons_lds64 give the chunk's masks and per-exon work words
//   carried     known (the sweep's break: later chunks are not looked at), stopped (a member the read lies before ended the
//               sweep), has-known-site, the reference transcript so far (the LAST member with an identical site), lfull / rfull /
//               lnoth / rnoth (sticky, src/update_gtf.c:629-696), and per exon the novel flags still standing (LDS, one byte per
//               staged position: a flag is cleared by the first chunk that has a member with the exon / junction / site no later
//               than the break)
//
// The tile's exons are copied from the slab to their positions in LDS once; the probe rounds of every chunk read them there.
// Its tiles (slab_tile_is_chunked, listed by TileLists between the walk and the probes): the ones k_walk_slab found no window record for ("window >
// 32" after it tried a 64-bit record, or dictionary slices beyond the one-window kernels' staging), and the ones k_probe_slab or
// k_probe_slab_wide flagged TD_CHUNK because their dictionary slices hold a key in several entries (SE_WIDE: the key's transcripts
// lie more than 64 apart in the annotation, l2r_engine.hip build_dict) -- the probes here OR the parts of a key, each re-based to the
// chunk.
// Of a tile's dictionary slices a chunk stages only the entries that can say something about its 63 members (the others re-base to
// nothing): a counting sort by bucket into up to CHUNK_KEY_CAP places per dictionary; a chunk beyond that sends the tile to the
// generic kernel.
// Same results as the one-window kernels, which the parity suites check.
#pragma once
#include "l2r_wide.hip.h"

namespace l2r {

constexpr int CHUNK_KEY_CAP = 320;                                  // dictionary entries staged per dictionary and chunk (39.5 KB of LDS per workgroup: 4 per CU)
struct ChunkLds { const WEnt *ent0, *ent1; const uint32_t *dir0, *dir1, *rdir; const int4 *hk, *hx; const int *win; };
constexpr int CHUNK_SCAN_TRIPS = 4096;                              // 64-transcript trips one chunk's scan may take (then: generic kernel)

struct ChunkVisit { m64_t vpre, lmask, rmask, k1mask; bool redo, stopped; };

// visit_window64 on one chunk; `stopped`: a member of the chunk lies behind the read (src/update_gtf.c:799-800 ends the sweep)
template <int LEVEL>
__device__ __forceinline__ ChunkVisit visit_chunk64(const ChunkLds &L, int w_n, bool work, uint32_t n, const ReadEnds &re, const m64_t *tilemask)
{
    ChunkVisit m{0ull, 0ull, 0ull, 0ull, false, false};
    m64_t m_aft = 0ull, m_bef = 0ull;
#pragma unroll 4
    for (int j = 0; j < w_n; ++j) {
        const int4 hk = L.hk[j];
        const m64_t bit = 1ull << j;
        m_aft |= re.el <= hk.x ? bit : 0ull;                                 // comp_trans <= (Q5): the read lies before the member
        m_bef |= hk.y <= re.s0 ? bit : 0ull;                                 // the member lies before the read
        if (LEVEL >= 1 && LEVEL <= 4) {
            const int4 hx = L.hx[j];
            if (LEVEL == 1) {
                m.lmask |= re.e0 == hx.y ? bit : 0ull;
                m.rmask |= re.sl == hx.z ? bit : 0ull;
            } else {
                m.lmask |= closed_overlap(re.s0, re.e0, hx.x, hx.y) ? bit : 0ull;
                if (LEVEL != 4) m.rmask |= closed_overlap(re.sl, re.el, hx.z, hx.w) ? bit : 0ull;
            }
        }
    }
    const m64_t below = (m_aft & (0ull - m_aft)) - 1ull;                     // all ones when no member ends the sweep
    m.vpre = work ? (~m_bef & below & (w_n >= 64 ? ~0ull : ((1ull << w_n) - 1ull))) : 0ull;
    m.stopped = work && m_aft != 0ull;
    m.lmask &= m.vpre; m.rmask &= m.vpre;
    const m64_t single = tilemask[0];
    if (n == 1) {
        m64_t c = m.vpre & single;
        while (c) {
            const int j = __ffsll((long long)c) - 1;
            c &= c - 1ull;
            const int4 hx = L.hx[j];
            if (overlap_frac(re.s0, re.e0, hx.x, hx.y) >= fast_args()->p.frac) m.k1mask |= 1ull << j;
        }
    } else if (m.vpre & tilemask[1] & ~single) m.redo = true;
    return m;
}

// probe_all64 for keys in several entries: the parts' masks are ORed
__device__ __forceinline__ void probe_parts64(const WEnt *ent, uint32_t lo, uint32_t hi, int32_t k1, int32_t k2, m64_t &pm, m64_t &sm)
{
    pm = 0ull; sm = 0ull;
    for (uint32_t r = lo; r < hi; ++r) {
        const WEnt q = ent[r];
        if (q.k1 == k1) { sm |= q.sm; if (q.k2 == k2) pm |= q.pm; }
    }
}

// probe_parts64 with a tolerance (-d > 0, src/update_gtf.c:717-779: see probe_near, l2r_kernels.hip.h): entries [lo, hi) = the staged
// entries of the buckets of k1 - dis .. k1 + dis, in NO particular order inside a bucket (the chunk's counting sort), a key possibly in
// several of them (parts: same keys, masks of different transcripts).  amb: members with two DIFFERENT sites within the tolerance.
__device__ __forceinline__ void probe_near_parts64(const WEnt *ent, uint32_t lo, uint32_t hi, int32_t k1, int32_t k2, int dis, int rs, int re, m64_t &pm, m64_t &sm, m64_t &amb)
{
    pm = 0ull; sm = 0ull;
    int first = INT32_MIN; bool several = false;
    for (uint32_t r = lo; r < hi; ++r) {
        const WEnt q = ent[r];
        if (__builtin_abs(q.k1 - k1) > dis) continue;
        if (q.k1 >= rs && q.k1 <= re) {
            if (first == INT32_MIN) first = q.k1; else several = several || q.k1 != first;
            sm |= q.sm;
        }
        if (__builtin_abs(q.k2 - k2) <= dis) pm |= q.pm;
    }
    if (several)                                             // (rare: two annotation sites within 2 dis + 1 bases -- which members have both?)
        for (uint32_t r = lo; r < hi; ++r) {
            const WEnt q = ent[r];
            if (__builtin_abs(q.k1 - k1) > dis || q.k1 < rs || q.k1 > re) continue;
            for (uint32_t u = r + 1u; u < hi; ++u) {
                const WEnt w = ent[u];
                if (__builtin_abs(w.k1 - k1) > dis || w.k1 < rs || w.k1 > re || w.k1 == q.k1) continue;
                amb |= q.sm & w.sm;
            }
        }
}

// map_exons_slab64 with the read's exons at their staged positions in LDS (A: start relative to the tile's base, L: length); the
// chunk's work word goes into the upper bits of A
__device__ __forceinline__ SiteMasks64 map_exons_lds64(const ChunkLds &L, const TileDesc &d, bool mapping, uint32_t *Ap, const uint16_t *Lp,
                                                       int32_t lo, uint32_t n, m64_t vpre, int dis = 0, int rs = 0, int re = 0)
{
    SiteMasks64 m{~0ull, 0ull, 0ull, 0ull, 0ull};
    const uint32_t none = (uint32_t)d.nbk + 1u;
    const int k_max = wave_max(mapping ? (int)n : 0);
    int s = 0, e = 0;
    if (mapping) { s = lo + (int)(Ap[0] & SLAB_REL_MASK); e = s + (int)Lp[0] - 1; }
    for (int k = 0; k < k_max; ++k) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        int s2 = 0, e2 = 0;
        if (junc) { s2 = lo + (int)(Ap[k + 1] & SLAB_REL_MASK); e2 = s2 + (int)Lp[k + 1] - 1; }
        m64_t xm, am, jm, dm;
        if (dis > 0) {                                  // (wave-uniform: -d)
            auto range = [&](const uint32_t *dir, bool on, int x, uint32_t &l_, uint32_t &h_) {
                const uint32_t i0 = on ? min((uint32_t)((max(x - dis, 0) >> SITE_SHIFT) + d.b_off), none) : none;
                const uint32_t i1 = on ? min((uint32_t)(((x + dis) >> SITE_SHIFT) + d.b_off), none) : none;
                l_ = dir[i0]; h_ = dir[i1 + 1u];
            };
            uint32_t ls, hs, le, he;
            range(L.dir0, live, s, ls, hs); range(L.dir1, junc, e, le, he);
            probe_near_parts64(L.ent0, ls, hs, s, e, dis, rs, re, xm, am, m.amb);
            probe_near_parts64(L.ent1, le, he, e, s2, dis, rs, re, jm, dm, m.amb);
        } else {
            const uint32_t is = live ? min((uint32_t)((s >> SITE_SHIFT) + d.b_off), none) : none;
            const uint32_t ie = junc ? min((uint32_t)((e >> SITE_SHIFT) + d.b_off), none) : none;
            const uint32_t ls = L.dir0[is], hs = L.dir0[is + 1u], le = L.dir1[ie], he = L.dir1[ie + 1u];
            probe_parts64(L.ent0, ls, hs, s, e, xm, am);
            probe_parts64(L.ent1, le, he, e, s2, jm, dm);
        }
        const m64_t amj = junc ? am : 0ull;
        uint32_t word = first_member64(xm & vpre);
        word |= first_member64(jm & vpre) << 6;
        word |= ((dm & vpre) ? 1u : 0u) << 12;
        word |= ((amj & vpre) ? 1u : 0u) << 13;
        m.kand &= junc ? (am & dm) : ~0ull;            // Q1: the acceptor probed with exon k is ITS OWN start, k < n-1
        m.kor |= amj | dm;
        // (with a tolerance a shared donor / acceptor does not say that the exons overlap: the full-length evidence asks the START slice)
        if (dis <= 0) { if (k == 0) m.dm_first = dm;
                        m.am_last = (live && !junc) ? am : m.am_last; }
        if (live) Ap[k] = (uint32_t)(s - lo) | (word << SLAB_REL_BITS);
        s = s2; e = e2;
    }
    return m;
}

// The next chunk of the tile's window: the scan of make_descriptor (l2r_window.hip.h) from transcript scan_j on, for up to
// max_members (<= WIDE_MEMBERS) members; ONE WAVE.  Leaves members, headers and masks in *W (W->d: n_win, j_lo = first member, TD_CONTIG) and
// returns where the next chunk's scan begins (n_tx: the window is exhausted; -1: the scan took too long).
__device__ __forceinline__ int chunk_window(PipeArgsK a, int lane, int32_t tid0, int32_t tlo, int32_t thi, int scan_j, uint32_t max_members, TileWin64 *W)
{
    const TxHdr *const hdr = a->f.hdr;
    const int32_t n_tx = a->f.p.n_tx;
    uint32_t n_win = 0;
    int first = -1, last = -1, next = n_tx;
    int trip = 0;
    for (int base = scan_j; base < n_tx; base += WAVE, ++trip) {
        if (trip == CHUNK_SCAN_TRIPS) { next = -1; break; }
        const int j = base + lane;
        bool ov = false, aft = false;
        if (j < n_tx) {
            const int4 h0 = *reinterpret_cast<const int4 *>(hdr + j);                 // {tid, start, end, .}
            aft = tid0 < h0.x || (tid0 == h0.x && thi <= h0.y);                       // comp_trans <= (Q5)
            const bool bef = h0.x < tid0 || (h0.x == tid0 && h0.z <= tlo && h0.y < tlo);
            ov = !aft && !bef;
        }
        const unsigned long long ma = __ballot(aft);
        const int stop = ma ? __ffsll((long long)ma) - 1 : WAVE;
        unsigned long long mo = __ballot(ov) & (stop < WAVE ? (1ull << stop) - 1ull : ~0ull);
        const uint32_t room = max_members - n_win;
        bool full = false;
        if ((uint32_t)__popcll(mo) > room) {
            // the chunk ends inside this trip: keep its first `room` members, the next chunk begins at the first one left out
            unsigned long long rest = mo;
            for (uint32_t i = 0; i < room; ++i) rest &= rest - 1ull;
            next = base + __ffsll((long long)rest) - 1;
            mo &= ~rest;
            full = true;
        }
        if ((mo >> lane) & 1ull) W->win[n_win + (uint32_t)__popcll(mo & ((1ull << lane) - 1ull))] = j;
        if (mo) {
            if (first < 0) first = base + __ffsll((long long)mo) - 1;
            last = base + 63 - __clzll((long long)mo);
        }
        n_win += (uint32_t)__popcll(mo);
        if (full) break;
        if (ma) break;                                   // (next stays n_tx: every read lies before what follows)
    }
    // the members' headers (the wave's own LDS writes above are visible to it: same wave, in order)
    bool single = false, loose = false;
    if (lane < (int)n_win) {
        const int j = W->win[lane];
        const int4 *hp = reinterpret_cast<const int4 *>(hdr + j);
        const int4 h0 = hp[0], h1 = hp[1], h2 = hp[2];
        int st = h0.y, en = h0.z;
        if (h0.x < tid0) { st = INT32_MIN; en = INT32_MIN; }            // another chromosome: before / after every read
        else if (h0.x > tid0) { st = INT32_MAX; en = INT32_MAX; }
        W->hk[lane] = make_int4(st, en, h1.x, (h1.z & 0xff) | (h1.y << 8));
        W->hx[lane] = h2;
        single = h1.x == 1; loose = !((h1.z & 0xff) & TX_COMPACT);
    }
    const unsigned long long b1 = __ballot(single), b2 = __ballot(loose);
    if (lane == 0) {
        W->d.n_win = n_win; W->d.j_lo = first < 0 ? scan_j : first;
        W->d.flags = (n_win == 0u || (uint32_t)(last - first + 1) == n_win) ? TD_CONTIG : 0u;
        W->mask[0] = b1; W->mask[1] = b2;
    }
    return next;
}

template <int LEVEL>
__global__ __launch_bounds__(TILE_THREADS, 4)
void k_probe_slab_chunked(SlabArgs kernarg_block, const uint32_t *__restrict__ u_tile_first, const int32_t *__restrict__ u_pos,
                          const uint32_t *__restrict__ u_tile_sbase, const TileWin *__restrict__ u_tw, const uint32_t *__restrict__ u_xbase)
{
    constexpr int DIR_N = FAST_DIR_BYTES;                   // directory words per dictionary (group 0 = the reach-back entries in front of the first bucket, bucket b = group b + 1, three closing words)
    static_assert(DIR_N >= DIR_CAP + 4, "directory words");
    constexpr uint32_t F_ALL = (uint32_t)(F_EXON | F_DON | F_ACC | F_JUNC);
    __shared__ __attribute__((aligned(16))) uint32_t s_A[SLAB_POS_CAP];
    __shared__ __attribute__((aligned(16))) uint16_t s_L[SLAB_POS_CAP];
    __shared__ __attribute__((aligned(16))) uint8_t s_F[SLAB_POS_CAP];
    __shared__ __attribute__((aligned(16))) WEnt s_ent[2 * CHUNK_KEY_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_dir[3 * DIR_N];
    __shared__ uint32_t s_kept[2];
    __shared__ __attribute__((aligned(16))) TileWin64 s_tw;
    __shared__ uint32_t s_next, s_lim;
    __shared__ int s_scan;
    (void)kernarg_block;
    const SlabArgsK sa = slab_args();
    const PipeArgsK a = pipe_args();
    const int lane = threadIdx.x & (WAVE - 1);
    WEnt *const s_ent0 = s_ent, *const s_ent1 = s_ent + CHUNK_KEY_CAP;
    uint32_t *const X0 = s_dir, *const X1 = s_dir + DIR_N, *const XR = s_dir + 2 * DIR_N;
    // the tiles of this kernel: chunk_list (TileLists + what the one-window kernels appended), taken from a cursor
    // (... = the entries k_describe_scan / TileLists made, then the ones the one-window kernels appended late: chunk_list_append_late)
    const uint32_t n_first = min(sa->list_cnt[1], (uint32_t)sa->n_tiles), n_list = n_first + min(sa->list_cnt[8], (uint32_t)sa->n_tiles);
    for (bool own = true;; own = false) {
        if (own && blockIdx.x >= n_list) break;
        if (threadIdx.x == 0) s_next = own ? blockIdx.x : gridDim.x + atomicAdd(sa->list_cnt + 3, 1u);
        __syncthreads();
        const uint32_t wi = s_next;
        if (wi >= n_list) break;
        const uint32_t t = wi < n_first ? sa->chunk_list[wi] : sa->chunk_list[sa->n_tiles + 1u + (wi - n_first)];
        // (one-kernel tile path: k_tile_chunk, l2r_tchunk.hip.h, has taken the tile from its CIGARs -- the same test as there)
        if (sa->chunk_direct_on && (wi < n_first ? (sa->tile_flags[t] & TD_CDIRECT) != 0u
                                                 : tile_chunk_direct(1u, sa->tile_flags[t], sa->chunk_on, u_tw[t].d, sa->tile_stat[t], u_tile_first[t + 1u] - u_tile_first[t],
                                                                     a->f.p.min_exon, a->f.p.min_intron, a->f.p.max_delet, a->f.p.ss_dis, a->f.p.ablate, true))) { __syncthreads(); continue; }
        const uint32_t r0 = u_tile_first[t], n_act = u_tile_first[t + 1u] - r0;
        const uint32_t sbase = u_tile_sbase[t], xbase = u_xbase[t], total = u_xbase[t + 1u] - xbase;
        const int32_t tile_lo = u_pos[r0] + 1;
        TileDesc d = u_tw[t].d;
        const int32_t thi = (int32_t)u_tw[t].pad[1];            // the tile's last base (k_walk_slab)
        // (the slices of the dictionaries are the tile's whatever the window)
        const int dis = a->f.p.ss_dis;
        const bool usable = d.nbk > 0 && dis >= 0 && dis <= DIS_MASK_MAX;
        const bool active = threadIdx.x < n_act;
        const uint32_t at = r0 + (active ? threadIdx.x : 0u);
        uint32_t pre = 0u, loc = 0u;
        const uint32_t *const xw = sa->slab_row;
        const uint32_t off = sbase + threadIdx.x;
        SlabRows q;
        q.last = SlabRow{0u};
#pragma unroll
        for (int i = 0; i < SLAB_AHEAD; ++i) q.x[i] = SlabRow{0u};
        if (active) { slab_preloc(sa, t, at, pre, loc); q.last = slab_load_row(xw, off); }
        const uint32_t n = pre >> PRE_N_SHIFT;
        const uint32_t r = r0 + (pre & 0xffu);
        const bool outlier = (pre & PRE_DENSE) != 0u, rev_in = (pre & PRE_REV) != 0u;
        if (threadIdx.x == 0) s_lim = min(total, (uint32_t)SLAB_POS_CAP);
        __syncthreads();
        const SlabOut out{a->f.ex_start, a->f.ex_end, a->f.ex_flag, xbase + loc};
        const SlabStage st{s_A, s_L, loc, tile_lo, active && loc + n <= (uint32_t)SLAB_POS_CAP && !(pre & PRE_DENSE)};
        if (active && loc + n > (uint32_t)SLAB_POS_CAP) atomicMin(&s_lim, loc);
        // ---- the tile's exons to their positions (a read that is not staged: straight into the result arrays)
        if (active) slab_copy_exons(sa, a, out, st, q, off, n, pre, r);
        uint32_t *const Ap = s_A + loc; const uint16_t *const Lp = s_L + loc; uint8_t *const Fp = s_F + loc;
        bool redo = active && (!usable || outlier || !st.fits || (n > 1 && (pre & PRE_INSANE) != 0u));
        const bool work0 = active && !redo;
        ReadEnds re{0, 0, 0, 0};
        if (work0) {
            re.s0 = tile_lo + (int)(Ap[0] & SLAB_REL_MASK); re.e0 = re.s0 + (int)Lp[0] - 1;
            re.sl = tile_lo + (int)(Ap[n - 1u] & SLAB_REL_MASK); re.el = re.sl + (int)Lp[n - 1u] - 1;
            for (uint32_t k = 0; k < n; ++k) Fp[k] = (uint8_t)F_ALL;
        }
        // ---- the sweep's state, carried from chunk to chunk
        bool known = false, stopped = false, ksite = false, lfull = false, rfull = false, lnoth = true, rnoth = true, out_rev = rev_in;
        int ref = -1;
        const ChunkLds L{s_ent0, s_ent1, X0 + 1, X1 + 1, XR, s_tw.hk, s_tw.hx, s_tw.win};      // (dir[b] = first entry of bucket b = group b + 1)
        // (members per chunk: halved, for the rest of the tile, when a chunk's members need more dictionary entries than are staged)
        uint32_t chunk_members = (uint32_t)WIDE_MEMBERS;
        int scan_from = d.j_lo;
        // The first entry that reaches into each bucket (reach-back directory) and its key do not change from chunk to chunk -- only the
        // staged place of its group does: two dependent round trips per chunk that are taken once per tile here (two buckets per thread).
        constexpr int RB_PER = (DIR_CAP + TILE_THREADS - 1) / TILE_THREADS;
        int32_t rb_k1[RB_PER]; bool rb_in[RB_PER];
#pragma unroll
        for (int u = 0; u < RB_PER; ++u) {
            const int i = (int)threadIdx.x + u * TILE_THREADS;
            rb_k1[u] = 0; rb_in[u] = false;
            if (usable && i < d.nbk) {
                const uint32_t e = ld32(a->f.st.rdir, (uint32_t)(d.b0 + i));
                rb_in[u] = e < d.st_r0 + d.st_nk;
                if (rb_in[u]) rb_k1[u] = a->f.st.ent[e].k1;
            }
        }
        while (usable) {
            if (threadIdx.x < (uint32_t)WAVE) {
                const int nx = chunk_window(a, lane, d.tid, tile_lo, thi, scan_from, chunk_members, &s_tw);
                if (lane == 0) s_scan = nx;
            }
            __syncthreads();
            const int scan_next = s_scan;
            const int w_n = (int)s_tw.d.n_win;
            TileDesc dc = d;
            dc.j_lo = s_tw.d.j_lo; dc.n_win = (uint32_t)w_n; dc.flags = (d.flags & ~TD_CONTIG) | (s_tw.d.flags & TD_CONTIG);
            // ---- stage the entries of the dictionary slices that can say something about this chunk's members, masks re-based to the
            //      chunk: a counting sort by bucket (the order inside a bucket does not matter: the probes OR over it)
            const int n_grp = d.nbk + 1;                                // groups 0 .. nbk
            const int tx_lo = dc.j_lo, tx_hi = w_n > 0 ? s_tw.win[w_n - 1] : dc.j_lo - 1;
            auto group_of = [&](int32_t k1) { return (uint32_t)(min(max((k1 >> SITE_SHIFT) + d.b_off, -1), d.nbk - 1) + 1); };
            auto relevant = [&](int32_t tx_base) { return tx_base <= tx_hi && tx_base + 63 >= tx_lo; };
            for (int i = (int)threadIdx.x; i < n_grp + 3; i += TILE_THREADS) { X0[i] = 0u; X1[i] = 0u; }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < d.st_nk; i += (uint32_t)TILE_THREADS) {
                const int4 xa = *reinterpret_cast<const int4 *>(a->f.st.ent + d.st_r0 + i);
                if (relevant(xa.z)) atomicAdd(&X0[group_of(xa.x)], 1u);
            }
            for (uint32_t i = threadIdx.x; i < d.en_nk; i += (uint32_t)TILE_THREADS) {
                const int4 xc = *reinterpret_cast<const int4 *>(a->f.en.ent + d.en_r0 + i);
                if (relevant(xc.z)) atomicAdd(&X1[group_of(xc.x)], 1u);
            }
            __syncthreads();
            if (threadIdx.x < 2u * (uint32_t)WAVE) {
                // inclusive prefix sums over the groups (wave 0: START, wave 1: END): X[g] = end of group g, the closing words = total
                uint32_t *const X = threadIdx.x < (uint32_t)WAVE ? X0 : X1;
                constexpr int PER = (DIR_N + WAVE - 1) / WAVE;
                uint32_t v[PER], sum = 0u;
#pragma unroll
                for (int u = 0; u < PER; ++u) { const int g = lane * PER + u; v[u] = g < n_grp + 3 ? X[g] : 0u; sum += v[u]; }
                uint32_t run = wave_inclusive_scan(sum) - sum;
#pragma unroll
                for (int u = 0; u < PER; ++u) { const int g = lane * PER + u; run += v[u]; if (g < n_grp + 3) X[g] = run; }
                if (lane == WAVE - 1) s_kept[threadIdx.x >> 6] = run;
            }
            __syncthreads();
            const bool crowded = s_kept[0] > (uint32_t)CHUNK_KEY_CAP || s_kept[1] > (uint32_t)CHUNK_KEY_CAP;
            if (crowded && scan_next >= 0 && chunk_members > 1u) {
                chunk_members >>= 1;                                     // the same stretch of the window again, with half the members
                __syncthreads();
                continue;
            }
            const bool bad = scan_next < 0 || crowded;                  // (a scan without end, one transcript with too many entries: the generic kernel)
            redo = redo || (active && bad);
            if (!bad) {
                for (uint32_t i = threadIdx.x; i < max(d.st_nk, d.en_nk); i += (uint32_t)TILE_THREADS) {
                    int4 xa = make_int4(0, 0, INT32_MIN / 2, 0), xb = make_int4(0, 0, 0, 0), xc = xa, xd = xb;
                    if (i < d.st_nk) { const int4 *qv = reinterpret_cast<const int4 *>(a->f.st.ent + d.st_r0 + i); xa = qv[0]; xb = qv[1]; }
                    if (i < d.en_nk) { const int4 *qv = reinterpret_cast<const int4 *>(a->f.en.ent + d.en_r0 + i); xc = qv[0]; xd = qv[1]; }
                    const bool has_st = i < d.st_nk && relevant(xa.z), has_en = i < d.en_nk && relevant(xc.z);
                    if (!has_st && !has_en) continue;
                    WEnt e0, e1;
                    e0.k1 = xa.x; e0.k2 = xa.y; e1.k1 = xc.x; e1.k2 = xc.y;
                    const m64_t pm0 = ((m64_t)(uint32_t)xb.y << 32) | (uint32_t)xb.x, sm0 = ((m64_t)(uint32_t)xb.w << 32) | (uint32_t)xb.z;
                    const m64_t pm1 = ((m64_t)(uint32_t)xd.y << 32) | (uint32_t)xd.x, sm1 = ((m64_t)(uint32_t)xd.w << 32) | (uint32_t)xd.z;
                    if (dc.flags & TD_CONTIG) {
                        e0.pm = rebase64(pm0, xa.z - dc.j_lo); e0.sm = rebase64(sm0, xa.z - dc.j_lo);
                        e1.pm = rebase64(pm1, xc.z - dc.j_lo); e1.sm = rebase64(sm1, xc.z - dc.j_lo);
                    } else {
                        m64_t mm[4] = {pm0, sm0, pm1, sm1};
                        rebase_gaps64(s_tw.win, w_n, mm, xa.z, xc.z);
                        e0.pm = mm[0]; e0.sm = mm[1]; e1.pm = mm[2]; e1.sm = mm[3];
                    }
                    // (places are handed out from the end of the group downwards: afterwards X[g] is the group's FIRST entry)
                    if (has_st) s_ent0[atomicSub(&X0[group_of(xa.x)], 1u) - 1u] = e0;
                    if (has_en) s_ent1[atomicSub(&X1[group_of(xc.x)], 1u) - 1u] = e1;
                }
            }
            __syncthreads();
            // reach-back directory: the first staged entry of the GROUP that holds the bucket's first reaching entry (what is scanned
            // from there on is filtered by the exon's own coordinates)
            if (!bad) {
#pragma unroll
                for (int u = 0; u < RB_PER; ++u) {
                    const int i = (int)threadIdx.x + u * TILE_THREADS;
                    if (i < d.nbk) XR[i] = rb_in[u] ? X0[group_of(rb_k1[u])] : X0[n_grp];
                }
            }
            __syncthreads();
            // ---- this chunk's part of the sweep
            const bool work = work0 && !redo && !known && !stopped;
            const int abl = a->f.p.ablate;                     // (timing diagnostics: 8192 no member pass, 4096 no probe rounds -- results wrong)
            const ChunkVisit vm = visit_chunk64<LEVEL>(L, (abl & 8192) ? 0 : w_n, work, n, re, s_tw.mask);
            redo = redo || vm.redo;
            const bool mapping = work && !vm.redo && n > 1 && !(abl & 4096);
            const SiteMasks64 sm = map_exons_lds64(L, dc, mapping, Ap, Lp, tile_lo, n, vm.vpre, dis, re.s0, re.el);
            // (-d > 0: a visited member with two sites within the tolerance of one read site -- its pair count is the generic kernel's)
            const bool amb = dis > 0 && mapping && (sm.amb & vm.vpre) != 0ull;
            redo = redo || amb;
            if (work && !vm.redo && !amb) {
                int jstar = -1;
                if (n > 1) {
                    m64_t c = sm.kand & vm.vpre;
                    while (c) {
                        const int j = __ffsll((long long)c) - 1;
                        c &= c - 1ull;
                        const int4 hk = L.hk[j];
                        if (hk.x <= re.e0 && re.sl <= hk.y) { jstar = j; break; }
                    }
                } else if (vm.k1mask) jstar = __ffsll((long long)vm.k1mask) - 1;
                const bool known_c = jstar >= 0;
                const m64_t upto = jstar >= 63 ? ~0ull : ((2ull << (known_c ? jstar : 0)) - 1ull);
                const m64_t V = known_c ? (vm.vpre & upto) : vm.vpre;
                const m64_t ks = (n > 1) ? (sm.kor & V) : 0ull;
                ksite = ksite || (ks & ~(known_c ? (1ull << jstar) : 0ull)) != 0ull;
                int jref = -1;
                if (n > 1) { if (ks) jref = 63 - __clzll((long long)ks); }
                else jref = jstar;
                if (jref >= 0) { ref = L.win[jref]; out_rev = ((L.hk[jref].w >> 8) & 1) != 0; }     // :825-831 (a later chunk's member is a later transcript)
                if (LEVEL >= 1 && LEVEL <= 4) { lfull = lfull || (vm.lmask & V) != 0ull; rfull = rfull || (vm.rmask & V) != 0ull; }
                if (LEVEL == 3 || LEVEL == 4) {
                    if (lnoth) {
                        if (sm.dm_first & V) lnoth = false;
                        else if (V) lnoth = (overlapping_exon_members64(L.rdir, L.dir0, L.ent0, dc.b_off, dc.nb, re.s0, re.e0) & V) == 0ull;
                    }
                    if (LEVEL == 3 && rnoth) {
                        if (sm.am_last & V) rnoth = false;
                        else if (V) rnoth = (overlapping_exon_members64(L.rdir, L.dir0, L.ent0, dc.b_off, dc.nb, re.sl, re.el) & V) == 0ull;
                    }
                }
                if (n > 1) {
                    const uint32_t lim = known_c ? (uint32_t)jstar : 62u;                   // (63 = no member)
                    for (int k = 0; k < (int)n; ++k) {
                        const uint32_t w = Ap[k] >> SLAB_REL_BITS;
                        uint32_t clr = ((w & 63u) <= lim ? (uint32_t)F_EXON : 0u) | (((w >> 6) & 63u) <= lim ? (uint32_t)F_JUNC : 0u);
                        if (!known_c) clr |= (((w >> 12) & 1u) ? (uint32_t)F_DON : 0u) | (((w >> 13) & 1u) ? (uint32_t)F_ACC : 0u);
                        Fp[k] = (uint8_t)(Fp[k] & ~clr);
                    }
                }
                known = known_c;
                stopped = vm.stopped;
            }
            __syncthreads();                                    // (the next chunk overwrites the window and the staged entries)
            if (bad || scan_next >= a->f.p.n_tx || w_n == 0) break;
            scan_from = scan_next;
        }
        // ---- verdicts; flag bytes into the staged positions
        uint32_t info = n << 8;
        if (work0 && !redo) {
            if (n > 1) {
                for (int k = 0; k < (int)n; ++k) {
                    uint32_t f = Fp[k];
                    if (known) f &= ~(uint32_t)(F_DON | F_ACC);                             // (every site of a known read is its transcript's)
                    f &= (k + 1 == (int)n) ? (uint32_t)F_EXON : 0xffu;                      // the last exon has no junction behind it
                    Ap[k] = (Ap[k] & SLAB_REL_MASK) | (f << SLAB_REL_BITS);
                }
            } else Ap[0] = (Ap[0] & SLAB_REL_MASK) | ((uint32_t)F_EXON << SLAB_REL_BITS);
            if (known) info |= I_KNOWN;
            if (ksite) info |= I_KSITE;
            if (full_decision(LEVEL, lfull, lnoth, rfull, rnoth)) info |= I_FULL;
            if (out_rev) info |= I_REV;
            if (fast_args()->p.n_sj == 0 && (info & (I_FULL | I_KNOWN | I_KSITE)) == (I_FULL | I_KSITE)) info |= I_ACCEPT;
        } else {
            ref = -1;
            if (work0) for (uint32_t k = 0; k < n; ++k) Ap[k] &= SLAB_REL_MASK;             // (flags 0: the generic kernel writes them)
        }
        redo = redo && active;
        {
            const unsigned long long m = __ballot(redo);
            if (m) {
                uint32_t pos_r = 0;
                if (lane == 0) pos_r = atomicAdd(a->f.redo_count, (uint32_t)__popcll(m));
                pos_r = __shfl(pos_r, 0, WAVE);
                if (redo) a->f.redo[pos_r + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = r;
            }
        }
        if (active) { a->f.info[r] = info; a->f.ref_tx[r] = ref; a->f.ex_off[r] = out.dst; }
        __syncthreads();
        slab_write_out(SlabOut{out.start, out.end, out.flag, xbase}, s_A, s_L, tile_lo, s_lim);
        __syncthreads();                                        // (the next entry of this workgroup overwrites the LDS image)
    }
}

}  // namespace l2r
