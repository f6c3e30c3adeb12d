// l2r_wide.hip.h -- the slab pipeline's probe kernel for tiles whose window holds 33 .. 63 transcripts (WIDE tiles).
//
// The mask path of l2r_kernels.hip.h / l2r_slab.hip.h keeps one bit per window member in 32-bit words; a tile whose reads can
// meet more than 32 annotation transcripts (a locus with many isoforms) went to the redo list, i.e. to k_classify_generic at
// about 1/70 of the speed.  This file is the same formulation on 64-bit masks, for those tiles only:
// k_walk_slab's last wave builds a 64-member window record (TileWin64) for them and appends the tile to a list;
// k_probe_slab skips them; k_probe_slab_wide<LEVEL>, a persistent grid, walks over the list.  A tile beyond 63 members has no
// window record at all: k_probe_slab_chunked (l2r_chunk.hip.h) takes its window 63 members at a time.
// With an isoform-rich annotation this is the kernel that classifies most reads, so it works like k_probe_slab: rows of the
// slab four exons ahead, exons + work words + flags staged at their positions in LDS, one coalesced write-out per entry.
//
// Same semantics as visit_window / map_exons_slab / decide, line by line, with these differences:
//   masks             unsigned long long; a window holds at most WIDE_MEMBERS = 63 transcripts, so that a "first member" index
//                     still takes 6 bits (63 = none) and a work word 6 + 6 + 2 bits: the upper 14 bits of a staged position
//   staged entries    {k1, k2, pair mask, single mask} = 24 bytes (two 8-byte key/mask reads) instead of one 16-byte vector
#pragma once
#include "l2r_slab.hip.h"

namespace l2r {

typedef unsigned long long m64_t;
struct WEnt { int32_t k1, k2; m64_t pm, sm; };                  // staged dictionary entry, masks in the tile's window frame
constexpr int WIDE_KEY_CAP = SLAB_KEY_CAP;
constexpr int WIDE_TW_VECS = (int)(sizeof(TileWin64) / 16);

struct VisitMasks64 { m64_t vpre, lmask, rmask, k1mask; bool redo; };
struct SiteMasks64 { m64_t kand, kor, dm_first, am_last, amb; };      // amb (-d > 0 only): members with TWO sites within the tolerance of one read site (probe_near64)
struct WideLds { const WEnt *ent0, *ent1; const uint8_t *dir0, *dir1, *rdir; const int4 *hk, *hx; const int *win; };

__device__ __forceinline__ m64_t rebase64(m64_t m, int d)
{
    if (d >= 0) return d < 64 ? m << d : 0ull;
    return -d < 64 ? m >> (-d) : 0ull;
}
// the masks of one START and one END entry (pair, single each) from their own frames (bit b = annotation transcript base + b) to
// the window's (bit j = member j): one pass over the members for the four of them, the loads of win[] eight at a time
__device__ __forceinline__ void rebase_gaps64(const int *win, int w_n, m64_t (&m)[4], int base_st, int base_en)
{
    m64_t out[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll 8
    for (int j = 0; j < WIDE_TX; ++j) {
        const int w = win[j];
        const uint32_t b0 = (uint32_t)(w - base_st), b1 = (uint32_t)(w - base_en);
        const uint32_t c0 = b0 < 64u ? b0 : 0u, c1 = b1 < 64u ? b1 : 0u;
        const m64_t k0 = b0 < 64u ? 1ull : 0ull, k1 = b1 < 64u ? 1ull : 0ull;
        out[0] |= ((m[0] >> c0) & k0) << j; out[1] |= ((m[1] >> c0) & k0) << j;
        out[2] |= ((m[2] >> c1) & k1) << j; out[3] |= ((m[3] >> c1) & k1) << j;
    }
    const m64_t keep = w_n >= 64 ? ~0ull : ((1ull << w_n) - 1ull);            // (win[] behind the members is not initialised)
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = out[i] & keep;
}
__device__ __forceinline__ uint32_t first_member64(m64_t x) { return x ? (uint32_t)(__ffsll((long long)x) - 1) : 63u; }      // (members 0 .. 62)

// visit_window (l2r_kernels.hip.h) on 64 members
template <int LEVEL>
__device__ __forceinline__ VisitMasks64 visit_window64(const WideLds &L, const TileDesc &d, int w_n, bool work, uint32_t n, int j0,
                                                       const ReadEnds &re, const m64_t *tilemask)
{
    VisitMasks64 m{0ull, 0ull, 0ull, 0ull, false};
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
    int jrel0 = j0 - d.j_lo;                       // first member the read's sweep reaches
    if (!(d.flags & TD_CONTIG)) {
        jrel0 = 0;
#pragma unroll 8
        for (int j = 0; j < w_n; ++j) jrel0 += L.win[j] < j0 ? 1 : 0;
    }
    const m64_t reach = jrel0 <= 0 ? ~0ull : (jrel0 >= 64 ? 0ull : ~((1ull << jrel0) - 1ull));
    const m64_t stop = m_aft & reach;                                        // the sweep ends at the lowest of these (:799-800)
    const m64_t below = (stop & (0ull - stop)) - 1ull;                       // all ones when there is none
    m.vpre = work ? (~m_bef & below & reach & (w_n >= 64 ? ~0ull : ((1ull << w_n) - 1ull))) : 0ull;
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

__device__ __forceinline__ void probe_all64(const WEnt *ent, uint32_t lo, uint32_t hi, int32_t k1, int32_t k2, m64_t &pm, m64_t &sm)
{
    pm = 0ull; sm = 0ull;
    for (uint32_t r = lo; r < hi; ++r) {
        const WEnt q = ent[r];
        if (q.k1 == k1) { sm = q.sm; if (q.k2 == k2) pm = q.pm; }
    }
}

// probe_near (l2r_kernels.hip.h: -d > 0, src/update_gtf.c:717-779 with dis > 0) on 64-bit masks; entries [lo, hi) = the staged entries of
// the buckets of k1 - dis .. k1 + dis, sorted by (key 1, key 2)
__device__ __forceinline__ void probe_near64(const WEnt *ent, uint32_t lo, uint32_t hi, int32_t k1, int32_t k2, int dis, int rs, int re, m64_t &pm, m64_t &sm, m64_t &amb)
{
    pm = 0ull; sm = 0ull;
    int last = INT32_MIN;
    for (uint32_t r = lo; r < hi; ++r) {
        const WEnt q = ent[r];
        if (q.k1 > k1 + dis) break;
        if (q.k1 < k1 - dis) continue;
        if (q.k1 >= rs && q.k1 <= re) {
            if (q.k1 != last) { amb |= sm & q.sm; last = q.k1; }
            sm |= q.sm;
        }
        if (__builtin_abs(q.k2 - k2) <= dis) pm |= q.pm;
    }
}

template <typename DirT>
__device__ __forceinline__ m64_t overlapping_exon_members64(const DirT *rdir, const DirT *dir, const WEnt *ent, int b_off, int nb, int s, int e)
{
    const int bs = s >> SITE_SHIFT;
    if (bs >= nb) return 0ull;
    const int be = min(e >> SITE_SHIFT, nb - 1);
    m64_t m = 0ull;
    const uint32_t i1 = dir[be + b_off + 1];
    for (uint32_t i = rdir[bs + b_off]; i < i1; ++i) {
        const WEnt q = ent[i];
        if (q.k1 <= e && q.k2 >= s) m |= q.pm;
    }
    return m;
}

// map_exons_slab on 64-bit masks: rows k .. k + 3 of the column in four register pairs with fixed roles, every round leaves its
// exon and its work word at the exon's position in LDS (SlabStage)
__device__ __forceinline__ SiteMasks64 map_exons_slab64(const WideLds &L, const TileDesc &d, bool mapping, const uint32_t *__restrict__ xw,
                                                        uint32_t off, uint32_t n, m64_t vpre, const SlabRows &q, const SlabStage &st, int dis = 0, int rs = 0, int re = 0)
{
    SiteMasks64 m{~0ull, 0ull, 0ull, 0ull, 0ull};
    uint32_t *const Ap = st.A + st.loc; uint16_t *const Lp = st.Ln + st.loc;
    SlabRow R[SLAB_AHEAD];
#pragma unroll
    for (int i = 0; i < SLAB_AHEAD; ++i) R[i] = n == (uint32_t)i + 1u ? q.last : q.x[i];
    const uint32_t none = (uint32_t)d.nbk + 1u;         // a bucket behind the staged ones: the staging leaves it empty
    const int k_max = wave_max(mapping ? (int)n : 0);
    const uint32_t nm1 = mapping ? n - 1u : 0u;
    int e_cur = slab_row_end(R[0], st.lo);
    auto round = [&](int k, SlabRow &cur, const SlabRow &nxt, bool reload) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        const int s = slab_row_start(cur, st.lo), e = e_cur, s2 = slab_row_start(nxt, st.lo), e2 = slab_row_end(nxt, st.lo);
        const uint32_t cw = cur.w;
        m64_t xm, am, jm, dm;
        if (dis > 0) {                                  // (wave-uniform: -d)
            uint32_t ls, hs, le, he;
            near_range(L.dir0, d.b_off, none, live, s, dis, ls, hs); near_range(L.dir1, d.b_off, none, junc, e, dis, le, he);
            probe_near64(L.ent0, ls, hs, s, e, dis, rs, re, xm, am, m.amb);
            probe_near64(L.ent1, le, he, e, s2, dis, rs, re, jm, dm, m.amb);
        } else {
            const uint32_t is = live ? min((uint32_t)((s >> SITE_SHIFT) + d.b_off), none) : none;
            const uint32_t ie = junc ? min((uint32_t)((e >> SITE_SHIFT) + d.b_off), none) : none;
            const uint32_t ls = L.dir0[is], hs = L.dir0[is + 1u], le = L.dir1[ie], he = L.dir1[ie + 1u];
            probe_all64(L.ent0, ls, hs, s, e, xm, am);
            probe_all64(L.ent1, le, he, e, s2, jm, dm);
        }
        if (reload) {                                   // exon k + SLAB_AHEAD into the registers of exon k
            const uint32_t j = (uint32_t)k + (uint32_t)SLAB_AHEAD;
            cur = slab_load_row(xw, off + (j < nm1 ? j + 1u : 0u) * SLAB_STRIDE);
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
        if (live) { Ap[k] = (cw & SLAB_REL_MASK) | (word << SLAB_REL_BITS); Lp[k] = (uint16_t)(cw >> SLAB_REL_BITS); }
        e_cur = e2;
    };
    // whole groups of SLAB_AHEAD rounds (one back edge, no exit inside: the wait in front of a row then counts the loads behind
    // it), then up to SLAB_AHEAD - 1 more rounds that ask for nothing
    int k = 0;
    for (; k + SLAB_AHEAD <= k_max; k += SLAB_AHEAD) {
#pragma unroll
        for (int i = 0; i < SLAB_AHEAD; ++i) round(k + i, R[i], R[(i + 1) % SLAB_AHEAD], true);
    }
#pragma unroll
    for (int i = 0; i < SLAB_AHEAD - 1; ++i) {
        if (k + i >= k_max) break;
        round(k + i, R[i], R[i + 1], false);
    }
    return m;
}

// decide (l2r_kernels.hip.h) on 64-bit masks; work words through getw(k), flag bytes through emit(k, f)
template <int LEVEL, typename GetW, typename Emit>
__device__ __forceinline__ Verdict decide64(const WideLds &L, const TileDesc &d, uint32_t n, const ReadEnds &re,
                                            const VisitMasks64 &vm, const SiteMasks64 &sm, bool rev_in, GetW getw, Emit emit)
{
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
    const bool known = jstar >= 0;
    const m64_t upto = jstar >= 63 ? ~0ull : ((2ull << (known ? jstar : 0)) - 1ull);
    const m64_t V = known ? (vm.vpre & upto) : vm.vpre;
    const m64_t ks = (n > 1) ? (sm.kor & V) : 0ull;
    const bool ksite = (ks & ~(known ? (1ull << jstar) : 0ull)) != 0ull;
    int jref = -1;
    if (n > 1) { if (ks) jref = 63 - __clzll((long long)ks); }
    else jref = jstar;
    bool lfull = false, rfull = false, lnoth = true, rnoth = true;
    if (LEVEL >= 1 && LEVEL <= 4) { lfull = (vm.lmask & V) != 0ull; rfull = (vm.rmask & V) != 0ull; }
    if (LEVEL == 3 || LEVEL == 4) {
        if (!lfull) {
            if (sm.dm_first & V) lnoth = false;
            else if (V) lnoth = (overlapping_exon_members64(L.rdir, L.dir0, L.ent0, d.b_off, d.nb, re.s0, re.e0) & V) == 0ull;
        }
        if (LEVEL == 3 && !rfull) {
            if (sm.am_last & V) rnoth = false;
            else if (V) rnoth = (overlapping_exon_members64(L.rdir, L.dir0, L.ent0, d.b_off, d.nb, re.sl, re.el) & V) == 0ull;
        }
    }
    const uint32_t lim = known ? (uint32_t)jstar : 62u;                        // (63 = no member)
    if (n > 1) {
        for (int k = 0; k < (int)n; ++k) {
            const uint32_t w = getw(k);
            uint32_t f = ((w & 63u) > lim ? (uint32_t)F_EXON : 0u) | (((w >> 6) & 63u) > lim ? (uint32_t)F_JUNC : 0u);
            if (!known) f |= (((w >> 12) & 1u) ? 0u : (uint32_t)F_DON) | (((w >> 13) & 1u) ? 0u : (uint32_t)F_ACC);
            f &= (k + 1 == (int)n) ? (uint32_t)F_EXON : 0xffu;                   // the last exon has no junction behind it
            emit(k, f);
        }
    } else emit(0, (uint32_t)F_EXON);
    int ref = -1;
    bool out_rev = rev_in;
    if (jref >= 0) { ref = L.win[jref]; out_rev = ((L.hk[jref].w >> 8) & 1) != 0; }     // :825-831
    uint32_t info = 0;
    if (known) info |= I_KNOWN;
    if (ksite) info |= I_KSITE;
    if (full_decision(LEVEL, lfull, lnoth, rfull, rnoth)) info |= I_FULL;
    if (out_rev) info |= I_REV;
    if (fast_args()->p.n_sj == 0 && (info & (I_FULL | I_KNOWN | I_KSITE)) == (I_FULL | I_KSITE)) info |= I_ACCEPT;
    return Verdict{info | (n << 8), ref};
}

struct WideArgs { const TileWin64 *tw64; };            // tw64[tile]: the 64-member window record k_walk_slab left for a TD_WIDE tile

template <int LEVEL>
__global__ __launch_bounds__(TILE_THREADS, 5)
void k_probe_slab_wide(SlabArgs kernarg_block, WideArgs wa, const uint32_t *__restrict__ u_tile_first, const int32_t *__restrict__ u_pos, const uint32_t *__restrict__ u_tile_sbase,
                       const TileWin *__restrict__ u_tw, const uint32_t *__restrict__ u_xbase, const TileStat *__restrict__ u_stat /* one-kernel tile path, else null */)
{
    constexpr int DIR_BYTES = FAST_DIR_BYTES;
    __shared__ __attribute__((aligned(16))) uint32_t s_A[SLAB_POS_CAP];
    __shared__ __attribute__((aligned(16))) uint16_t s_L[SLAB_POS_CAP];
    __shared__ __attribute__((aligned(16))) WEnt s_ent[2 * WIDE_KEY_CAP];
    __shared__ __attribute__((aligned(16))) uint8_t s_dir[3 * DIR_BYTES];
    __shared__ __attribute__((aligned(16))) TileWin64 s_tw;
    __shared__ uint32_t s_next, s_lim;
    (void)kernarg_block;
    const SlabArgsK sa = slab_args();
    const PipeArgsK a = pipe_args();
    const int lane = threadIdx.x & (WAVE - 1);
    WEnt *const s_ent0 = s_ent, *const s_ent1 = s_ent + WIDE_KEY_CAP;
    uint8_t *const s_dir0 = s_dir, *const s_dir1 = s_dir + DIR_BYTES, *const s_rdir = s_dir + 2 * DIR_BYTES;
    // the tiles of this kernel: wide_list (TileLists), taken from a cursor -- they differ in cost; a workgroup's FIRST entry is its
    // own number: an empty list costs no atomic
    // (one-kernel tile path: the wide tiles k_tile has given the slab form -- its list behind wide_list, list_cnt[5] entries; the others
    //  are its WIDE instance's)
    const uint32_t *const w_list = u_stat ? sa->wide_list + sa->n_tiles + 1u : sa->wide_list;
    const uint32_t n_wide = u_stat ? sa->list_cnt[5] : sa->list_cnt[0];
    for (bool own = true;; own = false) {
        if (own && blockIdx.x >= n_wide) break;
        if (threadIdx.x == 0) s_next = own ? blockIdx.x : gridDim.x + atomicAdd(sa->list_cnt + 2, 1u);
        __syncthreads();
        const uint32_t wi = s_next;
        if (wi >= n_wide) break;
        const uint32_t t = w_list[wi];
        const uint32_t tflags = u_tw[t].d.flags;
        const uint32_t r0 = u_tile_first[t], n_act = u_tile_first[t + 1u] - r0;
        const uint32_t sbase = u_tile_sbase[t], xbase = u_xbase[t], total = u_xbase[t + 1u] - xbase;
        const int32_t tile_lo = u_pos[r0] + 1;                   // the base of the tile's row words
        for (int i = (int)threadIdx.x; i < WIDE_TW_VECS; i += TILE_THREADS)
            reinterpret_cast<int4 *>(&s_tw)[i] = reinterpret_cast<const int4 *>(wa.tw64 + t)[i];
        const uint32_t row_max = max((u_tw[t].pad[0] >> (8u * (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)))) & 0xffu, 1u) - 1u;
        bool active = threadIdx.x < n_act;
        const uint32_t at = r0 + (active ? threadIdx.x : 0u);
        uint32_t pre = 0u, loc = 0u;
        const uint32_t *const xw = sa->slab_row;
        const uint32_t off = sbase + threadIdx.x;
        SlabRows q;
        q.last = SlabRow{0u};
#pragma unroll
        for (int i = 0; i < SLAB_AHEAD; ++i) q.x[i] = SlabRow{0u};
        if (active) {
            slab_preloc(sa, t, at, pre, loc);
            q.last = slab_load_row(xw, off);
#pragma unroll
            for (int i = 0; i < SLAB_AHEAD; ++i) q.x[i] = slab_load_row(xw, off + min((uint32_t)i + 1u, row_max) * SLAB_STRIDE);      // (an outlier's column holds nothing: read, not used)
        }
        const uint32_t idx = pre & 0xffu;
        const uint32_t n = pre >> PRE_N_SHIFT;
        const uint32_t r = r0 + idx;
        const bool outlier = (pre & PRE_DENSE) != 0u, rev_in = (pre & PRE_REV) != 0u;
        const SlabRow first = n == 1u ? q.last : q.x[0];
        const ReadEnds re{slab_row_start(first, tile_lo), slab_row_end(first, tile_lo), slab_row_start(q.last, tile_lo), slab_row_end(q.last, tile_lo)};
        __syncthreads();
        const TileDesc d = s_tw.d;
        const int w_n = (int)d.n_win;
        // ---- stage the dictionary slices, masks re-based to the tile's window (64-bit)
        const DictRegs dv = load_dict_slices(a, d);
        int my_wide = 0;
        if ((int)threadIdx.x < WIDE_KEY_CAP) {
            const bool has_st = threadIdx.x < d.st_nk, has_en = threadIdx.x < d.en_nk;
            WEnt e0, e1;
            e0.k1 = dv.xa.x; e0.k2 = dv.xa.y; e1.k1 = dv.xc.x; e1.k2 = dv.xc.y;
            const m64_t pm0 = ((m64_t)(uint32_t)dv.xb.y << 32) | (uint32_t)dv.xb.x, sm0 = ((m64_t)(uint32_t)dv.xb.w << 32) | (uint32_t)dv.xb.z;
            const m64_t pm1 = ((m64_t)(uint32_t)dv.xd.y << 32) | (uint32_t)dv.xd.x, sm1 = ((m64_t)(uint32_t)dv.xd.w << 32) | (uint32_t)dv.xd.z;
            if (d.flags & TD_CONTIG) {
                e0.pm = rebase64(pm0, dv.xa.z - d.j_lo); e0.sm = rebase64(sm0, dv.xa.z - d.j_lo);
                e1.pm = rebase64(pm1, dv.xc.z - d.j_lo); e1.sm = rebase64(sm1, dv.xc.z - d.j_lo);
            } else {
                m64_t mm[4] = {pm0, sm0, pm1, sm1};
                rebase_gaps64(s_tw.win, w_n, mm, dv.xa.z, dv.xc.z);
                e0.pm = mm[0]; e0.sm = mm[1]; e1.pm = mm[2]; e1.sm = mm[3];
            }
            if (has_st) { s_ent0[threadIdx.x] = e0; if (dv.xa.w & SE_WIDE) my_wide = 1; }
            if (has_en) { s_ent1[threadIdx.x] = e1; if (dv.xc.w & SE_WIDE) my_wide = 1; }
        }
        if (d.nbk > 0) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int i = (int)threadIdx.x + qq * TILE_THREADS;
                if (i <= d.nbk) {
                    s_dir0[i] = (uint8_t)(dv.dd[0][qq] - d.st_r0); s_dir1[i] = (uint8_t)(dv.dd[1][qq] - d.en_r0);
                    s_rdir[i] = (uint8_t)(dv.dd[2][qq] - d.st_r0);
                }
            }
        }
        if (threadIdx.x < 3u && (threadIdx.x > 0u || d.nbk == 0)) {
            s_dir0[d.nbk + (int)threadIdx.x] = (uint8_t)d.st_nk; s_dir1[d.nbk + (int)threadIdx.x] = (uint8_t)d.en_nk;
        }
        if (threadIdx.x == 0) s_lim = min(total, (uint32_t)SLAB_POS_CAP);
        const int any_wide = __syncthreads_or(my_wide);
        if (any_wide && sa->chunk_on) {                         // (a key with several entries: k_probe_slab_chunked ORs them)
            if (threadIdx.x == 0) { sa->tw[t].d.flags = tflags | TD_CHUNK; chunk_list_append_late(sa, t); }
            __syncthreads();
            continue;
        }
        // a read is staged when its positions fit and it is no outlier
        const SlabOut out{a->f.ex_start, a->f.ex_end, a->f.ex_flag, xbase + loc};
        const SlabStage st{s_A, s_L, loc, tile_lo, active && loc + n <= (uint32_t)SLAB_POS_CAP && !(pre & PRE_DENSE)};
        if (active && loc + n > (uint32_t)SLAB_POS_CAP) atomicMin(&s_lim, loc);
        // ---- classification
        uint32_t info = n << 8; int ref = -1;
        bool redo = active && (!(d.flags & TD_WIDE) || outlier || !st.fits || any_wide != 0 || (n > 1 && (pre & PRE_INSANE) != 0u));
        const bool work = active && !redo;
        const WideLds L{s_ent0, s_ent1, s_dir0, s_dir1, s_rdir, s_tw.hk, s_tw.hx, s_tw.win};
        const VisitMasks64 vm = visit_window64<LEVEL>(L, d, w_n, work, n, d.j_lo, re, s_tw.mask);
        redo = redo || vm.redo;
        const bool mapping = work && !redo && n > 1;
        const int dis = a->f.p.ss_dis;
        const SiteMasks64 sm = map_exons_slab64(L, d, mapping, xw, off, n, vm.vpre, q, st, dis, re.s0, re.el);
        if (active && !mapping) slab_copy_exons(sa, a, out, st, q, off, n, pre, r);
        // (-d > 0: a visited member with two sites within the tolerance of one read site -- its pair count is the generic kernel's)
        if (dis > 0 && mapping && (sm.amb & vm.vpre) != 0ull) redo = true;
        if (work && !redo) {
            uint32_t *const Ap = s_A + loc;
            // (the read's ends once more, from its staged exons: four registers that need not live through the probe rounds)
            const uint16_t *const Lq = s_L + loc;
            ReadEnds re2;
            re2.s0 = tile_lo + (int)(Ap[0] & SLAB_REL_MASK); re2.e0 = re2.s0 + (int)Lq[0] - 1;
            re2.sl = tile_lo + (int)(Ap[n - 1u] & SLAB_REL_MASK); re2.el = re2.sl + (int)Lq[n - 1u] - 1;
            const Verdict vd = decide64<LEVEL>(L, d, n, re2, vm, sm, rev_in, [&](int k) { return Ap[k] >> SLAB_REL_BITS; },
                                               [&](int k, uint32_t f) { Ap[k] = (Ap[k] & SLAB_REL_MASK) | (f << SLAB_REL_BITS); });
            info = vd.info; ref = vd.ref;
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
