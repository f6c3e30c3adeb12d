// l2r_slab.hip.h -- the one-walk pipeline as TWO light kernels that both run at (nearly) full occupancy (gfx950).
//
// Measured on MI355X (profiles/r02): one wave issues an instruction every 13-18 cycles in these kernels whatever is done to
// its instruction count -- they are bound by the latency of a wave's dependent instruction chains -- and a SIMD's issue
// rate grows with its resident waves up to 8 (ubench_valu_issue.txt).  k_classify_fast / k_fused keep a tile's exons in
// LDS (10 bytes each, 40 KB per workgroup): 4 waves per SIMD.  Here the exons go through HBM in a layout that both
// sides touch with full 256- / 512-byte rows, and neither kernel needs them in LDS:
//
//   k_order (l2r_fused.hip.h)   the tile's reads by falling CIGAR length = the SLOT of every read (lane order of both kernels)
//   k_walk_slab     one lane per read, CIGAR words in registers, ONE walk; exon k of the read in slot s goes to
//                   element k * 256 + s of the tile's SLAB of the result arrays (row k = exon k of all reads of the
//                   tile: a wave stores whole rows); read ends -> the tile's span.  No LDS, no barrier.
//   k_probe_slab    per tile: one wave makes the descriptor and window (make_descriptor), the dictionary slices are
//                   staged in LDS (21 KB with the per-exon work words: 7 workgroups per CU, <= 64 VGPRs); every lane
//                   reads its exons back row by row (coalesced), window pass, probes, verdicts with the device functions
//                   of the classic kernel; flag bytes go to the slab of ex_flag.
//
// The result arrays are slabs: ex_off[r] = element of exon 0, exon k at ex_off[r] + k * 256.  A tile's slab has as many rows
// as its longest read can have exons (upper bound from the CIGAR lengths, at most SLAB_ROWS); reads beyond that bound
// ("outliers") are walked literally into a dense area behind the slabs (ex_off | EXOFF_DENSE, stride 1) and classified
// by the generic kernel.  l2r_download() turns slabs into read order (k_linearize_slab).
// HBM traffic: CIGAR once, exons written once and read once.
#pragma once
#include "l2r_fused.hip.h"

namespace l2r {

constexpr int SLAB_ROWS = 24;                            // rows of a tile's slab at most (exons of its longest read it can hold)
constexpr uint32_t SLAB_STRIDE = TILE_THREADS;           // elements between exon k and exon k + 1 of a read
constexpr uint32_t EXOFF_DENSE = EX_DENSE_FLAG;          // ex_off flag (slab pipeline only): exons at stride 1
constexpr uint32_t I_PRE_INSANE = I_UNREL;               // k_walk_slab -> k_probe_slab, in info[]: first or last exon empty
constexpr uint32_t I_PRE_DIRECT = I_SJCHK;               // ... an outlier: its exons are in the dense area, the generic kernel classifies it

// c ops -> is the read an outlier of the slab layout?  (exon_bound with min_exon >= 1, the only case this pipeline takes)
__host__ __device__ __forceinline__ uint32_t slab_rows_of(uint32_t c) { return (c + 3u) >> 1; }

struct SlabArgs {
    FusedArgs g;
    const uint32_t *tile_sbase;                          // first element of every tile's slab
    unsigned long long *ovf_cursor; uint32_t ovf_base;   // dense area behind the slabs for outliers
    // the records' fields in slot order (k_order): c_lo, number of ops (65535: more), pos, strand
    const uint32_t *s_clo; const uint16_t *s_ncig; const int32_t *s_pos; const uint8_t *s_rev;
    uint32_t *pre;                                       // k_walk_slab -> k_probe_slab, slot order: exon count << 8 | I_PRE_*
    TileWin *tw;                                         // k_walk_slab -> k_probe_slab: descriptor + window per tile
    // tiles whose window holds 33 .. 64 transcripts (l2r_wide.hip.h): wide_cnt[0] counts the appends of k_walk_slab, k_probe_slab
    // moves the count to wide_cnt[1] (what k_probe_slab_wide reads) and clears [0] for the next run
    uint32_t *wide_cnt; uint32_t *wide_tile; TileWin64 *tw64; uint32_t wide_cap;
    uint32_t n_tiles;
};
typedef const __attribute__((address_space(4))) SlabArgs *SlabArgsK;
__device__ __forceinline__ SlabArgsK slab_args()
{
    SlabArgsK q = (SlabArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}
constexpr int SLAB_TW_VECS = (int)(sizeof(TileWin) / 16);
// Which 16-byte vectors of a window record carry something for a window of n_win members: the members' two header arrays, their
// transcript numbers (four per vector), and the descriptor + masks at the end.  Only those travel from k_walk_slab to k_probe_slab.
__device__ __forceinline__ bool tw_vec_used(int i, uint32_t n_win)
{
    if (i < WIN_TX) return (uint32_t)i < n_win;
    if (i < 2 * WIN_TX) return (uint32_t)(i - WIN_TX) < n_win;
    if (i < 2 * WIN_TX + WIN_TX / 4) return (uint32_t)(4 * (i - 2 * WIN_TX)) < n_win;
    return true;
}
// Workgroup -> tile.  Workgroups are handed to the 8 XCDs round robin (workgroup b runs on XCD b % 8), each XCD has an L2 of its
// own: with this mapping an XCD works through ONE contiguous eighth of the tiles, so the dictionary slices of neighbouring
// tiles (they overlap) are fetched into one L2 instead of all eight.
// The grid is 8 * ceil(n_tiles / 8) workgroups; the few that land behind the last tile leave at once.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t grid) { return (b & 7u) * (grid >> 3) + (b >> 3); }
constexpr int SLAB_KEY_CAP = 168;                        // dictionary entries staged per dictionary and tile (k_probe_slab's LDS: 20 KB = 8 workgroups per CU)
static_assert(sizeof(TileWin) % 16 == 0, "TileWin is copied in 16-byte pieces");

// Both kernels are laid out for SHORT dependent load chains (they are latency bound: a tile's time is the sum of its
// dependent round trips): everything a lane needs sits at "tile start + slot".
// The walk of one tile's reads, one lane per read (slot order): exon k of the read in slot s goes to row k of the tile's slab.
// Writes the exon rows and ex_off[r]; returns what the rest of the tile's work needs.
struct SlabWalk { uint32_t pre, r, n; int el; bool active, outlier; };
__device__ __forceinline__ SlabWalk slab_walk(SlabArgsK sa, FusedArgsK a, uint32_t r0, uint32_t n_act, uint32_t sbase, int32_t tid0,
                                              const uint8_t *__restrict__ u_order)
{
    const bool active = threadIdx.x < n_act;
    const uint32_t at = r0 + (active ? threadIdx.x : 0u);
    // ---- the read in this slot and the head of its CIGAR
    FusedRead v;
    v.src = active ? 0 : -1; v.c_lo = 0u; v.n_cig = 0u; v.lub = 0u; v.pos = 0; v.tid = tid0; v.rev = 0u;
    uint32_t r = r0;
    if (active) {
        v.c_lo = ld32(sa->s_clo, at); v.n_cig = v.c_lo + (uint32_t)ld32(sa->s_ncig, at); v.pos = ld32(sa->s_pos, at);
        r = r0 + (uint32_t)ld32(u_order, at);
    }
    fused_load_words(a, v);                       // (n_cig = number of ops from here on; 65535 stands for "more")
    fused_mask_words(v);
    DevParams p;
    p.min_exon = a->f.p.min_exon; p.min_intron = a->f.p.min_intron; p.max_delet = a->f.p.max_delet;
    const uint32_t t3 = ((uint32_t)p.min_intron << 4) | 3u, t2 = ((uint32_t)(p.max_delet + 1) << 4) | 2u;
    bool outlier = slab_rows_of(v.n_cig) > (uint32_t)SLAB_ROWS;
    const int c_max = wave_max((active && !outlier) ? (int)min(v.n_cig, (uint32_t)FUSED_HEAD) : 0);
    int32_t *const xs = a->f.ex_start, *const xe = a->f.ex_end;
    uint16_t *const xl = a->f.ex_len;                   // slab rows: {start, 16-bit length}
    uint32_t n = 0u, off = 0u;
    int el = INT32_MIN;
    bool sane = true;
    if (active && !outlier) {
        // one lane per read; exon k lands in row k of the tile's slab, at the read's slot
        off = sbase + threadIdx.x;
        int start = v.pos + 1, end = v.pos;
        int s0 = 0, e0 = 0;
        bool first = true;
        uint32_t longest = 0u;
        auto step = [&](uint32_t c) {
            const uint32_t op = c & 0xfu;
            const int len = (int)(c >> 4);
            const bool cut = ((op == 3u) & (c >= t3)) | ((op == 2u) & (c >= t2));
            const bool keep = cut & (first | (end - start >= p.min_exon - 1));
            if (keep) {
                const uint32_t xlen = (uint32_t)(end - start + 1);
                st32(xs, off + n * SLAB_STRIDE, start); st32(xl, off + n * SLAB_STRIDE, (uint16_t)xlen);
                longest = max(longest, xlen);
                if (first) { s0 = start; e0 = end; }
                first = false; ++n;
            }
            start = cut ? end + len + 1 : start;
            end += len & __builtin_amdgcn_sbfe(0x18d, op, 1u);
        };
#pragma unroll
        for (int q = 0; q < FUSED_HEAD_VEC; ++q)
            if (4 * q < c_max) { step(v.cg[4 * q]); step(v.cg[4 * q + 1]); step(v.cg[4 * q + 2]); step(v.cg[4 * q + 3]); }       // (wave-uniform)
        if (v.n_cig > (uint32_t)FUSED_HEAD) {
            const uint32_t *const words = a->f.cig + v.c_lo;
            for (uint32_t i = FUSED_HEAD; i < v.n_cig; ++i) step(words[i]);
        }
        {   const uint32_t xlen = (uint32_t)(end - start + 1);
            st32(xs, off + n * SLAB_STRIDE, start); st32(xl, off + n * SLAB_STRIDE, (uint16_t)xlen);
            longest = max(longest, xlen); }
        if (first) { s0 = start; e0 = end; }
        ++n;
        el = end;
        // kept inner exons are at least min_exon >= 1 long; the first and the last one are kept whatever their length
        sane = s0 <= e0 && start <= end;
        // an exon of 64 kb or more does not fit the row format: the read is stored densely after all (below)
        if (longest > 0xffffu) { outlier = true; n = 0u; sane = true; el = INT32_MIN; }
    }
    if (active && outlier) {
        // an outlier: the literal walk (l2r_kernels.hip.h), twice -- count, take a run of the dense area, store
        const int64_t *const p_off = a->f.cig_off;
        const uint32_t n_ops = (uint32_t)(ld32(p_off, r + 1u) - ld32(p_off, r));
        const uint32_t *const words = a->f.cig + v.c_lo;
        {
            WalkState w{v.pos + 1, v.pos, 0};
            auto none = [&](int, int, int) {};
            walk_ops<false>(w, words, 0, (int)n_ops, p, none);
            n = (uint32_t)w.n + 1u;
        }
        const uint32_t run = sa->ovf_base + (uint32_t)atomicAdd(sa->ovf_cursor, (unsigned long long)n);
        WalkState w{v.pos + 1, v.pos, 0};
        auto put = [&](int k, int s, int e) { xs[run + (uint32_t)k] = s; xe[run + (uint32_t)k] = e; sane = sane & (s <= e); el = e; };
        walk_ops<false>(w, words, 0, (int)n_ops, p, put);
        put(w.n, w.start, w.end);
        off = run | EXOFF_DENSE;
    }
    SlabWalk o;
    o.pre = (n << 8) | (sane ? 0u : I_PRE_INSANE) | (outlier ? I_PRE_DIRECT : 0u);
    o.r = r; o.n = n; o.el = el; o.active = active; o.outlier = outlier;
    // ex_off[r]: a slab read sits at "its tile's slab + its slot", which the kernels behind the classification only need for the
    // reads they touch -- every read with a junction table or an accepted list (k_validate_sj, k_gather_accepted), else only
    // the redo list's (written by the probe side when it lists a read) and the densely stored reads (here).
    if (active && (outlier || a->f.p.n_sj > 0 || (a->f.p.want & WANT_ACCEPTED))) a->f.ex_off[r] = off;
    return o;
}

__global__ __launch_bounds__(TILE_THREADS, 8)
void k_walk_slab(SlabArgs kernarg_block, const uint32_t *__restrict__ u_tile_first, const uint8_t *__restrict__ u_order,
                 const int32_t *__restrict__ u_tid, const int32_t *__restrict__ u_pos, const uint32_t *__restrict__ u_tile_sbase)
{
    __shared__ int s_wmax[TILE_THREADS / WAVE];
    __shared__ uint32_t s_wsum[TILE_THREADS / WAVE], s_wn[TILE_THREADS / WAVE];
    __shared__ __attribute__((aligned(16))) TileWin s_tw;
    __shared__ __attribute__((aligned(16))) TileWin64 s_tw64;
    (void)kernarg_block;
    const SlabArgsK sa = slab_args();
    const FusedArgsK a = fused_args();
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const uint32_t t = xcd_tile(blockIdx.x, gridDim.x);
    if (t >= sa->n_tiles) return;
    const uint32_t r0 = u_tile_first[t], n_act = u_tile_first[t + 1u] - r0;
    const int32_t tid0 = n_act ? u_tid[r0] : 0, pos0 = n_act ? u_pos[r0] : 0;
    const uint32_t sbase = u_tile_sbase[t];
    const SlabWalk w = slab_walk(sa, a, r0, n_act, sbase, tid0, u_order);
    const bool active = w.active, outlier = w.outlier;
    const uint32_t n = w.n;
    const int el = w.el;
    if (active) sa->pre[r0 + threadIdx.x] = w.pre;
    const int m = wave_max(active ? el : INT32_MIN);
    const uint32_t wsum = wave_sum(active ? n : 0u);
    const int wn = wave_max((active && !outlier) ? (int)n : 0);
    if (lane == 0) { s_wmax[wv] = m; s_wsum[wv] = wsum; s_wn[wv] = (uint32_t)min(wn, 255); }
    if (t == 0u && threadIdx.x == 0) {
        // the run's counters (this kernel is the first of a run, k_probe_slab the first to count): redo list, chunk cursor
        // of the accepted list, exon cursor.  (The cursor of the outlier area is cleared by k_probe_slab for the next run.)
        uint32_t *const cnt = a->f.redo_count;
        cnt[0] = 0u; cnt[1] = 0u; cnt[2] = 0u; cnt[3] = 0u; cnt[4] = 0u;
    }
    __syncthreads();
    // ---- the tile's descriptor and window, by the last wave alone (the others are done): nobody waits for its load chain
    if (wv != TILE_THREADS / WAVE - 1) return;
    if (lane == 0) a->tile_total[t] = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];   // (one word per tile: a single counter would serialise 156 k waves)
    make_descriptor(a, lane, tid0, pos0 + 1, max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])), true, &s_tw, (uint32_t)SLAB_KEY_CAP,
                    sa->tw64 ? &s_tw64 : nullptr);
    if (s_tw.d.flags & TD_WIDE) {
        // a window of 33 .. 64 members: the tile joins the list of k_probe_slab_wide, its 64-member record goes along
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(sa->wide_cnt, 1u);
        slot = __shfl(slot, 0, WAVE);
        if (slot < sa->wide_cap) {
            if (lane == 0) sa->wide_tile[slot] = t;
            for (int i = lane; i < (int)(sizeof(TileWin64) / 16); i += WAVE) reinterpret_cast<int4 *>(sa->tw64 + slot)[i] = reinterpret_cast<const int4 *>(&s_tw64)[i];
        } else if (lane == 0) {
            s_tw.d.flags = (s_tw.d.flags & ~(TD_WIDE | (7u << 8))) | (4u << 8);     // list full: the tile takes the generic kernel ("window > 32")
        }
    }
    if (lane == 0) s_tw.pad[0] = s_wn[0] | (s_wn[1] << 8) | (s_wn[2] << 16) | (s_wn[3] << 24);     // rows each wave of k_probe_slab has to look at
    {   const uint32_t n_win = (s_tw.d.flags & TD_FAST) ? s_tw.d.n_win : 0u;
        for (int i = lane; i < SLAB_TW_VECS; i += WAVE) if (tw_vec_used(i, n_win)) reinterpret_cast<int4 *>(sa->tw + t)[i] = reinterpret_cast<const int4 *>(&s_tw)[i]; }
}

// map_exons (l2r_kernels.hip.h) with the read's exons streamed from its slab column: row k at off + k * 256, the same row
// for the whole wave = coalesced.  Rows 0..3 come preloaded (SlabRows: the kernel asks for them with its first loads, before
// it knows the read), row k + 4 is asked for in round k.  Rows at and behind the read's exon count hold anything: not used.
// Work words at W[k * 256].
constexpr int SLAB_AHEAD = 4;                            // rows of a read's column in flight (6: no gain, measured)
struct SlabRows { int s[SLAB_AHEAD], e[SLAB_AHEAD]; };
__device__ __forceinline__ SiteMasks map_exons_slab(const TileLds &L, const TileDesc &d, bool mapping, const int32_t *__restrict__ xs,
                                                    const uint16_t *__restrict__ xl, uint32_t off, uint32_t n, uint32_t vpre,
                                                    const SlabRows &q)
{
    SiteMasks m{0xffffffffu, 0u, 0u, 0u};
    uint16_t *W = L.W + threadIdx.x;
    int s = q.s[0], e = q.e[0], s1 = q.s[1], e1 = q.e[1];
    int ps[SLAB_AHEAD - 2], pe[SLAB_AHEAD - 2];         // rows k + 2 .. k + SLAB_AHEAD - 1
#pragma unroll
    for (int i = 0; i < SLAB_AHEAD - 2; ++i) { ps[i] = q.s[i + 2]; pe[i] = q.e[i + 2]; }
    const uint32_t none = (uint32_t)d.nbk + 1u;         // a bucket behind the staged ones: the staging leaves it empty
    const int k_max = wave_max(mapping ? (int)n : 0);
    // The bucket ranges of exon k + 1 are looked up while exon k's entries are compared (the directory bytes and the entries
    // are two dependent LDS round trips: one of them per exon is taken off the chain).
    auto buckets = [&](int k, int sv, int ev, uint32_t &ls, uint32_t &hs, uint32_t &le, uint32_t &he) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        const uint32_t is = live ? min((uint32_t)((sv >> SITE_SHIFT) + d.b_off), none) : none;
        const uint32_t ie = junc ? min((uint32_t)((ev >> SITE_SHIFT) + d.b_off), none) : none;
        ls = L.dir0[is]; hs = L.dir0[is + 1u]; le = L.dir1[ie]; he = L.dir1[ie + 1u];
    };
    uint32_t ls, hs, le, he;
    buckets(0, s, e, ls, hs, le, he);
    for (int k = 0; k < k_max; ++k) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        int s4 = 0, e4 = 0;
        if (mapping) { const uint32_t i4 = off + min((uint32_t)k + (uint32_t)SLAB_AHEAD, n - 1u) * SLAB_STRIDE; s4 = ld32(xs, i4); e4 = (int)ld32(xl, i4); }   // SLAB_AHEAD rows in flight (e4: the length until the row is used)
        const int s2 = s1;
        const v4i_t qs0 = lds_entry(L.ent0, ls);
        const v4i_t qe0 = lds_entry(L.ent1, le), qe1 = lds_entry(L.ent1, le + 1u);
        uint32_t ls_n, hs_n, le_n, he_n;
        buckets(k + 1, s1, e1, ls_n, hs_n, le_n, he_n);
        uint32_t xm, am, jm, dm;
        {   const bool m0 = ls < hs && qs0.x == s;
            am = m0 ? (uint32_t)qs0.w : 0u; xm = (m0 && qs0.y == e) ? (uint32_t)qs0.z : 0u; }
        probe2(qe0, qe1, le, he, e, s2, jm, dm);
        if (__any(hs > ls + 1u || he > le + 2u)) { probe_rest(L.ent0, ls + 1u, hs, s, e, xm, am, 0u); probe_rest(L.ent1, le + 2u, he, e, s2, jm, dm, 0u); }
        const uint32_t amj = junc ? am : 0u;
        uint32_t word = first_member(xm & vpre);
        word |= first_member(jm & vpre) << 6;
        word |= nonzero(dm & vpre) << 12;
        word |= nonzero(amj & vpre) << 13;
        m.kand &= junc ? (am & dm) : 0xffffffffu;     // Q1: the acceptor probed with exon k is ITS OWN start, k < n-1
        m.kor |= amj | dm;
        if (k == 0) m.dm_first = dm;
        m.am_last = (live && !junc) ? am : m.am_last;
        if (live) W[(uint32_t)k * SLAB_STRIDE] = (uint16_t)word;
        s = s1; e = e1; s1 = ps[0]; e1 = pe[0];
#pragma unroll
        for (int i = 0; i + 1 < SLAB_AHEAD - 2; ++i) { ps[i] = ps[i + 1]; pe[i] = pe[i + 1]; }
        ps[SLAB_AHEAD - 3] = s4; pe[SLAB_AHEAD - 3] = s4 + e4 - 1;
        ls = ls_n; hs = hs_n; le = le_n; he = he_n;
    }
    return m;
}

// The tile's dictionary slices into LDS, re-based to the tile's window (entries loaded by fused_load_dict: one START and one
// END entry per thread, directory words two per thread).  `win`: the window's transcript numbers (gapped windows).
struct SlabLds { uint16_t *W; v4i_t *ent0, *ent1; uint8_t *dir0, *dir1, *rdir; };
__device__ __forceinline__ int slab_stage_dict(const TileDesc &d, const FusedDict &dv, const int *win, const SlabLds &S)
{
    const bool fast = (d.flags & TD_FAST) != 0;
    const int w_n = fast ? (int)d.n_win : 0;
    v4i_t *const s_ent0 = S.ent0, *const s_ent1 = S.ent1;
    uint8_t *const s_dir0 = S.dir0, *const s_dir1 = S.dir1, *const s_rdir = S.rdir;
    int my_wide = 0;
    if (fast) {
        if ((int)threadIdx.x < SLAB_KEY_CAP) {
            const bool has_st = threadIdx.x < d.st_nk, has_en = threadIdx.x < d.en_nk;
            v4i_t e0, e1;
            e0.x = dv.xa.x; e0.y = dv.xa.y; e1.x = dv.xc.x; e1.y = dv.xc.y;
            if (d.flags & TD_CONTIG) {
                e0.z = (int)rebase_mask((uint32_t)dv.xb.x, (uint32_t)dv.xb.y, dv.xa.z - d.j_lo);
                e0.w = (int)rebase_mask((uint32_t)dv.xb.z, (uint32_t)dv.xb.w, dv.xa.z - d.j_lo);
                e1.z = (int)rebase_mask((uint32_t)dv.xd.x, (uint32_t)dv.xd.y, dv.xc.z - d.j_lo);
                e1.w = (int)rebase_mask((uint32_t)dv.xd.z, (uint32_t)dv.xd.w, dv.xc.z - d.j_lo);
            } else {
                e0.z = (int)rebase_gaps(win, w_n, (uint32_t)dv.xb.x, (uint32_t)dv.xb.y, dv.xa.z);
                e0.w = (int)rebase_gaps(win, w_n, (uint32_t)dv.xb.z, (uint32_t)dv.xb.w, dv.xa.z);
                e1.z = (int)rebase_gaps(win, w_n, (uint32_t)dv.xd.x, (uint32_t)dv.xd.y, dv.xc.z);
                e1.w = (int)rebase_gaps(win, w_n, (uint32_t)dv.xd.z, (uint32_t)dv.xd.w, dv.xc.z);
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
    }
    return my_wide;
}

// Verdicts of one tile's reads (slot order) from the staged window + dictionaries: the device functions of the classic kernel
// on slab-resident exons; flags into the slab rows, info / ref_tx per read, redo list.
template <int LEVEL>
__device__ __forceinline__ void slab_classify(FusedArgsK a, const TileDesc &d, const SlabLds &S, const int4 *hk, const int4 *hx, const int *win,
                                              const uint32_t *tilemask, bool active, uint32_t pre, uint32_t r, bool rev_in, int32_t tid,
                                              uint32_t off, const SlabRows &q, const ReadEnds &re, int any_wide)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const bool fast = (d.flags & TD_FAST) != 0;
    const int w_n = fast ? (int)d.n_win : 0;
    const uint32_t n = pre >> 8;
    const bool outlier = (pre & I_PRE_DIRECT) != 0u;
    const int32_t *const xs = a->f.ex_start; const uint16_t *const xl = a->f.ex_len;
    uint16_t *const s_W = S.W;
    // ---- classification (device functions of the classic kernel)
    uint32_t info = n << 8; int ref = -1;
    bool redo = active && (!fast || outlier || any_wide != 0 || tid != d.tid || (n > 1 && (pre & I_PRE_INSANE) != 0u));
    const bool work = active && !redo;
    const TileLds L{nullptr, nullptr, s_W, S.ent0, S.ent1, S.dir0, S.dir1, S.rdir, hk, hx, win};
    const VisitMasks vm = visit_window<LEVEL>(L, d, w_n, work, n, d.j_lo, re, tilemask);
    redo = redo || vm.redo;
    const SiteMasks sm = map_exons_slab(L, d, work && !redo && n > 1, xs, xl, off, n, vm.vpre, q);
    uint8_t *const xf = a->f.ex_flag;
    if (work && !redo) {
        const Verdict vd = decide<LEVEL, (int)SLAB_STRIDE>(L, d, threadIdx.x, n, re, vm, sm, rev_in);
        info = vd.info; ref = vd.ref;
        for (uint32_t k = 0; k < n; ++k) st32(xf, off + k * SLAB_STRIDE, (uint8_t)s_W[k * SLAB_STRIDE + threadIdx.x]);       // (rows: coalesced)
    } else if (active && !outlier) {
        for (uint32_t k = 0; k < n; ++k) st32(xf, off + k * SLAB_STRIDE, (uint8_t)0);
    }
    redo = redo && active;
    {
        const unsigned long long m = __ballot(redo);
        if (m) {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(a->f.redo_count, (uint32_t)__popcll(m));
            at = __shfl(at, 0, WAVE);
            if (redo) { a->f.redo[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = r; if (!outlier) a->f.ex_off[r] = off; }      // (see slab_walk)
        }
    }
    if (active) { a->f.info[r] = info; a->f.ref_tx[r] = ref; }
}

template <int LEVEL>
__global__ __launch_bounds__(TILE_THREADS, 8)
void k_probe_slab(SlabArgs kernarg_block, const uint32_t *__restrict__ u_tile_first, const uint8_t *__restrict__ u_order,
                  const int32_t *__restrict__ u_tid, const uint32_t *__restrict__ u_tile_sbase, const TileWin *__restrict__ u_tw)
{
    constexpr int DIR_BYTES = FAST_DIR_BYTES;
    __shared__ __attribute__((aligned(16))) uint16_t s_W[SLAB_ROWS * TILE_THREADS];
    __shared__ __attribute__((aligned(16))) v4i_t s_ent[2 * SLAB_KEY_CAP];
    __shared__ __attribute__((aligned(16))) uint8_t s_dir[3 * DIR_BYTES];
    __shared__ __attribute__((aligned(16))) TileWin s_tw;
    (void)kernarg_block;
    const SlabArgsK sa = slab_args();
    const FusedArgsK a = fused_args();
    const uint32_t t = xcd_tile(blockIdx.x, gridDim.x);
    if (t >= sa->n_tiles) return;
    const uint32_t r0 = u_tile_first[t], n_act = u_tile_first[t + 1u] - r0;
    const int32_t tid0 = n_act ? u_tid[r0] : 0;
    const uint32_t sbase = u_tile_sbase[t];
    v4i_t *const s_ent0 = s_ent, *const s_ent1 = s_ent + SLAB_KEY_CAP;
    uint8_t *const s_dir0 = s_dir, *const s_dir1 = s_dir + DIR_BYTES, *const s_rdir = s_dir + 2 * DIR_BYTES;
    // ---- one round trip behind the descriptor (scalar loads: uniform address): the tile's dictionary slices, its window
    //      (k_walk_slab), the slot's read and the first four rows of its column -- all asked for before anything is looked at
    const TileDesc d = u_tw[t].d;
    if (t == 0u && threadIdx.x == 0 && sa->wide_cnt) {          // (k_walk_slab is done: its count of wide tiles moves on, the counter is cleared for the next run)
        sa->wide_cnt[1] = min(sa->wide_cnt[0], sa->wide_cap); sa->wide_cnt[0] = 0u;
    }
    if (d.flags & TD_WIDE) return;                               // k_probe_slab_wide takes the tile
    // the last row any read of this wave has (k_walk_slab): rows behind it are not asked for
    const uint32_t row_max = max((u_tw[t].pad[0] >> (8u * (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)))) & 0xffu, 1u) - 1u;
    const FusedDict dv = fused_load_dict(a, d);
    int4 twv = make_int4(0, 0, 0, 0);
    if ((int)threadIdx.x < SLAB_TW_VECS && tw_vec_used((int)threadIdx.x, (d.flags & TD_FAST) ? d.n_win : 0u)) twv = reinterpret_cast<const int4 *>(u_tw + t)[threadIdx.x];
    const bool active = threadIdx.x < n_act;
    const uint32_t at = r0 + (active ? threadIdx.x : 0u);
    uint32_t pre = 0u, r = r0;
    bool rev_in = false;
    const int32_t *const xs = a->f.ex_start; const uint16_t *const xl = a->f.ex_len;
    const uint32_t off = sbase + threadIdx.x;
    SlabRows q;
#pragma unroll
    for (int i = 0; i < SLAB_AHEAD; ++i) { q.s[i] = 0; q.e[i] = 0; }
    if (active) {
        pre = ld32(sa->pre, at); r = r0 + (uint32_t)ld32(u_order, at); rev_in = ld32(sa->s_rev, at) != 0;
#pragma unroll
        for (int i = 0; i < SLAB_AHEAD; ++i) {      // (an outlier's slab column holds nothing: read, not used)
            const uint32_t ix = off + min((uint32_t)i, row_max) * SLAB_STRIDE;
            q.s[i] = ld32(xs, ix); q.e[i] = (int)ld32(xl, ix);           // (the length: turned into the end below)
        }
    }
    if (threadIdx.x == 0 && t == 0u) *sa->ovf_cursor = 0ull;        // (k_walk_slab is done with the outlier area)
    const uint32_t n = pre >> 8;
    const bool outlier = (pre & I_PRE_DIRECT) != 0u;
    const int32_t tid = tid0;                                       // (sorted input: a tile is of one chromosome)
#pragma unroll
    for (int i = 0; i < SLAB_AHEAD; ++i) q.e[i] = q.s[i] + q.e[i] - 1;
    ReadEnds re{q.s[0], q.e[0], 0, 0};
    if (active && !outlier) { re.sl = ld32(xs, off + (n - 1u) * SLAB_STRIDE); re.el = re.sl + (int)ld32(xl, off + (n - 1u) * SLAB_STRIDE) - 1; }
    // ---- stage window and dictionary slices, re-based to the tile's window
    if ((int)threadIdx.x < SLAB_TW_VECS) reinterpret_cast<int4 *>(&s_tw)[threadIdx.x] = twv;
    const SlabLds S{s_W, s_ent0, s_ent1, s_dir0, s_dir1, s_rdir};
    // (the window's transcript numbers: straight from the loaded vectors' home, they are not in LDS yet)
    const int my_wide = slab_stage_dict(d, dv, reinterpret_cast<const int *>(u_tw[t].win), S);
    const int any_wide = __syncthreads_or(my_wide);
    slab_classify<LEVEL>(a, d, S, s_tw.hk, s_tw.hx, s_tw.win, s_tw.mask, active, pre, r, rev_in, tid, off, q, re, any_wide);
}

__global__ __launch_bounds__(TILE_THREADS)
void k_exon_counts(int64_t n_reads, const uint32_t *__restrict__ info, uint32_t *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * TILE_THREADS + threadIdx.x;
    if (r < n_reads) out[r] = info[r] >> 8;
}

// Slabs -> read order (l2r_download): one thread per read, dest[r] = running sum of the exon counts in read order.
__global__ __launch_bounds__(TILE_THREADS)
void k_linearize_slab(const uint32_t *__restrict__ tile_first, const uint32_t *__restrict__ tile_sbase, const uint8_t *__restrict__ order,
                      const uint32_t *__restrict__ pre, const uint32_t *__restrict__ ex_off, const uint32_t *__restrict__ info, const uint32_t *__restrict__ dest,
                      const int32_t *__restrict__ xs, const int32_t *__restrict__ xe, const uint8_t *__restrict__ xf,
                      int32_t *__restrict__ os, int32_t *__restrict__ oe, uint8_t *__restrict__ of, const uint16_t *__restrict__ xl)
{
    // one workgroup per tile, one thread per slot: a slab read sits at its tile's slab + its slot, a densely stored one at ex_off
    const uint32_t t = blockIdx.x, r0 = tile_first[t], n_act = tile_first[t + 1u] - r0;
    if (threadIdx.x >= n_act) return;
    const uint32_t at = r0 + threadIdx.x, r = r0 + order[at];
    const bool dense = (pre[at] & I_PRE_DIRECT) != 0u;
    uint32_t off = dense ? ex_off[r] & ~EXOFF_DENSE : tile_sbase[t] + threadIdx.x;
    const uint32_t n = info[r] >> 8, to = dest[r];
    const uint32_t st = dense ? 1u : SLAB_STRIDE;
    for (uint32_t k = 0; k < n; ++k) { os[to + k] = xs[off + k * st]; oe[to + k] = ex_end_at(xs, xe, xl, off + k * st, st); of[to + k] = xf[off + k * st]; }
}

}  // namespace l2r
