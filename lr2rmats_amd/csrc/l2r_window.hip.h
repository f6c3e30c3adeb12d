// l2r_window.hip.h -- what the one-walk (slab) pipeline shares between its kernels (gfx950): the argument block, a tile's
// WINDOW record (the annotation transcripts its reads can overlap + the slices of the site dictionaries it needs), the wave that
// makes it (make_descriptor), and the loads of a tile's dictionary slices.  The classic pipeline (l2r_kernels.hip.h) makes the
// same descriptor inside k_pass_a.
#pragma once
#include <cstddef>
#include "l2r_kernels.hip.h"

namespace l2r {

// What the slab kernels need beyond FastArgs (appended to it, so that the device functions of the classic kernel find their
// fields where they expect them).
struct PipeArgs {
    FastArgs f;
    CursorDir cd;
    const int32_t *tid_base; int32_t n_tid_dir;
    uint32_t *tile_total;                    // out: exons per tile (k_walk_slab), scanned in place into the tiles' first output slots
};
typedef const __attribute__((address_space(4))) PipeArgs *PipeArgsK;
__device__ __forceinline__ PipeArgsK pipe_args()
{
    PipeArgsK q = (PipeArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}

constexpr int PIPE_KEY_CAP = 200;                        // dictionary entries a descriptor may ask for per dictionary (default)

// A tile's window and descriptor (built in LDS by one wave of k_walk_slab, handed to k_probe_slab through HBM).
struct TileWin {
    int4 hk[WIN_TX];             // {start, end, n, flags | rev << 8} on the tile's chromosome
    int4 hx[WIN_TX];             // {s0, e0, sl, el}
    int win[WIN_TX];             // window member -> annotation index
    TileDesc d;
    uint32_t mask[2];            // members with one exon / without TX_COMPACT
    uint32_t pad[2];             // slab pipeline, pad[0]: byte w = largest exon count among the reads of wave w (k_walk_slab)
};
// A window of up to 63 members (WIDE_MEMBERS; l2r_wide.hip.h: tiles of loci with many isoforms); same fields, 64-bit member masks
constexpr int WIDE_TX = 64;
constexpr int WIDE_MEMBERS = 63;                          // members a 64-bit window may hold: 63 = "no member" in the 6-bit indices of a work word
struct TileWin64 {
    int4 hk[WIDE_TX];
    int4 hx[WIDE_TX];
    int win[WIDE_TX];
    TileDesc d;
    unsigned long long mask[2];
};

// A predicate's votes inside a GROUP of G consecutive lanes of the wave (G = 64: the wave), bit i = lane i of the group.
// gshift = first lane of the group.  Lanes of other groups that have left a divergent loop simply do not vote.
template <int G>
__device__ __forceinline__ unsigned long long group_ballot(bool p, int gshift)
{
    const unsigned long long b = __ballot(p);
    if (G == WAVE) return b;
    return (b >> gshift) & ((1ull << (G & 63)) - 1ull);
}

// The tile's dictionary slices and transcript window from its span [tlo, thi] on chromosome tid0 -- the descriptor the probe
// kernels start from -- made by ONE GROUP OF G LANES (k_describe_scan: four tiles per wave, 16 lanes each; everything here is
// group-uniform but `lane`, the lane's number inside its group).  The record is written straight into its home in HBM (Wg; Wg64 for
// a window of 33 .. 63 members, the tile then is flagged TD_WIDE instead of TD_FAST and Wg only carries the descriptor); only the
// window's transcript numbers pass through LDS (win_s: 64 words per group), where the lanes that found them and the lanes that
// fetch their headers meet.  Returns the descriptor's flags.
// jl = the cursor value of (tid0, tlo); tb, nb = the chromosome's first bucket and bucket count (k_walk_slab has looked them up).
template <int G>
__device__ __forceinline__ uint32_t make_descriptor(PipeArgsK a, int lane, int gshift, int32_t tid0, int32_t tlo, int32_t thi, TileWin *Wg, TileWin64 *Wg64,
                                                    int *win_s, uint32_t key_cap, int jl, int tb, int nb, uint32_t rows_word)
{
    const uint32_t win_cap = Wg64 ? (uint32_t)WIDE_MEMBERS : (uint32_t)WIN_TX;
    const TxHdr *const hdr = a->f.hdr;
    const int32_t n_tx = a->f.p.n_tx;
    // -d > 0: the mask kernels probe within the tolerance (probe_near), up to DIS_MASK_MAX; the staged buckets then reach that much
    // further on both sides (exon / junction flags look at annotation sites up to `dis` outside the reads' span)
    const int dis = a->f.p.ss_dis;
    int lo = INT32_MAX, hi = -1;
    if (nb > 0) { lo = min(max(tlo - max(dis, 0), 0) >> SITE_SHIFT, nb - 1); hi = min(max(thi + max(dis, 0), 0) >> SITE_SHIFT, nb - 1); }
    TileDesc d;
    d.tid = tid0; d.b_off = 0; d.nb = 0; d.b0 = 0; d.nbk = 0;
    d.st_r0 = d.st_nk = d.en_r0 = d.en_nk = 0u; d.n_win = 0u;
    bool fast = dis >= 0 && dis <= DIS_MASK_MAX && !(a->f.p.ablate & 1);
    uint32_t why = fast ? 0u : 7u;
    uint32_t sd_r0 = 0u, sd_r1 = 0u, ed_r0 = 0u, ed_r1 = 0u;
    const bool sliced = fast && hi >= 0 && hi - lo + 1 <= DIR_CAP;
    if (fast && hi >= 0 && !sliced) { fast = false; why = 2u; }
    if (sliced) {
        sd_r0 = ld32(a->f.st.rdir, (uint32_t)(tb + lo)); sd_r1 = ld32(a->f.st.dir, (uint32_t)(tb + hi + 1));
        ed_r0 = ld32(a->f.en.dir, (uint32_t)(tb + lo)); ed_r1 = ld32(a->f.en.dir, (uint32_t)(tb + hi + 1));
    }
    // sorted input: the smallest cursor value of the tile is the one of its first read, and no read needs
    // its own: a member below a read's cursor value lies entirely before that read, which visit_window sees by itself
    d.j_lo = jl;
    bool contig = true;
    uint32_t n_win = 0;
    if (fast) {
        int first = -1, last = -1;
        for (int base = jl, trip = 0; base < n_tx; ++trip) {
            const int j = base + lane;
            bool ov = false, aft = false;
            if (j < n_tx) {
                const int4 h0 = *reinterpret_cast<const int4 *>(hdr + j);                 // {tid, start, end, .}
                aft = tid0 < h0.x || (tid0 == h0.x && thi <= h0.y);                       // comp_trans <= (Q5)
                const bool bef = h0.x < tid0 || (h0.x == tid0 && h0.z <= tlo && h0.y < tlo);
                ov = !aft && !bef;
            }
            const unsigned long long ma = group_ballot<G>(aft, gshift);
            const int stop = ma ? __ffsll((long long)ma) - 1 : G;
            const unsigned long long mo = group_ballot<G>(ov, gshift) & (stop < 64 ? (1ull << stop) - 1ull : ~0ull);
            if ((mo >> lane) & 1ull) {
                const uint32_t rank = n_win + (uint32_t)__popcll(mo & ((1ull << lane) - 1ull));
                if (rank < win_cap) win_s[rank] = j;
            }
            if (mo) {
                if (first < 0) first = base + __ffsll((long long)mo) - 1;
                last = base + 63 - __clzll((long long)mo);
            }
            n_win += (uint32_t)__popcll(mo);
            if (ma) break;
            base += G;
            if (n_win > win_cap || trip == WIN_SCAN_TRIPS * (WAVE / G) - 1) { fast = false; why = n_win > win_cap ? 4u : 5u; break; }
        }
        if (fast && n_win > win_cap) { fast = false; why = 4u; }
        if (fast) {
            d.n_win = n_win;
            if (n_win) { d.j_lo = first; contig = (uint32_t)(last - first + 1) == n_win; }
        }
    }
    if (sliced) {
        d.b_off = -lo; d.nb = nb; d.b0 = tb + lo; d.nbk = hi - lo + 1;
        d.st_r0 = sd_r0; d.st_nk = sd_r1 - sd_r0;
        d.en_r0 = ed_r0; d.en_nk = ed_r1 - ed_r0;
        if (fast && (d.st_nk > key_cap || d.en_nk > key_cap)) { fast = false; why = 3u; }
    }
    const bool wide = fast && d.n_win > (uint32_t)WIN_TX;      // (only with Wg64)
    d.flags = (fast ? (wide ? TD_WIDE : TD_FAST) : 0u) | (contig ? TD_CONTIG : 0u) | (why << 8);
    // the members' headers, G members per round (the group's own LDS writes above are visible to it: same wave, in order), straight
    // into the record; masks of the members with one exon / without TX_COMPACT
    const int w_n = fast ? (int)d.n_win : 0;
    int4 *const hk_out = wide ? Wg64->hk : Wg->hk, *const hx_out = wide ? Wg64->hx : Wg->hx;
    int *const win_out = wide ? Wg64->win : Wg->win;
    unsigned long long b1 = 0ull, b2 = 0ull;
    for (int m0 = 0; m0 < w_n; m0 += G) {
        const int m = m0 + lane;
        bool single = false, loose = false;
        if (m < w_n) {
            const int j = win_s[m];
            const int4 *hp = reinterpret_cast<const int4 *>(hdr + j);
            const int4 h0 = hp[0], h1 = hp[1], h2 = hp[2];
            int st = h0.y, en = h0.z;
            if (h0.x < tid0) { st = INT32_MIN; en = INT32_MIN; }            // another chromosome: before / after every read
            else if (h0.x > tid0) { st = INT32_MAX; en = INT32_MAX; }
            hk_out[m] = make_int4(st, en, h1.x, (h1.z & 0xff) | (h1.y << 8));
            hx_out[m] = h2;
            win_out[m] = j;
            single = h1.x == 1; loose = !((h1.z & 0xff) & TX_COMPACT);
        }
        b1 |= group_ballot<G>(single, gshift) << m0; b2 |= group_ballot<G>(loose, gshift) << m0;
    }
    if (lane == 0) {
        // descriptor + masks + {rows per wave of the probe kernels, the tile's last base}: the last 64 bytes of the record
        int4 *const tail = reinterpret_cast<int4 *>(&Wg->d);
        const int4 v0 = make_int4(d.j_lo, d.tid, d.b_off, d.nb), v1 = make_int4(d.b0, d.nbk, (int)d.st_r0, (int)d.st_nk),
                   v2 = make_int4((int)d.en_r0, (int)d.en_nk, (int)d.flags, (int)d.n_win);
        tail[0] = v0; tail[1] = v1; tail[2] = v2;
        tail[3] = wide ? make_int4(0, 0, (int)rows_word, thi) : make_int4((int)(uint32_t)b1, (int)(uint32_t)b2, (int)rows_word, thi);
        if (wide) {
            int4 *const t64 = reinterpret_cast<int4 *>(&Wg64->d);
            t64[0] = v0; t64[1] = v1; t64[2] = v2;
            t64[3] = make_int4((int)(uint32_t)b1, (int)(uint32_t)(b1 >> 32), (int)(uint32_t)b2, (int)(uint32_t)(b2 >> 32));
        }
    }
    return d.flags;
}
static_assert(offsetof(TileWin, d) % 16 == 0 && sizeof(TileDesc) == 48 && offsetof(TileWin, mask) == offsetof(TileWin, d) + 48 && offsetof(TileWin, pad) == offsetof(TileWin, d) + 56,
              "make_descriptor writes the tail of a TileWin as four 16-byte vectors");
static_assert(offsetof(TileWin64, d) % 16 == 0 && offsetof(TileWin64, mask) == offsetof(TileWin64, d) + 48 && sizeof(TileWin64) == offsetof(TileWin64, d) + 64, "... and of a TileWin64");

// base[idx] = v with a 32-bit byte offset (see ld32): one shift per lane instead of a 64-bit multiply-add
template <typename T>
__device__ __forceinline__ void st32(T *base, uint32_t idx, T v)
{
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + (size_t)(idx * (uint32_t)sizeof(T))) = v;
}

// a tile's dictionary entries / bucket directory words as loaded, one START and one END entry per thread
struct DictRegs { int4 xa, xb, xc, xd; uint32_t dd[3][2]; };
__device__ __forceinline__ DictRegs load_dict_slices(PipeArgsK a, const TileDesc &d)
{
    DictRegs v;
    v.xa = v.xb = v.xc = v.xd = make_int4(0, 0, 0, 0);
    const bool fast = (d.flags & (TD_FAST | TD_WIDE)) != 0;
    if (fast && threadIdx.x < d.st_nk) { const int4 *q = reinterpret_cast<const int4 *>(a->f.st.ent + d.st_r0 + threadIdx.x); v.xa = q[0]; v.xb = q[1]; }
    if (fast && threadIdx.x < d.en_nk) { const int4 *q = reinterpret_cast<const int4 *>(a->f.en.ent + d.en_r0 + threadIdx.x); v.xc = q[0]; v.xd = q[1]; }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = (int)threadIdx.x + q * TILE_THREADS;
        v.dd[0][q] = v.dd[1][q] = v.dd[2][q] = 0u;
        if (fast && d.nbk > 0 && i <= d.nbk) {
            const uint32_t b = (uint32_t)(d.b0 + i);
            v.dd[0][q] = ld32(a->f.st.dir, b); v.dd[1][q] = ld32(a->f.en.dir, b); v.dd[2][q] = ld32(a->f.st.rdir, b);
        }
    }
    return v;
}

}  // namespace l2r
