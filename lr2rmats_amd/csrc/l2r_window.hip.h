// l2r_window.hip.h -- what the one-walk (slab) pipeline shares between its kernels (gfx950): the argument block, a tile's
// WINDOW record (the annotation transcripts its reads can overlap + the slices of the site dictionaries it needs), the wave that
// makes it (make_descriptor), and the loads of a tile's dictionary slices.  The classic pipeline (l2r_kernels.hip.h) makes the
// same descriptor inside k_pass_a.
#pragma once
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

// The tile's dictionary slices and transcript window from its span [tlo, thi] on chromosome tid0 -- the descriptor
// k_pass_a leaves in HBM, made by ONE WAVE of the workgroup (everything is wave-uniform but `lane`).  The window's
// member headers go straight into LDS.
// With W64 (slab pipeline): a window of 33 .. 63 members is collected into *W64 and the tile is flagged TD_WIDE instead of
// TD_FAST (W then only carries the descriptor); up to 32 members everything is as without it.
__device__ __forceinline__ void make_descriptor(PipeArgsK a, int lane, int32_t tid0, int32_t tlo, int32_t thi, bool in_lds, TileWin *W,
                                                uint32_t key_cap = (uint32_t)PIPE_KEY_CAP, TileWin64 *W64 = nullptr, int jl_known = INT32_MIN /* the cursor value of (tid0, tlo), when the caller has looked it up already */)
{
    const uint32_t win_cap = W64 ? (uint32_t)WIDE_MEMBERS : (uint32_t)WIN_TX;
    int *const win_out = W64 ? W64->win : W->win;
    const TxHdr *const hdr = a->f.hdr;
    const int32_t n_tx = a->f.p.n_tx;
    int tb = 0, nb = 0;
    if (tid0 >= 0 && tid0 < a->n_tid_dir) { tb = a->tid_base[tid0]; nb = a->tid_base[tid0 + 1] - tb; }
    int lo = INT32_MAX, hi = -1;
    if (nb > 0) { lo = min(max(tlo, 0) >> SITE_SHIFT, nb - 1); hi = min(max(thi, 0) >> SITE_SHIFT, nb - 1); }
    TileDesc d;
    d.tid = tid0; d.b_off = 0; d.nb = 0; d.b0 = 0; d.nbk = 0;
    d.st_r0 = d.st_nk = d.en_r0 = d.en_nk = 0u; d.n_win = 0u;
    bool fast = in_lds && a->f.p.ss_dis == 0 && !(a->f.p.ablate & 1);
    uint32_t why = fast ? 0u : (!in_lds ? 1u : 7u);
    uint32_t sd_r0 = 0u, sd_r1 = 0u, ed_r0 = 0u, ed_r1 = 0u;
    const bool sliced = fast && hi >= 0 && hi - lo + 1 <= DIR_CAP;
    if (fast && hi >= 0 && !sliced) { fast = false; why = 2u; }
    if (sliced) {
        sd_r0 = a->f.st.rdir[tb + lo]; sd_r1 = a->f.st.dir[tb + hi + 1];
        ed_r0 = a->f.en.dir[tb + lo]; ed_r1 = a->f.en.dir[tb + hi + 1];
    }
    // sorted input: the smallest cursor value of the tile is the one of its first read, and no read needs
    // its own: a member below a read's cursor value lies entirely before that read, which visit_window sees by itself
    CursorDir cd;
    cd.key = a->cd.key; cd.dir = a->cd.dir; cd.kb_base = a->cd.kb_base; cd.n_tid = a->cd.n_tid; cd.n_tx = a->cd.n_tx;
    const int jl = jl_known != INT32_MIN ? jl_known : cursor_value(cd, tid0, tlo);
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
            const unsigned long long ma = __ballot(aft);
            const int stop = ma ? __ffsll((long long)ma) - 1 : WAVE;
            const unsigned long long mo = __ballot(ov) & (stop < WAVE ? (1ull << stop) - 1ull : ~0ull);
            if ((mo >> lane) & 1ull) {
                const uint32_t rank = n_win + (uint32_t)__popcll(mo & ((1ull << lane) - 1ull));
                if (rank < win_cap) win_out[rank] = j;
            }
            if (mo) {
                if (first < 0) first = base + __ffsll((long long)mo) - 1;
                last = base + 63 - __clzll((long long)mo);
            }
            n_win += (uint32_t)__popcll(mo);
            if (ma) break;
            base += WAVE;
            if (n_win > win_cap || trip == WIN_SCAN_TRIPS - 1) { fast = false; why = n_win > win_cap ? 4u : 5u; break; }
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
    const bool wide = fast && d.n_win > (uint32_t)WIN_TX;      // (only with W64)
    d.flags = (fast ? (wide ? TD_WIDE : TD_FAST) : 0u) | (contig ? TD_CONTIG : 0u) | (why << 8);
    if (wide) {
        // all 64 lanes: one member each
        bool single = false, loose = false;
        if (lane < (int)d.n_win) {
            const int j = W64->win[lane];
            const int4 *hp = reinterpret_cast<const int4 *>(hdr + j);
            const int4 h0 = hp[0], h1 = hp[1], h2 = hp[2];
            int st = h0.y, en = h0.z;
            if (h0.x < tid0) { st = INT32_MIN; en = INT32_MIN; }
            else if (h0.x > tid0) { st = INT32_MAX; en = INT32_MAX; }
            W64->hk[lane] = make_int4(st, en, h1.x, (h1.z & 0xff) | (h1.y << 8));
            W64->hx[lane] = h2;
            single = h1.x == 1; loose = !((h1.z & 0xff) & TX_COMPACT);
        }
        const unsigned long long b1 = __ballot(single), b2 = __ballot(loose);
        if (lane == 0) { W64->d = d; W64->mask[0] = b1; W64->mask[1] = b2; W->d = d; W->mask[0] = 0u; W->mask[1] = 0u; }
        return;
    }
    if (W64 && lane < WIN_TX) W->win[lane] = W64->win[lane];                // (narrow after all: the members move to the 32-member record)
    // the members' headers (the wave's own LDS writes above are visible to it: same wave, in order)
    const int w_n = fast ? (int)d.n_win : 0;
    bool single = false, loose = false;
    if (lane < w_n) {
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
    if (lane == 0) { W->d = d; W->mask[0] = (uint32_t)b1; W->mask[1] = (uint32_t)b2; }
}

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
