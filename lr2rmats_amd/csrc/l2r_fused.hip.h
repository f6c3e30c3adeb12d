// l2r_fused.hip.h -- the ONE-WALK pipeline for short-CIGAR, coordinate-sorted input (gfx950).
//
// The classic pipeline (l2r_kernels.hip.h) walks every CIGAR twice: k_pass_a needs the exon counts (LDS placement,
// output offsets) and the read ends (which slice of the dictionaries and which transcripts a tile will need) before
// k_classify_fast can start, so the CIGAR stream is read from HBM twice and ~17 VALU per op are spent twice.  Here:
//
//   k_order     per tile of up to 256 reads, from the CIGAR LENGTHS alone (cig_off, 8 bytes per read): an upper bound
//               of every read's exon count -> its slot range in the LDS exon arrays; the tile's reads by falling length
//               (the order in which the classification kernel deals them to its lanes)
//   k_fused     persistent; per tile: CIGAR words straight into registers (one lane = one read), ONE walk -> exons in
//               LDS and the read ends; one wave turns the tile's span into the dictionary slices and the transcript
//               window (what k_pass_a's descriptor was); staging, window pass, probes, verdicts exactly as
//               k_classify_fast (same device functions); the tile's exons leave through an LDS map (output slot ->
//               LDS slot), coalesced, into a chunk of the result arrays taken from an atomic cursor once the tile's
//               exon count is known.  No k_pass_a, no scan.
//
// The result arrays are therefore made of one chunk per tile (reads of a chunk in read order, chunks in the order the
// cursor handed them out); ex_off[r] is explicit.  l2r_download() puts them into read order with k_linearize.
// Unsorted input (history-dependent cursors), long CIGARs (ONT) and every read the masks cannot decide take the classic
// kernels / the generic kernel, as before.
#pragma once
#include "l2r_kernels.hip.h"

namespace l2r {

constexpr int FUSED_HEAD_VEC = 6;                          // 16-byte CIGAR vectors a lane holds: 24 ops; longer reads finish from memory
constexpr int FUSED_HEAD = 4 * FUSED_HEAD_VEC;

// upper bound of a read's exon count from its number of CIGAR ops: with min_exon >= 1 every kept exon but the first and
// the last needs an op of its own next to its cut (src/bam2gtf.c:31-78), so n <= (c + 3) / 2; otherwise n <= c + 1
__device__ __forceinline__ uint32_t exon_bound(uint32_t c, int min_exon)
{
    return min_exon >= 1 ? (c + 3u) >> 1 : c + 1u;
}

__global__ __launch_bounds__(TILE_THREADS)
void k_order(const int64_t *__restrict__ cig_off, const uint32_t *__restrict__ tile_first, DevParams p,
             uint8_t *__restrict__ order_out, uint16_t *__restrict__ lub_out, uint32_t *__restrict__ tile_ub,
             uint32_t *__restrict__ redo_count /* [3]: redo list, chunk cursor of the accepted list */, unsigned long long *__restrict__ ex_cursor,
             int32_t *__restrict__ tile_thi /* slab pipeline: the tile's largest read end (atomicMax of k_walk_slab), else null */,
             unsigned long long *__restrict__ ovf_cursor /* slab pipeline: dense area of the outliers */,
             uint32_t *__restrict__ tile_total /* slab pipeline: exon count per tile (atomicAdd of k_walk_slab), else null */,
             const int32_t *__restrict__ r_pos, const uint8_t *__restrict__ r_rev,
             uint32_t *__restrict__ s_clo, uint16_t *__restrict__ s_ncig, int32_t *__restrict__ s_pos, uint8_t *__restrict__ s_rev
             /* slab pipeline: the records' fields in SLOT order (tile start + slot), so that its kernels need no indirection; else null */)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {       // the kernels that use these run after this one
        redo_count[0] = 0u; redo_count[1] = 0u; redo_count[2] = 0u; *ex_cursor = 0ull;
        if (ovf_cursor) *ovf_cursor = 0ull;
    }
    if (tile_total && threadIdx.x == 0) tile_total[blockIdx.x] = 0u;
    if (tile_thi && threadIdx.x == 0) tile_thi[blockIdx.x] = INT32_MIN;
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_hist[WAVE];
    if (threadIdx.x < WAVE) s_hist[threadIdx.x] = 0u;
    const uint32_t r0 = tile_first[blockIdx.x], n_act = tile_first[blockIdx.x + 1] - r0;
    const bool active = threadIdx.x < n_act;
    const int64_t r = (int64_t)r0 + threadIdx.x;
    uint32_t c = 0u;
    if (active) c = (uint32_t)min((int64_t)0x7fffffff, cig_off[r + 1] - cig_off[r]);
    const uint32_t ub = active ? min(exon_bound(c, p.min_exon), 0xffffu) : 0u;
    uint32_t total;
    const uint32_t local = block_exclusive_scan(ub, s_wave, total);          // (two barriers: s_hist is clear behind them)
    if (active) lub_out[r] = (uint16_t)min(local, 0xffffu);                   // (a tile beyond the LDS capacity does not use it)
    if (threadIdx.x == 0) tile_ub[blockIdx.x] = total;
    // the tile's reads by falling length (counting sort, ties in arrival order); threads without a read sort last
    const uint32_t est = active ? max(1u, min((c + 1u) >> 1, (uint32_t)(WAVE - 1))) : 0u;
    const uint32_t bin = (uint32_t)(WAVE - 1) - est;
    const uint32_t rank = atomicAdd(&s_hist[bin], 1u);
    __syncthreads();
    if (threadIdx.x < WAVE) { const uint32_t v = s_hist[threadIdx.x]; s_hist[threadIdx.x] = wave_inclusive_scan(v) - v; }
    __syncthreads();
    const uint32_t slot = s_hist[bin] + rank;
    if (active) {
        order_out[(int64_t)r0 + slot] = (uint8_t)threadIdx.x;
        if (s_clo) {
            const int64_t at = (int64_t)r0 + slot;
            s_clo[at] = (uint32_t)cig_off[r]; s_ncig[at] = (uint16_t)min(c, 0xffffu); s_pos[at] = r_pos[r]; s_rev[at] = r_rev[r];
        }
    }
}

// What the fused kernel needs beyond FastArgs (appended to it, so that the device functions of the classic kernel
// find their fields where they expect them).
struct FusedArgs {
    FastArgs f;
    CursorDir cd;
    const int32_t *tid_base; int32_t n_tid_dir;
    const uint16_t *lub;                     // k_order: first LDS slot of every read inside its tile
    const uint32_t *tile_ub;                 // k_order: slots the tile needs
    uint32_t *tile_start, *tile_total;       // out: the tile's chunk of the exon arrays
    unsigned long long *ex_cursor;           // next free exon slot
};
typedef const __attribute__((address_space(4))) FusedArgs *FusedArgsK;
__device__ __forceinline__ FusedArgsK fused_args()
{
    FusedArgsK q = (FusedArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}

constexpr int FUSED_EXON_CAP = 2880;                     // exon slots of a tile in LDS (10 bytes each): four workgroups per CU
constexpr int FUSED_KEY_CAP = 200;                       // dictionary entries staged per dictionary and tile
constexpr int FUSED_W_WORDS = FUSED_EXON_CAP / 2;
constexpr int FUSED_TAIL_WORDS = FUSED_W_WORDS + 2 * FUSED_KEY_CAP * 4 + 3 * FAST_DIR_BYTES / 4;
constexpr int FUSED_ALL_WORDS = 2 * FUSED_EXON_CAP + FUSED_TAIL_WORDS;
static_assert(2 * FUSED_KEY_CAP * 16 >= FUSED_EXON_CAP * 2, "the output map (16 bits per exon) lives in the dead dictionary slices");

// A tile's window and descriptor in LDS.  Two of them: while tile t is classified, one wave prepares tile t + 1's.
struct TileWin {
    int4 hk[WIN_TX];             // {start, end, n, flags | rev << 8} on the tile's chromosome
    int4 hx[WIN_TX];             // {s0, e0, sl, el}
    int win[WIN_TX];             // window member -> annotation index
    TileDesc d;
    uint32_t mask[2];            // members with one exon / without TX_COMPACT
    uint32_t pad[2];             // slab pipeline, pad[0]: byte w = largest exon count among the reads of wave w (k_walk_slab)
};
// A window of up to 64 members (l2r_wide.hip.h: tiles of loci with many isoforms); same fields, 64-bit member masks
constexpr int WIDE_TX = 64;
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
// With W64 (slab pipeline): a window of 33 .. 64 members is collected into *W64 and the tile is flagged TD_WIDE instead of
// TD_FAST (W then only carries the descriptor); up to 32 members everything is as without it.
__device__ __forceinline__ void make_descriptor(FusedArgsK a, int lane, int32_t tid0, int32_t tlo, int32_t thi, bool in_lds, TileWin *W,
                                                uint32_t key_cap = (uint32_t)FUSED_KEY_CAP, TileWin64 *W64 = nullptr)
{
    const uint32_t win_cap = W64 ? (uint32_t)WIDE_TX : (uint32_t)WIN_TX;
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
    const int jl = cursor_value(cd, tid0, tlo);
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
    const bool wide = fast && d.n_win > (uint32_t)WIN_TX;                  // (only with W64)
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

// One CIGAR op of the register walk (src/bam2gtf.c:31-78, the rules of walk_step) without a branch: the exon in progress
// is written to its LDS slot {start, end} at EVERY step, and a kept cut finalises the slot by moving on to the next one.
// t3 / t2 = the smallest words "N, length min_intron" / "D, length max_delet + 1": op and length compare as one number.
struct WalkRegs { int start, end; uint32_t n; bool first; };          // first: no exon kept yet (the chain through n stays one add per op)
__device__ __forceinline__ void walk_word(WalkRegs &w, uint32_t c, uint32_t t3, uint32_t t2, int min_len_m1, int2 *slots, uint32_t room_m1)
{
    const uint32_t op = c & 0xfu;
    const int len = (int)(c >> 4);
    const bool cut = ((op == 3u) & (c >= t3)) | ((op == 2u) & (c >= t2));
    const bool keep = cut & (w.first | (w.end - w.start >= min_len_m1));
    slots[min(w.n, room_m1)] = make_int2(w.start, w.end);
    w.first = w.first & !keep;
    w.n += keep ? 1u : 0u;
    w.start = cut ? w.end + len + 1 : w.start;
    w.end += len & __builtin_amdgcn_sbfe(0x18d, op, 1u);         // ops 0 2 3 7 8 advance the reference
}

// map_exons (l2r_kernels.hip.h) over exon PAIRS {start, end} in LDS: one 8-byte read per exon
__device__ __forceinline__ SiteMasks map_exons_se(const TileLds &L, const int2 *SE, const TileDesc &d, bool mapping, uint32_t local, uint32_t n, uint32_t vpre)
{
    SiteMasks m{0xffffffffu, 0u, 0u, 0u};
    uint16_t *W = L.W + local;
    int s = 0, e = 0;
    if (mapping) { const int2 x = SE[local]; s = x.x; e = x.y; }
    const uint32_t none = (uint32_t)d.nbk + 1u;         // a bucket behind the staged ones: the staging leaves it empty
    const int k_max = wave_max(mapping ? (int)n : 0);
    for (int k = 0; k < k_max; ++k) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        const uint32_t is = min((uint32_t)((s >> SITE_SHIFT) + d.b_off), none);
        const uint32_t ie = junc ? min((uint32_t)((e >> SITE_SHIFT) + d.b_off), none) : none;
        const int2 nx = SE[local + (uint32_t)k + 1u];         // (a slot behind the read's last exon: read, not used)
        const int s2 = nx.x, e2 = nx.y;
        const uint32_t ls = L.dir0[is], hs = L.dir0[is + 1u], le = L.dir1[ie], he = L.dir1[ie + 1u];
        const v4i_t qs0 = lds_entry(L.ent0, ls);
        const v4i_t qe0 = lds_entry(L.ent1, le), qe1 = lds_entry(L.ent1, le + 1u);
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
        if (live) W[k] = (uint16_t)word;
        s = s2; e = e2;
    }
    return m;
}

// per-thread inputs of a tile: the read a thread is dealt, and the head of its CIGAR (words behind the read's last op
// are replaced by "I, length 0", which changes nothing in the walk)
struct FusedRead {
    int32_t src;                 // the thread's read of the tile (k_order's order), -1: none
    uint32_t c_lo, n_cig, lub;
    int32_t pos, tid;
    uint32_t rev;
    uint32_t cg[FUSED_HEAD];
};
struct FusedTile { uint32_t r0, n_act; int32_t tid0, pos0; bool in_lds; };   // (wave-uniform)

__device__ __forceinline__ FusedTile fused_tile(const uint32_t *__restrict__ tile_first, const int32_t *__restrict__ r_tid,
                                                const int32_t *__restrict__ r_pos, const uint32_t *__restrict__ tile_ub, uint32_t t)
{
    FusedTile T;
    T.r0 = tile_first[t]; T.n_act = tile_first[t + 1u] - T.r0;
    T.tid0 = T.n_act ? r_tid[T.r0] : 0; T.pos0 = T.n_act ? r_pos[T.r0] : 0;
    T.in_lds = tile_ub[t] <= (uint32_t)FUSED_EXON_CAP;
    return T;
}
// thread -> slot of k_order's order, rotated by one wave per tile (see load_uniforms of the classic kernel)
__device__ __forceinline__ int32_t fused_src(const uint8_t *__restrict__ order, const FusedTile &T, uint32_t t)
{
    const uint32_t slot = (threadIdx.x - (((t + (t >> 10)) & 3u) << 6)) & (uint32_t)(TILE_THREADS - 1);
    return slot < T.n_act ? (int32_t)ld32(order, T.r0 + slot) : -1;
}
// The reads of a tile arrive in three steps, each issued a phase before its result is needed (order byte -> record
// fields -> CIGAR words are dependent loads): fields, words, then masking of the words behind the read's last op.
__device__ __forceinline__ void fused_load_fields(FusedArgsK a, const FusedTile &T, int32_t src, FusedRead &v)
{
    v.src = src; v.c_lo = 0u; v.n_cig = 0u; v.lub = 0u; v.pos = 0; v.tid = T.tid0; v.rev = 0u;
    if (src >= 0) {
        const uint32_t r = T.r0 + (uint32_t)src;
        const int64_t *const p_off = a->f.cig_off;
        v.c_lo = (uint32_t)ld32(p_off, r); v.n_cig = (uint32_t)ld32(p_off, r + 1u);        // (n_cig holds c_hi until fused_load_words)
        v.pos = ld32(a->f.r_pos, r); v.tid = ld32(a->f.r_tid, r); v.rev = ld32(a->f.r_rev, r);
        v.lub = ld32(a->lub, r);
    }
}
__device__ __forceinline__ void fused_load_words(FusedArgsK a, FusedRead &v)
{
    v.n_cig -= v.c_lo;
#pragma unroll
    for (int i = 0; i < FUSED_HEAD; ++i) v.cg[i] = 1u;
    if (v.src >= 0) {
        const uint32_t *const words = a->f.cig + v.c_lo;
#pragma unroll
        for (int q = 0; q < FUSED_HEAD_VEC; ++q)
            if ((uint32_t)(4 * q) < v.n_cig) {
                const v4i_a4 x = *reinterpret_cast<const v4i_a4 *>(words + 4 * q);
                v.cg[4 * q] = (uint32_t)x.x; v.cg[4 * q + 1] = (uint32_t)x.y; v.cg[4 * q + 2] = (uint32_t)x.z; v.cg[4 * q + 3] = (uint32_t)x.w;
            }
    }
}
__device__ __forceinline__ void fused_mask_words(FusedRead &v)
{
#pragma unroll
    for (int i = 0; i < FUSED_HEAD; ++i) v.cg[i] = (uint32_t)i < v.n_cig ? v.cg[i] : 1u;
}
__device__ __forceinline__ FusedRead fused_load_read(FusedArgsK a, const FusedTile &T, int32_t src)
{
    FusedRead v;
    fused_load_fields(a, T, src, v);
    fused_load_words(a, v);
    fused_mask_words(v);
    return v;
}
// upper bound of the read's end: every op counted as if it advanced the reference (exact for M / N / D / = / X CIGARs)
__device__ __forceinline__ int32_t fused_end_bound(FusedArgsK a, const FusedRead &v)
{
    uint32_t sum = 0u;
#pragma unroll
    for (int i = 0; i < FUSED_HEAD; i += 2) sum += (v.cg[i] >> 4) + (v.cg[i + 1] >> 4);
    if (v.n_cig > (uint32_t)FUSED_HEAD) {
        const uint32_t *const words = a->f.cig + v.c_lo;
        for (uint32_t i = FUSED_HEAD; i < v.n_cig; ++i) sum += words[i] >> 4;
    }
    return (int32_t)min((uint32_t)v.pos + sum, 0x7fffffffu);
}
// a tile's dictionary entries / bucket directory words as loaded, one START and one END entry per thread
struct FusedDict { int4 xa, xb, xc, xd; uint32_t dd[3][2]; };
__device__ __forceinline__ FusedDict fused_load_dict(FusedArgsK a, const TileDesc &d)
{
    FusedDict v;
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

// PERSISTENT and software pipelined over the workgroup's tiles t, t + grid, ...:
//     reads of tile t+G (fields, CIGAR head)        issued after tile t's walk (the order byte a tile earlier), in registers
//     span of tile t+G (upper bound of the ends)    from those registers after tile t's window pass, reduced over the workgroup
//     descriptor + window of tile t+G               made by the LAST wave (it holds the tile's shortest reads) while the others probe
//     dictionary entries / directory of tile t+G    issued at the top of tile t+G, consumed after its walk
template <int LEVEL>
__global__ __launch_bounds__(TILE_THREADS, 4)
void k_fused(FusedArgs kernarg_block /* read through fused_args() / fast_args() */, int64_t n_tiles, const uint32_t *__restrict__ u_tile_first,
             const uint8_t *__restrict__ u_order, const int32_t *__restrict__ u_tid, const int32_t *__restrict__ u_pos,
             const uint32_t *__restrict__ u_tile_ub)
{
    // LDS image of a tile:  S[cap] | E[cap] | W (16 bits per exon) | START entries | END entries | dir bytes x 3
    // (the output map, 16 bits per exon, takes the place of the entries once the verdicts are in)
    constexpr int DIR_BYTES = FAST_DIR_BYTES;
    __shared__ __attribute__((aligned(16))) uint32_t s_all[FUSED_ALL_WORDS];
    __shared__ __attribute__((aligned(16))) TileWin s_tw[2];
    __shared__ int s_wide;
    __shared__ int s_wmax[4];
    __shared__ __attribute__((aligned(16))) uint32_t s_nx[TILE_THREADS];        // per read, READ order: exon count, then (in place) exact exon offset inside the tile
    __shared__ uint32_t s_base[2];
    (void)kernarg_block;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    int2 *const s_SE = reinterpret_cast<int2 *>(s_all);                 // exon pairs {start, end}
    uint16_t *const s_W = reinterpret_cast<uint16_t *>(s_all + 2 * FUSED_EXON_CAP);
    v4i_t *const s_ent0 = reinterpret_cast<v4i_t *>(s_all + 2 * FUSED_EXON_CAP + FUSED_W_WORDS), *const s_ent1 = s_ent0 + FUSED_KEY_CAP;
    uint8_t *const s_dir0 = reinterpret_cast<uint8_t *>(s_all + 2 * FUSED_EXON_CAP + FUSED_W_WORDS + 2 * FUSED_KEY_CAP * 4);
    uint8_t *const s_dir1 = s_dir0 + DIR_BYTES, *const s_rdir = s_dir1 + DIR_BYTES;
    uint16_t *const s_map = reinterpret_cast<uint16_t *>(s_ent0);

    // diagnostics (L2R_STAMPS=1): cycles of wave 0 per phase [0 walk, 1 staging, 2 window pass + next span, 3 probes + verdicts,
    // 4 offsets + map + write-out], barrier waits of wave 0 [5] and of the last wave [6], the last wave's descriptor work [7]
    const bool stamping = fused_args()->f.stamps != nullptr;
    unsigned long long t_prev = stamping ? __builtin_readcyclecounter() : 0ull;
#define F_STAMP(i) do { if (stamping && threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
        atomicAdd(&fused_args()->f.stamps[(blockIdx.x & 1023u) * 8u + (i)], t_ - t_prev); t_prev = t_; } } while (0)
#define F_BARRIER() do { const unsigned long long b0_ = stamping ? __builtin_readcyclecounter() : 0ull; __syncthreads(); \
        if (stamping && lane == 0 && (wv == 0 || wv == TILE_THREADS / WAVE - 1)) { const unsigned long long b1_ = __builtin_readcyclecounter(); \
            atomicAdd(&fused_args()->f.stamps[(blockIdx.x & 1023u) * 8u + (wv == 0 ? 5u : 6u)], b1_ - b0_); t_prev += b1_ - b0_; } } while (0)
    uint32_t t = blockIdx.x;
    if ((int64_t)t >= n_tiles) return;
    // ---- prologue: the first tile's reads, span, descriptor (nothing to hide behind yet)
    FusedTile T = fused_tile(u_tile_first, u_tid, u_pos, u_tile_ub, t);
    FusedRead v = fused_load_read(fused_args(), T, fused_src(u_order, T, t));
    {
        const int m = wave_max((v.src >= 0 && v.tid == T.tid0) ? fused_end_bound(fused_args(), v) : INT32_MIN);
        if (lane == 0) s_wmax[wv] = m;
    }
    __syncthreads();
    if (wv == TILE_THREADS / WAVE - 1)
        make_descriptor(fused_args(), lane, T.tid0, T.pos0 + 1, max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])), T.in_lds, &s_tw[0]);
    __syncthreads();
    int cur = 0;                                   // which TileWin holds the current tile
    // ... and of the tile after it: uniforms, order byte, record fields (its CIGAR words follow inside the loop)
    FusedTile Tn = T;
    FusedRead vn = v;
    int32_t src_n1 = -1;                           // the thread's read of the next tile (its order byte)
    {
        const uint32_t t1 = t + gridDim.x;
        if ((int64_t)t1 < n_tiles) { Tn = fused_tile(u_tile_first, u_tid, u_pos, u_tile_ub, t1); src_n1 = fused_src(u_order, Tn, t1); }
    }

    for (; (int64_t)t < n_tiles; t += gridDim.x) {
        const FusedArgsK a = fused_args();
        const uint32_t t_next = t + gridDim.x, t_next2 = t_next + gridDim.x;
        const bool has_next = (int64_t)t_next < n_tiles, has_next2 = (int64_t)t_next2 < n_tiles;
        // uniforms of the tile after the next one (scalar loads; their order byte is fetched after this tile's walk)
        FusedTile Tn2 = Tn;
        if (has_next2) Tn2 = fused_tile(u_tile_first, u_tid, u_pos, u_tile_ub, t_next2);
        // record fields of the next tile's reads (their CIGAR words follow after this tile's walk)
        if (has_next) fused_load_fields(a, Tn, src_n1, vn);
        TileWin *const tw = &s_tw[cur], *const tw_next = &s_tw[cur ^ 1];
        const TileDesc d = tw->d;
        const bool fast = (d.flags & TD_FAST) != 0, in_lds = T.in_lds;
        const int w_n = fast ? (int)d.n_win : 0;
        // the tile's dictionary entries and directory words start their trip; they are staged after the walk
        const FusedDict dv = fused_load_dict(a, d);
        const bool active = v.src >= 0;
        const uint32_t r = T.r0 + (uint32_t)max(v.src, 0);
        const uint32_t lub = v.lub, n_cig = v.n_cig, c_lo = v.c_lo;
        const int32_t pos = v.pos, tid = v.tid;
        const bool rev_in = v.rev != 0u;
        const int32_t src = v.src;
        if (threadIdx.x == 0) s_wide = 0;
        // ---- phase 1: the ONE walk, CIGAR words out of registers; a wave stops where its longest CIGAR ends
        DevParams p;
        p.min_exon = a->f.p.min_exon; p.min_intron = a->f.p.min_intron; p.max_delet = a->f.p.max_delet;
        const uint32_t room = exon_bound(n_cig, p.min_exon);
        // the straight-line walk wants every kept inner exon non-empty by construction (min_exon >= 1, thresholds that fit a CIGAR word);
        // anything else takes the generic rules read by read (`over`)
        const bool plain = p.min_exon >= 1 && p.min_intron >= 0 && p.min_intron < (1 << 28) && p.max_delet >= -1 && p.max_delet < (1 << 28) - 1;
        const uint32_t t3 = ((uint32_t)p.min_intron << 4) | 3u, t2 = ((uint32_t)(p.max_delet + 1) << 4) | 2u;
        const int c_max = wave_max(active ? (int)min(n_cig, (uint32_t)FUSED_HEAD) : 0);
        uint32_t n = 0u;
        ReadEnds re{0, 0, 0, 0};
        bool sane = true, over = false;
        if (active && in_lds && plain && n_cig <= (uint32_t)FUSED_HEAD) {
            WalkRegs w{pos + 1, pos, 0u, true};
            int2 *const slots = s_SE + lub;
            const uint32_t room_m1 = room - 1u;
#pragma unroll
            for (int q = 0; q < FUSED_HEAD_VEC; ++q) {
                if (4 * q < c_max) {         // (wave-uniform)
                    walk_word(w, v.cg[4 * q], t3, t2, p.min_exon - 1, slots, room_m1); walk_word(w, v.cg[4 * q + 1], t3, t2, p.min_exon - 1, slots, room_m1);
                    walk_word(w, v.cg[4 * q + 2], t3, t2, p.min_exon - 1, slots, room_m1); walk_word(w, v.cg[4 * q + 3], t3, t2, p.min_exon - 1, slots, room_m1);
                }
            }
            over = w.n >= room;                                    // (cannot happen: exon_bound; the slot index was clamped)
            slots[min(w.n, room_m1)] = make_int2(w.start, w.end);
            n = w.n + 1u;
            const int2 x0 = slots[0];
            re.s0 = x0.x; re.e0 = x0.y; re.sl = w.start; re.el = w.end;
            // kept inner exons are at least min_exon >= 1 long; the first and the last one are kept whatever their length
            sane = re.s0 <= re.e0 && re.sl <= re.el;
        } else if (active) {
            // a CIGAR beyond the register head, unusual thresholds, or a tile beyond the LDS capacity: the literal walk
            WalkState w{pos + 1, pos, 0};
            auto emit = [&](int k, int s, int e) {
                if (in_lds && (uint32_t)k < room) s_SE[lub + (uint32_t)k] = make_int2(s, e);
                else over = true;
                sane = sane & (s <= e);
                re.sl = s; re.el = e;
            };
            walk_ops<false>(w, a->f.cig + c_lo, 0, (int)n_cig, p, emit);
            emit(w.n, w.start, w.end);
            n = (uint32_t)w.n + 1u;
            if (in_lds && !over) { const int2 x0 = s_SE[lub]; re.s0 = x0.x; re.e0 = x0.y; }
        }
        // the next tile's CIGAR heads (its record fields were issued a tile ago), the order byte of the tile after it
        if (has_next) fused_load_words(a, vn);
        const int32_t src_n2 = has_next2 ? fused_src(u_order, Tn2, t_next2) : -1;
        F_STAMP(0);
        F_BARRIER();                     // every walk is done
        // ---- stage the dictionary slices, re-based to the tile's window
        int my_wide = 0;
        if (fast) {
            if ((int)threadIdx.x < FUSED_KEY_CAP) {
                const bool has_st = threadIdx.x < d.st_nk, has_en = threadIdx.x < d.en_nk;
                v4i_t e0, e1;
                e0.x = dv.xa.x; e0.y = dv.xa.y; e1.x = dv.xc.x; e1.y = dv.xc.y;
                if (d.flags & TD_CONTIG) {
                    e0.z = (int)rebase_mask((uint32_t)dv.xb.x, (uint32_t)dv.xb.y, dv.xa.z - d.j_lo);
                    e0.w = (int)rebase_mask((uint32_t)dv.xb.z, (uint32_t)dv.xb.w, dv.xa.z - d.j_lo);
                    e1.z = (int)rebase_mask((uint32_t)dv.xd.x, (uint32_t)dv.xd.y, dv.xc.z - d.j_lo);
                    e1.w = (int)rebase_mask((uint32_t)dv.xd.z, (uint32_t)dv.xd.w, dv.xc.z - d.j_lo);
                } else {
                    e0.z = (int)rebase_gaps(tw->win, w_n, (uint32_t)dv.xb.x, (uint32_t)dv.xb.y, dv.xa.z);
                    e0.w = (int)rebase_gaps(tw->win, w_n, (uint32_t)dv.xb.z, (uint32_t)dv.xb.w, dv.xa.z);
                    e1.z = (int)rebase_gaps(tw->win, w_n, (uint32_t)dv.xd.x, (uint32_t)dv.xd.y, dv.xc.z);
                    e1.w = (int)rebase_gaps(tw->win, w_n, (uint32_t)dv.xd.z, (uint32_t)dv.xd.w, dv.xc.z);
                }
                if (has_st) { s_ent0[threadIdx.x] = e0; if (dv.xa.w & SE_WIDE) my_wide = 1; }
                if (has_en) { s_ent1[threadIdx.x] = e1; if (dv.xc.w & SE_WIDE) my_wide = 1; }
            }
            if (d.nbk > 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = (int)threadIdx.x + q * TILE_THREADS;
                    if (i <= d.nbk) {
                        s_dir0[i] = (uint8_t)(dv.dd[0][q] - d.st_r0); s_dir1[i] = (uint8_t)(dv.dd[1][q] - d.en_r0);
                        s_rdir[i] = (uint8_t)(dv.dd[2][q] - d.st_r0);
                    }
                }
            }
            if (threadIdx.x < 3u && (threadIdx.x > 0u || d.nbk == 0)) {
                s_dir0[d.nbk + (int)threadIdx.x] = (uint8_t)d.st_nk; s_dir1[d.nbk + (int)threadIdx.x] = (uint8_t)d.en_nk;
            }
        }
        if (my_wide) s_wide = 1;
        F_STAMP(1);
        F_BARRIER();
        const int any_wide = s_wide;
        // ---- phase 2: classification (the classic kernel's device functions; a read's slot range starts at `lub`)
        uint32_t info = n << 8; int ref = -1;
        bool redo = active && (!fast || !in_lds || over || any_wide != 0 || tid != d.tid || (n > 1 && !sane));
        const bool work = active && !redo;
        const TileLds L{nullptr, nullptr, s_W, s_ent0, s_ent1, s_dir0, s_dir1, s_rdir, tw->hk, tw->hx, tw->win};
        const VisitMasks vm = visit_window<LEVEL>(L, d, w_n, work, n, d.j_lo, re, tw->mask);
        redo = redo || vm.redo;
        // the next tile's span: upper bounds of its read ends (the registers loaded above), reduced over the workgroup
        if (has_next) {
            fused_mask_words(vn);
            const int m = wave_max((vn.src >= 0 && vn.tid == Tn.tid0) ? fused_end_bound(a, vn) : INT32_MIN);
            if (lane == 0) s_wmax[wv] = m;
        }
        F_STAMP(2);
        F_BARRIER();
        // ... and its descriptor and window, by the last wave, into the other TileWin, while the others start probing
        if (has_next && wv == TILE_THREADS / WAVE - 1) {
            const unsigned long long d0 = stamping ? __builtin_readcyclecounter() : 0ull;
            make_descriptor(a, lane, Tn.tid0, Tn.pos0 + 1, max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])), Tn.in_lds, tw_next);
            if (stamping && lane == 0) atomicAdd(&a->f.stamps[(blockIdx.x & 1023u) * 8u + 7u], __builtin_readcyclecounter() - d0);
        }
        const SiteMasks sm = map_exons_se(L, s_SE, d, work && !redo && n > 1, lub, n, vm.vpre);
        if (work && !redo) {
            const Verdict vd = decide<LEVEL>(L, d, lub, n, re, vm, sm, rev_in);
            info = vd.info; ref = vd.ref;
        } else if (active && in_lds && !over) {
            for (uint32_t k = 0; k < n; ++k) s_W[lub + k] = (uint16_t)0;
        }
        redo = redo && active;
        {
            const unsigned long long m = __ballot(redo);
            if (m) {
                uint32_t at = 0;
                if (lane == 0) at = atomicAdd(a->f.redo_count, (uint32_t)__popcll(m));
                at = __shfl(at, 0, WAVE);
                if (redo) a->f.redo[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = r;
            }
        }
        if (active) s_nx[src] = n;
        F_STAMP(3);
        F_BARRIER();
        // ---- exact exon offsets in read order (one wave, four reads per lane) and the tile's chunk of the result arrays
        if (wv == 0) {
            const uint4 quad = *reinterpret_cast<const uint4 *>(s_nx + 4 * lane);
            const uint32_t c0 = (uint32_t)(4 * lane) < T.n_act ? quad.x : 0u, c1 = (uint32_t)(4 * lane + 1) < T.n_act ? quad.y : 0u;
            const uint32_t c2 = (uint32_t)(4 * lane + 2) < T.n_act ? quad.z : 0u, c3 = (uint32_t)(4 * lane + 3) < T.n_act ? quad.w : 0u;
            const uint32_t mine = c0 + c1 + c2 + c3;
            const uint32_t inc = wave_inclusive_scan(mine), ex = inc - mine;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
            *reinterpret_cast<uint4 *>(s_nx + 4 * lane) = make_uint4(ex, ex + c0, ex + c0 + c1, ex + c0 + c1 + c2);
            if (lane == 0) {
                const unsigned long long at = total ? atomicAdd(a->ex_cursor, (unsigned long long)total) : 0ull;
                s_base[0] = (uint32_t)at; s_base[1] = total;
                a->tile_start[t] = (uint32_t)at; a->tile_total[t] = total;
            }
        }
        F_BARRIER();
        const uint32_t base = s_base[0], tile_total = s_base[1];
        // a tile whose exons could not be kept in LDS (capacity), or a read beyond its bound: walked again, straight to HBM
        const bool direct = !in_lds;
        if (active) {
            const uint32_t loc = s_nx[src];
            a->f.ex_off[r] = base + loc;
            a->f.info[r] = info;
            a->f.ref_tx[r] = ref;
            if (!direct && !over) for (uint32_t k = 0; k < n; ++k) s_map[loc + k] = (uint16_t)(lub + k);
            if (direct || over) {
                int32_t *const xs = a->f.ex_start, *const xe = a->f.ex_end; uint8_t *const xf = a->f.ex_flag;
                WalkState w{pos + 1, pos, 0};
                auto put = [&](int k, int s, int e) { xs[base + loc + (uint32_t)k] = s; xe[base + loc + (uint32_t)k] = e; xf[base + loc + (uint32_t)k] = 0; };
                walk_ops<false>(w, a->f.cig + c_lo, 0, (int)n_cig, p, put);
                put(w.n, w.start, w.end);
                if (!direct) for (uint32_t k = 0; k < n; ++k) s_map[loc + k] = (uint16_t)0xffffu;       // (skipped by the copy below)
            }
        }
        F_BARRIER();
        if (!direct) {
            int32_t *const xs = a->f.ex_start, *const xe = a->f.ex_end; uint8_t *const xf = a->f.ex_flag;
            for (uint32_t i = threadIdx.x; i < tile_total; i += TILE_THREADS) {
                const uint32_t q = s_map[i];
                if (q == 0xffffu) continue;
                const int2 x = s_SE[q];
                st32(xs, base + i, x.x);
                st32(xe, base + i, x.y);
                st32(xf, base + i, (uint8_t)s_W[q]);
            }
        }
        F_STAMP(4);
        F_BARRIER();                     // the tile's LDS image has been written out; the next descriptor is complete
        T = Tn; Tn = Tn2; v = vn; src_n1 = src_n2; cur ^= 1;
    }
#undef F_STAMP
#undef F_BARRIER
}

// The result arrays of the fused pipeline, tile chunk by tile chunk, into read order (l2r_download): tile t's exons
// [start[t], start[t] + total[t]) -> [dest[t], ...), dest = exclusive scan of the totals in tile order.
__global__ __launch_bounds__(TILE_THREADS)
void k_linearize(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ tile_dest, const uint32_t *__restrict__ tile_total,
                 const int32_t *__restrict__ xs, const int32_t *__restrict__ xe, const uint8_t *__restrict__ xf,
                 int32_t *__restrict__ os, int32_t *__restrict__ oe, uint8_t *__restrict__ of)
{
    const uint32_t from = tile_start[blockIdx.x], to = tile_dest[blockIdx.x], n = tile_total[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < n; i += TILE_THREADS) { os[to + i] = xs[from + i]; oe[to + i] = xe[from + i]; of[to + i] = xf[from + i]; }
}

}  // namespace l2r
