// l2r_tile.hip.h -- the ONE-KERNEL tile path for coordinate-sorted records with short CIGARs (gfx950): what k_walk_slab hands to
// k_probe_slab through HBM (4 bytes per exon each way, a word per read each way: 0.64 GB of the two-kernel step's 2.47 GB on 10 M
// reads) stays inside one workgroup.
//
//   k_tile_index            at UPLOAD, once per read set (below): per tile the span, op statistics and slot records -- functions of the
//                           records alone, no option has a say in them.
//   k_describe_scan<true>   first kernel of a run: the tiles' descriptors and windows from the spans the upload recorded, the tile lists
//                           of the 64-bit-mask / chunked kernels, the run's counters cleared, and -- for every tile whose exon count the
//                           op statistics settle under this run's thresholds (tile_exact: nearly all) -- the count itself, written into
//                           the words the later tiles read (lb_tile / lb_blk / lb_sup, l2r_slab.hip.h).
//   k_tile                  per tile: the slot records (one load by tile number), CIGAR heads into registers, dictionary slices and
//                           window into LDS; the CIGAR registers are walked once to PLACE the read's exons as row words at their
//                           read-order positions in LDS -- the image k_probe_slab builds from slab rows (an exact tile knows every
//                           read's place from its slot record; any other tile walks twice: once to COUNT, a scan, then to place).
//                           Window pass, probes, verdicts, the junction check (-j: rows staged over the dead dictionary slices) and
//                           the coalesced write-out (same device functions as the slab kernels).  Registers are max(walk, probe),
//                           not the sum: the 24 CIGAR words are dead before the probe rounds begin.
//   the tile's first result slot = the exon counts of all tiles in front: the counts of the tiles in front of it inside its block of
//                           16, the sums of the blocks in front inside its super-block of 64 blocks, the sums of the super-blocks in
//                           front (three 64-lane loads, one per wave).  Counts k_describe_scan wrote are complete for every reader
//                           from its first instruction (plain loads at the top of the kernel, nothing to wait for).  A tile that is
//                           not exact publishes its count from k_tile (agent-scope atomics on both sides, no fence) and later tiles
//                           poll for it: fine for a few, a convoy for many (launch_all sends runs with more than 2 % of such tiles
//                           to the two-kernel path).  Workgroup -> tile is blocked-cyclic over the XCDs (fused_tile: 16 consecutive
//                           tiles per XCD share their dictionary slices in one L2).  A tile that waits in vain (LB_POLLS) sets lb_err,
//                           which ends every other wait too: l2r_sync then does the run again on the slab pipeline.
//   k_tile<..., WIDE>       the same kernel with 64-bit masks, 24-byte entries and the 63-member window record, one workgroup per entry of
//                           wide_list, launched BESIDE the plain instance on a stream of its own: the windows of 33 .. 63 transcripts
//                           (tile_wide_direct, l2r_slab.hip.h: exact tiles; the plain instance returns at once for them).
//   k_tile_chunk            (l2r_tchunk.hip.h) the exact tiles with windows beyond 63 transcripts, likewise beside the plain instance, which
//                           returns at once for them (TD_CDIRECT: k_describe_scan's verdict).
//   tiles this kernel does not finish: windows beyond 32 members that are not exact (k_probe_slab_chunked / _wide),
//                           a dictionary key in several entries, a read of 255 exons or more, more exons than the staged positions
//                           hold (only outliers make such tiles).  For those the workgroup runs k_walk_slab's tile body instead
//                           (slab_walk_tile: slab rows, reads' words, span record) and lists the tile for k_probe_slab (fb_list).
//
// -e < 1 and long CIGARs keep the two-kernel path (l2r_slab.hip.h).
#pragma once
#include <type_traits>
#include "l2r_chunk.hip.h"

namespace l2r {

#ifndef L2R_TILE_GROUP_SHIFT
#define L2R_TILE_GROUP_SHIFT 4
#endif
constexpr uint32_t TILE_GROUP = 1u << L2R_TILE_GROUP_SHIFT;      // consecutive tiles per XCD
// Workgroup -> tile, blocked-cyclic: workgroup b runs on XCD b % 8 (observed placement: for speed only), so XCD x takes the tile groups
// x, x + 8, x + 16, ...  The grid is a whole number of rounds of 8 groups; workgroups behind the last tile leave at once.
__device__ __forceinline__ uint32_t fused_tile(uint32_t b)
{
    const uint32_t x = b & 7u, i = b >> 3;
    return ((((i >> L2R_TILE_GROUP_SHIFT) << 3) + x) << L2R_TILE_GROUP_SHIFT) + (i & (TILE_GROUP - 1u));
}
inline unsigned fused_grid(int64_t n_tiles)
{
    const int64_t per = 8 * (int64_t)TILE_GROUP;
    return (unsigned)(std::max<int64_t>((n_tiles + per - 1) / per, 1) * per);
}

// Upload time, once per read set: an index of every tile's CIGAR operations that no parameter has a say in.
//   TileRec::pad[0]  the tile's LAST base = the largest end among its reads = pos + the reference bases of the CIGAR (ops M D N = X),
//                    whatever the parameters cut or keep (src/bam2gtf.c:41-74: `end` only ever grows by these lengths):
//                    k_describe_scan<true> makes the tile's window from it before any CIGAR has been walked;
//   TileStat         how many N operations the tile's reads have, the shortest of them, the longest D operation, and the shortest
//                    stretch of reference bases between two N operations of one read.  With them a run knows a tile's EXON COUNT
//                    without its CIGARs whenever no threshold is borderline inside the tile (tile_exact): every N is an intron
//                    (-i <= the shortest N), no D cuts (-t >= the longest D), no inner exon is dropped (-e <= the shortest
//                    stretch) => exons = reads + N operations (src/bam2gtf.c:41-74).  Tiles for which that does not hold count in k_tile.
//   slot records     the tile's reads by falling CIGAR length (the counting sort k_walk_slab does per run: CIGAR lengths only), one
//                    12-byte record per SLOT at a fixed place -- tile t, thread p: u_slot[t * 256 + p] -- so k_tile asks for them with
//                    nothing but its tile number, beside its scalar loads, and the CIGAR heads are its second round trip.  Thread p
//                    holds slot (p + 64 rot) & 255, rot = a hash of the tile number: the slot groups (group 0 = the longest reads)
//                    are rotated over the waves.  A record also says where the read's exons begin among the tile's exons in read order
//                    IF the tile is exact (`loc`: the sum of 1 + N operations over the tile's reads in front of it): an exact tile
//                    needs neither a count walk nor a scan.
struct SlotRec { uint32_t c_lo; int32_t pos; uint32_t xw; };      // xw: SLOT_* fields
// xw: CIGAR length in bits 0-7 (255: that many or more), strand bit 8, "a read" bit 9, read number inside the tile bits 10-17, loc bits 18-29
constexpr uint32_t SLOT_REV = 1u << 8, SLOT_VALID = 1u << 9;
constexpr int SLOT_IDX_SHIFT = 10, SLOT_LOC_SHIFT = 18;
constexpr uint32_t SLOT_LOC_LIMIT = 1u << 12;
__device__ __forceinline__ uint32_t tile_rot(uint32_t t) { return (t ^ (t >> 3) ^ (t >> 7)) & 3u; }
// SUMMARY: the caller's per-record CIGAR summaries are there (l2r_reads::cig_summary) -- the host has made the tiles' statistics and last
// bases from them, the records' N operations come as one 16-bit column (sum_nn): no CIGAR is touched, the kernel is the scan and the
// counting sort over 15 bytes per record in, 12 out.
template <bool SUMMARY>
__global__ __launch_bounds__(TILE_THREADS)
void k_tile_index(TileRec *__restrict__ rec, TileStat *__restrict__ stat, SlotRec *__restrict__ slot_rec, uint32_t n_tiles,
                  const uint32_t *__restrict__ cig_off32, const int32_t *__restrict__ r_pos, const uint8_t *__restrict__ r_rev, const uint32_t *__restrict__ cig,
                  const uint16_t *__restrict__ sum_nn)
{
    __shared__ int s_m[5][TILE_THREADS / WAVE];
    __shared__ uint32_t s_hist[WAVE], s_wave[TILE_THREADS / WAVE];
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint32_t r0 = rec[t].r0, n_act = rec[t].n_act;
        const uint32_t i = threadIdx.x;
        const bool active = i < n_act;
        int end = INT32_MIN, n_n = 0, min_n = INT32_MAX, max_d = 0, min_seg = INT32_MAX;
        uint32_t c_lo = 0u, c = 0u, rev = 0u; int32_t pos = 0;
        if (active) {
            const uint32_t r = r0 + i;
            c_lo = cig_off32[r]; c = cig_off32[r + 1u] - c_lo; pos = r_pos[r]; rev = r_rev[r] ? 1u : 0u;
            end = pos;
            int seg = 0; bool first = true;
            auto step = [&](uint32_t w) {
                const uint32_t op = w & 0xfu; const int len = (int)(w >> 4);
                if (op == 3u) {
                    ++n_n; min_n = min(min_n, len);
                    if (!first) min_seg = min(min_seg, seg);      // (the first exon is kept whatever its length)
                    first = false; seg = 0;
                } else {
                    if (op == 2u) max_d = max(max_d, len);
                    seg += len & __builtin_amdgcn_sbfe(0x18d, op, 1u);
                }
                end += len & __builtin_amdgcn_sbfe(0x18d, op, 1u);
            };
            if (SUMMARY) n_n = (int)sum_nn[r];
            else {
                // (the head of the CIGAR as k_tile fetches it: six 16-byte vectors in flight at once, words behind the last op = "I, length 0";
                //  one 4-byte load per op in a serial loop had every thread wait for each word: 0.42 ms for 10 M reads)
                const uint32_t *const words = cig + c_lo;
                uint32_t cg[SLAB_HEAD];
#pragma unroll
                for (int k = 0; k < SLAB_HEAD; ++k) cg[k] = 1u;
#pragma unroll
                for (int q = 0; q < SLAB_HEAD_VEC; ++q)
                    if ((uint32_t)(4 * q) < c) {
                        const v4i_a4 x = *reinterpret_cast<const v4i_a4 *>(words + 4 * q);
                        cg[4 * q] = (uint32_t)x.x; cg[4 * q + 1] = (uint32_t)x.y; cg[4 * q + 2] = (uint32_t)x.z; cg[4 * q + 3] = (uint32_t)x.w;
                    }
#pragma unroll
                for (int k = 0; k < SLAB_HEAD; ++k) if ((uint32_t)k < c) step(cg[k]);
                for (uint32_t k = SLAB_HEAD; k < c; ++k) step(words[k]);
            }
        }
        // the read's place among the tile's exons in read order if every N operation is an intron and nothing is dropped
        uint32_t tot_x;
        const uint32_t loc = block_exclusive_scan(active ? (uint32_t)n_n + 1u : 0u, s_wave, tot_x);
        if (!SUMMARY) {      // (with summaries the host has applied both rules to the statistics it made)
        if (__syncthreads_or(active && n_n + 1 >= 255)) min_seg = INT32_MIN;      // (a read of 255 exons or more: never an exact tile -- k_tile counts it and gives it the slab form)
        if (tot_x >= SLOT_LOC_LIMIT) min_seg = INT32_MIN;         // (places the record cannot say: the tile counts in k_tile -- it keeps the slab form anyway)
        }
        // the counting sort of k_walk_slab (64 bins by CIGAR length, threads without a read last)
        if (i < (uint32_t)WAVE) s_hist[i] = 0u;
        __syncthreads();
        const uint32_t est = active ? max(1u, min((c + 1u) >> 1, (uint32_t)(WAVE - 1))) : 0u;
        const uint32_t bin = (uint32_t)(WAVE - 1) - est;
        const uint32_t rank = atomicAdd(&s_hist[bin], 1u);
        __syncthreads();
        if (i < (uint32_t)WAVE) { const uint32_t v = s_hist[i]; s_hist[i] = wave_inclusive_scan(v) - v; }
        __syncthreads();
        const uint32_t at = (s_hist[bin] + rank - (tile_rot(t) << 6)) & (uint32_t)(TILE_THREADS - 1);      // the thread of that slot
        slot_rec[(size_t)t * TILE_THREADS + at] = SlotRec{c_lo, pos, min(c, 255u) | (rev ? SLOT_REV : 0u) | (active ? SLOT_VALID : 0u) | (i << SLOT_IDX_SHIFT) |
                                                                         (min(loc, SLOT_LOC_LIMIT - 1u) << SLOT_LOC_SHIFT)};
        if (SUMMARY) continue;                                   // (statistics and last base: the host's, from the summaries)
        const int v[5] = {wave_max(end), (int)wave_sum((uint32_t)n_n), wave_min(min_n), wave_max(max_d), wave_min(min_seg)};
        if ((threadIdx.x & (WAVE - 1)) == 0) for (int k = 0; k < 5; ++k) s_m[k][threadIdx.x >> 6] = v[k];
        __syncthreads();
        if (threadIdx.x == 0) {
            rec[t].pad[0] = (uint32_t)max(max(s_m[0][0], s_m[0][1]), max(s_m[0][2], s_m[0][3]));
            TileStat st;
            st.n_ops_n = s_m[1][0] + s_m[1][1] + s_m[1][2] + s_m[1][3];
            st.min_n = min(min(s_m[2][0], s_m[2][1]), min(s_m[2][2], s_m[2][3]));
            st.max_d = max(max(s_m[3][0], s_m[3][1]), max(s_m[3][2], s_m[3][3]));
            st.min_seg = min(min(s_m[4][0], s_m[4][1]), min(s_m[4][2], s_m[4][3]));
            stat[t] = st;
        }
        __syncthreads();
    }
}

// ---- the tiles' exon counts -> first result slots (see the head of this file; the words' layout: l2r_slab.hip.h LB_*)
constexpr uint32_t LB_POLLS = 1u << 19;                  // (2^19 polls, 3.4 us apart in the end: two seconds)
__device__ __forceinline__ unsigned long long lb_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// thread 0 of a tile whose exon count k_describe_scan could not know (tile_exact), once the tile has counted
__device__ __forceinline__ void lb_publish(SlabArgsK sa, uint32_t t, uint32_t total)
{
    const unsigned long long one = 1ull << LB_SHIFT;
    __hip_atomic_store(sa->lb_tile + t, one | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t b = t >> LB_BLK_SHIFT, in_blk = min((uint32_t)LB_BLK, sa->n_tiles - (b << LB_BLK_SHIFT));
    const unsigned long long old = __hip_atomic_fetch_add(sa->lb_blk + b, one | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (the block's last count: its sum goes to the super-block)
    if ((uint32_t)(old >> LB_SHIFT) + 1u == in_blk)
        (void)__hip_atomic_fetch_add(sa->lb_sup + (t >> LB_SUP_SHIFT), one | ((old & LB_SUM_MASK) + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Waves 0 .. 2 of a tile: the words of one level in front of tile t -- wave 0 the tiles of its block, wave 1 the blocks of its super-block,
// wave 2 the super-blocks.  lb_words: the level's first word, how many of them, and the number of counts a complete one holds.
struct LbLevel { const unsigned long long *p; uint32_t cnt, want; };
__device__ __forceinline__ LbLevel lb_level(SlabArgsK sa, uint32_t t, int wv)
{
    if (wv == 0) return LbLevel{sa->lb_tile + (t & ~(uint32_t)(LB_BLK - 1)), t & (uint32_t)(LB_BLK - 1), 1u};
    if (wv == 1) return LbLevel{sa->lb_blk + ((t >> LB_SUP_SHIFT) << (LB_SUP_SHIFT - LB_BLK_SHIFT)), (t >> LB_BLK_SHIFT) & (uint32_t)((1 << (LB_SUP_SHIFT - LB_BLK_SHIFT)) - 1), (uint32_t)LB_BLK};
    return LbLevel{sa->lb_sup, t >> LB_SUP_SHIFT, 1u << (LB_SUP_SHIFT - LB_BLK_SHIFT)};
}
// The wave's share of the tile's first result slot = the sum of its level's words, every one of them complete; a word that is not yet
// is polled.  (Wave-uniform result; a shard's exon count is below 2^32, so the low 32 bits of every partial sum are exact.)
// TRY: one look, no poll -- `done` says whether every word was complete (the tile's first look, at its start: with every count in front
// known since k_describe_scan it is the only one).
template <bool TRY>
__device__ __forceinline__ uint32_t lb_share(SlabArgsK sa, uint32_t t, int wv, int lane, bool &done, uint32_t &n_polls)
{
    const LbLevel L = lb_level(sa, t, wv);
    n_polls = 0u; done = true;
    uint32_t sum = 0u;
    for (uint32_t base = 0u; base < L.cnt; base += (uint32_t)WAVE) {
        const uint32_t i = base + (uint32_t)lane;
        unsigned long long v = 0ull;
        bool ok = i >= L.cnt;
        // A poll is one load instruction of the lanes whose word is not complete yet; between polls the wave sleeps, longer every time
        // (thousands of waves that poll back to back take the memory system from the tiles they are waiting for).
        for (uint32_t polls = 0u;; ++polls) {
            if (!ok) { v = lb_load(L.p + i); ok = (uint32_t)(v >> LB_SHIFT) == L.want; }
            if (__all(ok)) break;
            if (TRY) { done = false; break; }
            ++n_polls;
            // (a tile that has waited in vain says so; the others stop waiting when they see that: a broken run ends in seconds)
            if (polls == LB_POLLS) { if (lane == 0) atomicOr(sa->lb_err, 1u); break; }
            if ((polls & 63u) == 63u && __hip_atomic_load(sa->lb_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (polls < 2u) __builtin_amdgcn_s_sleep(16); else if (polls < 6u) __builtin_amdgcn_s_sleep(48); else __builtin_amdgcn_s_sleep(127);
        }
        sum += i < L.cnt ? (uint32_t)v : 0u;
    }
    return wave_sum(sum);
}

// map_exons_slab (l2r_slab.hip.h) with the read's exons at their read-order positions in LDS (A[loc + j]: the row word the place walk
// left) instead of in slab rows: exon k + AHEAD is read into exon k's register when round k is done with it, every round leaves
// {start | work word, length} at the exon's position.  Positions of other lanes are never touched; a lane's reads behind its last exon
// re-read that one (clipped index), live or rewritten: not used.
#ifndef L2R_TILE_AHEAD
#define L2R_TILE_AHEAD 2
#endif
constexpr int TILE_AHEAD = L2R_TILE_AHEAD;
#ifndef L2R_TILE_NS
#define L2R_TILE_NS 1
#endif
#ifndef L2R_TILE_NE
#define L2R_TILE_NE 2
#endif
#ifndef L2R_WIDE_WGS
#define L2R_WIDE_WGS 6
#endif
#ifndef L2R_TILE_PRIO
#define L2R_TILE_PRIO 1
#endif
constexpr int TILE_PRIO = L2R_TILE_PRIO;                // k_tile: a wave's issue priority outside its classification (0 inside)
constexpr int TILE_NS = L2R_TILE_NS, TILE_NE = L2R_TILE_NE;      // entries of a START / END bucket every probe round looks at without a loop
template <bool DIS>
__device__ __forceinline__ SiteMasks map_exons_lds(const TileLds &L, const TileDesc &d, bool mapping, uint32_t n, uint32_t vpre, const SlabStage &st, int dis = 0, int rs = 0, int re = 0)
{
    SiteMasks m{0xffffffffu, 0u, 0u, 0u, 0u};
    uint32_t *const Ap = st.A + st.loc; uint16_t *const Lp = st.Ln + st.loc;
    const uint32_t nm1 = mapping ? n - 1u : 0u;
    SlabRow R[TILE_AHEAD];
#pragma unroll
    for (int i = 0; i < TILE_AHEAD; ++i) R[i] = SlabRow{Ap[min((uint32_t)i, nm1)]};
    const uint32_t none = (uint32_t)d.nbk + 1u;         // a bucket behind the staged ones: the staging leaves it empty
    const int k_max = wave_max(mapping ? (int)n : 0);
    auto buckets = [&](int k, int sv, int ev, uint32_t &ls, uint32_t &hs, uint32_t &le, uint32_t &he) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        if (DIS) { near_range(L.dir0, d.b_off, none, live, sv, dis, ls, hs); near_range(L.dir1, d.b_off, none, junc, ev, dis, le, he); return; }
        const uint32_t is = live ? min((uint32_t)((sv >> SITE_SHIFT) + d.b_off), none) : none;      // (a lane without exon k must not open the long-bucket path)
        const uint32_t ie = junc ? min((uint32_t)((ev >> SITE_SHIFT) + d.b_off), none) : none;
        ls = L.dir0[is]; hs = L.dir0[is + 1u]; le = L.dir1[ie]; he = L.dir1[ie + 1u];
    };
    uint32_t ls, hs, le, he;
    int e_cur = slab_row_end(R[0], st.lo);
    buckets(0, slab_row_start(R[0], st.lo), e_cur, ls, hs, le, he);
    auto round = [&](int k, SlabRow &cur, const SlabRow &nxt, bool reload) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        const int s = slab_row_start(cur, st.lo), e = e_cur, s2 = slab_row_start(nxt, st.lo), e2 = slab_row_end(nxt, st.lo);
        const uint32_t cw = cur.w;
        uint32_t ls_n, hs_n, le_n, he_n;
        uint32_t xm, am, jm, dm;
        if (DIS) {
            buckets(k + 1, s2, e2, ls_n, hs_n, le_n, he_n);
            probe_near(L.ent0, ls, hs, s, e, dis, rs, re, xm, am, m.amb);
            probe_near(L.ent1, le, he, e, s2, dis, rs, re, jm, dm, m.amb);
        } else {
            v4i_t qs[TILE_NS], qe[TILE_NE];
#pragma unroll
            for (int i = 0; i < TILE_NS; ++i) qs[i] = lds_entry(L.ent0, ls + (uint32_t)i);
#pragma unroll
            for (int i = 0; i < TILE_NE; ++i) qe[i] = lds_entry(L.ent1, le + (uint32_t)i);
            buckets(k + 1, s2, e2, ls_n, hs_n, le_n, he_n);
            am = 0u; xm = 0u; jm = 0u; dm = 0u;
#pragma unroll
            for (int i = 0; i < TILE_NS; ++i) {
                const bool mi = ls + (uint32_t)i < hs && qs[i].x == s;
                am = mi ? (uint32_t)qs[i].w : am; xm = (mi && qs[i].y == e) ? (uint32_t)qs[i].z : xm;
            }
#pragma unroll
            for (int i = 0; i < TILE_NE; ++i) {
                const bool mi = le + (uint32_t)i < he && qe[i].x == e;
                dm = mi ? (uint32_t)qe[i].w : dm; jm = (mi && qe[i].y == s2) ? (uint32_t)qe[i].z : jm;
            }
            if (__any(hs > ls + (uint32_t)TILE_NS || he > le + (uint32_t)TILE_NE)) { probe_rest(L.ent0, ls + (uint32_t)TILE_NS, hs, s, e, xm, am, 0u); probe_rest(L.ent1, le + (uint32_t)TILE_NE, he, e, s2, jm, dm, 0u); }
        }
        if (reload) cur = SlabRow{Ap[min((uint32_t)k + (uint32_t)TILE_AHEAD, nm1)]};      // exon k + TILE_AHEAD into the register of exon k
        const uint32_t amj = junc ? am : 0u;
        uint32_t word = first_member(xm & vpre);
        word |= first_member(jm & vpre) << 6;
        word |= nonzero(dm & vpre) << 12;
        word |= nonzero(amj & vpre) << 13;
        m.kand &= junc ? (am & dm) : 0xffffffffu;     // Q1: the acceptor probed with exon k is ITS OWN start, k < n-1
        m.kor |= amj | dm;
        if (!DIS) { if (k == 0) m.dm_first = dm;
                    m.am_last = (live && !junc) ? am : m.am_last; }
        if (live) { Ap[k] = (cw & SLAB_REL_MASK) | (word << SLAB_REL_BITS); Lp[k] = (uint16_t)(cw >> SLAB_REL_BITS); }
        ls = ls_n; hs = hs_n; le = le_n; he = he_n; e_cur = e2;
    };
    int k = 0;
    for (; k + TILE_AHEAD <= k_max; k += TILE_AHEAD) {
#pragma unroll
        for (int i = 0; i < TILE_AHEAD; ++i) round(k + i, R[i], R[(i + 1) % TILE_AHEAD], true);
    }
#pragma unroll
    for (int i = 0; i < TILE_AHEAD - 1; ++i) {
        if (k + i >= k_max) break;
        round(k + i, R[i], R[i + 1], false);
    }
    return m;
}

// Verdicts of the tile's reads from the staged window + dictionaries (slab_classify with the exons in LDS).  big: the read has an exon
// the row word cannot say (16 kb or longer, or 2^18 - 1 bases or more behind the tile's first base): the generic kernel classifies it,
// its exons are written from a literal walk once the tile's first slot is known, its positions are marked.
template <int LEVEL, bool DIS>
__device__ __forceinline__ SlabVerdict tile_classify(PipeArgsK a, const TileDesc &d, const SlabLds &S, const int4 *hk, const int4 *hx, const int *win, const uint32_t *tilemask,
                                                    bool active, uint32_t pre, bool big, uint32_t r, const ReadEnds &re, const SlabStage &st, int any_wide, SlabStamp &stamp)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const bool fast = (d.flags & TD_FAST) != 0;
    const int w_n = (fast && !(a->f.p.ablate & 8192)) ? (int)d.n_win : 0;          // (bits 12, 13: timing diagnostics -- no probe rounds, no member pass; results wrong)
    const uint32_t n = pre >> PRE_N_SHIFT;
    const bool rev_in = (pre & PRE_REV) != 0u;
    uint32_t info = n << 8; int ref = -1;
    bool redo = active && (!fast || big || any_wide != 0 || (n > 1 && (pre & PRE_INSANE) != 0u));
    const bool work = active && !redo;
    const TileLds L{nullptr, nullptr, nullptr, S.ent0, S.ent1, S.dir0, S.dir1, S.rdir, hk, hx, win};
    const VisitMasks vm = visit_window<LEVEL>(L, d, w_n, work, n, d.j_lo, re, tilemask);
    redo = redo || vm.redo;
    stamp.mark(2);
    const bool mapping = work && !redo && n > 1 && !(a->f.p.ablate & 4096);
    const SiteMasks sm = map_exons_lds<DIS>(L, d, mapping, n, vm.vpre, st, DIS ? a->f.p.ss_dis : 0, re.s0, re.el);
    stamp.mark(3);
    if (active && !mapping) {
        // no probe round has rewritten this read's row words: {start, flags 0} and the length apart, as the write-out reads them
        uint32_t *const Ap = st.A + st.loc; uint16_t *const Lp = st.Ln + st.loc;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t w = Ap[k];
            Ap[k] = big ? SLAB_POS_SKIP : (w & SLAB_REL_MASK); Lp[k] = (uint16_t)(w >> SLAB_REL_BITS);
        }
    }
    // (-d > 0: a visited member with two sites within the tolerance of one read site -- its pair count is the generic kernel's)
    if (DIS && mapping && (sm.amb & vm.vpre) != 0u) redo = true;
    if (work && !redo) {
        uint32_t *const Ap = st.A + st.loc;
        // (the read's ends once more, from its staged exons: four registers that need not live through the probe rounds)
        const uint16_t *const Lq = st.Ln + st.loc;
        ReadEnds re2;
        re2.s0 = st.lo + (int)(Ap[0] & SLAB_REL_MASK); re2.e0 = re2.s0 + (int)Lq[0] - 1;
        re2.sl = st.lo + (int)(Ap[n - 1u] & SLAB_REL_MASK); re2.el = re2.sl + (int)Lq[n - 1u] - 1;
        const Verdict vd = decide<LEVEL>(L, d, n, re2, vm, sm, rev_in, [&](int k) { return Ap[k] >> SLAB_REL_BITS; },
                                         [&](int k, uint32_t f) { Ap[k] = (Ap[k] & SLAB_REL_MASK) | (f << SLAB_REL_BITS); });
        info = vd.info; ref = vd.ref;
    }
    stamp.mark(4);
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
    if (active) a->f.ref_tx[r] = ref;                           // (info: by the caller, behind the junction check)
    return SlabVerdict{info, ref, redo};
}

// ---- the same for a tile whose window holds 33 .. 63 transcripts (k_tile<..., WIDE>): 64-bit masks, 24-byte entries
// probe_all64 (l2r_wide.hip.h) for keys that may come in several entries (SE_WIDE: the key's transcripts lie more than 64 apart in the
// annotation): the parts' masks are ORed, each re-based to the window -- a part that has nothing to say about it re-bases to nothing
// (k_probe_slab_chunked's rule, probe_parts64); such a tile need not be handed on.
__device__ __forceinline__ void probe_or64(const WEnt *ent, uint32_t lo, uint32_t hi, int32_t k1, int32_t k2, m64_t &pm, m64_t &sm)
{
    pm = 0ull; sm = 0ull;
    for (uint32_t r = lo; r < hi; ++r) {
        const WEnt q = ent[r];
        if (q.k1 == k1) { sm |= q.sm; if (q.k2 == k2) pm |= q.pm; }
    }
}
// map_exons_lds on 64-bit masks (map_exons_slab64 with the exons at their read-order positions in LDS)
__device__ __forceinline__ SiteMasks64 map_exons_lds_wide(const WideLds &L, const TileDesc &d, bool mapping, uint32_t n, m64_t vpre, const SlabStage &st, int dis, int rs, int re)
{
    SiteMasks64 m{~0ull, 0ull, 0ull, 0ull, 0ull};
    uint32_t *const Ap = st.A + st.loc; uint16_t *const Lp = st.Ln + st.loc;
    const uint32_t none = (uint32_t)d.nbk + 1u;         // a bucket behind the staged ones: the staging leaves it empty
    const int k_max = wave_max(mapping ? (int)n : 0);
    const uint32_t nm1 = mapping ? n - 1u : 0u;
    SlabRow cur{Ap[0]}, nxt{Ap[min(1u, nm1)]};
    for (int k = 0; k < k_max; ++k) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        const SlabRow nn{Ap[min((uint32_t)k + 2u, nm1)]};             // (behind the last exon: that one again -- live or rewritten, not used)
        const int s = slab_row_start(cur, st.lo), e = slab_row_end(cur, st.lo), s2 = slab_row_start(nxt, st.lo);
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
            probe_or64(L.ent0, ls, hs, s, e, xm, am);
            probe_or64(L.ent1, le, he, e, s2, jm, dm);
        }
        const m64_t amj = junc ? am : 0ull;
        uint32_t word = first_member64(xm & vpre);
        word |= first_member64(jm & vpre) << 6;
        word |= ((dm & vpre) ? 1u : 0u) << 12;
        word |= ((amj & vpre) ? 1u : 0u) << 13;
        m.kand &= junc ? (am & dm) : ~0ull;            // Q1: the acceptor probed with exon k is ITS OWN start, k < n-1
        m.kor |= amj | dm;
        if (dis <= 0) { if (k == 0) m.dm_first = dm;
                        m.am_last = (live && !junc) ? am : m.am_last; }
        if (live) { Ap[k] = (cw & SLAB_REL_MASK) | (word << SLAB_REL_BITS); Lp[k] = (uint16_t)(cw >> SLAB_REL_BITS); }
        cur = nxt; nxt = nn;
    }
    return m;
}
// slab_stage_dict on 64-bit masks (k_probe_slab_wide's staging): entries re-based to the tile's window, byte directories
__device__ __forceinline__ void wide_stage_dict(const TileDesc &d, const DictRegs &dv, const int *win, WEnt *s_ent0, WEnt *s_ent1, uint8_t *s_dir0, uint8_t *s_dir1, uint8_t *s_rdir)
{
    const int w_n = (int)d.n_win;
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
            rebase_gaps64(win, w_n, mm, dv.xa.z, dv.xc.z);
            e0.pm = mm[0]; e0.sm = mm[1]; e1.pm = mm[2]; e1.sm = mm[3];
        }
        if (has_st) s_ent0[threadIdx.x] = e0;
        if (has_en) s_ent1[threadIdx.x] = e1;
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
// tile_classify on 64-bit masks
template <int LEVEL, bool DIS>
__device__ __forceinline__ SlabVerdict tile_classify_wide(PipeArgsK a, const TileDesc &d, const WideLds &L, const m64_t *tilemask,
                                                         bool active, uint32_t pre, bool big, uint32_t r, const ReadEnds &re, const SlabStage &st)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const bool fast = (d.flags & TD_WIDE) != 0;
    const int w_n = (fast && !(a->f.p.ablate & 8192)) ? (int)d.n_win : 0;          // (bits 12, 13: timing diagnostics as in tile_classify)
    const uint32_t n = pre >> PRE_N_SHIFT;
    const bool rev_in = (pre & PRE_REV) != 0u;
    uint32_t info = n << 8; int ref = -1;
    bool redo = active && (!fast || big || (n > 1 && (pre & PRE_INSANE) != 0u));
    const bool work = active && !redo;
    const VisitMasks64 vm = visit_window64<LEVEL>(L, d, w_n, work, n, d.j_lo, re, tilemask);
    redo = redo || vm.redo;
    const bool mapping = work && !redo && n > 1 && !(a->f.p.ablate & 4096);
    const int dis = DIS ? a->f.p.ss_dis : 0;
    const SiteMasks64 sm = map_exons_lds_wide(L, d, mapping, n, vm.vpre, st, dis, re.s0, re.el);
    if (active && !mapping) {
        // no probe round has rewritten this read's row words: {start, flags 0} and the length apart, as the write-out reads them
        uint32_t *const Ap = st.A + st.loc; uint16_t *const Lp = st.Ln + st.loc;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t w = Ap[k];
            Ap[k] = big ? SLAB_POS_SKIP : (w & SLAB_REL_MASK); Lp[k] = (uint16_t)(w >> SLAB_REL_BITS);
        }
    }
    if (DIS && mapping && (sm.amb & vm.vpre) != 0ull) redo = true;
    if (work && !redo) {
        uint32_t *const Ap = st.A + st.loc;
        const uint16_t *const Lq = st.Ln + st.loc;
        ReadEnds re2;
        re2.s0 = st.lo + (int)(Ap[0] & SLAB_REL_MASK); re2.e0 = re2.s0 + (int)Lq[0] - 1;
        re2.sl = st.lo + (int)(Ap[n - 1u] & SLAB_REL_MASK); re2.el = re2.sl + (int)Lq[n - 1u] - 1;
        const Verdict vd = decide64<LEVEL>(L, d, n, re2, vm, sm, rev_in, [&](int k) { return Ap[k] >> SLAB_REL_BITS; },
                                           [&](int k, uint32_t f) { Ap[k] = (Ap[k] & SLAB_REL_MASK) | (f << SLAB_REL_BITS); });
        info = vd.info; ref = vd.ref;
    } else if (work && mapping) {
        // (probed, then found ambiguous: the work words go, flags 0 -- the generic kernel writes them)
        uint32_t *const Ap = st.A + st.loc;
        for (uint32_t k = 0; k < n; ++k) Ap[k] &= SLAB_REL_MASK;
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
    if (active) a->f.ref_tx[r] = ref;                           // (info: by the caller, behind the junction check)
    return SlabVerdict{info, ref, redo};
}

// LDS of k_tile beside the staged positions, dictionary slices, directories and window record of k_probe_slab: the reads' exon counts
// (one byte each, read order) and their exclusive scan (16 bit).
constexpr int TILE_LDS_BYTES = TILE_POS_CAP * 6 + 2 * SLAB_KEY_CAP * 16 + SLAB_AUX_BYTES + TILE_THREADS * 3 + 16 * 4 + 4 * 4;
static_assert(TILE_LDS_BYTES <= 23040, "k_tile: 7 workgroups per CU need 45 allocation granules of 512 bytes at most");
static_assert(TILE_POS_CAP >= 2 * TILE_THREADS + 16, "slab_walk_tile's words live in the staged positions");
static_assert(TILE_POS_CAP < 65536 && TILE_POS_CAP % 8 == 0, "16-bit places; 16-byte aligned arrays");

constexpr int SJ_STAGE = 2 * SLAB_KEY_CAP * 16 / 12;    // junction rows k_tile stages per tile: {donor, acceptor, running maximum of the acceptors} over the dead dictionary slices
static_assert(SLAB_AUX_BYTES >= (SJ_STAGE / 32 + 2 * 4 + 2) * 4 && SLAB_AUX_BYTES >= TILE_THREADS * 5 && 2 * SLAB_KEY_CAP * 16 >= TILE_POS_CAP,
              "the junction check's arrays fit the dead dictionary slices / directories");
// WIDE: the instance for the tiles of the 64-bit-mask kernel that tile_wide_direct (l2r_slab.hip.h) names -- launched behind the plain
// instance over wide_list (one workgroup per entry): the same tile, with 64-bit masks, 24-byte entries, a 63-member window record, at 5
// workgroups per CU.  The plain instance returns at once for those tiles instead of giving them the slab form.
template <int LEVEL, bool ACC, bool DIS, bool WIDE = false>
__global__ __launch_bounds__(TILE_THREADS, WIDE ? L2R_WIDE_WGS : 7)
void k_tile(SlabArgs kernarg_block, const TileRec *__restrict__ u_rec, const TileWin *__restrict__ u_tw, const TileStat *__restrict__ u_stat, const SlotRec *__restrict__ u_slot,
            uint32_t *__restrict__ u_xbase)
{
    constexpr int DIR_BYTES = FAST_DIR_BYTES;
    using WinT = typename std::conditional<WIDE, TileWin64, TileWin>::type;
    using EntT = typename std::conditional<WIDE, WEnt, v4i_t>::type;
    constexpr int AUX_BYTES = SLAB_DIR_BYTES + (int)sizeof(WinT);
    static_assert(AUX_BYTES >= SLAB_AUX_BYTES && WIDE_KEY_CAP == SLAB_KEY_CAP, "the WIDE instance's arrays hold what the plain one's do");
    __shared__ __attribute__((aligned(16))) uint32_t s_A[TILE_POS_CAP];
    __shared__ __attribute__((aligned(16))) uint16_t s_L[TILE_POS_CAP];
    __shared__ __attribute__((aligned(16))) EntT s_ent[2 * SLAB_KEY_CAP];
    __shared__ __attribute__((aligned(16))) uint8_t s_aux[AUX_BYTES];            // directories, then the window record
    __shared__ __attribute__((aligned(16))) uint8_t s_cnt[TILE_THREADS];        // exon counts, read order (255: that many or more)
    __shared__ __attribute__((aligned(16))) uint16_t s_loc[TILE_THREADS];       // ... and their exclusive scan
    __shared__ uint32_t s_flagw[TILE_THREADS / WAVE], s_redow[TILE_THREADS / WAVE];
    __shared__ uint32_t s_lb[4];
    __shared__ uint32_t s_chunk[2];
    // (in a tile that keeps the slab form the staged positions hold slab_walk_tile's words)
    uint8_t *const s_dir = s_aux;
    WinT &s_tw = *reinterpret_cast<WinT *>(s_aux + SLAB_DIR_BYTES);
    (void)kernarg_block;
    const SlabArgsK sa = slab_args();
    const PipeArgsK a = pipe_args();
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    uint32_t t;
    if (WIDE) { if (blockIdx.x >= sa->list_cnt[0]) return; t = sa->wide_list[blockIdx.x]; }
    else { t = fused_tile(blockIdx.x); if (t >= sa->n_tiles) return; }
    // diagnostics (L2R_STAMPS=1), wave 0: [0] records, CIGAR heads asked for, staging  [1] (count walk + barrier)  [6] scan, count
    // published, place walk  [2] window pass  [3] probe rounds  [4] verdicts  [7] the tile's first slot (exon counts in front)  [5] write-out
    SlabStamp stamp; stamp.start(a->f.stamps); if (stamp.who == 3) stamp.who = -1;
    // Issue priority: a wave is above the others everywhere but in its classification (window pass, probe rounds, verdicts: the part of a
    // tile that is bound by vector and LDS issue).  The waves that are asking for their records and CIGARs, walking, waiting for the
    // counts in front or writing out get their few instructions in first -- their round trips start earlier, the probing waves lose
    // nothing they could use (measured: k_tile 0.484 -> 0.471 ms; priority 3 instead of 1 the same within noise; high only up to the
    // end of the staging: 0.478).
    __builtin_amdgcn_s_setprio(TILE_PRIO);
    const uint32_t clk0 = stamp.p ? (uint32_t)__builtin_amdgcn_s_memrealtime() : 0u;
    // The thread's slot record lies at a place the tile number alone says (k_tile_index): asked for beside the scalar loads below.
    const v3u_a4 srec = *reinterpret_cast<const v3u_a4 *>(u_slot + ((size_t)t * TILE_THREADS + threadIdx.x));
    // The tile's record from the upload and its descriptor from k_describe_scan: every scalar load of the prologue leaves before the
    // first one is waited for (see k_probe_slab).
    const TileRec rec = u_rec[t];
    const TileDesc d0 = WIDE ? sa->tw64[t].d : u_tw[t].d;
    const TileStat tst = u_stat[t];
    const uint32_t chunk_on = sa->chunk_on; const int32_t ablate = a->f.p.ablate;
    const uint32_t n_tiles = sa->n_tiles;
    asm volatile("" :: "s"(tst.n_ops_n), "s"(tst.min_n), "s"(tst.max_d), "s"(tst.min_seg), "s"(chunk_on), "s"(ablate), "s"(n_tiles), "s"(rec.r0), "s"(rec.n_act), "s"(rec.sbase), "s"(rec.rows), "s"(rec.tid0), "s"(rec.lo),
                       "s"(d0.j_lo), "s"(d0.b_off), "s"(d0.nb), "s"(d0.b0), "s"(d0.nbk), "s"(d0.st_r0), "s"(d0.st_nk), "s"(d0.en_r0), "s"(d0.en_nk), "s"(d0.flags), "s"(d0.n_win));
    const uint32_t r0 = rec.r0, n_act = rec.n_act;
    const int32_t tid0 = rec.tid0, pos0 = rec.lo - 1;
    const int32_t tile_lo = rec.lo;                              // the base of the tile's row words: its first read's first base
    // a tile of the 64-bit-mask or the chunked kernel (on their lists since k_describe_scan): slab form, nothing is staged here
    const bool pre_slab = !WIDE && ((d0.flags & TD_WIDE) != 0u || (chunk_on && slab_tile_is_chunked(d0.flags)));
    // The tile is EXACT: no threshold is borderline in it, so a read's exon count is 1 + its N operations (the upload's read_n) and
    // the tile's count is known to the later tiles since k_describe_scan -- no count walk, nothing to publish.
    const bool counted = tile_exact(tst, a->f.p.min_exon, a->f.p.min_intron, a->f.p.max_delet) && !(ablate & 256);
    {   // (a wide tile that the WIDE instance takes whole, behind this launch: nothing of it happens in the plain one -- and the other way round)
        const bool direct = tile_wide_direct(sa->wide_direct_on, d0.flags, chunk_on, tst, n_act, a->f.p.min_exon, a->f.p.min_intron, a->f.p.max_delet, ablate);
        if (WIDE ? !direct : direct) return;
        // (... or k_tile_chunk, l2r_tchunk.hip.h: an exact tile of the chunked kernel)
        if (!WIDE && (d0.flags & TD_CDIRECT) != 0u) return;
    }
    TileDesc d = d0;
    if (pre_slab) d.flags = 0u;
    EntT *const s_ent0 = s_ent, *const s_ent1 = s_ent + SLAB_KEY_CAP;
    uint8_t *const s_dir0 = s_dir, *const s_dir1 = s_dir + DIR_BYTES, *const s_rdir = s_dir + 2 * DIR_BYTES;
    // A first look at the exon counts in front of the tile (waves 0 .. 2, one level each): asked for here, looked at further down.
    // PLAIN loads: what k_describe_scan wrote (the launch in front) is visible to them, and a word is complete only once -- a stale
    // copy of a word that a tile of this launch is still adding to merely looks incomplete, and the second look (agent-scope loads,
    // behind the probe rounds) takes over.
    LbLevel lv{nullptr, 0u, 0u};
    unsigned long long ev = 0ull;
    if (wv < 3 && !pre_slab) { lv = lb_level(sa, t, wv); if ((uint32_t)lane < lv.cnt) ev = lv.p[lane]; }
    // the dictionary slices and the window record: they travel while the CIGAR heads are asked for
    const DictRegs dv = load_dict_slices(a, d);
    int4 twv = make_int4(0, 0, 0, 0);
    if (WIDE) { if ((int)threadIdx.x < WIDE_TW_VECS) twv = reinterpret_cast<const int4 *>(sa->tw64 + t)[threadIdx.x]; }
    else if ((int)threadIdx.x < SLAB_TW_VECS && tw_vec_used((int)threadIdx.x, (d.flags & TD_FAST) ? d.n_win : 0u)) twv = reinterpret_cast<const int4 *>(u_tw + t)[threadIdx.x];
    // ---- the thread's slot record (asked for at the top of the kernel) and the head of its read's CIGAR: six 16-byte vectors, all in
    //      flight at once; words behind the last op become "I, length 0"
    const uint32_t c_lo = srec.x, xs = srec.z;
    const int32_t pos = (int32_t)srec.y;
    const bool active = (xs & SLOT_VALID) != 0u;
    const uint32_t n_cig = xs & 0xffu, idx = (xs >> SLOT_IDX_SHIFT) & 0xffu;         // (n_cig 255: that many or more)
    uint32_t cg[SLAB_HEAD];
#pragma unroll
    for (int i = 0; i < SLAB_HEAD; ++i) cg[i] = 1u;
    if (active) {
        const uint32_t *const words = a->f.cig + c_lo;
#pragma unroll
        for (int q = 0; q < SLAB_HEAD_VEC; ++q)
            if ((uint32_t)(4 * q) < n_cig) {
                const v4i_a4 x = *reinterpret_cast<const v4i_a4 *>(words + 4 * q);
                cg[4 * q] = (uint32_t)x.x; cg[4 * q + 1] = (uint32_t)x.y; cg[4 * q + 2] = (uint32_t)x.z; cg[4 * q + 3] = (uint32_t)x.w;
            }
    }
    // the first look: complete words only (a level of more than 64 words -- beyond 65 k tiles -- is left to the second look)
    bool first_look = false;                                     // (of this wave's level)
    uint32_t first_share = 0u;
    if (wv < 3 && !pre_slab) {
        const bool in = (uint32_t)lane < lv.cnt;
        first_look = __all(!in || (uint32_t)(ev >> LB_SHIFT) == lv.want) && lv.cnt <= (uint32_t)WAVE;
        first_share = wave_sum(in ? (uint32_t)ev : 0u);
    }
    // ---- window and dictionary slices into LDS, re-based to the tile's window (the CIGAR words travel)
    if (!pre_slab && (int)threadIdx.x < (WIDE ? WIDE_TW_VECS : SLAB_TW_VECS)) reinterpret_cast<int4 *>(&s_tw)[threadIdx.x] = twv;
    int my_wide = 0;
    if constexpr (WIDE) wide_stage_dict(d, dv, reinterpret_cast<const int *>(sa->tw64[t].win), s_ent0, s_ent1, s_dir0, s_dir1, s_rdir);
    else if (!pre_slab) my_wide = slab_stage_dict(d, dv, reinterpret_cast<const int *>(u_tw[t].win), SlabLds{nullptr, s_ent0, s_ent1, s_dir0, s_dir1, s_rdir});
    stamp.mark(0);
#pragma unroll
    for (int i = 0; i < SLAB_HEAD; ++i) cg[i] = (uint32_t)i < n_cig ? cg[i] : 1u;
    DevParams p;
    p.min_exon = a->f.p.min_exon; p.min_intron = a->f.p.min_intron; p.max_delet = a->f.p.max_delet;
    const uint32_t t3 = ((uint32_t)p.min_intron << 4) | 3u, t2 = ((uint32_t)(p.max_delet + 1) << 4) | 2u;      // op and length compare as one number
    const int c_max = wave_max(active ? (int)min(n_cig, (uint32_t)SLAB_HEAD) : 0);
    // ---- the reads' places among the tile's exons in read order: an exact tile has them in its records; else the first of two walks
    //      COUNTS (src/bam2gtf.c:31-78 with nothing stored), the waves meet and every wave scans the counts.  They also meet when the
    //      annotation has dictionary keys in several entries (rare: a site shared by transcripts more than 64 apart): whether this
    //      tile staged one is every wave's business.
    uint32_t n = 0u;
    const bool meet = !WIDE && !pre_slab && (!counted || sa->has_wide_keys != 0u);       // (else the waves meet behind the place walk: the staged slices must be whole before the probes)
    if (!pre_slab && !counted) {
        if (active) {
            int start = pos + 1, end = pos;
            bool first = true;
            auto step = [&](uint32_t c) {
                const uint32_t op = c & 0xfu;
                const int len = (int)(c >> 4);
                const bool cut = ((op == 3u) & (c >= t3)) | ((op == 2u) & (c >= t2));
                const bool keep = cut & (first | (end - start >= p.min_exon - 1));
                n += keep ? 1u : 0u;
                first = first & !keep;
                start = cut ? end + len + 1 : start;
                end += len & __builtin_amdgcn_sbfe(0x18d, op, 1u);         // ops 0 2 3 7 8 advance the reference
            };
#pragma unroll
            for (int q = 0; q < SLAB_HEAD_VEC; ++q)
                if (4 * q < c_max) { step(cg[4 * q]); step(cg[4 * q + 1]); step(cg[4 * q + 2]); step(cg[4 * q + 3]); }       // (wave-uniform)
            if (n_cig > (uint32_t)SLAB_HEAD) {
                const uint32_t n_ops = ld32(sa->cig_off32, r0 + idx + 1u) - c_lo;
                const uint32_t *const words = a->f.cig + c_lo;
                for (uint32_t i = SLAB_HEAD; i < n_ops; ++i) step(words[i]);
            }
            ++n;
        }
        s_cnt[idx] = (uint8_t)min(n, 255u);                 // (every entry is written: idx is a permutation of 0 .. 255)
    }
    if (meet) {
        // (what the staged positions cannot hold: a read of 255 exons or more -- bit 0; a dictionary key in several entries -- bit 1)
        const uint32_t fw = (__any(n >= 255u) ? 1u : 0u) | (__any(my_wide != 0) ? 2u : 0u);
        if (lane == 0) s_flagw[wv] = fw;
        __syncthreads();
    }
    stamp.mark(1);
    uint32_t total = 0u, loc = 0u;
    bool late_slab = false, wide_key = false;
    int any_wide = 0;
    if (!pre_slab) {
        if (counted) { total = n_act + (uint32_t)tst.n_ops_n; loc = (xs >> SLOT_LOC_SHIFT) & (SLOT_LOC_LIMIT - 1u); }
        else {
            // each wave scans the 256 counts (four per lane) for itself
            const uint32_t c4 = reinterpret_cast<const uint32_t *>(s_cnt)[lane];
            const uint32_t b0 = c4 & 0xffu, b1 = (c4 >> 8) & 0xffu, b2 = (c4 >> 16) & 0xffu, b3 = c4 >> 24;
            const uint32_t sum = b0 + b1 + b2 + b3;
            const uint32_t inc = wave_inclusive_scan(sum), ex = inc - sum;
            reinterpret_cast<uint2 *>(s_loc)[lane] = make_uint2(ex | ((ex + b0) << 16), (ex + b0 + b1) | ((ex + b0 + b1 + b2) << 16));     // (the four waves write the same values)
            total = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
            loc = s_loc[idx];
        }
        // (wave-uniform for the compiler too: a branch it takes for divergent keeps the CIGAR registers alive through the probe rounds)
        const uint32_t fl = meet ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(s_flagw[0] | s_flagw[1] | s_flagw[2] | s_flagw[3])) : 0u;
        // a dictionary key in several entries: k_probe_slab_chunked ORs them (the tile keeps the slab form); with chunked windows off
        // the tile's reads take the generic kernel
        late_slab = total > (uint32_t)TILE_POS_CAP || (fl & 1u) != 0u || ((fl & 2u) != 0u && chunk_on != 0u);
        wide_key = (fl & 2u) != 0u;
        any_wide = (wide_key && !late_slab) ? 1 : 0;
    }
    // (the staged form first in the source, the slab form behind it: see the note at the end of the kernel)
    auto staged_form = [&]() {
    // ---- the tile's exon count is known: published for every later tile's first slot
    // (L2R_ABLATE bit 15, tests: tile 3 never publishes its count -- the tiles behind it wait in vain, the engine falls back to the slab pipeline)
    if (threadIdx.x == 0 && !counted && !(ablate & 128) && !((ablate & 32768) && t == 3u)) lb_publish(sa, t, total);
    const uint32_t clk1 = stamp.p ? (uint32_t)__builtin_amdgcn_s_memrealtime() : 0u;
    // ---- second walk: PLACE the exons as row words at their positions in LDS
    const SlabStage st{s_A, s_L, loc, tile_lo, true};
    ReadEnds re{0, 0, 0, 0};
    bool sane = true, big = false;
    n = 0u;
    if (active) {
        uint32_t *const Ap = s_A + loc;
        int start = pos + 1, end = pos;
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
                Ap[n] = slab_pack(start - tile_lo, xlen);
                longest = max(longest, xlen);
                if (first) { s0 = start; e0 = end; }
                first = false; ++n;
            }
            start = cut ? end + len + 1 : start;
            end += len & __builtin_amdgcn_sbfe(0x18d, op, 1u);
        };
#pragma unroll
        for (int q = 0; q < SLAB_HEAD_VEC; ++q)
            if (4 * q < c_max) { step(cg[4 * q]); step(cg[4 * q + 1]); step(cg[4 * q + 2]); step(cg[4 * q + 3]); }       // (wave-uniform)
        if (n_cig > (uint32_t)SLAB_HEAD) {
            const uint32_t n_ops = ld32(sa->cig_off32, r0 + idx + 1u) - c_lo;
            const uint32_t *const words = a->f.cig + c_lo;
            for (uint32_t i = SLAB_HEAD; i < n_ops; ++i) step(words[i]);
        }
        {   const uint32_t xlen = (uint32_t)(end - start + 1);
            Ap[n] = slab_pack(start - tile_lo, xlen);
            longest = max(longest, xlen);
            // (starts rise along the read: the last one is the furthest)
            if ((uint32_t)(start - tile_lo) >= SLAB_REL_MASK) longest = 0xffffffffu; }
        if (first) { s0 = start; e0 = end; }
        ++n;
        // with min_exon >= 1 kept inner exons are at least one base long; the first and the last one are kept whatever their length
        sane = s0 <= e0 && start <= end;
        big = longest > SLAB_LEN_MAX;
        re = ReadEnds{s0, e0, start, end};
    }
    if (!meet) __syncthreads();                                  // (the dictionary slices and the window record are whole)
    stamp.mark(6);
    const uint32_t pre = idx | ((xs & SLOT_REV) ? PRE_REV : 0u) | (sane ? 0u : PRE_INSANE) | (n << PRE_N_SHIFT);
    const uint32_t r = r0 + idx;
    // ---- classification (a lane probes the positions it has placed itself; the dictionary slices were whole at the barrier above)
    __builtin_amdgcn_s_setprio(0);
    SlabVerdict vd;
    if constexpr (WIDE) vd = tile_classify_wide<LEVEL, DIS>(a, d, WideLds{s_ent0, s_ent1, s_dir0, s_dir1, s_rdir, s_tw.hk, s_tw.hx, s_tw.win}, s_tw.mask, active, pre, big, r, re, st);
    else vd = tile_classify<LEVEL, DIS>(a, d, SlabLds{nullptr, s_ent0, s_ent1, s_dir0, s_dir1, s_rdir}, s_tw.hk, s_tw.hx, s_tw.win, s_tw.mask, active, pre, big, r, re, st, any_wide, stamp);
    __builtin_amdgcn_s_setprio(TILE_PRIO);
    if (ACC) { const int w_redo = __any(vd.redo) ? 1 : 0; if (lane == 0) s_redow[wv] = (uint32_t)w_redo; }
    // ---- short-read junction support (-j: src/update_gtf.c:698-709 check_with_short_sj, :609-627 check_short_sj) for the reads whose
    //      verdict is final here, on the tile's LDS image: k_validate_sj's three steps without its passes over the results in HBM.
    //      (1) thread = read: is it a candidate (full, not known, has a known site), its cursor row (:613-614) and the Q7 test on it;
    //      every position of the tile learns its read; (2) the tile's exon POSITIONS across the threads: the table lookups of the
    //      candidates' novel junctions, spread over all lanes; (3) the read's verdict.  A read on the redo list gets its verdict from
    //      the generic kernel and its junction check from k_validate_sj behind that (which skips the reads checked here).
    if (a->f.p.n_sj > 0 && !(ablate & 1024)) {
        __syncthreads();                                        // (the dictionary slices, directories and window record are dead: what follows lives there)
        SjDir sd;
        sd.cur.key = sa->sj.cur.key; sd.cur.dir = sa->sj.cur.dir; sd.cur.kb_base = sa->sj.cur.kb_base; sd.cur.n_tid = sa->sj.cur.n_tid; sd.cur.n_tx = sa->sj.cur.n_tx;
        sd.ddir = sa->sj.ddir; sd.dbase = sa->sj.dbase; sd.d_ntid = sa->sj.d_ntid; sd.row = sa->sj.row;
        const int n_sj = a->f.p.n_sj;
        DevParams pj;
        pj.ss_dis = a->f.p.ss_dis; pj.use_multi = a->f.p.use_multi; pj.min_sj_cnt = a->f.p.min_sj_cnt; pj.n_sj = n_sj;
        const SjTid ti0 = sj_tid_rows(sd, tid0, n_sj);         // (a tile is of one chromosome)
        const bool cand = active && !vd.redo && (vd.info & (I_FULL | I_KNOWN | I_KSITE)) == (I_FULL | I_KSITE) && !(ablate & 512);
        // The table's rows whose donor lies in the tile's span (+- the tolerance), whole 512-bp buckets of the donor directory, and the
        // row in front of them: [rlo - 1, rhi).  Every lookup of the tile's reads ends inside them -- a junction's donor is a base of
        // its read -- and so does every read's cursor row unless it lies in front (the first row whose running maximum of the acceptors
        // is above the read's start has a donor in front of the read's start or is that row).  Up to SJ_STAGE rows are staged.
        const int dis_j = max(pj.ss_dis, 0);
        int rlo = ti0.end, rhi = ti0.end, c0 = ti0.end;
        if (ti0.nb > 0) {
            const int b_lo = max(tile_lo - dis_j, 0) >> SITE_SHIFT, b_hi = max((int)rec.pad[0] + dis_j, 0) >> SITE_SHIFT;
            c0 = (int)sd.ddir[ti0.db];
            if (b_lo < ti0.nb) rlo = min((int)sd.ddir[ti0.db + b_lo], ti0.end);
            if (b_hi + 1 < ti0.nb) rhi = min((int)sd.ddir[ti0.db + b_hi + 1], ti0.end);
        }
        const bool have_prev = rlo > c0;                        // (the chromosome has a row in front of the staged ones)
        const int m = rhi - rlo + 1;                            // staged entries: index 0 = the row in front (if any), 1 .. m - 1 = rows rlo .. rhi - 1
        // (-d < 0 and introns of no length -- -i < 1: a junction's acceptor may lie in front of its donor -- keep the literal scan from the cursor row)
        if (m <= SJ_STAGE && pj.ss_dis >= 0 && a->f.p.min_intron >= 1) {
            int *const s_don = reinterpret_cast<int *>(s_ent), *const s_acc = s_don + SJ_STAGE, *const s_pm = s_acc + SJ_STAGE;
            uint32_t *const s_okb = reinterpret_cast<uint32_t *>(s_aux);          // bit i: row i's read count reaches -J
            for (int i0 = 0; i0 < m; i0 += TILE_THREADS) {
                const int i = i0 + (int)threadIdx.x;
                bool okc = false;
                if (i < m && (i > 0 || have_prev)) {
                    const int4 q = sd.row[rlo - 1 + i];
                    s_don[i] = q.x; s_acc[i] = q.y; s_pm[i] = (int)(uint32_t)sd.cur.key[rlo - 1 + i];      // (low word of the key: the running maximum of the chromosome's acceptors)
                    okc = (pj.use_multi ? q.z + q.w : q.z) >= pj.min_sj_cnt;
                } else if (i == 0) { s_don[0] = INT32_MIN; s_acc[0] = INT32_MIN; s_pm[0] = INT32_MIN; }
                const unsigned long long mb = __ballot(okc);
                if (lane == 0) { s_okb[(i0 >> 5) + 2 * wv] = (uint32_t)mb; s_okb[(i0 >> 5) + 2 * wv + 1] = (uint32_t)(mb >> 32); }
            }
            __syncthreads();
            if (cand) {
                const int r_start = tile_lo + (int)(s_A[loc] & SLAB_REL_MASK);
                const int r_end = tile_lo + (int)(s_A[loc + n - 1u] & SLAB_REL_MASK) + (int)s_L[loc + n - 1u] - 1;
                // the cursor row (:613-614), clipped to the staged rows: first entry >= 1 whose running maximum is above the read's start
                int lo = 1, hi = m;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_pm[mid] > r_start) hi = mid; else lo = mid + 1; }
                const int fi = lo;                              // (m: the row behind the staged ones -- its donor lies behind every read of the tile)
                // Q7: the cursor row lies behind the read (or there is none) -> unsupported, no unreliable flag.  A cursor row in front of
                // the staged ones (the running maximum in front of entry 1 is above the start already) has its donor in front of the read;
                // one behind them (fi = m) has its donor behind every read of the tile.
                const bool exact = fi > 1 || s_pm[0] <= r_start;
                bool ok = !exact || (fi < m && s_don[fi] < r_end);
                if (ok) {
                    // The read's novel junctions as bits first (independent LDS reads), then one lookup per set bit: the lanes of a wave
                    // take their j-th novel junction together -- a loop over the exons with the lookup inside ran the lookup's two
                    // dependent searches for nearly every exon index (some lane of the 64 has a novel junction there): 7 us per tile.
                    unsigned long long nov = 0ull;
                    const uint32_t nj = n - 1u, n64 = min(nj, 64u);
                    for (uint32_t k = 0; k < n64; ++k) nov |= (unsigned long long)(((s_A[loc + k] >> SLAB_REL_BITS) & (uint32_t)F_JUNC) ? 1u : 0u) << k;
                    if (ablate & 2048) nov = 0ull;              // (timing diagnostics: no lookups)
                    uint32_t k_tail = 64u;                      // (a read of more than 65 exons: the ones behind, one by one)
                    while (nov != 0ull || k_tail < nj) {
                        uint32_t k;
                        if (nov != 0ull) { k = (uint32_t)__ffsll((long long)nov) - 1u; nov &= nov - 1ull; }
                        else { k = k_tail++; if (!((s_A[loc + k] >> SLAB_REL_BITS) & (uint32_t)F_JUNC)) continue; }
                        const uint32_t av = s_A[loc + k];
                        const int don = tile_lo + (int)(av & SLAB_REL_MASK) + (int)s_L[loc + k], acc = tile_lo + (int)(s_A[loc + k + 1u] & SLAB_REL_MASK) - 1;
                        // src/update_gtf.c:589-603 from the later of the cursor row and the first row with a donor >= don - dis
                        int l2 = fi, h2 = m;
                        const int want = don - dis_j;
                        while (l2 < h2) { const int mid = (l2 + h2) >> 1; if (s_don[mid] < want) l2 = mid + 1; else h2 = mid; }
                        bool sup = false;
                        for (int i = l2; i < m; ++i) {
                            const int dd = s_don[i];
                            if (dd >= acc || dd - don > dis_j) break;
                            if (__builtin_abs(dd - don) <= dis_j && __builtin_abs(s_acc[i] - acc) <= dis_j && ((s_okb[i >> 5] >> (i & 31)) & 1u)) { sup = true; break; }
                        }
                        if (!sup) { s_A[loc + k] = av | ((uint32_t)F_UNREL << SLAB_REL_BITS); ok = false; }
                    }
                }
                vd.info |= I_SJCHK | (ok ? I_SJPASS : I_UNREL);
                if (ok || a->f.p.split_trans) vd.info |= I_ACCEPT;
            }
        } else {
            // (more rows than the staging holds, or a negative -d: the lookups go to the table in HBM -- k_validate_sj's three steps)
            uint8_t *const s_owner = reinterpret_cast<uint8_t *>(s_ent);
            int *const s_sjfrom = reinterpret_cast<int *>(s_aux);
            uint8_t *const s_bad = s_aux + TILE_THREADS * 4;
            int from = -1;
            bool ok0 = false;
            if (cand) {
                // (the read's first start and last end from its staged exons: registers that need not live through the probe rounds)
                const int r_start = tile_lo + (int)(s_A[loc] & SLAB_REL_MASK);
                const int r_end = tile_lo + (int)(s_A[loc + n - 1u] & SLAB_REL_MASK) + (int)s_L[loc + n - 1u] - 1;
                from = cursor_value(sd.cur, tid0, r_start);        // (first row whose prefix-max key is above (tid, start))
                // Q7: cursor row beyond the read -> unsupported, no unreliable flag
                if (from < n_sj) ok0 = !(from >= ti0.end || sd.row[from].x >= r_end);
            }
            s_sjfrom[threadIdx.x] = (cand && ok0) ? from : -1;
            s_bad[threadIdx.x] = 0;
            if (active) for (uint32_t k = 0; k < n; ++k) s_owner[loc + k] = (uint8_t)threadIdx.x;
            __syncthreads();
            for (uint32_t q = threadIdx.x; q < total; q += (uint32_t)TILE_THREADS) {
                const uint32_t av = s_A[q];
                if (av == SLAB_POS_SKIP || !((av >> SLAB_REL_BITS) & F_JUNC)) continue;
                const uint32_t who = s_owner[q];
                const int fr = s_sjfrom[who];
                if (fr < 0) continue;
                // (a junction flag only stands on an exon that is not its read's last: position q + 1 is the same read's)
                const int don = tile_lo + (int)(av & SLAB_REL_MASK) + (int)s_L[q], acc = tile_lo + (int)(s_A[q + 1u] & SLAB_REL_MASK) - 1;
                if (!junction_supported(ti0, don, acc, fr, pj, sd)) { s_A[q] = av | ((uint32_t)F_UNREL << SLAB_REL_BITS); s_bad[who] = 1; }
            }
            __syncthreads();
            if (cand) {
                const bool ok = ok0 && s_bad[threadIdx.x] == 0;
                vd.info |= I_SJCHK | (ok ? I_SJPASS : I_UNREL);
                if (ok || a->f.p.split_trans) vd.info |= I_ACCEPT;
            }
        }
    }
    if (active) a->f.info[r] = vd.info;
    // ---- the tile's first result slot
    uint32_t clk2 = 0u;
    {
        uint32_t share = first_share;
        uint32_t n_polls = 0u;
        bool done;
        clk2 = stamp.p ? (uint32_t)__builtin_amdgcn_s_memrealtime() : 0u;
        // (a wave whose first look found every count of its level in front has nothing to do here: the rule -- a tile in front that is
        //  not exact publishes its count from this launch, and the waves that need it look again, with agent-scope loads)
        if (wv < 3 && !first_look && !(ablate & 64)) share = lb_share<false>(sa, t, wv, lane, done, n_polls);
        if (stamp.p && lane == 0 && wv < 3) { atomicAdd(&stamp.p[8192 + wv], (unsigned long long)n_polls); atomicAdd(&stamp.p[8192 + 3 + wv], n_polls ? 1ull : 0ull); }
        // (L2R_ABLATE bit 6, timing diagnostics only: no look at the counts in front -- the results land at made-up slots)
        if (ablate & 64) share = wv == 0 ? t * (uint32_t)TILE_POS_CAP : 0u;
        if (lane == 0 && wv < 3) s_lb[wv] = share;
    }
    __syncthreads();
    // (diagnostics, per tile, for l2r_debug_tile_times: the 100 MHz clock at the tile's start | the XCD's number, at its count's
    //  publication, at the begin and the end of its wait for the counts in front.  Stored here: a store in front of the prologue's scalar
    //  loads turns every one of them into a vector load.)
    if (stamp.p && threadIdx.x == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a->tile_total[t] = (clk0 << 3) | (xcc & 7u); a->f.tile_acc[t] = clk1; a->f.tile_acc_ex[t] = clk2;
        sa->tile_flags[t] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    }
    stamp.mark(7);
    const uint32_t xbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(s_lb[0] + s_lb[1] + s_lb[2]));
    if (threadIdx.x == 0) {
        u_xbase[t] = xbase;
        if (t + 1u == n_tiles) { u_xbase[n_tiles] = xbase + total; *sa->exon_total = xbase + total; }
    }
    const SlabOut out{a->f.ex_start, a->f.ex_end, a->f.ex_flag, xbase + loc};
    if (active) a->f.ex_off[r] = out.dst;
    if (active && big) {
        // an exon the row word cannot say: the literal walk (l2r_kernels.hip.h) straight into the result arrays (flags: the generic kernel's)
        const uint32_t n_ops = ld32(sa->cig_off32, r + 1u) - c_lo;
        const uint32_t *const words = a->f.cig + c_lo;
        WalkState w{pos + 1, pos, 0};
        auto put = [&](int k, int s_, int e_) { out.start[out.dst + (uint32_t)k] = s_; out.end[out.dst + (uint32_t)k] = e_; out.flag[out.dst + (uint32_t)k] = 0; };
        walk_ops<false>(w, words, 0, (int)n_ops, p, put);
        put(w.n, w.start, w.end);
    }
    if (!ACC) {
        slab_write_out(SlabOut{out.start, out.end, out.flag, xbase}, s_A, s_L, tile_lo, total);
        stamp.mark(5);
        return;
    }
    // ---- the tile's accepted chunk (k_probe_slab).  Not fused: the tile stays CHUNK_DEFERRED (k_describe_scan) for k_gather_accepted.
    const bool fused = __builtin_amdgcn_readfirstlane((int)(s_redow[0] | s_redow[1] | s_redow[2] | s_redow[3])) == 0 && !(ablate & 2);
    if (!fused) { slab_write_out(SlabOut{out.start, out.end, out.flag, xbase}, s_A, s_L, tile_lo, total); return; }
    uint32_t *const s_racc = reinterpret_cast<uint32_t *>(s_aux), *const s_rscan = s_racc + TILE_THREADS;
    uint16_t *const s_map = reinterpret_cast<uint16_t *>(s_ent);
    const bool acc = active && (vd.info & I_ACCEPT) != 0u;
    // accepted reads (high half) and their exons (low half), by read number inside the tile
    s_racc[idx] = acc ? ((1u << 16) | n) : 0u;                  // (every entry is written: idx is a permutation of 0 .. 255)
    __syncthreads();
    uint32_t ca, cx;
    {   // every wave scans the 256 words for itself (four per lane)
        const uint4 c4 = reinterpret_cast<const uint4 *>(s_racc)[lane];
        const uint32_t sum = c4.x + c4.y + c4.z + c4.w;
        const uint32_t inc = wave_inclusive_scan(sum), ex = inc - sum;
        reinterpret_cast<uint4 *>(s_rscan)[lane] = make_uint4(ex, ex + c4.x, ex + c4.x + c4.y, ex + c4.x + c4.y + c4.z);     // (the four waves write the same values)
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
        ca = tot >> 16; cx = tot & 0xffffu;
    }
    unsigned long long chunk = 0ull;                 // {first record slot, first exon slot} of the tile's chunk
    if (threadIdx.x == 0) {
        a->f.tile_acc[t] = 0u; a->f.tile_acc_ex[t] = 0u;            // nothing of this tile is left for k_gather_accepted
        if (ca) chunk = atomicAdd(a->f.chunk_cursor, ((unsigned long long)ca << 32) | cx);      // (the answer travels during the write-out below)
    }
    const uint32_t mine = acc ? s_rscan[idx] : 0u;                  // records / exons of the tile's accepted reads in front of this one
    if (acc) for (uint32_t k = 0; k < n; ++k) s_map[(mine & 0xffffu) + k] = (uint16_t)(loc + k);
    slab_write_out(SlabOut{out.start, out.end, out.flag, xbase}, s_A, s_L, tile_lo, total);
    if (threadIdx.x == 0) {
        s_chunk[0] = (uint32_t)chunk; s_chunk[1] = (uint32_t)(chunk >> 32);
        a->f.tile_chunk[t] = (uint32_t)chunk; a->f.tile_rchunk[t] = (uint32_t)(chunk >> 32);
    }
    if (ca == 0u) return;
    __syncthreads();
    const uint32_t to = s_chunk[0], to_r = s_chunk[1];
    if (acc) {                                      // the record of the thread's own read
        const uint32_t rslot = to_r + (mine >> 16);
        const uint64_t gidx = (uint64_t)(a->f.first_read + (int64_t)r);
        AccRec rc; rc.read_lo = (uint32_t)gidx; rc.read_hi = (uint32_t)(gidx >> 32); rc.info = vd.info; rc.ref_tx = vd.ref;
        a->f.acc_rec[rslot] = rc;
        a->f.acc_ex_off[rslot] = to + (mine & 0xffffu);
    }
    {   // the chunk's exons: thread j takes slots 4j .. 4j + 3 (16-byte stores at whatever alignment the chunk has)
        int32_t *const o_s = a->f.acc_start, *const o_e = a->f.acc_end; uint8_t *const o_f = a->f.acc_flag;
        for (uint32_t p4 = threadIdx.x * 4u; p4 < cx; p4 += (uint32_t)TILE_THREADS * 4u) {
            const uint2 m4 = *reinterpret_cast<const uint2 *>(s_map + p4);
            const uint32_t src[4] = {m4.x & 0xffffu, m4.x >> 16, m4.y & 0xffffu, m4.y >> 16};
            int sv[4], ev[4]; uint32_t fv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t q_ = p4 + (uint32_t)i < cx ? src[i] : 0u;      // (map entries behind the chunk are stale)
                const uint32_t av = s_A[q_]; const uint32_t lv = s_L[q_];
                sv[i] = tile_lo + (int)(av & SLAB_REL_MASK); ev[i] = sv[i] + (int)lv - 1; fv[i] = (av >> SLAB_REL_BITS) & 0xffu;
            }
            const uint32_t at_ = to + p4;
            if (p4 + 4u <= cx) {
                v4i_t s4, e4; s4.x = sv[0]; s4.y = sv[1]; s4.z = sv[2]; s4.w = sv[3]; e4.x = ev[0]; e4.y = ev[1]; e4.z = ev[2]; e4.w = ev[3];
                *reinterpret_cast<v4i_a4 *>(o_s + at_) = s4;
                *reinterpret_cast<v4i_a4 *>(o_e + at_) = e4;
                *reinterpret_cast<u32_a1 *>(o_f + at_) = fv[0] | (fv[1] << 8) | (fv[2] << 16) | (fv[3] << 24);
            } else {
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) if (p4 + i < cx) { o_s[at_ + i] = sv[i]; o_e[at_ + i] = ev[i]; o_f[at_ + i] = (uint8_t)fv[i]; }
            }
        }
    }

    };
    // (For its register allocation the compiler lays the two forms of a tile out one behind the other -- "the staged form, then, if a flag
    //  says so, the slab form" -- whatever the order here: what the slab form needs is alive through the staged form's probe rounds.)
    if (WIDE || !(pre_slab || late_slab)) { staged_form(); return; }
    if constexpr (!WIDE) {
    // ---- the tile keeps the slab form: k_walk_slab's body on the CIGAR registers (its LDS words behind the sort's arrays, which a
    //      slower wave may still be reading), then the tile's first slot and the list of the kernel that takes it
    // (the thread's slot once more, from a tile number the compiler cannot recognise: kept from above it would be spilled through the staged form)
    uint32_t t_again = t;
    asm volatile("" : "+s"(t_again));
    const uint32_t slot_s = (threadIdx.x + (tile_rot(t_again) << 6)) & (uint32_t)(TILE_THREADS - 1);
    const bool active_s = active;
    uint32_t *const w0 = s_A;
    const WalkLds W{w0, w0 + TILE_THREADS, reinterpret_cast<int *>(w0 + 2 * TILE_THREADS), w0 + 2 * TILE_THREADS + 4, w0 + 2 * TILE_THREADS + 8};
    // (the CIGAR heads are fetched once more: for the compiler this form FOLLOWS the staged one -- see below -- and words kept for it
    //  would be alive, i.e. spilled, through the probe rounds of every tile; these tiles are rare)
    uint32_t cgs[SLAB_HEAD];
#pragma unroll
    for (int i = 0; i < SLAB_HEAD; ++i) cgs[i] = 1u;
    if (active_s) {
        const uint32_t *const words = a->f.cig + c_lo;
#pragma unroll
        for (int q = 0; q < SLAB_HEAD_VEC; ++q)
            if ((uint32_t)(4 * q) < n_cig) {
                const v4i_a4 x = *reinterpret_cast<const v4i_a4 *>(words + 4 * q);
                cgs[4 * q] = (uint32_t)x.x; cgs[4 * q + 1] = (uint32_t)x.y; cgs[4 * q + 2] = (uint32_t)x.z; cgs[4 * q + 3] = (uint32_t)x.w;
            }
    }
#pragma unroll
    for (int i = 0; i < SLAB_HEAD; ++i) cgs[i] = (uint32_t)i < n_cig ? cgs[i] : 1u;
    const uint32_t tot = slab_walk_tile<false, false>(sa, a, t, r0, n_act, rec.sbase, rec.rows, tid0, pos0, slot_s, active_s, c_lo, pos,
                                                      n_cig | ((xs & SLOT_REV) ? 1u << 16 : 0u) | (idx << 24) /* k_walk_slab's record word */, cgs, W);
    if (threadIdx.x == 0 && !counted) lb_publish(sa, t, tot);
    uint32_t share = 0u;
    uint32_t n_polls_s = 0u;
    bool done_s;
    if (wv < 3) share = lb_share<false>(sa, t, wv, lane, done_s, n_polls_s);
    if (lane == 0 && wv < 3) s_lb[wv] = share;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t xbase = s_lb[0] + s_lb[1] + s_lb[2];
        u_xbase[t] = xbase;
        if (t + 1u == n_tiles) { u_xbase[n_tiles] = xbase + tot; *sa->exon_total = xbase + tot; }
        // the rows each wave of the probe kernels has to look at (k_describe_scan could not know them)
        sa->tw[t].pad[0] = W.wn[0] | (W.wn[1] << 8) | (W.wn[2] << 16) | (W.wn[3] << 24);
        // which kernel takes the tile: the 64-bit-mask and the chunked kernel have theirs on their lists (k_describe_scan); a key in
        // several entries makes it k_probe_slab_chunked's (as k_probe_slab decides), anything else k_probe_slab's
        if (!pre_slab) {
            if (wide_key && chunk_on) { sa->tw[t].d.flags = d0.flags | TD_CHUNK; chunk_list_append_late(sa, t); }
            else sa->fb_list[atomicAdd(sa->list_cnt + 4, 1u)] = t;
        } else if (d0.flags & TD_WIDE) sa->wide_list[n_tiles + 1u + atomicAdd(sa->list_cnt + 5, 1u)] = t;      // (k_probe_slab_wide's, behind the WIDE instance's tiles)
    }
    }

}

}  // namespace l2r
