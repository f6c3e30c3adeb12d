// l2r_tchunk.hip.h -- the one-kernel tile path for tiles whose window does not fit ANY mask width (gfx950): k_tile_chunk.
//
// A tile whose reads can meet more than 63 annotation transcripts (a locus with very many isoforms) was handed over in slab form
// (k_tile -> k_probe_slab_chunked, l2r_chunk.hip.h): per chunk of 63 window members that kernel re-stages the tile's dictionary
// slices (three dependent global round trips and seven barriers) and probes every exon again against the re-staged entries, although
// WHICH entries an exon meets does not depend on the chunk.  Here the tile stays in one workgroup from its CIGARs to its block of the
// result arrays, and what does not depend on the chunk is done once per tile:
//
//   once per tile   slot records, CIGAR heads, place walk: k_tile's (exact tiles only: a read's place comes from its slot record, the
//                   tile's first result slot from the words k_describe_scan wrote -- nothing of the plain instance is needed, the
//                   launch runs BESIDE it on a stream of its own);
//                   the KEYS {k1, k2} of the tile's dictionary slices and their bucket directories into LDS, every entry at its own
//                   place (no filter, no sort); the entries' masks stay in the REGISTERS of the threads that loaded them;
//                   the key lookup: per exon one word -- where the parts of its START key / END key lie among the staged entries
//                   (first part, number of parts, "the pair matches") -- kept beside the exon's row word (s_R); a staged key is one
//                   64-bit word, both dictionaries' lower bounds run in one loop, one round of reads behind them says the rest
//                   (tc_lookup2);
//                   the window's CHUNKS: stretches of 63 consecutive transcripts from the tile's cursor value on, the ones with a
//                   transcript that overlaps the tile's span listed up to the first transcript every read lies before (four waves, a
//                   stretch each per round trip).
//   per chunk       the 63 transcripts' headers (asked for one chunk ahead); every thread re-bases ITS entries' masks to the chunk
//                   (two shifts per mask, from registers) and leaves them at the entries' places; waves 0 and 1 group the 63 members
//                   by first / last exon (TcGroup); one barrier; the member pass (tc_visit: before / behind member by member, the
//                   full-length evidence of levels 1-4 once per distinct terminal exon); per exon the ORs over the parts its word
//                   names (no directory, no key compare) and the flag clears the chunk before left to do; the sweep's carried state
//                   exactly as in k_probe_slab_chunked (src/update_gtf.c:792-835).  A workgroup leaves the loop when none of its
//                   reads is still sweeping (known / stopped).
//
// A chunk here is 63 CONSECUTIVE transcripts (not 63 members): a transcript of the stretch that does not overlap the tile's span lies
// before every read or behind every read, which the member pass sees per read like for any other member (m_bef / m_aft), so masks are
// re-based by shifts alone (no gapped windows) and the chunk's window is its first transcript's number.
// Not taken here (they keep the slab form and k_probe_slab_chunked): tiles that are not exact under the run's thresholds, -d > 0,
// slices of more than TC_ST_CAP / TC_EN_CAP entries, and the tiles a one-window kernel flags late (a key in several entries).
// Junction support (-j) is left to k_validate_sj, the accepted chunk to k_gather_accepted (as for every list-driven kernel).
#pragma once
#include "l2r_tile.hip.h"

namespace l2r {

constexpr int TC_ENT_POOL = TC_ST_CAP + TC_EN_CAP;                // (the caps: l2r_slab.hip.h, beside tile_chunk_direct)
constexpr int TC_PER_THREAD = (TC_ENT_POOL + TILE_THREADS - 1) / TILE_THREADS;
constexpr int TC_MEMBERS_PER = WIDE_MEMBERS;                     // transcripts per chunk: 63, so that a "first member" still takes 6 bits (63 = none)
constexpr int TC_TRIPS = 256;                            // stretches of 63 transcripts the window scan may look at (16 k transcripts)
constexpr int TC_CHUNKS = 64;                            // ... and the ones with members a tile may have (4 k members)
constexpr int TC_DIR_N = DIR_CAP + 4;                    // directory words per dictionary (bucket b's first entry, two closing words)
static_assert(TC_ST_CAP <= 512 && TC_EN_CAP <= 512, "9-bit entry numbers in a lookup word");
// A position's lookup word (s_R): START half in bits 0-13, END half in bits 14-27 -- first part (9 bits) | parts (4 bits) << 9 | "the pair
// matches" << 13 --, the exon's novel flags still standing in bits 28-31 (F_EXON | F_DON | F_ACC | F_JUNC).
constexpr int TC_HALF_BITS = 14, TC_F_SHIFT = 28;
constexpr uint32_t TC_R_OVER = 1u << 13;                 // "the pair matches" without a part: a key of more than 15 parts (the read goes to the generic kernel)
constexpr uint32_t TC_ROW_FIRST = 1u << 30, TC_ROW_LAST = 1u << 31;      // row word, until the first chunk's work words: the exon begins / ends its read
constexpr uint32_t TC_HALF_MASK = (1u << TC_HALF_BITS) - 1u;

// A chunk's members by their FIRST exon and by their LAST exon (the full-length evidence of levels 1-4 compares the read's terminal
// exons with them): the isoforms of a locus share few terminal exons (measured: 6 + 8 distinct ones per 63 transcripts on cfg3_iso100,
// 10 + 11 on cfg3_iso40, 30 + 30 where a stretch spans several genes), so a read tests each distinct exon once and ORs the mask of
// the members that have it -- 7 vector instructions per group instead of 7 per member.  Wave 0 groups the staged headers per chunk
// (one ballot per distinct exon); more than TC_GROUPS of a kind: the member pass compares member by member as before.
constexpr int TC_GROUPS = 24;
struct TcGroup { int2 key; m64_t members; };               // 16 bytes; first exons at [0, TC_GROUPS), last exons behind them
struct TcMask { m64_t pm, sm; };                         // a staged entry's masks in the chunk's frame (bit j = transcript chunk base + j)
struct TcLds { const unsigned long long *key0, *key1; const TcMask *msk0, *msk1; const uint16_t *dir0, *dir1, *rdir; const int4 *hk, *hx; };

// The two lookups of one exon in the tile's key staging (once per tile): START key (start, end) and END key (end, next start).  Per
// key: the parts of the pair (k1, k2) if the dictionary has it, else the parts of the first pair with key 1 (their single masks cover
// every transcript with that site: build_dict walks a pair's parts until both member lists are through).  The entries of a bucket are
// sorted by (key 1, key 2), a pair's parts lie side by side.  The kernel is bound by its VECTOR INSTRUCTIONS (6.8 k per wave, four
// waves per SIMD), so a staged key is ONE 64-bit word key 1 << 32 | key 2 (coordinates are not negative: l2r_set_annotation) and a
// step of the lower bound one 64-bit compare; both dictionaries' searches run in one loop (two independent LDS reads in flight).
// Behind the lower bound l, ONE round of six reads per dictionary says the rest: the entries l .. l+2 (the pair and its parts, or --
// nothing in front with key 1 -- the first pair with key 1 behind it) and l-1 .. l-3 (the entries with key 1 and a smaller key 2: the
// first of them is the first pair with key 1).  Longer runs take a loop (rare).  A second search for the first entry with key 1 cost
// four more dependent steps for every wave in which one lane's pair was missing.
// Half word: first part (9 bits) | parts (4 bits) << 9 | "the pair matches" << 13; 0 = nothing.  over: more than 15 parts (the read
// goes to the generic kernel).
typedef unsigned long long tckey_t;
__device__ __forceinline__ tckey_t tc_key(int32_t k1, int32_t k2) { return ((tckey_t)(uint32_t)k1 << 32) | (tckey_t)(uint32_t)k2; }
__device__ __forceinline__ uint32_t tc_k1(tckey_t v) { return (uint32_t)(v >> 32); }
struct TcKey { const tckey_t *key; const uint16_t *dir; };
__device__ __forceinline__ uint32_t tc_half(const tckey_t *key, uint32_t lo, uint32_t hi, uint32_t l, tckey_t T, uint32_t zero_slot, bool &over)
{
    const uint32_t k1 = tc_k1(T);
    const int li = (int)l;                               // (three words in front of the staging and two behind it may be read: not used)
    const tckey_t f0 = key[li], f1 = key[li + 1], f2 = key[li + 2], g1 = key[li - 1], g2 = key[li - 2], g3 = key[li - 3];
    const bool at = l < hi && tc_k1(f0) == k1, pair = at && f0 == T;
    const bool p1 = at && l + 1u < hi && f1 == f0, p2 = p1 && l + 2u < hi && f2 == f0;
    const bool b1 = !pair && l > lo && tc_k1(g1) == k1, b2 = b1 && l - 1u > lo && tc_k1(g2) == k1, b3 = b2 && l - 2u > lo && tc_k1(g3) == k1;
    const bool q1 = b3 ? g2 == g3 : (b2 && g1 == g2), q2 = b3 && q1 && g1 == g3;           // the run of the first entry in front
    uint32_t base = b1 ? l - (1u + (b2 ? 1u : 0u) + (b3 ? 1u : 0u)) : l;
    uint32_t cnt = b1 ? 1u + (q1 ? 1u : 0u) + (q2 ? 1u : 0u) : 1u + (p1 ? 1u : 0u) + (p2 ? 1u : 0u);
    const bool found = b1 || at;
    const bool slow = b3 || (!b1 && p2);                 // maybe more in front / more parts
    if (__any(slow)) {
        if (slow) {
            while (b1 && base > lo && tc_k1(key[(int)base - 1]) == k1) --base;
            const tckey_t kb = key[base];
            cnt = 1u;
            while (base + cnt < hi && key[base + cnt] == kb) ++cnt;
            if (cnt > 15u) { over = true; cnt = 15u; }
        }
    }
    return found ? base | (cnt << 9) | (pair ? 1u << 13 : 0u) : zero_slot;             // (nothing: the slot behind the entries, whose masks are 0; 0 parts)
}
__device__ __forceinline__ uint32_t tc_lookup2(const TcKey &K0, const TcKey &K1, int b_off, uint32_t none, uint32_t zero0, uint32_t zero1, bool on0, bool on1, int32_t s, int32_t e, int32_t s2, bool &over)
{
    const uint32_t ib0 = on0 ? min((uint32_t)((s >> SITE_SHIFT) + b_off), none) : none, ib1 = on1 ? min((uint32_t)((e >> SITE_SHIFT) + b_off), none) : none;
    const uint32_t lo0 = K0.dir[ib0], hi0 = K0.dir[ib0 + 1u], lo1 = K1.dir[ib1], hi1 = K1.dir[ib1 + 1u];
    const tckey_t T0 = tc_key(s, e), T1 = tc_key(e, s2);
    uint32_t l0 = lo0, h0 = hi0, l1 = lo1, h1 = hi1;
    while (l0 < h0 || l1 < h1) {
        const uint32_t m0 = (l0 + h0) >> 1, m1 = (l1 + h1) >> 1;
        const bool less0 = K0.key[m0] < T0, less1 = K1.key[m1] < T1;           // (an empty range reads an entry of the staging or the word behind it: not used)
        if (l0 < h0) { l0 = less0 ? m0 + 1u : l0; h0 = less0 ? h0 : m0; }
        if (l1 < h1) { l1 = less1 ? m1 + 1u : l1; h1 = less1 ? h1 : m1; }
    }
    return tc_half(K0.key, lo0, hi0, l0, T0, zero0, over) | (tc_half(K1.key, lo1, hi1, l1, T1, zero1, over) << TC_HALF_BITS);
}

// overlapping_exon_members64 (l2r_wide.hip.h) on the split key / mask arrays
__device__ __forceinline__ m64_t tc_overlapping_exon_members(const TcLds &L, int b_off, int nb, int s, int e)
{
    const int bs = s >> SITE_SHIFT;
    if (bs >= nb) return 0ull;
    const int be = min(e >> SITE_SHIFT, nb - 1);
    m64_t m = 0ull;
    const uint32_t i1 = L.dir0[be + b_off + 1];
    for (uint32_t i = L.rdir[bs + b_off]; i < i1; ++i) {
        const unsigned long long q = L.key0[i];
        if ((int)(uint32_t)(q >> 32) <= e && (int)(uint32_t)q >= s) m |= L.msk0[i].pm;
    }
    return m;
}

// visit_chunk64 (l2r_chunk.hip.h) for a chunk of 63 transcripts whose headers are all staged (a transcript behind the chunk's last or
// behind the annotation: "behind every read").  The member pass is bound by vector issue (four waves per SIMD walk the same 63 headers):
// members 0 .. 31 and 32 .. 62 are collected in two 32-bit words, highest member first, each predicate by ONE compare and ONE
// add-with-carry (m = m + m + predicate) -- 10 vector instructions per member at level 3 where the 64-bit select-and-or form takes 24.
template <int LEVEL>
__device__ __forceinline__ ChunkVisit tc_visit(const int2 *s_se, const int4 *s_hx, const TcGroup *s_grp, uint32_t n_grp /* first | last << 8; 0xffff: member by member */,
                                              int w_n, bool work, uint32_t n, const ReadEnds &re, const m64_t *tilemask)
{
    ChunkVisit m{0ull, 0ull, 0ull, 0ull, false, false};
    if (!__any(work) || w_n <= 0) return m;
    uint32_t aft[2] = {0u, 0u}, bef[2] = {0u, 0u}, lm[2] = {0u, 0u}, rm[2] = {0u, 0u};
    // A few members' headers are asked for together: left to itself the compiler puts every
    // LDS read right in front of its use and waits for it there -- two exposed LDS latencies per member (measured: 200 cycles per member
    // and wave, all of the pass).  The empty asm statements name the loaded registers: the loads cannot sink below them.
    constexpr bool LR = LEVEL >= 1 && LEVEL <= 4;
    constexpr int MB = 3;                                                     // members per block (4: 128 registers and 5 spilled; the 63 members of a chunk are 21 blocks of 3)
    auto ask = [&](int j_top, int2 (&se)[MB], int4 (&hx)[MB]) {              // members j_top, j_top - 1, .. (clipped to 0: read, not used)
#pragma unroll
        for (int u = 0; u < MB; ++u) { const int j = max(j_top - u, 0); se[u] = s_se[j]; if (LR) hx[u] = s_hx[j]; }
    };
    auto take = [&](int half, int j_top, int j_lo, const int2 (&se)[MB], const int4 (&hx)[MB]) {
#pragma unroll
        for (int u = 0; u < MB; ++u) {
            if (j_top - u < j_lo) break;                                     // (wave-uniform)
            shift_in_le(aft[half], re.el, se[u].x);                          // comp_trans <= (Q5): the read lies before the member
            shift_in_le(bef[half], se[u].y, re.s0);                          // the member lies before the read
            if (LEVEL == 1) { shift_in_eq(lm[half], re.e0, hx[u].y); shift_in_eq(rm[half], re.sl, hx[u].z); }
            else if (LR) {
                shift_in_le2(lm[half], re.s0, hx[u].y, hx[u].x, re.e0);      // closed_overlap(s0, e0, hx.x, hx.y)
                if (LEVEL != 4) shift_in_le2(rm[half], re.sl, hx[u].w, hx[u].z, re.el);
            }
        }
    };
    const bool grouped = LR && n_grp != 0xffffu;                              // (wave-uniform)
    if (grouped) {
        // before / behind member by member (the staged {start, end} pairs alone; the next block's pairs are in flight), the terminal
        // exons group by group (a group's 16 bytes in two reads, four groups asked for at once; `&`, not `&&`: the compiler turns a
        // short-circuit into a branch around the second key's LDS read)
#pragma unroll
        for (int half = 1; half >= 0; --half) {
            const int j_hi = half ? TC_MEMBERS_PER - 1 : 31, j_lo = half ? 32 : 0;
            int2 se[MB], sn[MB];
#pragma unroll
            for (int u = 0; u < MB; ++u) se[u] = s_se[max(j_hi - u, 0)];
            for (int j = j_hi; j >= j_lo; j -= MB) {
#pragma unroll
                for (int u = 0; u < MB; ++u) sn[u] = s_se[max(j - MB - u, 0)];
                asm volatile("" :: "v"(se[0].x), "v"(se[1].x), "v"(se[2].x), "v"(sn[0].x), "v"(sn[1].x), "v"(sn[2].x));
#pragma unroll
                for (int u = 0; u < MB; ++u) {
                    if (j - u < j_lo) break;
                    shift_in_le(aft[half], re.el, se[u].x);
                    shift_in_le(bef[half], se[u].y, re.s0);
                }
#pragma unroll
                for (int u = 0; u < MB; ++u) se[u] = sn[u];
            }
        }
        const int n1 = (int)(n_grp & 0xffu), n2 = LEVEL == 4 ? 0 : (int)(n_grp >> 8);
        constexpr int GB = 4;
        auto groups = [&](const TcGroup *grp, bool first_kind, int cnt, int qa, int qb, uint32_t (&acc)[2]) {      // the read's terminal exon [qa, qb]
            for (int g0 = 0; g0 < cnt; g0 += GB) {
                int2 k[GB]; uint2 mm[GB];
#pragma unroll
                for (int u = 0; u < GB; ++u) { const TcGroup *q = grp + min(g0 + u, TC_GROUPS - 1); k[u] = q->key; mm[u] = *reinterpret_cast<const uint2 *>(&q->members); }
                asm volatile("" :: "v"(k[0].x), "v"(k[1].x), "v"(k[2].x), "v"(k[3].x), "v"(mm[0].x), "v"(mm[1].x), "v"(mm[2].x), "v"(mm[3].x));
#pragma unroll
                for (int u = 0; u < GB; ++u) {
                    const bool c = (g0 + u < cnt) & (LEVEL == 1 ? (first_kind ? qb == k[u].y : qa == k[u].x) : ((qa <= k[u].y) & (k[u].x <= qb)));
                    acc[0] |= c ? mm[u].x : 0u; acc[1] |= c ? mm[u].y : 0u;
                }
            }
        };
        groups(s_grp, true, n1, re.s0, re.e0, lm);
        groups(s_grp + TC_GROUPS, false, n2, re.sl, re.el, rm);
    } else {
#pragma unroll
    for (int half = 1; half >= 0; --half) {
        const int j_hi = half ? TC_MEMBERS_PER - 1 : 31, j_lo = half ? 32 : 0;
        int2 seA[MB]; int4 hxA[MB];
#pragma unroll
        for (int u = 0; u < MB; ++u) hxA[u] = make_int4(0, 0, 0, 0);
        for (int j = j_hi; j >= j_lo; j -= MB) {
            ask(j, seA, hxA);
            asm volatile("" :: "v"(seA[0].x), "v"(seA[1].x), "v"(seA[2].x), "v"(hxA[0].x), "v"(hxA[1].x), "v"(hxA[2].x));
            take(half, j, j_lo, seA, hxA);
        }
    }
    }
    const m64_t m_aft = ((m64_t)aft[1] << 32) | aft[0], m_bef = ((m64_t)bef[1] << 32) | bef[0];
    m.lmask = ((m64_t)lm[1] << 32) | lm[0]; m.rmask = ((m64_t)rm[1] << 32) | rm[0];
    const m64_t keep = w_n >= 64 ? ~0ull : ((1ull << w_n) - 1ull);
    const m64_t stop = m_aft & keep;
    const m64_t below = (stop & (0ull - stop)) - 1ull;                       // all ones when no member ends the sweep
    m.vpre = work ? (~m_bef & below & keep) : 0ull;
    m.stopped = work && stop != 0ull;
    m.lmask &= m.vpre; m.rmask &= m.vpre;
    const m64_t single = tilemask[0];
    if (n == 1) {
        m64_t c = m.vpre & single;
        while (c) {
            const int j = __ffsll((long long)c) - 1;
            c &= c - 1ull;
            const int4 hx = s_hx[j];
            if (overlap_frac(re.s0, re.e0, hx.x, hx.y) >= fast_args()->p.frac) m.k1mask |= 1ull << j;
        }
    } else if (m.vpre & tilemask[1] & ~single) m.redo = true;
    return m;
}

// map_exons_lds64 (l2r_chunk.hip.h) with the lookups done: per exon the ORs over the parts its word names.  The first two parts of both
// keys are read without a loop (four independent 16-byte reads in flight; a part that is not there is read and not used), the next
// exon's word is asked for before this exon's masks are looked at.
// pend: the read swept the chunk before this one and goes on (not known there): that chunk's work words, still in the row words, clear
// the novel flags of the exons / junctions / sites one of its visited members had -- here, where both words pass by anyway (a loop
// of its own behind the carried state cost 4% of the kernel).
__device__ __forceinline__ uint32_t tc_flag_clears(uint32_t w, uint32_t lim, bool sites)
{
    uint32_t clr = ((w & 63u) <= lim ? (uint32_t)F_EXON : 0u) | (((w >> 6) & 63u) <= lim ? (uint32_t)F_JUNC : 0u);
    if (sites) clr |= (((w >> 12) & 1u) ? (uint32_t)F_DON : 0u) | (((w >> 13) & 1u) ? (uint32_t)F_ACC : 0u);
    return clr << TC_F_SHIFT;
}
__device__ __forceinline__ SiteMasks64 tc_map_exons(const TcLds &L, bool mapping, bool pend, uint32_t *Ap, uint32_t *Rp, uint32_t n, m64_t vpre, uint32_t zero0, uint32_t zero1)
{
    SiteMasks64 m{~0ull, 0ull, 0ull, 0ull, 0ull};
    const int k_max = wave_max(mapping ? (int)n : 0);
    const uint32_t nm1 = mapping ? n - 1u : 0u;
    uint32_t R = mapping ? Rp[0] : 0u;
    for (int k = 0; k < k_max; ++k) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        // (a key without an entry, an exon the lane does not have: the slot behind the dictionary's entries, whose masks are 0 -- and a
        // second part that is not there reads that slot too: ORs without selects)
        const uint32_t R0 = live ? R & TC_HALF_MASK : zero0, R1 = junc ? (R >> TC_HALF_BITS) & TC_HALF_MASK : zero1;
        const uint32_t i0 = R0 & 511u, c0 = (R0 >> 9) & 15u, i1 = R1 & 511u, c1 = (R1 >> 9) & 15u;
        const TcMask a0 = L.msk0[i0], a1 = L.msk0[c0 > 1u ? i0 + 1u : zero0], b0 = L.msk1[i1], b1 = L.msk1[c1 > 1u ? i1 + 1u : zero1];
        const uint32_t Rn = mapping ? Rp[min((uint32_t)k + 1u, nm1)] : 0u;
        m64_t xm = a0.pm | a1.pm, am = a0.sm | a1.sm, jm = b0.pm | b1.pm, dm = b0.sm | b1.sm;
        if (__any(c0 > 2u || c1 > 2u)) {
            for (uint32_t c = 2u; c < c0; ++c) { const TcMask q = L.msk0[i0 + c]; xm |= q.pm; am |= q.sm; }
            for (uint32_t c = 2u; c < c1; ++c) { const TcMask q = L.msk1[i1 + c]; jm |= q.pm; dm |= q.sm; }
        }
        if (!((R0 >> 13) & 1u)) xm = 0ull;
        if (!((R1 >> 13) & 1u)) jm = 0ull;
        const m64_t amj = junc ? am : 0ull;
        uint32_t word = first_member64(xm & vpre);
        word |= first_member64(jm & vpre) << 6;
        word |= ((dm & vpre) ? 1u : 0u) << 12;
        word |= ((amj & vpre) ? 1u : 0u) << 13;
        m.kand &= junc ? (am & dm) : ~0ull;            // Q1: the acceptor probed with exon k is ITS OWN start, k < n-1
        m.kor |= amj | dm;
        if (k == 0) m.dm_first = dm;
        m.am_last = (live && !junc) ? am : m.am_last;
        if (live) {
            const uint32_t A = Ap[k];
            if (pend) { const uint32_t clr = tc_flag_clears(A >> SLAB_REL_BITS, 62u, true); if (R & clr) Rp[k] = R & ~clr; }      // (63 = no member)
            Ap[k] = (A & SLAB_REL_MASK) | (word << SLAB_REL_BITS);
        }
        R = Rn;
    }
    return m;
}

constexpr int TC_LDS_BYTES = TILE_POS_CAP * (4 + 2 + 4) + (TC_ST_CAP + 6) * 8 + (TC_ENT_POOL + 2) * 16 + 3 * TC_DIR_N * 2 + 2 * 64 * 16 + 64 * 8 + TC_TRIPS + TC_CHUNKS * 2 + 64;
static_assert(TC_LDS_BYTES <= 40960, "k_tile_chunk: 4 workgroups per CU need 80 allocation granules of 512 bytes at most");

template <int LEVEL>
__global__ __launch_bounds__(TILE_THREADS, 4)
void k_tile_chunk(SlabArgs kernarg_block, const TileRec *__restrict__ u_rec, const TileWin *__restrict__ u_tw, const TileStat *__restrict__ u_stat, const SlotRec *__restrict__ u_slot,
                  uint32_t *__restrict__ u_xbase, uint32_t late /* 1: the launch behind the list kernels, over the tiles they handed on late */)
{
    constexpr uint32_t F_ALL = (uint32_t)(F_EXON | F_DON | F_ACC | F_JUNC);
    __shared__ __attribute__((aligned(16))) uint32_t s_A[TILE_POS_CAP];          // row words: start - base | the chunk's work word << 18 (at the end: | flag byte << 18)
    __shared__ __attribute__((aligned(16))) uint16_t s_L[TILE_POS_CAP];          // lengths
    __shared__ __attribute__((aligned(16))) uint32_t s_R[TILE_POS_CAP];          // lookup words (tc_lookup): START | END << 14 | novel flags still standing << 28
    __shared__ __attribute__((aligned(16))) tckey_t s_key0_[TC_ST_CAP + 6];         // START keys, key 1 << 32 | key 2 (the full-length evidence asks them in every chunk); four words in front, two behind: tc_half
    // the entries' masks in the chunk's frame, START entries first; until the first chunk the END entries' KEYS live in the END part (the
    // lookups are through before a mask is written); two entries more: a part that is not there may be read
    __shared__ __attribute__((aligned(16))) TcMask s_msk[TC_ENT_POOL + 2];
    __shared__ __attribute__((aligned(16))) uint16_t s_dir[3 * TC_DIR_N];
    __shared__ __attribute__((aligned(16))) int4 s_hk[64], s_hx[64];
    __shared__ __attribute__((aligned(16))) int2 s_se[64];                        // the members' {start, end} once more (the member pass is bound by its LDS reads)
    __shared__ __attribute__((aligned(16))) uint8_t s_trip[TC_TRIPS];                                         // per stretch: 1 has a member | 2 the sweeps end in it
    __shared__ uint16_t s_chunk[TC_CHUNKS];
    __shared__ m64_t s_mask[2];
    __shared__ uint32_t s_lb[4], s_nchunk, s_bad, s_ngrp[2];
    tckey_t *const s_key0 = s_key0_ + 4, *const s_key1 = reinterpret_cast<tckey_t *>(s_msk + TC_ST_CAP);
    (void)kernarg_block;
    const SlabArgsK sa = slab_args();
    const PipeArgsK a = pipe_args();
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const uint32_t n_tiles = sa->n_tiles;
    if (blockIdx.x >= sa->list_cnt[late ? 8 : 1]) return;
    const uint32_t t = sa->chunk_list[late ? n_tiles + 1u + blockIdx.x : blockIdx.x];
    if (t >= n_tiles) return;
    // diagnostics (L2R_STAMPS=1), wave 0: [0] records, loads asked for, window scan  [1] place walk + barrier  [2] chunk list, key lookups + barrier
    // [3] per chunk: headers + masks staged, barrier  [4] member pass  [5] mask ORs  [6] carried state + barrier  [7] verdicts, first slot, write-out
    SlabStamp stamp; stamp.start(a->f.stamps); if (stamp.who == 3) stamp.who = -1;
    __builtin_amdgcn_s_setprio(TILE_PRIO);
    const v3u_a4 srec = *reinterpret_cast<const v3u_a4 *>(u_slot + ((size_t)t * TILE_THREADS + threadIdx.x));
    const TileRec rec = u_rec[t];
    const TileDesc d = u_tw[t].d;
    const TileStat tst = u_stat[t];
    const uint32_t flags0 = sa->tile_flags[t];
    const uint32_t chunk_on = sa->chunk_on; const int32_t ablate = a->f.p.ablate;
    const int32_t n_tx = a->f.p.n_tx;
    const uint32_t r0 = rec.r0, n_act = rec.n_act;
    const int32_t tid0 = rec.tid0, tile_lo = rec.lo, tile_hi = (int32_t)rec.pad[0];
    // (the tiles of the first launch carry k_describe_scan's verdict as a bit of their flags; the late ones are tested here)
    if (late ? !tile_chunk_direct(sa->chunk_direct_on, flags0, chunk_on, d, tst, n_act, a->f.p.min_exon, a->f.p.min_intron, a->f.p.max_delet, a->f.p.ss_dis, ablate, true)
             : (flags0 & TD_CDIRECT) == 0u) {
        if (threadIdx.x == 0) atomicAdd(sa->list_cnt + (late ? 10 : 9), 1u);            // (left to k_probe_slab_chunked: the host skips that launch while nobody counts here)
        return;
    }
    // ---- the counts in front of the tile (waves 0 .. 2, one level each): plain loads, complete unless a tile in front is not exact
    LbLevel lv{nullptr, 0u, 0u};
    unsigned long long ev = 0ull;
    if (wv < 3) { lv = lb_level(sa, t, wv); if ((uint32_t)lane < lv.cnt) ev = lv.p[lane]; }
    // ---- the thread's slot record and the head of its read's CIGAR
    const uint32_t c_lo = srec.x, xs = srec.z;
    const int32_t pos = (int32_t)srec.y;
    const bool active = (xs & SLOT_VALID) != 0u;
    const uint32_t n_cig = xs & 0xffu, idx = (xs >> SLOT_IDX_SHIFT) & 0xffu;
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
    // ---- the thread's dictionary entries (entry e = thread + 256 u: START entries first, then END): keys -> LDS, masks stay here
    const uint32_t n_ent = d.st_nk + d.en_nk;
    int32_t e_base[TC_PER_THREAD]; m64_t e_pm[TC_PER_THREAD], e_sm[TC_PER_THREAD];
#pragma unroll
    for (int u = 0; u < TC_PER_THREAD; ++u) {
        const uint32_t e = threadIdx.x + (uint32_t)(u * TILE_THREADS);
        e_base[u] = INT32_MIN / 2; e_pm[u] = 0ull; e_sm[u] = 0ull;
        if (e < n_ent) {
            const int4 *const qv = reinterpret_cast<const int4 *>(e < d.st_nk ? a->f.st.ent + d.st_r0 + e : a->f.en.ent + d.en_r0 + (e - d.st_nk));
            const int4 xa = qv[0], xb = qv[1];
            if (e < d.st_nk) s_key0[e] = tc_key(xa.x, xa.y); else s_key1[e - d.st_nk] = tc_key(xa.x, xa.y);
            e_base[u] = xa.z;
            e_pm[u] = ((m64_t)(uint32_t)xb.y << 32) | (uint32_t)xb.x; e_sm[u] = ((m64_t)(uint32_t)xb.w << 32) | (uint32_t)xb.z;
        }
    }
    uint16_t *const s_dir0 = s_dir, *const s_dir1 = s_dir + TC_DIR_N, *const s_rdir = s_dir + 2 * TC_DIR_N;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        const int i = (int)threadIdx.x + qq * TILE_THREADS;
        if (i <= d.nbk) {
            const uint32_t b = (uint32_t)(d.b0 + i);
            s_dir0[i] = (uint16_t)(ld32(a->f.st.dir, b) - d.st_r0); s_dir1[i] = (uint16_t)(ld32(a->f.en.dir, b) - d.en_r0);
            s_rdir[i] = (uint16_t)(ld32(a->f.st.rdir, b) - d.st_r0);
        }
    }
    if (threadIdx.x >= 1u && threadIdx.x < 3u) { s_dir0[d.nbk + (int)threadIdx.x] = (uint16_t)d.st_nk; s_dir1[d.nbk + (int)threadIdx.x] = (uint16_t)d.en_nk; }
    s_trip[wv + 4 * lane] = 2u;                                                   // (the stretches this wave may look at; one it leaves out: the sweeps' end)
    if (threadIdx.x == 0) s_bad = 0u;
    // ---- the window's stretches: wave w looks at stretches w, w + 4, ...; every one up to the first that ends the sweeps gets its word
    {
        const TxHdr *const hdr = a->f.hdr;
        for (int trip = wv; trip < TC_TRIPS; trip += TILE_THREADS / WAVE) {
            const int base = d.j_lo + trip * TC_MEMBERS_PER;
            const int j = base + lane;
            bool ov = false, aft = false;
            if (lane < TC_MEMBERS_PER && j < n_tx) {
                const int4 h0 = *reinterpret_cast<const int4 *>(hdr + j);                 // {tid, start, end, .}
                aft = tid0 < h0.x || (tid0 == h0.x && tile_hi <= h0.y);                   // comp_trans <= (Q5)
                const bool bef = h0.x < tid0 || (h0.x == tid0 && h0.z <= tile_lo && h0.y < tile_lo);
                ov = !aft && !bef;
            }
            const unsigned long long ma = __ballot(aft);
            const int stop = ma ? __ffsll((long long)ma) - 1 : WAVE;
            const unsigned long long mo = __ballot(ov) & (stop < WAVE ? (1ull << stop) - 1ull : ~0ull);
            const bool last = ma != 0ull || base + TC_MEMBERS_PER >= n_tx;
            if (lane == 0) s_trip[trip] = (uint8_t)((mo ? 1u : 0u) | (last ? 2u : 0u));
            if (last) break;
        }
    }
    stamp.mark(0);
#pragma unroll
    for (int i = 0; i < SLAB_HEAD; ++i) cg[i] = (uint32_t)i < n_cig ? cg[i] : 1u;
    DevParams p;
    p.min_exon = a->f.p.min_exon; p.min_intron = a->f.p.min_intron; p.max_delet = a->f.p.max_delet;
    const uint32_t t3 = ((uint32_t)p.min_intron << 4) | 3u, t2 = ((uint32_t)(p.max_delet + 1) << 4) | 2u;
    const int c_max = wave_max(active ? (int)min(n_cig, (uint32_t)SLAB_HEAD) : 0);
    const uint32_t total = n_act + (uint32_t)tst.n_ops_n;
    const uint32_t loc = (xs >> SLOT_LOC_SHIFT) & (SLOT_LOC_LIMIT - 1u);
    // first look at the counts in front
    bool first_look = false; uint32_t first_share = 0u;
    if (wv < 3) {
        const bool in = (uint32_t)lane < lv.cnt;
        first_look = __all(!in || (uint32_t)(ev >> LB_SHIFT) == lv.want) && lv.cnt <= (uint32_t)WAVE;
        first_share = wave_sum(in ? (uint32_t)ev : 0u);
    }
    // ---- the place walk (k_tile's): the read's exons as row words at their read-order positions
    uint32_t *const Ap = s_A + loc; uint16_t *const Lp = s_L + loc; uint32_t *const Rp = s_R + loc;
    ReadEnds re{0, 0, 0, 0};
    bool sane = true, big = false;
    uint32_t n = 0u;
    if (active) {
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
                Ap[n] = (uint32_t)(start - tile_lo) & SLAB_REL_MASK; Lp[n] = (uint16_t)xlen;
                longest = max(longest, xlen);
                if (first) { s0 = start; e0 = end; }
                first = false; ++n;
            }
            start = cut ? end + len + 1 : start;
            end += len & __builtin_amdgcn_sbfe(0x18d, op, 1u);
        };
#pragma unroll
        for (int q = 0; q < SLAB_HEAD_VEC; ++q)
            if (4 * q < c_max) { step(cg[4 * q]); step(cg[4 * q + 1]); step(cg[4 * q + 2]); step(cg[4 * q + 3]); }
        if (n_cig > (uint32_t)SLAB_HEAD) {
            const uint32_t n_ops = ld32(sa->cig_off32, r0 + idx + 1u) - c_lo;
            const uint32_t *const words = a->f.cig + c_lo;
            for (uint32_t i = SLAB_HEAD; i < n_ops; ++i) step(words[i]);
        }
        {   const uint32_t xlen = (uint32_t)(end - start + 1);
            Ap[n] = ((uint32_t)(start - tile_lo) & SLAB_REL_MASK) | TC_ROW_LAST; Lp[n] = (uint16_t)xlen;
            longest = max(longest, xlen);
            if ((uint32_t)(start - tile_lo) >= SLAB_REL_MASK) longest = 0xffffffffu; }
        if (first) { s0 = start; e0 = end; }
        ++n;
        Ap[0] |= TC_ROW_FIRST;                                   // (until the first chunk's work words: which exons begin / end a read, for the lookups)
        sane = s0 <= e0 && start <= end;
        big = longest > SLAB_LEN_MAX;
        re = ReadEnds{s0, e0, start, end};
    }
    const uint32_t r = r0 + idx;
    const bool rev_in = (xs & SLOT_REV) != 0u;
    __syncthreads();                                         // (keys, directories, the stretches' words are whole)
    stamp.mark(1);
    // ---- the chunk list (wave 0, four stretches per lane) while the other waves begin their lookups
    if (wv == 0) {
        const uint32_t f4 = reinterpret_cast<const uint32_t *>(s_trip)[lane];
        const uint32_t stops = (f4 >> 1) & 0x01010101u;
        const unsigned long long ms = __ballot(stops != 0u);
        // (a word behind the first stretch that ends the sweeps says nothing: its wave may have left earlier)
        const int sl = ms ? __ffsll((long long)ms) - 1 : WAVE;
        uint32_t keep = lane < sl ? 0x01010101u : 0u;
        if (lane == sl) { const uint32_t fs = (uint32_t)__ffs((int)stops) - 1u; keep = fs >= 24u ? 0x01010101u : ((2u << fs) - 1u) & 0x01010101u; }
        const uint32_t has = f4 & keep;
        const uint32_t cnt = __popc(has);
        const uint32_t inc = wave_inclusive_scan(cnt);
        uint32_t at = inc - cnt;
#pragma unroll
        for (int q = 0; q < 4; ++q) if ((has >> (8 * q)) & 1u) { if (at < (uint32_t)TC_CHUNKS) s_chunk[at] = (uint16_t)(4 * lane + q); ++at; }
        const uint32_t nc = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
        if (lane == 0) { s_nchunk = nc; if (ms == 0ull || nc > (uint32_t)TC_CHUNKS) s_bad = 1u; }      // (no end in sight, too many chunks: the generic kernel)
    }
    // ---- the key lookups, once per tile, a POSITION per thread and round (every thread the same number of lookups, whatever its read's
    //      exon count; an exon's successor is the next position unless it ends its read)
    bool redo = active && (big || (n > 1 && !sane));
    const bool work0 = active && !redo;
    const uint32_t none = (uint32_t)d.nbk + 1u;
    const tckey_t *const key0 = s_key0, *const key1 = s_key1;
    {
        const TcKey K0{key0, s_dir0}, K1{key1, s_dir1};
        for (uint32_t q = threadIdx.x; q < total; q += (uint32_t)TILE_THREADS) {
            const uint32_t aw = s_A[q];
            const bool first_x = (aw & TC_ROW_FIRST) != 0u, last_x = (aw & TC_ROW_LAST) != 0u;
            const int s = tile_lo + (int)(aw & SLAB_REL_MASK), e = s + (int)s_L[q] - 1;
            const int s2 = last_x ? 0 : tile_lo + (int)(s_A[q + 1u] & SLAB_REL_MASK);
            bool over = false;
            uint32_t w = (ablate & 16384) ? 0u : tc_lookup2(K0, K1, d.b_off, none, d.st_nk, d.en_nk, !(first_x && last_x), !last_x, s, e, s2, over);      // (bit 14, timing diagnostics: no lookups -- results wrong)
            if (over) w = TC_R_OVER;
            s_R[q] = w | (F_ALL << TC_F_SHIFT);
        }
    }
    if (threadIdx.x < 2u) s_msk[threadIdx.x ? (uint32_t)TC_ST_CAP + d.en_nk : d.st_nk] = TcMask{0ull, 0ull};      // (the masks of "no entry": never written again)
    __syncthreads();                                         // (the chunk list)
    stamp.mark(2);
    const uint32_t n_chunk = s_nchunk;
    const bool bad = s_bad != 0u;
    redo = redo || (active && bad);
    if (work0) { bool over = false; for (uint32_t k = 0; k < n; ++k) over = over || (Rp[k] & ((1u << TC_F_SHIFT) - 1u)) == TC_R_OVER; redo = redo || over; }
    // ---- the sweep, chunk by chunk
    __builtin_amdgcn_s_setprio(0);
    bool known = false, stopped = false, ksite = false, lfull = false, rfull = false, lnoth = true, rnoth = true, out_rev = rev_in;
    int ref = -1;
    bool pend = false; uint32_t lim_last = 62u;
    const TcLds L{key0, key1, s_msk, s_msk + TC_ST_CAP, s_dir0, s_dir1, s_rdir, s_hk, s_hx};
    // (the END directory is through with the lookups: the chunks' terminal-exon groups live there)
    static_assert(2 * TC_GROUPS * (int)sizeof(TcGroup) <= TC_DIR_N * 2 && (TC_DIR_N * 2) % 8 == 0, "the groups take the END directory's place");
    TcGroup *const s_grp = reinterpret_cast<TcGroup *>(s_dir1);
    const TxHdr *const hdr = a->f.hdr;
    // (the headers of the chunk's transcripts: thread j < 63 asks for transcript cb + j one chunk ahead)
    int4 h0 = make_int4(0, 0, 0, 0), h1 = h0, h2 = h0;
    auto ask_headers = [&](uint32_t ci) {
        if (ci < n_chunk && threadIdx.x < (uint32_t)TC_MEMBERS_PER) {
            const int j = d.j_lo + (int)s_chunk[ci] * TC_MEMBERS_PER + (int)threadIdx.x;
            if (j < n_tx) { const int4 *hp = reinterpret_cast<const int4 *>(hdr + j); h0 = hp[0]; h1 = hp[1]; h2 = hp[2]; }
            else { h0 = make_int4(INT32_MAX, 0, 0, 0); h1 = make_int4(2, 0, TX_COMPACT, 0); h2 = make_int4(0, 0, 0, 0); }      // (behind the annotation: another chromosome, behind every read)
        }
        // (wave 1 groups the members by their last exon while wave 0 groups them by their first: it asks for the terminal exons as well)
        if (LEVEL >= 1 && LEVEL <= 3 && ci < n_chunk && wv == 1 && lane < TC_MEMBERS_PER) {
            const int j = d.j_lo + (int)s_chunk[ci] * TC_MEMBERS_PER + lane;
            h2 = j < n_tx ? reinterpret_cast<const int4 *>(hdr + j)[2] : make_int4(0, 0, 0, 0);
        }
    };
    if (!bad) ask_headers(0u);
    for (uint32_t ci = 0; !bad && ci < n_chunk; ++ci) {
        const int cb = d.j_lo + (int)s_chunk[ci] * TC_MEMBERS_PER;
        const int w_n = min(TC_MEMBERS_PER, n_tx - cb);
        if (threadIdx.x < (uint32_t)WAVE) {
            bool single = false, loose = false;
            if (lane < TC_MEMBERS_PER) {
                int st = h0.y, en = h0.z;
                if (h0.x < tid0) { st = INT32_MIN; en = INT32_MIN; }            // another chromosome: before / after every read
                else if (h0.x > tid0) { st = INT32_MAX; en = INT32_MAX; }
                s_hk[lane] = make_int4(st, en, h1.x, (h1.z & 0xff) | (h1.y << 8)); s_se[lane] = make_int2(st, en);
                s_hx[lane] = h2;
                single = h1.x == 1 && lane < w_n; loose = !((h1.z & 0xff) & TX_COMPACT) && lane < w_n;
            }
            const unsigned long long b1 = __ballot(single), b2 = __ballot(loose);
            if (lane == 0) { s_mask[0] = b1; s_mask[1] = b2; }
        }
        // the members by first exon (wave 0) and by last exon (wave 1), TcGroup: one ballot per distinct exon.  The loop runs in one wave
        // while the others wait at the barrier below: one 64-bit compare per step, its lane mask taken as it is; a member only notes
        // its group's number -- the groups' words are written behind the loop, by the members (LDS atomics for the masks)
        if (LEVEL >= 1 && LEVEL <= 4 && wv < (LEVEL == 4 ? 1 : 2)) {
            if (ablate & 131072) { if (lane == 0) s_ngrp[wv] = 0xffu; }
            else {
                __builtin_amdgcn_s_setprio(3);
                TcGroup *const grp = s_grp + wv * TC_GROUPS;
                unsigned long long rem = __ballot(lane < w_n);
                const int kx = wv ? h2.z : h2.x, ky = wv ? h2.w : h2.y;
                const unsigned long long key = ((unsigned long long)(uint32_t)ky << 32) | (uint32_t)kx;
                uint32_t mine = 0u, g = 0u;
                while (rem != 0ull && g < (uint32_t)TC_GROUPS) {
                    const int ld = __builtin_ctzll(rem);
                    const unsigned long long v = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(ky, ld) << 32) | (uint32_t)__builtin_amdgcn_readlane(kx, ld);
                    const unsigned long long same = __builtin_amdgcn_uicmpl(key, v, 32) & rem;      // (32: equal)
                    mine = ((same >> lane) & 1ull) ? g : mine;
                    rem &= ~same; ++g;
                }
                if (rem == 0ull) {
                    if (lane < TC_GROUPS) grp[lane].members = 0ull;
                    if (lane < w_n) { grp[mine].key = make_int2(kx, ky); atomicOr(&grp[mine].members, 1ull << lane); }      // (every member of a group writes the same pair)
                }
                if (lane == 0) s_ngrp[wv] = rem != 0ull ? 0xffu : g;
                __builtin_amdgcn_s_setprio(0);
            }
        }
        // this thread's entries in the chunk's frame
        const m64_t keepm = w_n >= 64 ? ~0ull : ((1ull << w_n) - 1ull);
#pragma unroll
        for (int u = 0; u < TC_PER_THREAD; ++u) {
            const uint32_t e = threadIdx.x + (uint32_t)(u * TILE_THREADS);
            if (e < n_ent) {
                const int dd = e_base[u] - cb;
                TcMask q; q.pm = rebase64(e_pm[u], dd) & keepm; q.sm = rebase64(e_sm[u], dd) & keepm;
                s_msk[e < d.st_nk ? e : e - d.st_nk + (uint32_t)TC_ST_CAP] = q;
            }
        }
        __syncthreads();
        stamp.mark(3);
        ask_headers(ci + 1u);
        const bool work = work0 && !redo && !known && !stopped;
        const uint32_t ng1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ngrp[0]), ng2 = LEVEL == 4 ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ngrp[1]);
        const uint32_t n_grp = ((ablate & 65536) || ng1 == 0xffu || ng2 == 0xffu) ? 0xffffu : (ng1 | (ng2 << 8));      // (bit 16: member by member)
        const ChunkVisit vm = tc_visit<LEVEL>(s_se, s_hx, s_grp, n_grp, (ablate & 8192) ? 0 : w_n, work, n, re, s_mask);
        redo = redo || vm.redo;
        stamp.mark(4);
        const bool mapping = work && !vm.redo && n > 1 && !(ablate & 4096);
        const SiteMasks64 sm = tc_map_exons(L, mapping, pend, Ap, Rp, n, vm.vpre, d.st_nk, d.en_nk);
        stamp.mark(5);
        if (work && !vm.redo) {
            int jstar = -1;
            if (n > 1) {
                m64_t c = sm.kand & vm.vpre;
                while (c) {
                    const int j = __ffsll((long long)c) - 1;
                    c &= c - 1ull;
                    const int4 hk = s_hk[j];
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
            if (jref >= 0) { ref = cb + jref; out_rev = ((s_hk[jref].w >> 8) & 1) != 0; }     // :825-831 (a later chunk's member is a later transcript)
            if (LEVEL >= 1 && LEVEL <= 4) { lfull = lfull || (vm.lmask & V) != 0ull; rfull = rfull || (vm.rmask & V) != 0ull; }
            if (LEVEL == 3 || LEVEL == 4) {
                if (lnoth) {
                    if (sm.dm_first & V) lnoth = false;
                    else if (V) lnoth = (tc_overlapping_exon_members(L, d.b_off, d.nb, re.s0, re.e0) & V) == 0ull;
                }
                if (LEVEL == 3 && rnoth) {
                    if (sm.am_last & V) rnoth = false;
                    else if (V) rnoth = (tc_overlapping_exon_members(L, d.b_off, d.nb, re.sl, re.el) & V) == 0ull;
                }
            }
            pend = n > 1; lim_last = known_c ? (uint32_t)jstar : 62u;      // (this chunk's flag clears: tc_map_exons of the next chunk, or the verdicts)
            known = known_c;
            stopped = vm.stopped;
        }
        // (the next chunk overwrites headers and masks; nobody left who sweeps on: the loop ends for the whole workgroup)
        const bool on = work0 && !redo && !known && !stopped;
        const int go_on = __syncthreads_or(on ? 1 : 0);
        stamp.mark(6);
        if (!go_on) break;
    }
    __builtin_amdgcn_s_setprio(TILE_PRIO);
    // ---- verdicts; flag bytes into the row words
    uint32_t info = n << 8;
    if (work0 && !redo) {
        if (n > 1) {
            for (int k = 0; k < (int)n; ++k) {
                uint32_t f = Rp[k];
                if (pend) f &= ~tc_flag_clears(Ap[k] >> SLAB_REL_BITS, lim_last, !known);       // (the last chunk this read swept)
                f >>= TC_F_SHIFT;
                if (known) f &= ~(uint32_t)(F_DON | F_ACC);                             // (every site of a known read is its transcript's)
                f &= (k + 1 == (int)n) ? (uint32_t)F_EXON : 0xffu;                      // the last exon has no junction behind it
                Ap[k] = (Ap[k] & SLAB_REL_MASK) | (f << SLAB_REL_BITS);
            }
        } else Ap[0] = (Ap[0] & SLAB_REL_MASK) | ((uint32_t)F_EXON << SLAB_REL_BITS);
        if (known) info |= I_KNOWN;
        if (ksite) info |= I_KSITE;
        if (full_decision(LEVEL, lfull, lnoth, rfull, rnoth)) info |= I_FULL;
        if (out_rev) info |= I_REV;
        if (a->f.p.n_sj == 0 && (info & (I_FULL | I_KNOWN | I_KSITE)) == (I_FULL | I_KSITE)) info |= I_ACCEPT;
    } else {
        ref = -1;
        if (active) for (uint32_t k = 0; k < n; ++k) Ap[k] = big ? SLAB_POS_SKIP : (Ap[k] & SLAB_REL_MASK);      // (flags 0: the generic kernel writes them)
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
    if (active) { a->f.info[r] = info; a->f.ref_tx[r] = ref; }
    // ---- the tile's first result slot (every tile in front exact: the first look was all of it)
    {
        uint32_t share = first_share, n_polls = 0u;
        bool done;
        if (wv < 3 && !first_look) share = lb_share<false>(sa, t, wv, lane, done, n_polls);
        if (lane == 0 && wv < 3) s_lb[wv] = share;
    }
    __syncthreads();
    const uint32_t xbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(s_lb[0] + s_lb[1] + s_lb[2]));
    if (threadIdx.x == 0) {
        u_xbase[t] = xbase;
        if (t + 1u == n_tiles) { u_xbase[n_tiles] = xbase + total; *sa->exon_total = xbase + total; }
    }
    const SlabOut out{a->f.ex_start, a->f.ex_end, a->f.ex_flag, xbase + loc};
    if (active) a->f.ex_off[r] = out.dst;
    if (active && big) {
        const uint32_t n_ops = ld32(sa->cig_off32, r + 1u) - c_lo;
        const uint32_t *const words = a->f.cig + c_lo;
        WalkState w{pos + 1, pos, 0};
        auto put = [&](int k, int s_, int e_) { out.start[out.dst + (uint32_t)k] = s_; out.end[out.dst + (uint32_t)k] = e_; out.flag[out.dst + (uint32_t)k] = 0; };
        walk_ops<false>(w, words, 0, (int)n_ops, p, put);
        put(w.n, w.start, w.end);
    }
    slab_write_out(SlabOut{out.start, out.end, out.flag, xbase}, s_A, s_L, tile_lo, total);
    stamp.mark(7);
}

}  // namespace l2r
