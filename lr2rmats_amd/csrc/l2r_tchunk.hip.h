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
//                   (first part, number of parts, "the pair matches") -- kept beside the exon's row word (s_R);
//                   the window's CHUNKS: stretches of 63 consecutive transcripts from the tile's cursor value on, the ones with a
//                   transcript that overlaps the tile's span listed up to the first transcript every read lies before (four waves, a
//                   stretch each per round trip).
//   per chunk       the 63 transcripts' headers (asked for one chunk ahead); every thread re-bases ITS entries' masks to the chunk
//                   (two shifts per mask, from registers) and leaves them at the entries' places; one barrier; the member pass
//                   (visit_chunk64); per exon the ORs over the parts its word names (no directory, no key compare); the sweep's
//                   carried state exactly as in k_probe_slab_chunked (src/update_gtf.c:792-835).  A workgroup leaves the loop when
//                   none of its reads is still sweeping (known / stopped).
//
// A chunk here is 63 CONSECUTIVE transcripts (not 63 members): a transcript of the stretch that does not overlap the tile's span lies
// before every read or behind every read, which the member pass sees per read like for any other member (m_bef / m_aft), so masks are
// re-based by shifts alone (no gapped windows) and the chunk's window is its first transcript's number.
// Not taken here (they keep the slab form and k_probe_slab_chunked): tiles that are not exact under the run's thresholds, -d > 0,
// slices of more than TC_ENT_POOL entries, and the tiles a one-window kernel flags late (a key in several entries).
// Junction support (-j) is left to k_validate_sj, the accepted chunk to k_gather_accepted (as for every list-driven kernel).
#pragma once
#include "l2r_tile.hip.h"

namespace l2r {

constexpr int TC_PER_THREAD = TC_ENT_POOL / TILE_THREADS;         // (TC_ENT_POOL: l2r_slab.hip.h, beside tile_chunk_direct)
constexpr int TC_MEMBERS_PER = WIDE_MEMBERS;                     // transcripts per chunk: 63, so that a "first member" still takes 6 bits (63 = none)
constexpr int TC_TRIPS = 256;                            // stretches of 63 transcripts the window scan may look at (16 k transcripts)
constexpr int TC_CHUNKS = 64;                            // ... and the ones with members a tile may have (4 k members)
constexpr int TC_DIR_N = DIR_CAP + 4;                    // directory words per dictionary (bucket b's first entry, two closing words)
static_assert(TC_ENT_POOL % TILE_THREADS == 0 && TC_ENT_POOL <= 1024, "entries per thread; 10-bit entry numbers in a lookup word");

struct TcMask { m64_t pm, sm; };                         // a staged entry's masks in the chunk's frame (bit j = transcript chunk base + j)
struct TcLds { const int2 *key0, *key1; const TcMask *msk0, *msk1; const uint16_t *dir0, *dir1, *rdir; const int4 *hk, *hx; };

// One lookup of the tile's key staging (once per exon and dictionary): the parts of the pair (k1, k2) if the dictionary has it, else the
// parts of the first pair with key 1 (their single masks cover every transcript with that site: build_dict walks a pair's parts until
// both member lists are through).  Word: first part (10 bits) | parts (5 bits) << 10 | "the pair matches" << 15; 0 = nothing.
// over: more than 31 parts (the read goes to the generic kernel).
__device__ __forceinline__ uint32_t tc_lookup(const int2 *key, const uint16_t *dir, int b_off, uint32_t none, bool on, int32_t k1, int32_t k2, bool &over)
{
    const uint32_t ib = on ? min((uint32_t)((k1 >> SITE_SHIFT) + b_off), none) : none;
    const uint32_t lo = dir[ib], hi = dir[ib + 1u];
    uint32_t a0 = 0xffffu, x0 = 0xffffu;
    for (uint32_t r = lo; r < hi; ++r) {
        const int2 q = key[r];
        const bool m1 = q.x == k1;
        a0 = (m1 && a0 == 0xffffu) ? r : a0;
        x0 = (m1 && q.y == k2 && x0 == 0xffffu) ? r : x0;
    }
    const bool pair = x0 != 0xffffu;
    const uint32_t base = pair ? x0 : a0;
    if (base == 0xffffu) return 0u;
    const int2 kb = key[base];
    uint32_t cnt = 1u;
    for (uint32_t r = base + 1u; r < hi; ++r) { const int2 q = key[r]; if (q.x != kb.x || q.y != kb.y) break; ++cnt; }
    if (cnt > 31u) { over = true; cnt = 31u; }
    return base | (cnt << 10) | (pair ? 1u << 15 : 0u);
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
        const int2 q = L.key0[i];
        if (q.x <= e && q.y >= s) m |= L.msk0[i].pm;
    }
    return m;
}

// map_exons_lds64 (l2r_chunk.hip.h) with the lookups done: per exon the ORs over the parts its word names
__device__ __forceinline__ SiteMasks64 tc_map_exons(const TcLds &L, bool mapping, uint32_t *Ap, const uint32_t *Rp, uint32_t n, m64_t vpre)
{
    SiteMasks64 m{~0ull, 0ull, 0ull, 0ull, 0ull};
    const int k_max = wave_max(mapping ? (int)n : 0);
    for (int k = 0; k < k_max; ++k) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        const uint32_t R = live ? Rp[k] : 0u;
        m64_t xm = 0ull, am = 0ull, jm = 0ull, dm = 0ull;
        {   const uint32_t i0 = R & 1023u, c0 = (R >> 10) & 31u;
            for (uint32_t c = 0; c < c0; ++c) { const TcMask q = L.msk0[i0 + c]; xm |= q.pm; am |= q.sm; }
            if (!((R >> 15) & 1u)) xm = 0ull; }
        {   const uint32_t R1 = junc ? R >> 16 : 0u;
            const uint32_t i0 = R1 & 1023u, c0 = (R1 >> 10) & 31u;
            for (uint32_t c = 0; c < c0; ++c) { const TcMask q = L.msk1[i0 + c]; jm |= q.pm; dm |= q.sm; }
            if (!((R1 >> 15) & 1u)) jm = 0ull; }
        const m64_t amj = junc ? am : 0ull;
        uint32_t word = first_member64(xm & vpre);
        word |= first_member64(jm & vpre) << 6;
        word |= ((dm & vpre) ? 1u : 0u) << 12;
        word |= ((amj & vpre) ? 1u : 0u) << 13;
        m.kand &= junc ? (am & dm) : ~0ull;            // Q1: the acceptor probed with exon k is ITS OWN start, k < n-1
        m.kor |= amj | dm;
        if (k == 0) m.dm_first = dm;
        m.am_last = (live && !junc) ? am : m.am_last;
        if (live) Ap[k] = (Ap[k] & SLAB_REL_MASK) | (word << SLAB_REL_BITS);
    }
    return m;
}

constexpr int TC_LDS_BYTES = TILE_POS_CAP * (4 + 2 + 4 + 1) + TC_ENT_POOL * (8 + 16) + 3 * TC_DIR_N * 2 + 2 * 64 * 16 + TC_TRIPS + TC_CHUNKS * 2 + 128;
static_assert(TC_LDS_BYTES <= 54272, "k_tile_chunk: 3 workgroups per CU need 106 allocation granules of 512 bytes at most");

template <int LEVEL>
__global__ __launch_bounds__(TILE_THREADS, 3)
void k_tile_chunk(SlabArgs kernarg_block, const TileRec *__restrict__ u_rec, const TileWin *__restrict__ u_tw, const TileStat *__restrict__ u_stat, const SlotRec *__restrict__ u_slot,
                  uint32_t *__restrict__ u_xbase)
{
    constexpr uint32_t F_ALL = (uint32_t)(F_EXON | F_DON | F_ACC | F_JUNC);
    __shared__ __attribute__((aligned(16))) uint32_t s_A[TILE_POS_CAP];          // row words: start - base | the chunk's work word << 18 (at the end: | flag byte << 18)
    __shared__ __attribute__((aligned(16))) uint16_t s_L[TILE_POS_CAP];          // lengths
    __shared__ __attribute__((aligned(16))) uint32_t s_R[TILE_POS_CAP];          // lookup words (tc_lookup): START | END << 16
    __shared__ __attribute__((aligned(16))) uint8_t s_F[TILE_POS_CAP];           // novel flags still standing
    __shared__ __attribute__((aligned(16))) int2 s_key[TC_ENT_POOL];
    __shared__ __attribute__((aligned(16))) TcMask s_msk[TC_ENT_POOL];
    __shared__ __attribute__((aligned(16))) uint16_t s_dir[3 * TC_DIR_N];
    __shared__ __attribute__((aligned(16))) int4 s_hk[64], s_hx[64];
    __shared__ __attribute__((aligned(16))) uint8_t s_trip[TC_TRIPS];                                         // per stretch: 1 has a member | 2 the sweeps end in it
    __shared__ uint16_t s_chunk[TC_CHUNKS];
    __shared__ m64_t s_mask[2];
    __shared__ uint32_t s_lb[4], s_nchunk, s_bad;
    (void)kernarg_block;
    const SlabArgsK sa = slab_args();
    const PipeArgsK a = pipe_args();
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    if (blockIdx.x >= sa->list_cnt[1]) return;
    const uint32_t t = sa->chunk_list[blockIdx.x];
    const uint32_t n_tiles = sa->n_tiles;
    if (t >= n_tiles) return;
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
    if (!tile_chunk_direct(sa->chunk_direct_on, flags0, chunk_on, d, tst, n_act, a->f.p.min_exon, a->f.p.min_intron, a->f.p.max_delet, a->f.p.ss_dis, ablate)) {
        if (threadIdx.x == 0) atomicAdd(sa->list_cnt + 9, 1u);            // (left to k_probe_slab_chunked: the host skips that launch while nobody counts here)
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
            s_key[e] = make_int2(xa.x, xa.y);
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
    uint32_t *const Ap = s_A + loc; uint16_t *const Lp = s_L + loc; uint32_t *const Rp = s_R + loc; uint8_t *const Fp = s_F + loc;
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
            Ap[n] = (uint32_t)(start - tile_lo) & SLAB_REL_MASK; Lp[n] = (uint16_t)xlen;
            longest = max(longest, xlen);
            if ((uint32_t)(start - tile_lo) >= SLAB_REL_MASK) longest = 0xffffffffu; }
        if (first) { s0 = start; e0 = end; }
        ++n;
        sane = s0 <= e0 && start <= end;
        big = longest > SLAB_LEN_MAX;
        re = ReadEnds{s0, e0, start, end};
    }
    const uint32_t r = r0 + idx;
    const bool rev_in = (xs & SLOT_REV) != 0u;
    __syncthreads();                                         // (keys, directories, the stretches' words are whole)
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
    // ---- the key lookups, once per tile
    bool redo = active && (big || (n > 1 && !sane));
    const bool work0 = active && !redo;
    const uint32_t none = (uint32_t)d.nbk + 1u;
    const int2 *const key0 = s_key, *const key1 = s_key + d.st_nk;
    if (work0) {
        bool over = false;
        int s = re.s0, e = re.e0;
        for (uint32_t k = 0; k < n; ++k) {
            const bool junc = k + 1u < n;
            int s2 = 0, e2 = 0;
            if (junc) { s2 = tile_lo + (int)Ap[k + 1u]; e2 = s2 + (int)Lp[k + 1u] - 1; }
            const uint32_t w0 = (n > 1u && !(ablate & 16384)) ? tc_lookup(key0, s_dir0, d.b_off, none, true, s, e, over) : 0u;
            const uint32_t w1 = (ablate & 16384) ? 0u : tc_lookup(key1, s_dir1, d.b_off, none, junc, e, s2, over);      // (bit 14, timing diagnostics: no lookups -- results wrong)
            Rp[k] = w0 | (w1 << 16);
            Fp[k] = (uint8_t)F_ALL;
            s = s2; e = e2;
        }
        redo = over;
    }
    __syncthreads();                                         // (the chunk list)
    const uint32_t n_chunk = s_nchunk;
    const bool bad = s_bad != 0u;
    redo = redo || (active && bad);
    // ---- the sweep, chunk by chunk
    __builtin_amdgcn_s_setprio(0);
    bool known = false, stopped = false, ksite = false, lfull = false, rfull = false, lnoth = true, rnoth = true, out_rev = rev_in;
    int ref = -1;
    const TcLds L{key0, key1, s_msk, s_msk + d.st_nk, s_dir0, s_dir1, s_rdir, s_hk, s_hx};
    const TxHdr *const hdr = a->f.hdr;
    // (the headers of the chunk's transcripts: thread j < 63 asks for transcript cb + j one chunk ahead)
    int4 h0 = make_int4(0, 0, 0, 0), h1 = h0, h2 = h0;
    auto ask_headers = [&](uint32_t ci) {
        if (ci < n_chunk && threadIdx.x < (uint32_t)TC_MEMBERS_PER) {
            const int j = d.j_lo + (int)s_chunk[ci] * TC_MEMBERS_PER + (int)threadIdx.x;
            if (j < n_tx) { const int4 *hp = reinterpret_cast<const int4 *>(hdr + j); h0 = hp[0]; h1 = hp[1]; h2 = hp[2]; }
            else { h0 = make_int4(INT32_MAX, 0, 0, 0); h1 = make_int4(2, 0, TX_COMPACT, 0); h2 = make_int4(0, 0, 0, 0); }      // (behind the annotation: another chromosome, behind every read)
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
                s_hk[lane] = make_int4(st, en, h1.x, (h1.z & 0xff) | (h1.y << 8));
                s_hx[lane] = h2;
                single = h1.x == 1 && lane < w_n; loose = !((h1.z & 0xff) & TX_COMPACT) && lane < w_n;
            }
            const unsigned long long b1 = __ballot(single), b2 = __ballot(loose);
            if (lane == 0) { s_mask[0] = b1; s_mask[1] = b2; }
        }
        // this thread's entries in the chunk's frame
        const m64_t keepm = w_n >= 64 ? ~0ull : ((1ull << w_n) - 1ull);
#pragma unroll
        for (int u = 0; u < TC_PER_THREAD; ++u) {
            const uint32_t e = threadIdx.x + (uint32_t)(u * TILE_THREADS);
            if (e < n_ent) {
                const int dd = e_base[u] - cb;
                TcMask q; q.pm = rebase64(e_pm[u], dd) & keepm; q.sm = rebase64(e_sm[u], dd) & keepm;
                s_msk[e] = q;
            }
        }
        __syncthreads();
        ask_headers(ci + 1u);
        const bool work = work0 && !redo && !known && !stopped;
        const ChunkLds CL{nullptr, nullptr, nullptr, nullptr, nullptr, s_hk, s_hx, nullptr};
        const ChunkVisit vm = visit_chunk64<LEVEL>(CL, (ablate & 8192) ? 0 : w_n, work, n, re, s_mask);
        redo = redo || vm.redo;
        const bool mapping = work && !vm.redo && n > 1 && !(ablate & 4096);
        const SiteMasks64 sm = tc_map_exons(L, mapping, Ap, Rp, n, vm.vpre);
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
        // (the next chunk overwrites headers and masks; nobody left who sweeps on: the loop ends for the whole workgroup)
        const bool on = work0 && !redo && !known && !stopped;
        if (!__syncthreads_or(on ? 1 : 0)) break;
    }
    __builtin_amdgcn_s_setprio(TILE_PRIO);
    // ---- verdicts; flag bytes into the row words
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
}

}  // namespace l2r
