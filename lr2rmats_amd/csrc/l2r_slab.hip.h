// l2r_slab.hip.h -- the one-walk pipeline for coordinate-sorted records with short CIGARs: TWO light kernels at 6 .. 8 workgroups
// per CU, with the exons handed over through HBM in a layout both sides touch with whole rows (gfx950).
//
// Measured on MI355X (profiles/r02, profiles/r03): kernels of this path are bound by the latency of a wave's dependent chains, and
// a SIMD's issue rate grows with its resident waves up to 8.  An LDS tile of exons (10 bytes each, 40 KB per workgroup: the classic
// kernel) caps that at 4 waves per SIMD.  Here the walk kernel keeps no exons in LDS, the probe kernel 6 bytes per exon (the tile's
// results on their way into read order):
//
//   k_walk_slab   per tile of up to 256 reads: the tile's reads by falling CIGAR length (a counting sort in LDS: the SLOT of a read =
//                 its lane, so the lanes of a wave get similar trip counts); one lane per read, CIGAR words in registers, ONE walk;
//                 exon k of the read in slot s goes to element row * 256 + s of the tile's SLAB (a row = the k-th exons of all reads of
//                 the tile: a wave stores and loads whole rows).  Then every read's place in the READ-ORDER result arrays inside its
//                 tile (`loc`: a scan of the exon counts in read order), the tile's exon count, and -- by the last wave alone, the
//                 others have left -- the tile's descriptor and window (make_descriptor).
//   k_scan_u32    exclusive scan of the tiles' exon counts: the first result slot of every tile (l2r_kernels.hip.h)
//   k_probe_slab  per tile: dictionary slices and window staged in LDS; every lane streams its read's exons row by row (coalesced),
//                 window pass, probes, verdicts with the device functions of the classic kernel; exons, work words and flags wait at
//                 their read-order POSITIONS in LDS (22 KB per workgroup: 6 or 7 workgroups per CU, no scratch), and the tile's block
//                 of the result arrays is written coalesced: ex_start / ex_end / ex_flag / ex_off in READ ORDER (exon k of read r
//                 at ex_off[r] + k), info, ref_tx.  Nothing is left in an intermediate layout: l2r_download / l2r_device_view_get
//                 hand out these arrays as they are.
//   k_probe_slab_wide (l2r_wide.hip.h)      the same for tiles whose window holds 33 .. 63 transcripts (64-bit masks).
//   k_probe_slab_chunked (l2r_chunk.hip.h)  tiles beyond that, tiles with a dictionary key in several entries, tiles whose dictionary
//                 slices do not fit the staging of the two above: the window 63 members at a time, the sweep's state carried per read.
//
// A slab row is one word per exon: start relative to the tile's first base (18 bits) | length (14 bits) -- 4 bytes per exon cross HBM
// between the kernels.  A read's LAST exon sits in row 0 of its column and exon k < n - 1 in row k + 1: the probe side needs the
// first and the last exon before anything else, and finds both at addresses that do not depend on the exon count (one dependent round
// trip less in front of its first barrier).  A tile's slab has as many rows as its longest read can have exons (bound from the CIGAR
// lengths at upload, at most SLAB_ROWS); reads beyond that, and reads with an exon the row word cannot say, are OUTLIERS: walked
// literally into a dense area (dense_start / dense_end), copied into the result arrays by the probe side, classified by the generic
// kernel.
#pragma once
#include "l2r_window.hip.h"

namespace l2r {

constexpr int SLAB_ROWS = 24;                            // rows of a tile's slab at most (exons of its longest read it can hold)
constexpr uint32_t SLAB_STRIDE = TILE_THREADS;           // elements between two rows of a column
constexpr int SLAB_HEAD_VEC = 6;                         // 16-byte CIGAR vectors a lane holds: 24 ops; longer reads finish from memory
constexpr int SLAB_HEAD = 4 * SLAB_HEAD_VEC;
// k_walk_slab -> k_probe_slab, one word per slot: the read's index inside its tile (bits 0-7), its strand bit, two flags, its exon count
// A row word: the exon's start relative to the tile's first base in the low 18 bits (k_probe_slab stages exactly these 18 bits), its
// length in the upper 14 (slab_pack).  The upload cuts tiles so that their reads BEGIN less than SLAB_TILE_SPAN = 2^17 bases apart
// (l2r_engine.hip); a read with an exon that does not fit -- 16 kb or longer, or starting 2^18 - 1 bases or more behind the tile's
// first base -- is an outlier (dense area).
constexpr int SLAB_REL_BITS = 18;
constexpr uint32_t SLAB_REL_MASK = (1u << SLAB_REL_BITS) - 1u;
constexpr uint32_t SLAB_LEN_MAX = (1u << (32 - SLAB_REL_BITS)) - 1u;
constexpr int32_t SLAB_TILE_SPAN = 1 << (SLAB_REL_BITS - 1);
__device__ __forceinline__ uint32_t slab_pack(int rel, uint32_t len) { return (uint32_t)rel | (len << SLAB_REL_BITS); }
constexpr uint32_t PRE_REV = 1u << 8;
constexpr uint32_t PRE_INSANE = 1u << 9;                 // first or last exon empty (or, with -e < 1, any exon): the generic kernel decides
constexpr uint32_t PRE_DENSE = 1u << 10;                 // an outlier: its exons are in the dense area (row 0 of its column holds the run's index)
constexpr int PRE_N_SHIFT = 11;
constexpr int PL_LOC_SHIFT = 19;                         // packed word: PRE_* bits and an 8-bit exon count below, `loc` (13 bits) above
constexpr uint32_t PL_PRE_MASK = (1u << PL_LOC_SHIFT) - 1u;
constexpr uint32_t PL_N_LIMIT = 1u << (PL_LOC_SHIFT - PRE_N_SHIFT), PL_LOC_LIMIT = 1u << (32 - PL_LOC_SHIFT);

// c ops -> rows its read needs at most when every kept inner exon is at least one base long (min_exon >= 1): each kept exon but
// the first and the last needs an op of its own next to its cut (src/bam2gtf.c:31-78), so n <= (c + 3) / 2.  With -e < 1 the walk
// itself watches the rows (k_walk_slab<true>).
__host__ __device__ __forceinline__ uint32_t slab_rows_of(uint32_t c) { return (c + 3u) >> 1; }
// row of exon j of a read with n exons (the last exon in row 0)
__device__ __forceinline__ uint32_t slab_row(uint32_t j, uint32_t n) { return j + 1u < n ? j + 1u : 0u; }

// What the upload knows of a tile, one 32-byte record (k_walk_slab starts from it: one scalar load instead of a chain of three):
// its reads [r0, r0 + n_act), the first element of its slab and the rows it has, its chromosome and first base (its first read's).
struct TileRec { uint32_t r0, n_act, sbase, rows; int32_t tid0, lo; uint32_t pad[2]; };
// k_walk_slab -> k_describe_scan, one 16-byte record per tile: chromosome, first and last base of the tile's reads, and (one byte per
// wave of the probe kernels) the largest exon count among the reads of each slot group
// ... and what of the descriptor's load chain does not need the tile's last base: the cursor value of its first read, its chromosome's
// first bucket and bucket count (looked up by one wave of k_walk_slab while its CIGAR vectors are in flight)
// ... and, for the probe kernels, the tile's reads and slab once more (one record instead of three arrays + the first read's position)
struct TileSpan { int32_t tid, lo, hi; uint32_t rows; int32_t jl, tb, nb, pad; uint32_t r0, n_act, sbase, fat; };     // fat: see SlabArgs::pl
// What the upload knows of a tile's CIGAR operations whatever the parameters (k_tile_index, l2r_tile.hip.h): the N operations of its
// reads, the shortest of them, the longest D operation, and the shortest stretch of reference bases between two N operations of one
// read.  With them a run knows the tile's EXON COUNT without its CIGARs whenever no threshold is borderline inside the tile: every N is
// an intron (-i <= the shortest N), no D cuts (-t >= the longest D), no inner exon is dropped (-e <= the shortest stretch) => exons =
// reads + N operations (src/bam2gtf.c:41-74).
struct TileStat { int32_t n_ops_n, min_n, max_d, min_seg; };
__host__ __device__ __forceinline__ bool tile_exact(const TileStat &st, int min_exon, int min_intron, int max_delet)
{
    return st.min_n >= min_intron && st.max_d <= max_delet && st.min_seg >= min_exon;
}
// One-kernel tile path: the words a tile's exon count travels through to the later tiles (SlabArgs::lb_tile / lb_blk / lb_sup): a sum
// in bits 0-39, from bit 40 on the number of counts it holds.  A block = the 16 tiles of one k_describe_scan workgroup (and of one XCD
// group of k_tile), a super-block = 64 blocks.
constexpr int LB_SHIFT = 40;
constexpr unsigned long long LB_SUM_MASK = (1ull << LB_SHIFT) - 1ull;
constexpr int LB_BLK_SHIFT = 4, LB_BLK = 1 << LB_BLK_SHIFT, LB_SUP_SHIFT = 10;
struct SlabArgs {
    PipeArgs g;
    const uint32_t *tile_sbase;                          // first element of every tile's slab (+ a closing entry)
    uint32_t *slab_row;                                  // the slabs: one word per exon (slab_pack)
    int32_t *dense_start, *dense_end;                    // outliers: exon k of a run at run + k
    unsigned long long *ovf_cursor;                      // next free element of the dense area
    // k_walk_slab -> the probe kernels, slot order, ONE word per read (slab_preloc): the PRE_* word with the exon count in 8 bits below the
    // exons of the tile's reads in front of this one (read order) in 13 bits.  A tile with a read of 256 exons or more or with 8192 exons
    // or more (outliers make such tiles) is FAT: its reads' two words go to pre_x / loc_x instead, whole.
    uint32_t *pl, *pre_x, *loc_x;
    const uint32_t *cig_off32;                           // the records' CIGAR offsets once more, 32 bit (made at upload: a shard has < 2^32 words)
    TileWin *tw;                                         // k_describe_scan -> the probe kernels: descriptor + window per tile
    TileSpan *span;                                      // k_walk_slab -> k_describe_scan: what a tile's descriptor is made from
    // tw64[tile]: the 64-member window record of a tile whose window holds 33 .. 63 transcripts (TD_WIDE).  The tiles of
    // k_probe_slab_wide / k_probe_slab_chunked are listed by block 1 of the scan launch between the walk and the probes (TileLists,
    // l2r_kernels.hip.h): wide_list / chunk_list, list_cnt[0] / [1] = entries, [2] / [3] = the kernels' work cursors.  A one-window
    // kernel that finds a dictionary key in several entries appends its tile to chunk_list (rare).  chunk_on 0: no chunked windows.
    TileWin64 *tw64;
    uint32_t *wide_list, *chunk_list, *list_cnt;
    uint32_t *list_cnt_next;                             // one-kernel tile path: the list counters take turns run by run like lb_sup (16 words each): this run's k_describe_scan<true> clears the entry counts of the NEXT run's block, so no launch behind the list kernels is needed for that
    uint32_t *tile_flags;                                // every tile's descriptor flags once more, densely (what TileLists reads)
    uint32_t chunk_on;
    uint32_t n_tiles;
    // one-kernel tile path (l2r_tile.hip.h): the tiles' exon counts on their way to every later tile's first result slot -- lb_tile[t]
    // (one word per tile), lb_blk[t >> 4] and lb_sup[t >> 10] (sums over 16 tiles / 64 blocks), LB_* above.  k_describe_scan<true> writes
    // the words of the tiles and blocks (with the counts it knows from tile_stat: all of them unless a threshold is borderline) and adds
    // the complete blocks to lb_sup.  lb_sup is one of TWO arrays that take turns run by run: the words add up during a run, so they
    // have to start from zero -- this run's k_describe_scan<true> clears the other array (lb_sup_next, n_sup words), which nobody
    // touches meanwhile, for the next run; no launch behind k_tile is needed for that.  lb_err: set by a tile that waited in vain
    // (diagnostics; never seen).  fb_list: the tiles k_tile left in slab form for k_probe_slab (list_cnt[4] entries).
    unsigned long long *lb_tile, *lb_blk, *lb_sup, *lb_sup_next;
    uint32_t n_sup;
    const TileStat *tile_stat;
    // per super-block of 1024 tiles (made with tile_stat): n_ops_n = its exon count if every tile in it is exact (reads + N operations),
    // the other three fields the extremes over its tiles -- tile_exact(sup_stat[s], ...) says "every tile of s is exact"
    const TileStat *sup_stat;
    uint32_t *lb_err, *fb_list;
    SjDir sj;                                            // the junction table's directories and rows (k_tile's junction check)
    uint32_t has_wide_keys;                              // the annotation has dictionary keys in several entries (SE_WIDE)
    uint32_t chunk_direct_on;                            // ... and k_tile_chunk (l2r_tchunk.hip.h) the exact tiles of the chunked kernel
    uint32_t wide_direct_on;                             // one-kernel tile path: k_tile's WIDE instance takes the exact 64-bit-mask tiles straight from their CIGARs
    uint32_t *exon_total;                                // the run's exon count (k_tile: written by the last tile)
};
typedef const __attribute__((address_space(4))) SlabArgs *SlabArgsK;
__device__ __forceinline__ SlabArgsK slab_args()
{
    SlabArgsK q = (SlabArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}

// A one-window kernel that finds a dictionary key in several entries hands its tile to k_probe_slab_chunked LATE: behind the entries
// k_describe_scan / TileLists made (list_cnt[1] of them, which k_tile_chunk may be walking over beside the caller), in a region of its own:
// chunk_list[n_tiles + 1 + i], i < list_cnt[8].
__device__ __forceinline__ void chunk_list_append_late(SlabArgsK sa, uint32_t t) { sa->chunk_list[sa->n_tiles + 1u + atomicAdd(sa->list_cnt + 8, 1u)] = t; }
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

// a read's PRE_* word and `loc` as k_walk_slab left them (the probe kernels behind k_probe_slab; that one spells it out around its loads)
__device__ __forceinline__ void slab_preloc(SlabArgsK sa, uint32_t t, uint32_t at, uint32_t &pre, uint32_t &loc)
{
    if (sa->span[t].fat) { pre = ld32(sa->pre_x, at); loc = ld32(sa->loc_x, at); }
    else { const uint32_t w = ld32(sa->pl, at); pre = w & PL_PRE_MASK; loc = w >> PL_LOC_SHIFT; }
}

// ---------------------------------------------------------------------------------------------------------- slab_walk_tile
// What k_walk_slab does with a tile once every slot holds its read's record and the head of its CIGAR (cg: SLAB_HEAD words in registers,
// words behind the read's last op = "I, length 0"): the walk into the tile's slab, the outliers' literal walk into the dense area, the
// reads' places in read order, their words (pl), the tile's span record and exon count.  A device function because the one-kernel tile
// path (l2r_tile.hip.h) hands the tiles it cannot finish itself -- windows beyond 32 members, outliers, fat tiles -- over in exactly this
// form.  `slot`: the thread's slot (the column of the slab and the place of its word; k_walk_slab: its thread number); W: 2 x 256 + 12 words
// of LDS.  FIRST: this kernel is the first of a run (clears the run's counters).  Returns the tile's exon count (every thread).
struct WalkLds { uint32_t *cnt, *loc; int *wmax; uint32_t *wn, *nmax; };
template <bool GENERAL, bool FIRST>
__device__ __forceinline__ uint32_t slab_walk_tile(SlabArgsK sa, PipeArgsK a, uint32_t t, uint32_t r0, uint32_t n_act, uint32_t sbase, uint32_t rows_tile, int32_t tid0, int32_t pos0,
                                                   uint32_t slot, bool active, uint32_t c_lo, int32_t pos, uint32_t xw, uint32_t (&cg)[SLAB_HEAD], const WalkLds &W)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const uint32_t n_cig = xw & 0xffffu, idx = xw >> 24;             // (n_cig 65535: that many or more)
    DevParams p;
    p.min_exon = a->f.p.min_exon; p.min_intron = a->f.p.min_intron; p.max_delet = a->f.p.max_delet;
    const uint32_t t3 = ((uint32_t)p.min_intron << 4) | 3u, t2 = ((uint32_t)(p.max_delet + 1) << 4) | 2u;      // op and length compare as one number
    // beyond the rows a slab can have: an outlier from the start (-e < 1: only when the CIGAR cannot be walked out of the registers)
    bool outlier = GENERAL ? n_cig > (uint32_t)SLAB_HEAD : slab_rows_of(n_cig) > (uint32_t)SLAB_ROWS;
    const int c_max = wave_max((active && !outlier) ? (int)min(n_cig, (uint32_t)SLAB_HEAD) : 0);
    uint32_t *const rows = sa->slab_row;
    const uint32_t off = sbase + slot;
    const int32_t base = pos0 + 1;                           // the tile's first base (sorted records: the first read's)
    uint32_t n = 0u;
    int el = INT32_MIN;
    bool sane = true;
    if (active && !outlier) {
        // one lane per read; kept exon k lands in row k + 1 of the read's column, the last exon in row 0
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
                if (!GENERAL || n + 1u < rows_tile) st32(rows, off + (n + 1u) * SLAB_STRIDE, slab_pack(start - base, xlen));
                longest = max(longest, xlen);
                if (GENERAL) sane = sane & (start <= end);
                if (first) { s0 = start; e0 = end; }
                first = false; ++n;
            }
            start = cut ? end + len + 1 : start;
            end += len & __builtin_amdgcn_sbfe(0x18d, op, 1u);         // ops 0 2 3 7 8 advance the reference
        };
#pragma unroll
        for (int q = 0; q < SLAB_HEAD_VEC; ++q)
            if (4 * q < c_max) { step(cg[4 * q]); step(cg[4 * q + 1]); step(cg[4 * q + 2]); step(cg[4 * q + 3]); }       // (wave-uniform)
        if (!GENERAL && n_cig > (uint32_t)SLAB_HEAD) {
            const uint32_t *const words = a->f.cig + c_lo;
            for (uint32_t i = SLAB_HEAD; i < n_cig; ++i) step(words[i]);
        }
        {   const uint32_t xlen = (uint32_t)(end - start + 1);
            st32(rows, off, slab_pack(start - base, xlen));
            longest = max(longest, xlen);
            // (starts rise along the read: the last one is the furthest)
            if ((uint32_t)(start - base) >= SLAB_REL_MASK) longest = 0xffffffffu; }
        if (first) { s0 = start; e0 = end; }
        ++n;
        el = end;
        // with min_exon >= 1 kept inner exons are at least one base long; the first and the last one are kept whatever their length
        sane = GENERAL ? (sane & (start <= end)) : (s0 <= e0 && start <= end);
        // an exon of 16 kb or more or one that starts 256 kb behind the base does not fit the row format, and (-e < 1 only) the read
        // may have outgrown the tile's rows
        if (longest > SLAB_LEN_MAX || (GENERAL && n > rows_tile)) { outlier = true; n = 0u; sane = true; el = INT32_MIN; }
    }
    if (active && outlier) {
        // an outlier: the literal walk (l2r_kernels.hip.h), twice -- count, take a run of the dense area, store
        const uint32_t r = r0 + idx;
        const uint32_t *const p_off = sa->cig_off32;
        const uint32_t n_ops = ld32(p_off, r + 1u) - ld32(p_off, r);
        const uint32_t *const words = a->f.cig + c_lo;
        {
            WalkState w{pos + 1, pos, 0};
            auto none = [&](int, int, int) {};
            walk_ops<false>(w, words, 0, (int)n_ops, p, none);
            n = (uint32_t)w.n + 1u;
        }
        const uint32_t run = (uint32_t)atomicAdd(sa->ovf_cursor, (unsigned long long)n);
        int32_t *const ds = sa->dense_start, *const de = sa->dense_end;
        WalkState w{pos + 1, pos, 0};
        auto put = [&](int k, int s, int e) { ds[run + (uint32_t)k] = s; de[run + (uint32_t)k] = e; sane = sane & (s <= e); el = e; };
        walk_ops<false>(w, words, 0, (int)n_ops, p, put);
        put(w.n, w.start, w.end);
        st32(rows, off, run);                     // (row 0 of the unused column: where the probe side finds the run)
    }
    W.cnt[idx] = active ? n : 0u;                        // (every entry is written: idx is a permutation of 0 .. 255)
    const int m = wave_max(active ? el : INT32_MIN);
    const int wn = wave_max((active && !outlier) ? (int)n : 0);
    const int wn_all = wave_max(active ? (int)min(n, 0x7fffffffu) : 0);
    // (wn: by SLOT group -- the probe kernels ask "which rows do the slots of my wave have"; k_tile's waves hold rotated slot groups)
    if (lane == 0) { W.wmax[wv] = m; W.wn[slot >> 6] = (uint32_t)min(wn, 255); W.nmax[wv] = (uint32_t)wn_all; }
    if (FIRST && t == 0u && threadIdx.x == 0) {
        // the run's counters (this kernel is the first of a run): redo list, chunk cursor of the accepted list.
        // (The cursor of the outlier area is cleared by k_probe_slab for the next run.)
        uint32_t *const cnt = a->f.redo_count;
        cnt[0] = 0u; cnt[1] = 0u; cnt[2] = 0u;
        // ... and the tile lists of k_probe_slab_wide / k_probe_slab_chunked with their work cursors (k_describe_scan appends)
        uint32_t *const lc = sa->list_cnt;
        lc[0] = 0u; lc[1] = 0u; lc[2] = 0u; lc[3] = 0u; lc[8] = 0u; lc[9] = 0u;
    }
    __syncthreads();
    // ---- every read's place among the tile's exons in READ order: each wave scans the 256 counts (four per lane) for itself
    uint32_t total;
    {
        const uint4 c4 = reinterpret_cast<const uint4 *>(W.cnt)[lane];
        const uint32_t sum = c4.x + c4.y + c4.z + c4.w;
        const uint32_t inc = wave_inclusive_scan(sum), ex = inc - sum;
        reinterpret_cast<uint4 *>(W.loc)[lane] = make_uint4(ex, ex + c4.x, ex + c4.x + c4.y, ex + c4.x + c4.y + c4.z);     // (the four waves write the same values)
        total = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
    }
    // (a fat tile: some read's exon count or place does not fit the packed word -- only outliers make such tiles)
    const bool fat = total >= PL_LOC_LIMIT || max(max(W.nmax[0], W.nmax[1]), max(W.nmax[2], W.nmax[3])) >= PL_N_LIMIT;
    if (active) {
        const uint32_t at = r0 + slot;
        const uint32_t word = idx | (((xw >> 16) & 1u) ? PRE_REV : 0u) | (sane ? 0u : PRE_INSANE) | (outlier ? PRE_DENSE : 0u) | (n << PRE_N_SHIFT);
        if (!fat) sa->pl[at] = word | (W.loc[idx] << PL_LOC_SHIFT);
        else { sa->pl[at] = 0u; sa->pre_x[at] = word; sa->loc_x[at] = W.loc[idx]; }
    }
    // ---- what the tile's descriptor is made from (k_describe_scan, one wave per tile, in the launch of the scan): its chromosome, its
    //      first and last base, the rows each wave of the probe kernels has to look at; and its exon count
    //      (Until round 4 the last wave of this workgroup made the descriptor itself, with the other three gone: its chain of five
    //       dependent round trips kept the workgroup's LDS and wave slots for 0.047 of the kernel's 0.284 ms.)
    if (threadIdx.x == 0) {
        a->tile_total[t] = total;                        // (one word per tile, scanned by k_describe_scan: a single counter would serialise 156 k waves)
        const int32_t tile_hi = max(max(W.wmax[0], W.wmax[1]), max(W.wmax[2], W.wmax[3]));
        const uint32_t rows = W.wn[0] | (W.wn[1] << 8) | (W.wn[2] << 16) | (W.wn[3] << 24);
        reinterpret_cast<int4 *>(sa->span + t)[0] = make_int4(tid0, pos0 + 1, tile_hi, (int)rows);
        reinterpret_cast<int4 *>(sa->span + t)[2] = make_int4((int)r0, (int)n_act, (int)sbase, fat ? 1 : 0);
    }
    return total;
}

// ---------------------------------------------------------------------------------------------------------- k_walk_slab
// GENERAL: -e < 1 (an inner exon may be empty: every exon's sanity is checked, and the read leaves the slab when it has more exons
// than the tile's slab has rows -- with min_exon >= 1 the row bound from the CIGAR length makes that impossible).
template <bool GENERAL>
__global__ __launch_bounds__(TILE_THREADS, 8)
void k_walk_slab(SlabArgs kernarg_block, const TileRec *__restrict__ u_rec)
{
    __shared__ uint32_t s_hist[WAVE];
    __shared__ uint32_t s_x0[TILE_THREADS], s_x1[TILE_THREADS], s_x2[TILE_THREADS];       // the records' fields, slot order
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[TILE_THREADS], s_loc[TILE_THREADS];      // exon counts / their exclusive scan, READ order
    __shared__ int s_wmax[TILE_THREADS / WAVE];
    __shared__ uint32_t s_wn[TILE_THREADS / WAVE], s_nmax[TILE_THREADS / WAVE];
    (void)kernarg_block;
    const SlabArgsK sa = slab_args();
    const PipeArgsK a = pipe_args();
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const uint32_t t = xcd_tile(blockIdx.x, gridDim.x);
    if (t >= sa->n_tiles) return;
    // (the tile's record from the upload: the vector loads below are the kernel's second round trip, not its fourth)
    const TileRec rec = u_rec[t];
    asm volatile("" :: "s"(rec.r0), "s"(rec.n_act), "s"(rec.sbase), "s"(rec.rows), "s"(rec.tid0), "s"(rec.lo));
    const uint32_t r0 = rec.r0, n_act = rec.n_act;
    const int32_t tid0 = rec.tid0, pos0 = rec.lo - 1;
    const uint32_t sbase = rec.sbase, rows_tile = rec.rows;
    // ---- the tile's reads by falling CIGAR length (counting sort, 64 bins): thread i brings read i, slot s takes what landed there
    {
        const uint32_t i = threadIdx.x;
        uint32_t c_lo = 0u, c = 0u, rev = 0u; int32_t pos = 0;
        if (i < n_act) {
            const uint32_t *const p_off = sa->cig_off32;
            c_lo = ld32(p_off, r0 + i); c = ld32(p_off, r0 + i + 1u) - c_lo;
            pos = ld32(a->f.r_pos, r0 + i); rev = ld32(a->f.r_rev, r0 + i) ? 1u : 0u;
        }
        if (i < (uint32_t)WAVE) s_hist[i] = 0u;
        __syncthreads();
        const uint32_t est = i < n_act ? max(1u, min((c + 1u) >> 1, (uint32_t)(WAVE - 1))) : 0u;      // threads without a read sort last
        const uint32_t bin = (uint32_t)(WAVE - 1) - est;
        const uint32_t rank = atomicAdd(&s_hist[bin], 1u);
        __syncthreads();
        if (i < (uint32_t)WAVE) { const uint32_t v = s_hist[i]; s_hist[i] = wave_inclusive_scan(v) - v; }
        __syncthreads();
        const uint32_t slot = s_hist[bin] + rank;
        s_x0[slot] = c_lo; s_x1[slot] = (uint32_t)pos; s_x2[slot] = min(c, 0xffffu) | (rev << 16) | (i << 24);
        __syncthreads();
    }
    const bool active = threadIdx.x < n_act;
    const uint32_t c_lo = s_x0[threadIdx.x], xw = s_x2[threadIdx.x];
    const int32_t pos = (int32_t)s_x1[threadIdx.x];
    const uint32_t n_cig = xw & 0xffffu;                            // (65535: that many or more)
    // ---- the head of the read's CIGAR: six 16-byte vectors, all in flight at once; words behind the last op become "I, length 0"
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
    // The head of the tile descriptor's load chain (k_describe_scan): the cursor value of the tile's first read (first transcript its
    // sweep can reach: three dependent SCALAR loads, on their own counter) and the chromosome's bucket range, by one wave, while the
    // CIGAR vectors above are in flight.
    if (wv == TILE_THREADS / WAVE - 1 && n_act) {
        CursorDir cd;
        cd.key = a->cd.key; cd.dir = a->cd.dir; cd.kb_base = a->cd.kb_base; cd.n_tid = a->cd.n_tid; cd.n_tx = a->cd.n_tx;
        const int jl = cursor_value(cd, tid0, pos0 + 1);
        int tb = 0, nb = 0;
        if (tid0 >= 0 && tid0 < a->n_tid_dir) { tb = a->tid_base[tid0]; nb = a->tid_base[tid0 + 1] - tb; }
        if (lane == 0) reinterpret_cast<int4 *>(sa->span + t)[1] = make_int4(jl, tb, nb, 0);
    }
#pragma unroll
    for (int i = 0; i < SLAB_HEAD; ++i) cg[i] = (uint32_t)i < n_cig ? cg[i] : 1u;
    const WalkLds W{s_cnt, s_loc, s_wmax, s_wn, s_nmax};
    (void)slab_walk_tile<GENERAL, true>(sa, a, t, r0, n_act, sbase, rows_tile, tid0, pos0, threadIdx.x, active, c_lo, pos, xw, cg, W);
}

// ---------------------------------------------------------------------------------------------------------- k_walk_slab_long
// k_walk_slab for LONG CIGARs (ONT-like input: hundreds of operations per read; min_exon >= 1): the same outputs -- slab rows, the
// reads' words, the tile's span record and exon count -- so that k_describe_scan and the probe kernels behind it do not know the
// difference, with the walk of k_pass_a<true>: ONE WAVE walks one read at a time as a scan over its op stream (wave_chunk_walk:
// six or eight words per lane and round, reference ends by prefix sum, exon starts by prefix maximum, kept cuts by ballot), the first
// round of the wave's next read in flight meanwhile.  A read's first WALK_SLAB exons wait in an exon-major LDS slab, later ones in a list;
// when every read of the tile has been walked, the reads get their slots (by falling exon count: counting sort) and their columns go
// to the tile's slab with whole rows.  No per-read cursor value, no window here (k_pass_a makes both): the tile's descriptor is
// k_describe_scan's.  Tiles are `reads_per_tile` reads (128 for ONT-like input: the upload cannot bound a read's exons by its CIGAR
// length); a tile's slab has SLAB_ROWS rows.
#ifndef L2R_WALKLONG_WGS
#define L2R_WALKLONG_WGS 8
#endif
#ifndef L2R_WALK_MIXED
#define L2R_WALK_MIXED 1
#endif
__global__ __launch_bounds__(TILE_THREADS, L2R_WALKLONG_WGS)
void k_walk_slab_long(SlabArgs kernarg_block, const TileRec *__restrict__ u_rec)
{
    __shared__ uint32_t s_hist[WAVE];
    // per read, three words (8 workgroups per CU: 18.7 KB with the dynamic part): going into the walk {first CIGAR word, ops, pos}, coming out
    // {., exon count | flags << 16, read end}; behind the walk s_ra holds the exon counts in read order and s_rc their exclusive scan
    __shared__ __attribute__((aligned(16))) uint32_t s_ra[TILE_THREADS], s_rb[TILE_THREADS], s_rc[TILE_THREADS];
    __shared__ uint8_t s_slot[TILE_THREADS];                                     // read -> slot
    __shared__ uint8_t s_nslot[TILE_THREADS];                                    // slot -> exon count of a slab read (0: outlier / none)
    __shared__ int s_wmax[TILE_THREADS / WAVE];
    __shared__ uint32_t s_wn[TILE_THREADS / WAVE], s_nmax[TILE_THREADS / WAVE];
    __shared__ uint32_t s_ovf_n;
    uint32_t *const s_cnt = s_ra, *const s_loc = s_rc;
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    (void)kernarg_block;
    const SlabArgsK sa = slab_args();
    const PipeArgsK a = pipe_args();
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const uint32_t t = xcd_tile(blockIdx.x, gridDim.x);
    if (t >= sa->n_tiles) return;
    const TileRec rec = u_rec[t];
    asm volatile("" :: "s"(rec.r0), "s"(rec.n_act), "s"(rec.sbase), "s"(rec.rows), "s"(rec.tid0), "s"(rec.lo));
    const uint32_t r0 = rec.r0, n_act = rec.n_act;
    const int32_t tid0 = rec.tid0, pos0 = rec.lo - 1;
    const uint32_t sbase = rec.sbase, rows_tile = rec.rows;
    DevParams p;
    p.min_exon = a->f.p.min_exon; p.min_intron = a->f.p.min_intron; p.max_delet = a->f.p.max_delet;
    const int slab_w = a->f.p.reads_per_tile;                                    // reads per exon row of the LDS slab
    int32_t *const s_slab_s = reinterpret_cast<int32_t *>(s_dyn);            // exon-major: start ...
    int *const s_ovf = s_slab_s + WALK_SLAB * slab_w;                        // exons beyond the slab: {read | k << 8, start, end}
    uint16_t *const s_slab_l = reinterpret_cast<uint16_t *>(s_ovf + 3 * WALK_OVF);      // ... and length
    const bool active = threadIdx.x < n_act;
    const uint32_t r = r0 + threadIdx.x;
    uint32_t c_lo = 0u, rev = 0u; int32_t pos = 0;
    if (active) {
        const uint32_t *const p_off = sa->cig_off32;
        c_lo = ld32(p_off, r); const uint32_t c = ld32(p_off, r + 1u) - c_lo;
        pos = ld32(a->f.r_pos, r); rev = ld32(a->f.r_rev, r) ? 1u : 0u;
        s_ra[threadIdx.x] = c_lo; s_rb[threadIdx.x] = min(c, 0x7fffffffu); s_rc[threadIdx.x] = (uint32_t)pos;
    }
    if (threadIdx.x == 0) s_ovf_n = 0u;
    if (threadIdx.x < (uint32_t)WAVE) s_hist[threadIdx.x] = 0u;
    // the head of the tile descriptor's load chain, by one wave, while the first chunks are on their way (see k_walk_slab)
    if (wv == TILE_THREADS / WAVE - 1 && n_act) {
        CursorDir cd;
        cd.key = a->cd.key; cd.dir = a->cd.dir; cd.kb_base = a->cd.kb_base; cd.n_tid = a->cd.n_tid; cd.n_tx = a->cd.n_tx;
        const int jl = cursor_value(cd, tid0, pos0 + 1);
        int tb = 0, nb = 0;
        if (tid0 >= 0 && tid0 < a->n_tid_dir) { tb = a->tid_base[tid0]; nb = a->tid_base[tid0 + 1] - tb; }
        if (lane == 0) reinterpret_cast<int4 *>(sa->span + t)[1] = make_int4(jl, tb, nb, 0);
    }
    __syncthreads();
    const int32_t base = pos0 + 1;                           // the tile's first base (sorted records: the first read's)
    {   // ---- the walk: the reads of the tile are dealt to the four waves
        const uint32_t *const cig = a->f.cig;
        constexpr uint32_t ROUND = (uint32_t)(WCHUNK * WAVE);
        uint32_t q = (uint32_t)wv;
        auto meta_of = [&](uint32_t qq) { return qq < n_act ? make_int4((int)s_ra[qq], (int)s_rb[qq], (int)s_rc[qq], 0) : make_int4(0, 0, 0, 0); };
        // (a read of up to 384 ops -- most ONT-like reads -- takes one round of six words per lane, longer ones rounds of eight)
        auto words_per_lane = [](int n_ops) { return L2R_WALK_MIXED && n_ops <= WCHUNK_SHORT * WAVE ? WCHUNK_SHORT : WCHUNK; };
        // The first round of the wave's next read is asked for before this read is walked.  Two buffers with fixed registers take turns (the
        // read loop is written out twice): handing the next read's chunk over to "the current one" would be eight register copies per read.
        struct Ahead { int4 meta; CigarWindow cw; WaveChunk ch; };
        auto fetch = [&](Ahead &b, uint32_t qq) {
            b.meta = meta_of(qq);
            b.cw = cigar_window(cig, (uint32_t)b.meta.x, (uint32_t)b.meta.y);
            b.ch = wave_chunk_load(b.cw, 0u, lane, words_per_lane(__builtin_amdgcn_readfirstlane(b.meta.y)));
        };
        auto process = [&](const Ahead &mine) {
            const int4 meta = mine.meta; const CigarWindow &cw = mine.cw; const WaveChunk &cur = mine.ch;
            const uint32_t n_cig = (uint32_t)__builtin_amdgcn_readfirstlane(meta.y);
            bool bad = false;                            // the read cannot live in the slab: an exon the row word cannot say, the list full
            // (inlined nine times into an issue-bound walk: every instruction here is paid nine times per read.  The slab keeps a length
            //  saturated to 16 bits; whether it fits the row word is looked at once per read behind the walk.)
            auto emit = [&](int k, int s_, int e_) {
                const uint32_t len = (uint32_t)(e_ - s_ + 1);            // (0: an empty exon -- start = end + 1, never further apart)
                if (k < WALK_SLAB) { const int at_ = __mul24(k, slab_w) + (int)q; s_slab_s[at_] = s_; s_slab_l[at_] = (uint16_t)min(len, 0xffffu); }
                else {
                    bad = bad | (len > SLAB_LEN_MAX);
                    const uint32_t at = atomicAdd(&s_ovf_n, 1u);
                    if (at < (uint32_t)WALK_OVF) { s_ovf[3 * at] = (int)q | (k << 8); s_ovf[3 * at + 1] = s_; s_ovf[3 * at + 2] = e_; }
                    else bad = true;
                }
            };
            WaveWalk st{meta.z, meta.z + 1, 0u, false};
#if L2R_WALK_MIXED
            if (n_cig <= (uint32_t)(WCHUNK_SHORT * WAVE)) {
                // (the first six words of both rounds are the same arithmetic: kept apart, or the compiler computes them in front of the
                //  branch and carries a dozen registers into both arms)
                WaveChunk mine = cur;
#pragma unroll
                for (int j = 0; j < WCHUNK_SHORT; ++j) asm volatile("" : "+v"(mine.w[j]));
                wave_chunk_walk<WCHUNK_SHORT>(st, cw, mine, 0u, p, lane, emit);
            } else
#endif
            {
                wave_chunk_walk(st, cw, cur, 0u, p, lane, emit);
                for (uint32_t b_ = ROUND; b_ < n_cig; b_ += ROUND) {             // (reads beyond 512 ops: round by round)
                    const WaveChunk more = wave_chunk_load(cw, b_, lane);
                    wave_chunk_walk(st, cw, more, b_, p, lane, emit);
                }
            }
            bool insane = false;                         // the read's last exon is empty (its first one: looked at in the slab, below)
            if (lane == 0) {
                emit((int)st.n_kept, st.cur_start, st.ref_end);
                insane = st.cur_start > st.ref_end;
                // (starts rise along the read: the last one is the furthest from the tile's base)
                bad = bad | ((uint32_t)(st.cur_start - base) >= SLAB_REL_MASK) | (st.n_kept + 1u > rows_tile);
            }
            const bool bad_any = __any(bad), insane_any = __any(insane);
            // (the read's own words are not read again before the barrier: this wave was their only reader)
            if (lane == 0) { s_rb[q] = (st.n_kept + 1u) | ((bad_any ? 1u : 0u) << 16) | ((insane_any ? 2u : 0u) << 16); s_rc[q] = (uint32_t)st.ref_end; }
        };
        Ahead b0, b1;
        fetch(b0, q);
        for (;;) {
            if (q >= n_act) break;
            fetch(b1, q + TILE_THREADS / WAVE); process(b0); q += TILE_THREADS / WAVE;
            if (q >= n_act) break;
            fetch(b0, q + TILE_THREADS / WAVE); process(b1); q += TILE_THREADS / WAVE;
        }
    }
    __syncthreads();
    uint32_t n = 0u; int el = INT32_MIN; bool outlier = false, sane = true;
    if (active) {
        const uint32_t v = s_rb[threadIdx.x]; n = v & 0xffffu; el = (int)s_rc[threadIdx.x]; outlier = ((v >> 16) & 1u) != 0u;
        // first and last exon not empty (with min_exon >= 1 the kept inner ones never are): the first one's length waits in the slab's row 0
        sane = ((v >> 16) & 2u) == 0u && s_slab_l[threadIdx.x] != 0;
        // an exon of 16 kb or more does not fit the row word: an outlier (the exons behind the LDS slab's rows were tested as they came)
        const int n_chk = min((int)n, WALK_SLAB);
        uint32_t longest = 0u;
        for (int k = 0; k < n_chk; ++k) longest = max(longest, (uint32_t)s_slab_l[k * slab_w + (int)threadIdx.x]);
        outlier = outlier || longest > SLAB_LEN_MAX;
    }
    uint32_t dense_run = 0u;
    if (active && outlier) {
        // an outlier: the literal walk by this lane alone into a run of the dense area (as in k_walk_slab; such reads are rare)
        const uint32_t n_ops = ld32(sa->cig_off32, r + 1u) - c_lo;
        const uint32_t *const words = a->f.cig + c_lo;
        {   WalkState w{pos + 1, pos, 0};
            auto none = [&](int, int, int) {};
            walk_ops<true>(w, words, 0, (int)n_ops, p, none);
            n = (uint32_t)w.n + 1u; }
        dense_run = (uint32_t)atomicAdd(sa->ovf_cursor, (unsigned long long)n);
        int32_t *const ds = sa->dense_start, *const de = sa->dense_end;
        WalkState w{pos + 1, pos, 0};
        sane = true;
        auto put = [&](int k, int s_, int e_) { ds[dense_run + (uint32_t)k] = s_; de[dense_run + (uint32_t)k] = e_; sane = sane & (s_ <= e_); el = e_; };
        walk_ops<true>(w, words, 0, (int)n_ops, p, put);
        put(w.n, w.start, w.end);
    }
    // ---- slots: the tile's reads by falling exon count (outliers and threads without a read last)
    const uint32_t key = (active && !outlier) ? min(n, (uint32_t)(WAVE - 2)) + 1u : (active ? 1u : 0u);
    const uint32_t bin = (uint32_t)(WAVE - 1) - key;
    const uint32_t rank = atomicAdd(&s_hist[bin], 1u);
    s_cnt[threadIdx.x] = active ? n : 0u;
    __syncthreads();
    if (threadIdx.x < (uint32_t)WAVE) { const uint32_t c = s_hist[threadIdx.x]; s_hist[threadIdx.x] = wave_inclusive_scan(c) - c; }
    __syncthreads();
    const uint32_t slot = s_hist[bin] + rank;
    s_slot[threadIdx.x] = (uint8_t)slot;
    s_nslot[slot] = (uint8_t)((active && !outlier) ? n : 0u);
    {   const int m = wave_max(active ? el : INT32_MIN);
        const int wn_all = wave_max(active ? (int)min(n, 0x7fffffffu) : 0);
        if (lane == 0) { s_wmax[wv] = m; s_nmax[wv] = (uint32_t)wn_all; } }
    if (t == 0u && threadIdx.x == 0) {
        uint32_t *const cnt = a->f.redo_count;
        cnt[0] = 0u; cnt[1] = 0u; cnt[2] = 0u;
        uint32_t *const lc = sa->list_cnt;
        lc[0] = 0u; lc[1] = 0u; lc[2] = 0u; lc[3] = 0u; lc[8] = 0u; lc[9] = 0u;
    }
    __syncthreads();
    {   const int wn = wave_max((int)s_nslot[threadIdx.x]);            // (thread = slot here: the rows each wave of the probe kernels has to look at)
        if (lane == 0) s_wn[wv] = (uint32_t)min(wn, 255); }
    uint32_t total;
    {
        const uint4 c4 = reinterpret_cast<const uint4 *>(s_cnt)[lane];
        const uint32_t sum = c4.x + c4.y + c4.z + c4.w;
        const uint32_t inc = wave_inclusive_scan(sum), ex = inc - sum;
        reinterpret_cast<uint4 *>(s_loc)[lane] = make_uint4(ex, ex + c4.x, ex + c4.x + c4.y, ex + c4.x + c4.y + c4.z);     // (the four waves write the same values)
        total = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
    }
    // ---- the read's column: exon k to row k + 1 of its slot, the last one to row 0 (threads of a wave: neighbouring reads, slots all
    //      over the tile's 256 -- every row store of the wave still lands in the row's 1 KB)
    uint32_t *const rows = sa->slab_row;
    const uint32_t off = sbase + slot;
    if (active && !outlier) {
        const int n_slab = min((int)n, WALK_SLAB);
        for (int k = 0; k < n_slab; ++k) {
            const int s0 = s_slab_s[k * slab_w + (int)threadIdx.x];
            st32(rows, off + slab_row((uint32_t)k, n) * SLAB_STRIDE, slab_pack(s0 - base, (uint32_t)s_slab_l[k * slab_w + (int)threadIdx.x]));
        }
    } else if (active) st32(rows, off, dense_run);            // (row 0 of the unused column: where the probe side finds the run)
    {   const uint32_t n_ovf = min(s_ovf_n, (uint32_t)WALK_OVF);
        for (uint32_t i = threadIdx.x; i < n_ovf; i += TILE_THREADS) {
            const uint32_t who = (uint32_t)s_ovf[3 * i] & 0xffu, k = (uint32_t)s_ovf[3 * i] >> 8;
            const uint32_t v = s_rb[who];
            if ((v >> 16) & 1u) continue;                     // (an outlier's exons are in the dense area)
            st32(rows, sbase + (uint32_t)s_slot[who] + slab_row(k, v & 0xffffu) * SLAB_STRIDE, slab_pack(s_ovf[3 * i + 1] - base, (uint32_t)(s_ovf[3 * i + 2] - s_ovf[3 * i + 1] + 1)));
        }
    }
    const bool fat = total >= PL_LOC_LIMIT || max(max(s_nmax[0], s_nmax[1]), max(s_nmax[2], s_nmax[3])) >= PL_N_LIMIT;
    if (active) {
        const uint32_t at = r0 + slot;
        const uint32_t word = threadIdx.x | (rev ? PRE_REV : 0u) | (sane ? 0u : PRE_INSANE) | (outlier ? PRE_DENSE : 0u) | (n << PRE_N_SHIFT);
        if (!fat) sa->pl[at] = word | (s_loc[threadIdx.x] << PL_LOC_SHIFT);
        else { sa->pl[at] = 0u; sa->pre_x[at] = word; sa->loc_x[at] = s_loc[threadIdx.x]; }
    }
    if (threadIdx.x == 0) {
        a->tile_total[t] = total;
        const int32_t tile_hi = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
        const uint32_t rw = s_wn[0] | (s_wn[1] << 8) | (s_wn[2] << 16) | (s_wn[3] << 24);
        reinterpret_cast<int4 *>(sa->span + t)[0] = make_int4(tid0, pos0 + 1, tile_hi, (int)rw);
        reinterpret_cast<int4 *>(sa->span + t)[2] = make_int4((int)r0, (int)n_act, (int)sbase, fat ? 1 : 0);
    }
}

// ---------------------------------------------------------------------------------------------------------- k_describe_scan
// The launch between the walk and the probes, 256-thread workgroups in two roles.
//   Workgroups [0, n_scan): the exclusive scan of the tiles' exon counts (every tile's first slot in the read-order result arrays), a
//       segment of DESCRIBE_SEG counts each.  A workgroup adds up the counts in front of its segment itself (a few 16-byte loads per
//       thread, all in flight: the segments do not wait for each other) -- for the 39 k tiles of 10 M reads that is ten workgroups and
//       about one round trip, where one workgroup took three dependent rounds (13 us).  Inputs beyond DESCRIBE_SCAN_MAX tiles scan
//       with k_scan_u32 in a launch of its own (n_scan = 0).
//   Workgroups [n_scan, ...): the tiles' descriptors and windows, SIXTEEN LANES PER TILE, four tiles per wave (make_descriptor:
//       dictionary slice bounds, window scan from the cursor value k_walk_slab looked up, member headers, masks), written straight
//       into tw[tile] (tw64[tile] for a window of 33 .. 63 members).  With a wave per tile the launch took 31 us for the 39 k tiles:
//       each is a chain of three round trips, and only 8192 waves are resident at a time; with sixteen lanes per tile nearly all
//       tiles are in flight at once.  A workgroup's 16 tiles that belong to k_probe_slab_wide / k_probe_slab_chunked are appended to
//       those kernels' lists with ONE reservation per workgroup and list (no particular order; list_cnt is cleared by k_walk_slab).
__device__ __forceinline__ bool tile_chunk_direct(uint32_t on, uint32_t flags, uint32_t chunk_on, const TileDesc &d, const TileStat &st, uint32_t n_act,
                                                  int min_exon, int min_intron, int max_delet, int dis, int ablate, bool late = false);      // (below, beside tile_wide_direct)
typedef SegScan DescribeScan;
constexpr int DESCRIBE_G = 16;                           // lanes per tile
constexpr int DESCRIBE_TILES = TILE_THREADS / DESCRIBE_G;       // tiles per workgroup
constexpr int DESCRIBE_SEG = SEG_COUNT;                 // counts per scanning workgroup
constexpr int64_t DESCRIBE_SCAN_MAX = (int64_t)DESCRIBE_SEG * 64;    // (262 k tiles = 67 M reads: beyond that the summing in front costs more than it saves)
// FIRST (the one-kernel tile path, l2r_tile.hip.h): this launch is the FIRST kernel of a run -- a tile's span comes from the record the
// upload made (u_rec: chromosome, first base, and the largest read end, which no parameter changes), the head of the descriptor's load
// chain (cursor value, bucket range) is looked up here, the words k_tile's exon counts travel through are cleared, and so are the
// run's counters; no scan role (n_scan = 0: k_tile finds the tiles' first slots itself).
template <bool FIRST>
__global__ __launch_bounds__(TILE_THREADS, 8)
void k_describe_scan(SlabArgs kernarg_block, DescribeScan job, uint32_t n_scan, const TileRec *__restrict__ u_rec)
{
    __shared__ int s_win[DESCRIBE_TILES][64];
    __shared__ uint32_t s_flags[DESCRIBE_TILES];
    __shared__ uint32_t s_tot[DESCRIBE_TILES][2];
    __shared__ uint32_t s_part[2][TILE_THREADS / WAVE];
    static_assert(DESCRIBE_TILES == LB_BLK, "a workgroup's tiles are one block of the exon counts' words");
    (void)kernarg_block;
    if (!FIRST && blockIdx.x < n_scan) { scan_segment(job, blockIdx.x, n_scan, s_part); return; }      // ---- scan role: segment blockIdx.x of the counts
    // ---- describe role
    const SlabArgsK sa = slab_args();
    const PipeArgsK a = pipe_args();
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const int g = lane / DESCRIBE_G, gl = lane % DESCRIBE_G, slot = wv * (WAVE / DESCRIBE_G) + g;
    const uint32_t t0 = (blockIdx.x - (FIRST ? 0u : n_scan)) * (uint32_t)DESCRIBE_TILES;
    const uint32_t t = t0 + (uint32_t)slot;
    uint32_t flags = TD_FAST;                            // (behind the last tile: nobody's)
    unsigned long long my_total = 0ull;                  // FIRST, lane 0 of a tile's group: {1, its exon count} where it is known here
    if (FIRST && blockIdx.x == 0 && threadIdx.x == 0) {
        // the run's counters: redo list, chunk cursor of the accepted list, the outlier area's cursor, the work cursors of the list-driven
        // kernels and k_probe_slab's list.  (The lists of the 64-bit-mask / chunked kernel are appended to by THIS launch: their
        // counts are cleared behind their readers, by k_classify_generic.)
        uint32_t *const cnt = a->f.redo_count;
        cnt[0] = 0u; cnt[1] = 0u; cnt[2] = 0u;
        *sa->ovf_cursor = 0ull; *sa->lb_err = 0u;
        *sa->exon_total = 0u;                            // (k_tile's last tile writes the run's exon count: an upload without reads has none)
        uint32_t *const lc = sa->list_cnt;
        lc[2] = 0u; lc[3] = 0u; lc[4] = 0u; lc[5] = 0u; lc[8] = 0u; lc[9] = 0u; lc[10] = 0u;
        if (sa->list_cnt_next) { sa->list_cnt_next[0] = 0u; sa->list_cnt_next[1] = 0u; }      // (the next run's entry counts: k_describe_scan appends to them from its first workgroup on)
    }
    if (FIRST && blockIdx.x == 0) for (uint32_t i = threadIdx.x; i < sa->n_sup; i += TILE_THREADS) sa->lb_sup_next[i] = 0ull;      // (the next run's super-block words)
    if (t < sa->n_tiles) {
        int4 spv, spw;
        TileStat st_first{0, 0, 0, 0}; uint32_t n_act_first = 0u;      // FIRST: the tile's op statistics and reads (tile_chunk_direct below)
        if (FIRST) {
            // (every load that needs nothing but the tile number leaves together, in front of the cursor's chain of three)
            const int4 r0v = reinterpret_cast<const int4 *>(u_rec + t)[0];             // {r0, n_act, sbase, rows}
            const int4 rv = reinterpret_cast<const int4 *>(u_rec + t)[1];              // {tid0, lo, hi, .}
            const int4 sv = *reinterpret_cast<const int4 *>(sa->tile_stat + t);
            asm volatile("" :: "v"(r0v.y), "v"(rv.x), "v"(sv.x));
            st_first = TileStat{sv.x, sv.y, sv.z, sv.w}; n_act_first = (uint32_t)r0v.y;
            CursorDir cd;
            cd.key = a->cd.key; cd.dir = a->cd.dir; cd.kb_base = a->cd.kb_base; cd.n_tid = a->cd.n_tid; cd.n_tx = a->cd.n_tx;
            const int jl = cursor_value(cd, rv.x, rv.y);
            int tb = 0, nb = 0;
            if (rv.x >= 0 && rv.x < a->n_tid_dir) { tb = a->tid_base[rv.x]; nb = a->tid_base[rv.x + 1] - tb; }
            spv = make_int4(rv.x, rv.y, rv.z, 0); spw = make_int4(jl, tb, nb, 0);
            // the tile's exon count, where no threshold is borderline inside it: known to every later tile from here on
            if (gl == 0) {
                const TileStat st{sv.x, sv.y, sv.z, sv.w};
                const bool exact = tile_exact(st, a->f.p.min_exon, a->f.p.min_intron, a->f.p.max_delet) && !(a->f.p.ablate & 256);
                my_total = exact ? (1ull << LB_SHIFT) | (unsigned long long)(uint32_t)(r0v.y + st.n_ops_n) : 0ull;
                sa->lb_tile[t] = my_total;
            }
        } else { spv = reinterpret_cast<const int4 *>(sa->span + t)[0]; spw = reinterpret_cast<const int4 *>(sa->span + t)[1]; }
        flags = make_descriptor<DESCRIBE_G>(a, gl, g * DESCRIBE_G, spv.x, spv.y, spv.z, sa->tw + t, sa->tw64 ? sa->tw64 + t : nullptr, s_win[slot],
                                            (uint32_t)SLAB_KEY_CAP, spw.x, spw.y, spw.z, (uint32_t)spv.w);
        // (one-kernel tile path: is the tile k_tile_chunk's?  Said once, here, as a bit of the descriptor's flags: the kernels test the bit)
        if (FIRST && gl == 0 && sa->chunk_direct_on != 0u) {
            const TileDesc dd = sa->tw[t].d;                     // (this lane's own stores of a moment ago)
            if (tile_chunk_direct(1u, flags, sa->chunk_on, dd, st_first, n_act_first, a->f.p.min_exon, a->f.p.min_intron, a->f.p.max_delet, a->f.p.ss_dis, a->f.p.ablate)) {
                flags |= TD_CDIRECT;
                sa->tw[t].d.flags = flags;
            }
        }
        if (gl == 0) sa->tile_flags[t] = flags;
        // (accepted list: every tile's chunk is k_gather_accepted's until a probe kernel has left it itself)
        if (gl == 0 && (a->f.p.want & WANT_ACCEPTED)) a->f.tile_chunk[t] = CHUNK_DEFERRED;
    }
    // ---- the workgroup's tiles for the 64-bit-mask and the chunked kernel
    if (gl == 0) s_flags[slot] = flags;
    if (FIRST && gl == 0) { s_tot[slot][0] = (uint32_t)my_total; s_tot[slot][1] = (uint32_t)(my_total >> 32); }
    __syncthreads();
    if (FIRST && wv == 1 && t0 < sa->n_tiles) {
        // the block's word: the sum of the counts known here (k_tile adds the others); a block that is complete goes to its super-block
        const uint32_t lo = lane < DESCRIBE_TILES ? s_tot[lane][0] : 0u, hi = lane < DESCRIBE_TILES ? s_tot[lane][1] : 0u;
        const unsigned long long sum = ((unsigned long long)wave_sum(hi) << 32) + (unsigned long long)wave_sum(lo);      // (an exact tile has fewer than 4096 exons: no carry out of the low words)
        if (lane == 0) {
            sa->lb_blk[t0 >> LB_BLK_SHIFT] = sum;
            // A super-block whose tiles are ALL exact (the upload's sup_stat says so for these thresholds) is written whole by the workgroup
            // of its first block; in any other, every complete block adds itself (64 adds to one word: 8 us of this launch when every
            // block did that).
            const int4 qv = *reinterpret_cast<const int4 *>(sa->sup_stat + (t0 >> LB_SUP_SHIFT));
            const TileStat q{qv.x, qv.y, qv.z, qv.w};
            const bool whole = tile_exact(q, a->f.p.min_exon, a->f.p.min_intron, a->f.p.max_delet) && !(a->f.p.ablate & 256) &&
                               ((t0 >> LB_SUP_SHIFT) + 1u) << LB_SUP_SHIFT <= sa->n_tiles;       // (the last, partial super-block is nobody's "front")
            const uint32_t in_blk = min((uint32_t)LB_BLK, sa->n_tiles - t0);
            if (whole) {
                if ((t0 & ((1u << LB_SUP_SHIFT) - 1u)) == 0u)
                    sa->lb_sup[t0 >> LB_SUP_SHIFT] = ((unsigned long long)(1u << (LB_SUP_SHIFT - LB_BLK_SHIFT)) << LB_SHIFT) | (unsigned long long)(uint32_t)q.n_ops_n;
            } else if ((uint32_t)(sum >> LB_SHIFT) == in_blk) atomicAdd(sa->lb_sup + (t0 >> LB_SUP_SHIFT), (1ull << LB_SHIFT) | (sum & LB_SUM_MASK));
        }
    }
    if (wv == 0) {
        const uint32_t f = lane < DESCRIBE_TILES ? s_flags[lane] : TD_FAST;
        const uint32_t tl = t0 + (uint32_t)lane;
        const bool is_w = (f & TD_WIDE) != 0u, is_c = sa->chunk_on && slab_tile_is_chunked(f);
        const unsigned long long mw = __ballot(is_w), mc = __ballot(is_c);
        if (mw | mc) {
            uint32_t bw = 0u, bc = 0u;
            if (lane == 0) { if (mw) bw = atomicAdd(sa->list_cnt + 0, (uint32_t)__popcll(mw)); if (mc) bc = atomicAdd(sa->list_cnt + 1, (uint32_t)__popcll(mc)); }
            bw = (uint32_t)__builtin_amdgcn_readfirstlane((int)bw); bc = (uint32_t)__builtin_amdgcn_readfirstlane((int)bc);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (is_w) sa->wide_list[bw + (uint32_t)__popcll(mw & below)] = tl;
            if (is_c) sa->chunk_list[bc + (uint32_t)__popcll(mc & below)] = tl;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- k_probe_slab
// A row of a column as loaded: {start, length word}.  The 16-bit length is fetched as the low half of a 4-byte load at its 2-byte
// boundary and masked where the row is READ: a 16-bit load is zero-extended by the compiler where it is issued, i.e. waited for there.
struct SlabRow { uint32_t w; };
__device__ __forceinline__ SlabRow slab_load_row(const uint32_t *__restrict__ xw, uint32_t i) { return SlabRow{ld32(xw, i)}; }
// (start / end are formed where they are used, from the word as loaded: anything done to a loaded register at the place of the
//  load is a wait for that load)
__device__ __forceinline__ int slab_row_start(const SlabRow &r, int32_t base) { return base + (int)(r.w & SLAB_REL_MASK); }
__device__ __forceinline__ uint32_t slab_row_len(const SlabRow &r) { return r.w >> SLAB_REL_BITS; }
__device__ __forceinline__ int slab_row_end(const SlabRow &r, int32_t base) { return slab_row_start(r, base) + (int)slab_row_len(r) - 1; }
#ifndef L2R_SLAB_AHEAD
#define L2R_SLAB_AHEAD 4
#endif
constexpr int SLAB_AHEAD = L2R_SLAB_AHEAD;               // exons of a read in flight (rows k .. k + SLAB_AHEAD - 1 in registers, one word each)
struct SlabRows { SlabRow last, x[SLAB_AHEAD]; };        // a column's row 0 (the last exon) and rows 1 .. 4 as loaded by the kernel's first loads

// The tile's results on their way out: the lanes of k_probe_slab hold one READ each (slot order), the result arrays want the exons
// in read order (exon k of a read at ex_off + k) -- written lane by lane that is one scattered 4-byte store per lane and exon
// (measured: 3 x the whole kernel).  So every exon is left in LDS at its POSITION inside the tile (pos = exons of the tile's reads
// before this one, in read order, + k), in 6 bytes: A[pos] = start relative to the tile's first read (18 bits: a fast tile spans
// less than 384 buckets of 512 bases) below the exon's 14-bit work word (later its flag byte), L[pos] = the 16-bit length.  After
// one barrier the tile's block of the result arrays is written with 16-byte stores, thread j positions 4j .. 4j + 3.
// SLAB_POS_CAP positions fit (with the dictionary slices 22.3 KB per workgroup, 7 workgroups per CU); a read whose exons lie
// behind that or further than 2^18 - 1 bases from the tile's start is written directly and classified by the generic kernel.
// (Measured: 8 bytes per position at 6 workgroups per CU 0.474 ms, this form 0.462; 2112 positions at 8 workgroups per CU and
// 64 VGPRs: 0.594 -- 17 spilled registers and 13 % more tiles.)
constexpr int SLAB_POS_CAP = 2536;                       // (k_probe_slab's LDS = 23040 bytes = 45 granules of 512: 7 workgroups per CU exactly; 2416 before: 1.1 % more tiles)
constexpr int TILE_POS_CAP = 2400;                       // ... of k_tile (l2r_tile.hip.h), which also keeps the reads' counts and places there: what the upload cuts the tiles of short CIGARs by
// One-kernel tile path: a tile of the 64-bit-mask kernel that k_tile's WIDE instance classifies straight from its CIGARs (the plain
// instance returns at once for it, k_probe_slab_wide skips it): exact under the run's thresholds -- its reads' places are in their slot
// records, its exon count is known since k_describe_scan -- and no more exons than the staged positions hold.
__device__ __forceinline__ bool tile_wide_direct(uint32_t on, uint32_t flags, uint32_t chunk_on, const TileStat &st, uint32_t n_act, int min_exon, int min_intron, int max_delet, int ablate)
{
    return on != 0u && (flags & TD_WIDE) != 0u && !(chunk_on && slab_tile_is_chunked(flags)) && tile_exact(st, min_exon, min_intron, max_delet) && !(ablate & 256) &&
           n_act + (uint32_t)st.n_ops_n <= (uint32_t)TILE_POS_CAP;
}
constexpr int TC_ST_CAP = 128, TC_EN_CAP = 512;          // k_tile_chunk (l2r_tchunk.hip.h): START / END dictionary entries of a tile it stages
// The tile is taken by k_tile_chunk (the plain instance returns at once for it, k_probe_slab_chunked skips it).  flags: the tile's
// descriptor flags as k_describe_scan left them (SlabArgs::tile_flags: nobody changes those).
// late: the tile is one a one-window kernel handed on late (a key in several entries; chunk_list_append_late): k_tile_chunk's second
// launch, behind the list kernels, takes those -- whatever their descriptor's flags said.
__device__ __forceinline__ bool tile_chunk_direct(uint32_t on, uint32_t flags, uint32_t chunk_on, const TileDesc &d, const TileStat &st, uint32_t n_act,
                                                  int min_exon, int min_intron, int max_delet, int dis, int ablate, bool late)
{
    return on != 0u && chunk_on != 0u && (late || (slab_tile_is_chunked(flags) && !(flags & TD_CHUNK))) && dis == 0 && d.nbk > 0 && d.st_nk < (uint32_t)TC_ST_CAP && d.en_nk <= (uint32_t)TC_EN_CAP &&      // (START: one slot behind the entries stays free -- the masks of "no entry", tc_map_exons)
          
           tile_exact(st, min_exon, min_intron, max_delet) && !(ablate & 256) && n_act + (uint32_t)st.n_ops_n <= (uint32_t)TILE_POS_CAP;
}

constexpr uint32_t SLAB_POS_SKIP = 0xffffffffu;          // A of a position that was written directly (a staged start is below 2^18 - 1)
struct SlabStage { uint32_t *A; uint16_t *Ln; uint32_t loc; int32_t lo; bool fits; };          // loc: the lane's first position, lo: the tile's first start

// map_exons (l2r_kernels.hip.h) with the read's exons streamed from its slab column: a row is the same exon number for the whole
// wave = coalesced.  Exons k .. k + 3 sit in four register pairs with FIXED roles per unrolled round (no register rotation: a copy
// of a loaded register is a wait for its load).  Round k reads exon k (current) and exon k + 1 (next) and, when it is done with
// exon k, asks for exon k + 4 into exon k's registers: that load has three rounds to arrive.  Every round leaves its exon and
// its work word at the exon's position in LDS (SlabStage).
// DIS: -d > 0 -- every probe looks at the entries within the tolerance of its coordinate (probe_near), `dis` is the tolerance and [rs, re]
// the read's span.
template <bool DIS>
__device__ __forceinline__ SiteMasks map_exons_slab(const TileLds &L, const TileDesc &d, bool mapping, const uint32_t *__restrict__ xw,
                                                    uint32_t off, uint32_t n, uint32_t vpre, const SlabRows &q, const SlabStage &st, int dis = 0, int rs = 0, int re = 0)
{
    SiteMasks m{0xffffffffu, 0u, 0u, 0u, 0u};
    uint32_t *const Ap = st.A + st.loc; uint16_t *const Lp = st.Ln + st.loc;
    // exon j of the read: row j + 1, the last one row 0
    SlabRow R[SLAB_AHEAD];
#pragma unroll
    for (int i = 0; i < SLAB_AHEAD; ++i) R[i] = n == (uint32_t)i + 1u ? q.last : q.x[i];
    const uint32_t none = (uint32_t)d.nbk + 1u;         // a bucket behind the staged ones: the staging leaves it empty
    const int k_max = wave_max(mapping ? (int)n : 0);
    const uint32_t nm1 = mapping ? n - 1u : 0u;         // (lanes that map nothing keep loading their row 0: a valid address, no branch)
    // The bucket ranges of exon k + 1 are looked up while exon k's entries are compared (the directory bytes and the entries
    // are two dependent LDS round trips: one of them per exon is taken off the chain).
    auto buckets = [&](int k, int sv, int ev, uint32_t &ls, uint32_t &hs, uint32_t &le, uint32_t &he) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        if (DIS) { near_range(L.dir0, d.b_off, none, live, sv, dis, ls, hs); near_range(L.dir1, d.b_off, none, junc, ev, dis, le, he); return; }
        const uint32_t is = live ? min((uint32_t)((sv >> SITE_SHIFT) + d.b_off), none) : none;      // (a lane without exon k must not open the long-bucket path)
        const uint32_t ie = junc ? min((uint32_t)((ev >> SITE_SHIFT) + d.b_off), none) : none;
        ls = L.dir0[is]; hs = L.dir0[is + 1u]; le = L.dir1[ie]; he = L.dir1[ie + 1u];
    };
    uint32_t ls, hs, le, he;
    int e_cur = slab_row_end(R[0], st.lo);              // (the end of the current exon: formed once, as the "next" of the round before)
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
            const v4i_t qs0 = lds_entry(L.ent0, ls);
            const v4i_t qe0 = lds_entry(L.ent1, le), qe1 = lds_entry(L.ent1, le + 1u);
            buckets(k + 1, s2, e2, ls_n, hs_n, le_n, he_n);
            {   const bool m0 = ls < hs && qs0.x == s;
                am = m0 ? (uint32_t)qs0.w : 0u; xm = (m0 && qs0.y == e) ? (uint32_t)qs0.z : 0u; }
            probe2(qe0, qe1, le, he, e, s2, jm, dm);
            if (__any(hs > ls + 1u || he > le + 2u)) { probe_rest(L.ent0, ls + 1u, hs, s, e, xm, am, 0u); probe_rest(L.ent1, le + 2u, he, e, s2, jm, dm, 0u); }
        }
        if (reload) {                                   // exon k + SLAB_AHEAD into the registers of exon k
            const uint32_t j = (uint32_t)k + (uint32_t)SLAB_AHEAD;
            cur = slab_load_row(xw, off + (j < nm1 ? j + 1u : 0u) * SLAB_STRIDE);
        }
        const uint32_t amj = junc ? am : 0u;
        uint32_t word = first_member(xm & vpre);
        word |= first_member(jm & vpre) << 6;
        word |= nonzero(dm & vpre) << 12;
        word |= nonzero(amj & vpre) << 13;
        m.kand &= junc ? (am & dm) : 0xffffffffu;     // Q1: the acceptor probed with exon k is ITS OWN start, k < n-1
        m.kor |= amj | dm;
        // (with a tolerance a shared donor / acceptor does not say that the exons overlap: the full-length evidence asks the START slice)
        if (!DIS) { if (k == 0) m.dm_first = dm;
                    m.am_last = (live && !junc) ? am : m.am_last; }
        if (live) { Ap[k] = (cw & SLAB_REL_MASK) | (word << SLAB_REL_BITS); Lp[k] = (uint16_t)(cw >> SLAB_REL_BITS); }      // (a staged read's base is the tile's)
        ls = ls_n; hs = hs_n; le = le_n; he = he_n; e_cur = e2;
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

// The tile's dictionary slices into LDS, re-based to the tile's window (entries loaded by load_dict_slices: one START and one
// END entry per thread, directory words two per thread).  `win`: the window's transcript numbers (gapped windows).
struct SlabLds { uint16_t *W; v4i_t *ent0, *ent1; uint8_t *dir0, *dir1, *rdir; };
__device__ __forceinline__ int slab_stage_dict(const TileDesc &d, const DictRegs &dv, const int *win, const SlabLds &S)
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

// diagnostics (L2R_STAMPS=1): cycles of wave 0 per phase of k_probe_slab summed over the tiles -- [0] loads until the staging,
// [1] staging + barrier, [2] window pass, [3] probe rounds, [4] verdicts, [5] write-out -- and of the last wave: [6] whole kernel,
// [7] probe rounds.  One lane of the two waves stamps; nothing is stamped (one uniform branch per phase) without the buffer.
struct SlabStamp {
    unsigned long long *p; unsigned long long t, t0; int who;       // who: 0 wave 0, 3 last wave, -1 nobody
    __device__ __forceinline__ void start(unsigned long long *buf)
    {
        p = buf; who = -1;
        if (buf && (threadIdx.x & (WAVE - 1)) == 0) { const int wv = threadIdx.x >> 6; who = wv == 0 ? 0 : (wv == TILE_THREADS / WAVE - 1 ? 3 : -1); }
        t = t0 = buf ? __builtin_readcyclecounter() : 0ull;
    }
    __device__ __forceinline__ void mark(int phase)
    {
        if (!p) return;
        const unsigned long long now = __builtin_readcyclecounter();
        if (who == 0) atomicAdd(&p[(blockIdx.x & 1023u) * 8u + (uint32_t)phase], now - t);
        if (who == 3 && phase == 3) atomicAdd(&p[(blockIdx.x & 1023u) * 8u + 7u], now - t);
        if (who == 3 && phase == 5) atomicAdd(&p[(blockIdx.x & 1023u) * 8u + 6u], now - t0);
        t = now;
    }
};

// Exons that no probe round has left in LDS: a slab read that maps nothing (one exon, a read for the generic kernel), or an outlier
// (its exons wait in the dense area).  Into the staged positions when the read fits there, else straight into the result arrays.
// Flags 0: the generic kernel writes them for the reads it takes, a one-exon read's flag follows from its verdict.
struct SlabOut { int32_t *start, *end; uint8_t *flag; uint32_t dst; };     // the read-order arrays, exon k of the lane's read at dst + k
__device__ __forceinline__ void slab_copy_exons(SlabArgsK sa, PipeArgsK a, const SlabOut &out, const SlabStage &st, const SlabRows &q, uint32_t off, uint32_t n,
                                                uint32_t pre, uint32_t r)
{
    // (an outlier's exons may be 16 kb and longer, which the length of a staged position cannot say: written directly, its
    //  positions marked so that the tile's write-out leaves them alone)
    const bool dense = (pre & PRE_DENSE) != 0u;
    auto put = [&](uint32_t k, int s, int e) {
        if (st.fits) { st.A[st.loc + k] = (uint32_t)(s - st.lo); st.Ln[st.loc + k] = (uint16_t)(e - s + 1); }
        else {
            out.start[out.dst + k] = s; out.end[out.dst + k] = e; out.flag[out.dst + k] = 0;
            if (st.loc + k < (uint32_t)SLAB_POS_CAP) st.A[st.loc + k] = SLAB_POS_SKIP;
        }
    };
    if (dense) {
        const uint32_t run = q.last.w;
        for (uint32_t k = 0; k < n; ++k) put(k, sa->dense_start[run + k], sa->dense_end[run + k]);
    } else {
        const int32_t base = st.lo;
        if (n == 1u) put(0u, slab_row_start(q.last, base), slab_row_end(q.last, base));
        else
            for (uint32_t k = 0; k < n; ++k) {
                const SlabRow w = slab_load_row(sa->slab_row, off + slab_row(k, n) * SLAB_STRIDE);
                put(k, slab_row_start(w, base), slab_row_end(w, base));
            }
    }
}

// Verdicts of one tile's reads (slot order) from the staged window + dictionaries: the device functions of the classic kernel
// on slab-resident exons; exons, work words and then flag bytes at their positions in LDS (or, for a read that does not fit
// there, in the result arrays), info / ref_tx / ex_off per read, redo list.
struct SlabVerdict { uint32_t info; int ref; bool redo; };
template <int LEVEL, bool DIS>
__device__ __forceinline__ SlabVerdict slab_classify(SlabArgsK sa, PipeArgsK a, const TileDesc &d, const SlabLds &S, const int4 *hk, const int4 *hx, const int *win,
                                              const uint32_t *tilemask, bool active, uint32_t pre, uint32_t r, uint32_t off, const SlabRows &q,
                                              const ReadEnds &re, const SlabOut &out, const SlabStage &st, int any_wide, SlabStamp &stamp)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const bool fast = (d.flags & TD_FAST) != 0;
    const int w_n = fast ? (int)d.n_win : 0;
    const uint32_t n = pre >> PRE_N_SHIFT;
    const bool outlier = (pre & PRE_DENSE) != 0u, rev_in = (pre & PRE_REV) != 0u;
    const uint32_t *const xw = sa->slab_row;
    // ---- classification (device functions of the classic kernel)
    uint32_t info = n << 8; int ref = -1;
    // (sorted input: a tile is of one chromosome, the descriptor's)
    bool redo = active && (!fast || outlier || !st.fits || any_wide != 0 || (n > 1 && (pre & PRE_INSANE) != 0u));
    const bool work = active && !redo;
    const TileLds L{nullptr, nullptr, nullptr, S.ent0, S.ent1, S.dir0, S.dir1, S.rdir, hk, hx, win};
    const VisitMasks vm = visit_window<LEVEL>(L, d, w_n, work, n, d.j_lo, re, tilemask);
    redo = redo || vm.redo;
    stamp.mark(2);
    const bool mapping = work && !redo && n > 1;
    const SiteMasks sm = map_exons_slab<DIS>(L, d, mapping, xw, off, n, vm.vpre, q, st, DIS ? a->f.p.ss_dis : 0, re.s0, re.el);
    stamp.mark(3);
    if (active && !mapping) slab_copy_exons(sa, a, out, st, q, off, n, pre, r);
    // (-d > 0: a visited member with two sites within the tolerance of one read site -- its pair count is the generic kernel's)
    if (DIS && mapping && (sm.amb & vm.vpre) != 0u) redo = true;
    if (work && !redo) {
        // work words: the upper 14 bits of the read's A words, replaced by the flag byte
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
    if (active) { a->f.info[r] = info; a->f.ref_tx[r] = ref; a->f.ex_off[r] = out.dst; }
    return SlabVerdict{info, ref, redo};
}

// The staged positions [0, lim) of the tile -> the tile's block of the read-order result arrays (first slot xbase): 16-byte stores
// of starts and ends, 4-byte stores of four flag bytes, at whatever alignment xbase has.
__device__ __forceinline__ void slab_write_out(const SlabOut &out0 /* dst = xbase */, const uint32_t *A, const uint16_t *Ln, int32_t lo, uint32_t lim)
{
    for (uint32_t p = threadIdx.x * 4u; p < lim; p += (uint32_t)TILE_THREADS * 4u) {
        const v4i_t a4 = *reinterpret_cast<const v4i_t *>(A + p);
        const uint2 l4 = *reinterpret_cast<const uint2 *>(Ln + p);
        const uint32_t av[4] = {(uint32_t)a4.x, (uint32_t)a4.y, (uint32_t)a4.z, (uint32_t)a4.w};
        const uint32_t lv[4] = {l4.x & 0xffffu, l4.x >> 16, l4.y & 0xffffu, l4.y >> 16};
        v4i_t s4, e4;
        s4.x = lo + (int)(av[0] & SLAB_REL_MASK); s4.y = lo + (int)(av[1] & SLAB_REL_MASK); s4.z = lo + (int)(av[2] & SLAB_REL_MASK); s4.w = lo + (int)(av[3] & SLAB_REL_MASK);
        e4.x = s4.x + (int)lv[0] - 1; e4.y = s4.y + (int)lv[1] - 1; e4.z = s4.z + (int)lv[2] - 1; e4.w = s4.w + (int)lv[3] - 1;
        const uint32_t f4 = ((av[0] >> SLAB_REL_BITS) & 0xffu) | (((av[1] >> SLAB_REL_BITS) & 0xffu) << 8) | (((av[2] >> SLAB_REL_BITS) & 0xffu) << 16) | ((av[3] >> SLAB_REL_BITS) << 24);
        const uint32_t at = out0.dst + p;
        const bool skip = av[0] == SLAB_POS_SKIP || av[1] == SLAB_POS_SKIP || av[2] == SLAB_POS_SKIP || av[3] == SLAB_POS_SKIP;
        if (p + 4u <= lim && !skip) {
            *reinterpret_cast<v4i_a4 *>(out0.start + at) = s4;
            *reinterpret_cast<v4i_a4 *>(out0.end + at) = e4;
            *reinterpret_cast<u32_a1 *>(out0.flag + at) = f4;
        } else {
            const int sv[4] = {s4.x, s4.y, s4.z, s4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i)
                if (p + i < lim && av[i] != SLAB_POS_SKIP) { out0.start[at + i] = sv[i]; out0.end[at + i] = ev[i]; out0.flag[at + i] = (uint8_t)(av[i] >> SLAB_REL_BITS); }
        }
    }
}

// Workgroups per CU: a spilled register is a scratch access, and scratch accesses count on the same in-order counter as the row
// loads (a reload inside the probe rounds waits for every row asked for before it): the kernel has to fit its registers.  Levels 3
// and 4 needed 74 .. 78 and ran at 6 workgroups per CU (0.418 ms against 0.435 at 7 with 32 bytes of scratch) until the reads' ends
// stopped living through the probe rounds (slab_classify reads them back from the staged exons): 72 registers, 7 per CU, 0.371 ms.
constexpr int slab_probe_wgs(int) { return 7; }
// ACC: the caller wants the compacted accepted-novel list and no junction table decides later (update_gtf.c:945-962 with sj_n == 0): a
// tile whose verdicts are all final here (no read on the redo list) leaves its accepted chunk FROM ITS LDS IMAGE -- records in read
// order, exons by position -- instead of k_gather_accepted reading the results back (0.72 GB read to write 0.72 GB on config 3).
// The dictionary slices, directories and the window record are dead behind the classification: the per-read counts (by read number),
// their scan and the slot -> position map of the accepted exons live there.
constexpr int SLAB_DIR_BYTES = (3 * FAST_DIR_BYTES + 15) & ~15;
constexpr int SLAB_AUX_BYTES = SLAB_DIR_BYTES + (int)sizeof(TileWin);
static_assert(SLAB_AUX_BYTES >= 2 * TILE_THREADS * 4, "the accepted counts and their scan (one word per read each) reuse the directories + window record");
static_assert(2 * SLAB_KEY_CAP * 16 >= SLAB_POS_CAP * 2, "the slot -> position map of a tile's accepted exons reuses the staged dictionary entries");
// LIST (behind k_tile, l2r_tile.hip.h): the workgroups share the list_cnt[4] tiles of u_list -- the tiles k_tile left in slab form, few or none;
// else workgroup b takes tile xcd_tile(b).  (A template parameter: as a run-time choice the loop cost the hot form 12 spilled registers.)
template <int LEVEL, bool ACC, bool DIS, bool LIST = false>
__global__ __launch_bounds__(TILE_THREADS, slab_probe_wgs(LEVEL))
void k_probe_slab(SlabArgs kernarg_block, const TileSpan *__restrict__ u_span, const TileWin *__restrict__ u_tw,
                  const uint32_t *__restrict__ u_xbase /* first result slot of every tile: the scanned exon counts (+ the total) */,
                  const uint32_t *__restrict__ u_list /* LIST only */)
{
    constexpr int DIR_BYTES = FAST_DIR_BYTES;
    __shared__ __attribute__((aligned(16))) uint32_t s_A[SLAB_POS_CAP];
    __shared__ __attribute__((aligned(16))) uint16_t s_L[SLAB_POS_CAP];
    __shared__ __attribute__((aligned(16))) v4i_t s_ent[2 * SLAB_KEY_CAP];
    __shared__ __attribute__((aligned(16))) uint8_t s_aux[SLAB_AUX_BYTES];       // directories, then the window record
    __shared__ int s_widew[TILE_THREADS / WAVE];
    __shared__ uint32_t s_redow[TILE_THREADS / WAVE];
    __shared__ uint32_t s_lim, s_chunk[2];
    uint8_t *const s_dir = s_aux;
    TileWin &s_tw = *reinterpret_cast<TileWin *>(s_aux + SLAB_DIR_BYTES);
    (void)kernarg_block;
    const SlabArgsK sa = slab_args();
    const PipeArgsK a = pipe_args();
    auto one_tile = [&](const uint32_t t) {
    SlabStamp stamp; stamp.start(a->f.stamps);
    __builtin_amdgcn_s_setprio(1);                              // (issue priority above the waves that are in their classification: see k_tile, l2r_tile.hip.h)
    // The tile's reads, slab and first base from ONE record of k_walk_slab, its descriptor, its first result slot: every scalar load
    // of the prologue is asked for before the first one is waited for (the empty asm "uses" them all here: the compiler would sink
    // some of them behind the early returns below, i.e. into a second and a third round trip).
    const TileSpan sp = u_span[t];
    const TileDesc d = u_tw[t].d;
    const uint32_t xbase = u_xbase[t], xnext = u_xbase[t + 1u];
    const uint32_t chunk_on = sa->chunk_on; const int32_t ablate = a->f.p.ablate;
    asm volatile("" :: "s"(chunk_on), "s"(ablate), "s"(sp.lo), "s"(sp.rows), "s"(sp.r0), "s"(sp.n_act), "s"(sp.sbase), "s"(sp.fat), "s"(xbase), "s"(xnext),
                       "s"(d.j_lo), "s"(d.b_off), "s"(d.nb), "s"(d.b0), "s"(d.nbk), "s"(d.st_r0), "s"(d.st_nk), "s"(d.en_r0), "s"(d.en_nk), "s"(d.flags), "s"(d.n_win));
    const uint32_t r0 = sp.r0, n_act = sp.n_act, sbase = sp.sbase, total = xnext - xbase, rows_w = sp.rows;
    const int32_t tile_lo = sp.lo;                               // the base of the tile's row words: its first read's first base (coordinate-sorted records)
    v4i_t *const s_ent0 = s_ent, *const s_ent1 = s_ent + SLAB_KEY_CAP;
    uint8_t *const s_dir0 = s_dir, *const s_dir1 = s_dir + DIR_BYTES, *const s_rdir = s_dir + 2 * DIR_BYTES;
    if (!LIST && t == 0u && threadIdx.x == 0) *sa->ovf_cursor = 0ull;      // (k_walk_slab is done: the outlier area's cursor is cleared for the next run)
    if (d.flags & TD_WIDE) return;                               // k_probe_slab_wide takes the tile
    // no window record fits the tile's window, or its dictionary slices do not fit the staging here: k_probe_slab_chunked takes it,
    // 63 members at a time, with the entries that matter for them (it finds the tile by these flags)
    if (chunk_on && slab_tile_is_chunked(d.flags)) return;
    // the rows any read of this wave has (k_walk_slab): rows behind them are not asked for
    // thread -> slot: the slot groups (k_walk_slab: by falling CIGAR length, group 0 = the tile's longest reads) are rotated over
    // the waves by a hash of the tile number (L2R_ABLATE bit 3: off)
    const uint32_t rot = (ablate & 8) ? 0u : ((t ^ (t >> 3) ^ (t >> 7)) & 3u);
    const uint32_t slot = (threadIdx.x + (rot << 6)) & (uint32_t)(TILE_THREADS - 1);
    const uint32_t row_max = max((rows_w >> (8u * (uint32_t)__builtin_amdgcn_readfirstlane((int)(slot >> 6)))) & 0xffu, 1u) - 1u;
    const DictRegs dv = load_dict_slices(a, d);
    int4 twv = make_int4(0, 0, 0, 0);
    if ((int)threadIdx.x < SLAB_TW_VECS && tw_vec_used((int)threadIdx.x, (d.flags & TD_FAST) ? d.n_win : 0u)) twv = reinterpret_cast<const int4 *>(u_tw + t)[threadIdx.x];
    const bool active = slot < n_act;
    const uint32_t at = r0 + (active ? slot : 0u);
    uint32_t pre = 0u, loc = 0u, plw = 0u;
    const uint32_t *const xw = sa->slab_row;
    const uint32_t off = sbase + slot;
    SlabRows q;
    q.last = SlabRow{0u};
#pragma unroll
    for (int i = 0; i < SLAB_AHEAD; ++i) q.x[i] = SlabRow{0u};
    if (active) {
        plw = ld32(sa->pl, at);
        if (sp.fat) { pre = ld32(sa->pre_x, at); loc = ld32(sa->loc_x, at); }
        q.last = slab_load_row(xw, off);
#pragma unroll
        for (int i = 0; i < SLAB_AHEAD; ++i) q.x[i] = slab_load_row(xw, off + min((uint32_t)i + 1u, row_max) * SLAB_STRIDE);      // (an outlier's column holds nothing: read, not used)
    }
    if (threadIdx.x == 0) s_lim = min(total, (uint32_t)SLAB_POS_CAP);
    if (!sp.fat) { pre = plw & PL_PRE_MASK; loc = plw >> PL_LOC_SHIFT; }
    const uint32_t n = pre >> PRE_N_SHIFT;
    const uint32_t r = r0 + (pre & 0xffu);
    const SlabRow first = n == 1u ? q.last : q.x[0];
    const ReadEnds re{slab_row_start(first, tile_lo), slab_row_end(first, tile_lo), slab_row_start(q.last, tile_lo), slab_row_end(q.last, tile_lo)};      // (not of an outlier: those are not classified here)
    const SlabOut out{a->f.ex_start, a->f.ex_end, a->f.ex_flag, xbase + loc};
    if (stamp.p) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp.mark(0);
    // ---- stage window and dictionary slices, re-based to the tile's window
    if ((int)threadIdx.x < SLAB_TW_VECS) reinterpret_cast<int4 *>(&s_tw)[threadIdx.x] = twv;
    const SlabLds S{nullptr, s_ent0, s_ent1, s_dir0, s_dir1, s_rdir};
    // (the window's transcript numbers: straight from the loaded vectors' home, they are not in LDS yet)
    const int my_wide = slab_stage_dict(d, dv, reinterpret_cast<const int *>(u_tw[t].win), S);
    {   // one barrier: every wave leaves its "some entry has members beyond its 64-bit masks" word, all read the four
        const int w_any = __any(my_wide) ? 1 : 0;
        if ((threadIdx.x & (WAVE - 1)) == 0) s_widew[threadIdx.x >> 6] = w_any;
    }
    __syncthreads();
    const int any_wide = s_widew[0] | s_widew[1] | s_widew[2] | s_widew[3];
    if (any_wide && sa->chunk_on) {
        // a key of the staged slices has several entries (its transcripts lie more than 64 apart): k_probe_slab_chunked ORs them
        if (threadIdx.x == 0) { sa->tw[t].d.flags = d.flags | TD_CHUNK; chunk_list_append_late(sa, t); }
        return;
    }
    // a read is staged when its positions fit and its rows have the tile's base
    const SlabStage st{s_A, s_L, loc, tile_lo, loc + n <= (uint32_t)SLAB_POS_CAP && !(pre & PRE_DENSE)};
    // the first read that does not fit the staged positions ends the block that is written from LDS (reads are in read order there)
    if (active && loc + n > (uint32_t)SLAB_POS_CAP) atomicMin(&s_lim, loc);      // (a read that is written directly for another reason marks its positions instead)
    stamp.mark(1);
    __builtin_amdgcn_s_setprio(0);
    const SlabVerdict vd = slab_classify<LEVEL, DIS>(sa, a, d, S, s_tw.hk, s_tw.hx, s_tw.win, s_tw.mask, active, pre, r, off, q, re, out, st, any_wide, stamp);
    __builtin_amdgcn_s_setprio(1);
    if (ACC) { const int w_redo = __any(vd.redo) ? 1 : 0; if ((threadIdx.x & (WAVE - 1)) == 0) s_redow[threadIdx.x >> 6] = (uint32_t)w_redo; }
    __syncthreads();
    if (!ACC) {
        slab_write_out(SlabOut{out.start, out.end, out.flag, xbase}, s_A, s_L, tile_lo, s_lim);
        stamp.mark(5);
        return;
    }
    // ---- the tile's accepted chunk (see above).  Not fused: the tile stays CHUNK_DEFERRED (k_describe_scan) for k_gather_accepted.
    const bool fused = (s_redow[0] | s_redow[1] | s_redow[2] | s_redow[3]) == 0u && !(ablate & 2);
    if (!fused) { slab_write_out(SlabOut{out.start, out.end, out.flag, xbase}, s_A, s_L, tile_lo, s_lim); return; }
    uint32_t *const s_racc = reinterpret_cast<uint32_t *>(s_aux), *const s_rscan = s_racc + TILE_THREADS;
    uint16_t *const s_map = reinterpret_cast<uint16_t *>(s_ent);
    const int lane = threadIdx.x & (WAVE - 1);
    const bool acc = active && (vd.info & I_ACCEPT) != 0u;
    // accepted reads (high half) and their exons (low half), by read number inside the tile
    if (active) s_racc[pre & 0xffu] = acc ? ((1u << 16) | n) : 0u;
    if (threadIdx.x >= n_act) s_racc[threadIdx.x] = 0u;
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
    const uint32_t mine = acc ? s_rscan[pre & 0xffu] : 0u;          // records / exons of the tile's accepted reads in front of this one
    if (acc) for (uint32_t k = 0; k < n; ++k) s_map[(mine & 0xffffu) + k] = (uint16_t)(loc + k);
    slab_write_out(SlabOut{out.start, out.end, out.flag, xbase}, s_A, s_L, tile_lo, s_lim);
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
        AccRec rec; rec.read_lo = (uint32_t)gidx; rec.read_hi = (uint32_t)(gidx >> 32); rec.info = vd.info; rec.ref_tx = vd.ref;
        a->f.acc_rec[rslot] = rec;
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
    if (!LIST) {
        const uint32_t t = xcd_tile(blockIdx.x, gridDim.x);
        if (t < sa->n_tiles) one_tile(t);
    } else {
        const uint32_t n_list = sa->list_cnt[4];
        for (uint32_t wi = blockIdx.x; wi < n_list; wi += gridDim.x) {
            one_tile(u_list[wi]);
            __syncthreads();                             // (the next tile of this workgroup overwrites the LDS image)
        }
    }
}

}  // namespace l2r
