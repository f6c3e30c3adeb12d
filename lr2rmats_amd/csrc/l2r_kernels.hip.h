// l2r_kernels.hip.h -- gfx950 device code of the read-vs-annotation path.
//
// All int32 interval arithmetic, no contraction -> no MFMA.  A TILE is up to 256 consecutive alignment
// records of one chromosome (tile_first, built at upload) handled by one 256-thread workgroup (4 wave64), one
// thread per record.
//
//   k_pass_a           CIGAR -> exon count per record, cursor value per record, tile sums, the
//                      tile's reads ordered by exon count, and a TILE DESCRIPTOR: the slice of the site
//                      dictionaries and the window of annotation transcripts (copies of their headers) the tile
//                      is going to need, so that the classification kernel can issue every one of its loads when
//                      it starts.  Long-CIGAR inputs: also the exons themselves, tile-compact (the CIGAR is read once)
//   k_scan_u32         exclusive scan of per-tile sums (one workgroup per array)
//   k_classify_fast    persistent, software pipelined.  CIGAR -> exons into an LDS tile; dictionary slices and
//                      transcript window staged in LDS, re-based into the tile's own transcript frame; per read:
//                      visit mask over the window, one START and one END dictionary probe per exon, known /
//                      known-site / flags from 32-bit membership masks; coalesced write-out of the per-read
//                      results AND of the tile's chunk of the accepted list (records + exons, compacted in LDS).
//                      Anything it cannot decide exactly goes to a redo list.
//   k_classify_generic wave per listed read, literal loops of the reference (any -d, any annotation)
//   k_validate_sj      short-read junction support for accepted candidates
//   k_count_accepted / k_gather_accepted   accepted-list chunks of the tiles the classification kernel could not
//                      finish itself (redo reads, junction table)
//   k_count_cut_ops    upload-time sizing of the exon buffers for long CIGARs
//
// Semantics follow the reference line by line where bytes of the graded outputs depend on it; each device
// function cites the lines it restates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace l2r {

typedef int v4i_t __attribute__((ext_vector_type(4)));                    // (LDS copy of a dictionary entry {k1, k2, pm, sm}, masks in the tile frame)
typedef int v4i_a4 __attribute__((ext_vector_type(4), aligned(4)));       // a 16-byte access at any 4-byte boundary (global memory)
typedef uint32_t v3u_a4 __attribute__((ext_vector_type(3), aligned(4)));   // a 12-byte access at any 4-byte boundary (global memory)
typedef uint32_t u32_a1 __attribute__((aligned(1)));                      // a 4-byte access at any address (global memory)

constexpr int TILE_THREADS = 256;
constexpr int WAVE = 64;
constexpr int LDS_EXON_CAP = 3072;      // exons of one tile staged in LDS (10 B each)
constexpr int DIR_CAP = 384;            // 512-bp buckets staged per tile (span up to ~196 kb)
constexpr int KEY_CAP = 224;            // dictionary entries staged per dictionary and tile (threads 224..255 stage the headers)
constexpr int WIN_TX = 32;              // annotation transcripts in a tile's window = bits of a tile-frame membership mask
constexpr int SITE_SHIFT = 9;
constexpr int DIS_MASK_MAX = 64;        // largest -d the mask kernels take (probe_near: a probe then looks at two 512-bp buckets at most)

// One annotation transcript (file order), 48 B = three 16-byte loads.
struct TxHdr {
    int32_t tid, start, end, ex_off;          // h0
    int32_t n, rev, flags, pad;               // h1
    int32_t s0, e0, sl, el;                   // h2  first and last exon
};
constexpr int TX_MONO = 1;      // exon starts and ends strictly increasing
constexpr int TX_COMPACT = 2;   // TX_MONO, start <= end for every exon, header span == exon span, tid >= 0, n >= 2

// Site dictionaries (built once per annotation on the host, multi-exon transcripts with tid >= 0 only).
//   START dictionary: the distinct exons sorted by (tid, start, end).  pm = transcripts that contain the exon,
//                     sm = transcripts in which `start` begins a non-first exon (an acceptor);
//   END dictionary:   the distinct junctions sorted by (tid, end, next start).  pm = transcripts that contain
//                     the junction, sm = transcripts in which `end` closes a non-last exon (a donor).
// Masks are 64 bits relative to tx_base (bit b = transcript tx_base + b, file order); SE_WIDE when a member does
// not fit.  Each dictionary has a directory over 512-bp coordinate buckets: dir[tid_base[tid] + (k1 >> 9)] = first
// entry of the bucket.  The reads of a tile (one locus) need a handful of buckets and entries; the classification
// kernel stages them in LDS with the masks re-based to the tile's transcript window.
struct SiteEnt {
    int32_t k1, k2, tx_base, flags;
    uint32_t pm[2], sm[2];
};
constexpr int SE_WIDE = 1;
struct SiteDict {
    const SiteEnt *ent;
    const uint32_t *dir;       // bucket -> first entry; one extra word closes the last bucket
    const uint32_t *rdir;      // START only: bucket -> first entry whose exon reaches into the bucket (<= dir[bucket])
};
struct SiteTabs {
    SiteDict st, en;           // START and END dictionaries
    const int32_t *tid_base;   // [n_tid + 1] first bucket of every tid (one bucket grid for both)
    int32_t n_tid;
};

// Cursor directory: the prefix-max keys of the annotation with the same kind of 512-bp bucket
// directory: dir[kb_base[tid] + c] = first j with key_j >= (tid, c << 9).
struct CursorDir {
    const int64_t *key;        // [n_tx] non-decreasing
    const uint32_t *dir;
    const int32_t *kb_base;    // [n_tid + 1]
    int32_t n_tid, n_tx;
};

// What k_pass_a leaves for every tile.
struct TileDesc {
    int32_t j_lo;              // first transcript of the tile's window (no member: the tile's smallest cursor value)
    int32_t tid;               // the tile's chromosome = the one of its first read (other reads: generic kernel)
    int32_t b_off;             // bucket of x in the staged slice = (x >> 9) + b_off
    int32_t nb;                // buckets the annotation has on this chromosome
    int32_t b0, nbk;           // first staged bucket (absolute), number of staged buckets
    uint32_t st_r0, st_nk;     // START entries [st_r0, st_r0 + st_nk)
    uint32_t en_r0, en_nk;     // END entries
    uint32_t flags, n_win;     // n_win: transcripts in the tile's window (win_hdr[tile * WIN_TX + 0 .. n_win))
};
constexpr uint32_t CHUNK_DEFERRED = 0xffffffffu;   // tile_chunk: the tile's accepted exons are compacted by k_gather_accepted
constexpr uint32_t TD_FAST = 1;    // exons fit the LDS tile, dictionary slices fit DIR_CAP / KEY_CAP, window fits WIN_TX
constexpr uint32_t TD_WALKED = 4;  // long-CIGAR input: pass A has left the tile's exons in `walked` (tile * LDS_EXON_CAP + in-tile offset)
constexpr uint32_t TD_WIDE = 8;    // slab pipeline: the window holds 33 .. 63 transcripts (l2r_wide.hip.h takes the tile), TD_FAST is not set
constexpr uint32_t TD_CHUNK = 16;  // slab pipeline: set by k_probe_slab / k_probe_slab_wide on a tile that stages a dictionary key in several entries: k_probe_slab_chunked takes it
constexpr uint32_t TD_CDIRECT = 32; // one-kernel tile path: k_tile_chunk (l2r_tchunk.hip.h) takes the tile from its CIGARs -- decided by k_describe_scan<true> (tile_chunk_direct), so that k_tile's plain instance only tests this bit
constexpr uint32_t TD_CONTIG = 2;  // the window's transcripts are consecutive in the annotation: j_lo, j_lo + 1, ...
constexpr int WIN_SCAN_TRIPS = 64; // pass A looks at up to 64 * WIN_SCAN_TRIPS transcripts for a tile's window

struct DevParams {
    int32_t min_exon, min_intron, max_delet, ss_dis;
    int32_t full_level, use_multi, min_sj_cnt, split_trans;
    float   frac;
    int32_t n_tx, n_sj, reads_per_tile;
    int32_t ablate;          // diagnostics (env L2R_ABLATE): 1 = every read through the generic kernel
    int32_t want;            // l2r_set_outputs: bit 0 per-read results, bit 1 the compacted accepted list
};
constexpr int32_t WANT_RESULTS = 1, WANT_ACCEPTED = 2;

// info / exon-flag bit layout: keep in sync with include/lr2rmats_hip.h
constexpr uint32_t I_KNOWN = 1u, I_KSITE = 2u, I_FULL = 4u, I_REV = 8u, I_UNREL = 16u,
                   I_SJCHK = 32u, I_SJPASS = 64u, I_ACCEPT = 128u;
constexpr uint8_t F_EXON = 1, F_DON = 2, F_ACC = 4, F_JUNC = 8, F_UNREL = 16;

__device__ __forceinline__ int64_t pack_key(int32_t tid, int32_t x)
{
    return ((int64_t)(tid + 1) << 32) | (uint32_t)x;
}

// ------------------------------------------------------------------ wave / block primitives

// Wave-wide scans and reductions on the DPP data path (no LDS round trips): shifts inside the 16-lane rows, then the
// last lane of a row is added to the following row(s).  All 64 lanes must be active.
struct OpAdd { static constexpr int identity = 0; static __device__ __forceinline__ int apply(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); } };
struct OpMin { static constexpr int identity = INT32_MAX; static __device__ __forceinline__ int apply(int a, int b) { return min(a, b); } };
struct OpMax { static constexpr int identity = INT32_MIN; static __device__ __forceinline__ int apply(int a, int b) { return max(a, b); } };

template <typename Op, int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_step(int v)
{
    return Op::apply(v, __builtin_amdgcn_update_dpp(Op::identity, v, CTRL, ROW_MASK, 0xf, false));
}
template <typename Op>
__device__ __forceinline__ int wave_scan(int v)                 // inclusive
{
    v = dpp_step<Op, 0x111, 0xf>(v);        // row_shr:1
    v = dpp_step<Op, 0x112, 0xf>(v);        // row_shr:2
    v = dpp_step<Op, 0x114, 0xf>(v);        // row_shr:4
    v = dpp_step<Op, 0x118, 0xf>(v);        // row_shr:8
    v = dpp_step<Op, 0x142, 0xa>(v);        // row_bcast:15 into rows 1 and 3
    v = dpp_step<Op, 0x143, 0xc>(v);        // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) { return (uint32_t)wave_scan<OpAdd>((int)v); }
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane(wave_scan<OpAdd>((int)v), WAVE - 1); }
__device__ __forceinline__ int wave_min(int v) { return __builtin_amdgcn_readlane(wave_scan<OpMin>(v), WAVE - 1); }
__device__ __forceinline__ int wave_max(int v) { return __builtin_amdgcn_readlane(wave_scan<OpMax>(v), WAVE - 1); }

// Exclusive scan over the 256 threads of a workgroup; returns the exclusive prefix, `total` = workgroup sum.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_wave /*[4]*/, uint32_t &total)
{
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == WAVE - 1) s_wave[w] = inc;
    __syncthreads();
    const uint32_t w0 = s_wave[0], w1 = s_wave[1], w2 = s_wave[2], w3 = s_wave[3];
    const uint32_t base = (w > 0 ? w0 : 0) + (w > 1 ? w1 : 0) + (w > 2 ? w2 : 0);
    total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + inc - v;
}

// ------------------------------------------------------------------ CIGAR -> exons

// src/bam2gtf.c:31-78 gen_exon.  emit(k, start, end) is called for every exon kept.
// CIGAR words are fetched eight (then four) at a time so that the loads of one read are in flight together; the state update
// is written without short-circuit logic so that it compiles to selects (one predicated region per op: the emit).
// WIDE: the words come from HBM, one read per lane; a lane then fetches 16 words (a whole 64-byte sector) at a time,
// otherwise every 16-byte load of a long CIGAR drags in a sector of its own (ONT reads: hundreds of ops).
struct WalkState { int start, end, n; };

template <typename Emit>
__device__ __forceinline__ void walk_step(WalkState &w, uint32_t c, const DevParams &p, Emit &emit)
{
    const int len = (int)(c >> 4);
    const uint32_t op = c & 0xfu;
    // N (3) cuts at len >= min_intron, D (2) at len > max_delet; M,=,X,N,D advance the reference
    const bool cut = ((op == 3u) & (len >= p.min_intron)) | ((op == 2u) & (len > p.max_delet));
    const bool keep = cut & ((w.n == 0) | (w.end - w.start + 1 >= p.min_exon));
    if (keep) emit(w.n, w.start, w.end);
    w.n += keep ? 1 : 0;
    w.start = cut ? w.end + len + 1 : w.start;
    w.end += len & __builtin_amdgcn_sbfe(0x18d, op, 1u);         // ops 0 2 3 7 8 advance: bit `op` of 0x18d as 0 / -1
}

// ops [k, n_cig) of a read
template <bool WIDE, typename Ptr, typename Emit>
__device__ __forceinline__ void walk_ops(WalkState &w, Ptr cig, int k, int n_cig, const DevParams &p, Emit &emit)
{
    if (WIDE) {
        for (; k + 16 <= n_cig; k += 16) {
            uint32_t c[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u] = cig[k + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) walk_step(w, c[u], p, emit);
        }
    }
    for (; k + 8 <= n_cig; k += 8) {
        const uint32_t c0 = cig[k], c1 = cig[k + 1], c2 = cig[k + 2], c3 = cig[k + 3];
        const uint32_t c4 = cig[k + 4], c5 = cig[k + 5], c6 = cig[k + 6], c7 = cig[k + 7];
        walk_step(w, c0, p, emit); walk_step(w, c1, p, emit); walk_step(w, c2, p, emit); walk_step(w, c3, p, emit);
        walk_step(w, c4, p, emit); walk_step(w, c5, p, emit); walk_step(w, c6, p, emit); walk_step(w, c7, p, emit);
    }
    for (; k + 4 <= n_cig; k += 4) {
        const uint32_t c0 = cig[k], c1 = cig[k + 1], c2 = cig[k + 2], c3 = cig[k + 3];
        walk_step(w, c0, p, emit); walk_step(w, c1, p, emit); walk_step(w, c2, p, emit); walk_step(w, c3, p, emit);
    }
    for (; k < n_cig; ++k) walk_step(w, cig[k], p, emit);
}

template <bool WIDE, typename Ptr, typename Emit>
__device__ __forceinline__ int walk_cigar(Ptr cig, int n_cig, int pos0, const DevParams &p, Emit emit)
{
    WalkState w{pos0 + 1, pos0, 0};
    walk_ops<WIDE>(w, cig, 0, n_cig, p, emit);
    emit(w.n, w.start, w.end);
    return w.n + 1;
}

// The first WALK_HEAD ops of a read, fetched ahead of use (pass A issues them together with its cursor lookup).
// Words behind the read's last op become "I, length 0", which changes nothing (the array is padded by four words).
constexpr int WALK_SLAB = 15;             // long-CIGAR pass A: exons per read staged in LDS before they go to HBM (6 bytes each)
constexpr int WALK_OVF = 256;             // ... and exons beyond that, per tile
constexpr int WALK_SENTINEL = INT32_MIN;    // first slot of a read whose exons pass A could not hand over
constexpr int WALK_HEAD = 16;
struct CigarHead { uint32_t c[WALK_HEAD]; };
__device__ __forceinline__ CigarHead load_cigar_head(const uint32_t *__restrict__ cig, int n_cig)
{
    CigarHead h;
#pragma unroll
    for (int q = 0; q < WALK_HEAD / 4; ++q) {
        uint32_t c0 = 1u, c1 = 1u, c2 = 1u, c3 = 1u;
        if (4 * q < n_cig) { c0 = cig[4 * q]; c1 = cig[4 * q + 1]; c2 = cig[4 * q + 2]; c3 = cig[4 * q + 3]; }
        h.c[4 * q] = c0; h.c[4 * q + 1] = 4 * q + 1 < n_cig ? c1 : 1u; h.c[4 * q + 2] = 4 * q + 2 < n_cig ? c2 : 1u; h.c[4 * q + 3] = 4 * q + 3 < n_cig ? c3 : 1u;
    }
    return h;
}
template <bool WIDE, typename Emit>
__device__ __forceinline__ int walk_cigar_headed(const CigarHead &h, const uint32_t *__restrict__ cig, int n_cig, int pos0, const DevParams &p, Emit emit)
{
    WalkState w{pos0 + 1, pos0, 0};
#pragma unroll
    for (int u = 0; u < WALK_HEAD; ++u) walk_step(w, h.c[u], p, emit);
    if (n_cig > WALK_HEAD) walk_ops<WIDE>(w, cig, WALK_HEAD, n_cig, p, emit);
    emit(w.n, w.start, w.end);
    return w.n + 1;
}

// ONE READ walked by ONE WAVE (long CIGARs: ONT-like reads, hundreds of ops): src/bam2gtf.c:31-78 as a scan over the op stream.
// Lane L takes eight (six: a read of up to 384 ops, one round) consecutive CIGAR words per round (two 16-byte loads; the wave reads
// 2 KB of the stream at a time, coalesced -- a lane per read fetches a 64-byte sector of its own per load).  A round:
//   reference end before every op     = pos + sum of the reference-consuming lengths in front of it    (wave prefix sum + carry)
//   a cut op (N >= min_intron, D > max_delet) closes the candidate exon [start behind the previous cut, end before this op];
//   "start behind the previous cut"    = running maximum of end + len + 1 over the cut ops in front      (wave prefix max + carry)
//   the candidate is kept iff it is the read's first or min_exon long (a dropped one is skipped, not merged: Q4), its exon
//   number = kept candidates in front  (two ballots + carry: a lane keeps the first and the last cut op of its words, see
//   wave_chunk_try; a lane with three is walked again two words per lane); the cut ops are found with a wave ballot.
// emit(k, start, end) is called by the lane that holds the cut op.  The caller keeps the first round of the wave's NEXT read in
// flight while this one is walked (wave_chunk_load / wave_chunk_walk): a wave on its own has one round trip to HBM per round.
constexpr int WCHUNK = 8;                                // CIGAR words per lane and round
struct WaveChunk { uint32_t w[WCHUNK]; };
struct WaveWalk { int ref_end, cur_start; uint32_t n_kept; bool seen_cut; };    // (wave-uniform carries of one read)

// A read's CIGAR as a buffer of its own (the four descriptor words are wave-uniform: one wave walks one read): a word behind the
// read's last op is outside the buffer and the hardware's range check loads it as 0 -- "M, length 0", an op that neither advances
// the reference nor cuts -- so the walk needs neither a branch around the loads nor a mask per word.
typedef uint32_t v4u_buf __attribute__((ext_vector_type(4)));
struct CigarWindow { __amdgpu_buffer_rsrc_t rs; };
__device__ __forceinline__ CigarWindow cigar_window(const uint32_t *__restrict__ cig, uint32_t first_word, uint32_t n_cig)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)first_word);
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(n_cig, 0x3fffffffu));
    CigarWindow w;
    w.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(cig + lo), (short)0, (int)(n * 4u), 0x00020000);
    return w;
}

// words [base + W lane, + 8) of a read's CIGAR (a walk of W < 8 words per lane leaves the last ones alone: they are the next lane's)
__device__ __forceinline__ WaveChunk wave_chunk_load(const CigarWindow &cw, uint32_t base, int lane, int W = WCHUNK)
{
    WaveChunk k;
    // (the round's first word goes into the per-lane offset, not into the instruction's scalar offset: the range check that zeroes the
    //  words behind the read's last op is then the one of offset against num_records whatever the scalar offset's part in it is)
    const int voff = W * 4 * lane + (int)(base * 4u);
    const v4u_buf x = __builtin_amdgcn_raw_buffer_load_b128(cw.rs, voff, 0, 0);
    const v4u_buf y = __builtin_amdgcn_raw_buffer_load_b128(cw.rs, voff + 16, 0, 0);
    k.w[0] = x.x; k.w[1] = x.y; k.w[2] = x.z; k.w[3] = x.w;
    k.w[4] = y.x; k.w[5] = y.y; k.w[6] = y.z; k.w[7] = y.w;
    return k;
}

// W words per lane: a read of up to 64 W ops is walked in one round, and every word less is a ninth of the round's instructions less
constexpr int WCHUNK_SHORT = 6;
// One round.  A lane needs the first and the last cut op among its words only -- the reference bases in front of each and their
// lengths, picked up word by word -- as long as no lane holds more than two (three take two micro-exons next to each other): no
// per-word arrays, and the exon numbers come from two ballots.  Returns false, with `st` untouched, when a lane holds three.
// (a lane's pick between two values by its bit of a wave-wide mask: the cut tests below are kept as masks in scalar registers, where
//  "a second cut", "a third cut" are two scalar instructions per word; as per-lane booleans the compiler makes them vector arithmetic)
__device__ __forceinline__ int pick(unsigned long long mask, int if_clear, int if_set)
{
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}

template <int WCHUNK, typename Emit>
__device__ __forceinline__ bool wave_chunk_try(WaveWalk &st, const WaveChunk &ch, const DevParams &p, int lane, Emit &emit)
{
    int A = 0, ab1 = 0, ab2 = 0, l1 = 0, l2 = 0;
    unsigned long long any_c = 0ull, two = 0ull, three = 0ull;               // lanes with one, two, three cut ops so far
#pragma unroll
    for (int j = 0; j < WCHUNK; ++j) {
        const uint32_t w = ch.w[j];                                          // (behind the read's last op: 0, see cigar_window)
        const uint32_t op = w & 0xfu;
        const int len = (int)(w >> 4);
        const int a = len & __builtin_amdgcn_sbfe(0x18d, op, 1u);            // ops 0 2 3 7 8 advance the reference
        const unsigned long long c = (__ballot(op == 3u) & __ballot(len >= p.min_intron)) | (__ballot(op == 2u) & __ballot(len > p.max_delet));
        const unsigned long long f = c & ~any_c;
        ab1 = pick(f, ab1, A); l1 = pick(f, l1, len);
        ab2 = pick(c, ab2, A); l2 = pick(c, l2, len);
        three |= c & two; two |= c & any_c; any_c |= c;
        A += a;
    }
    if (WCHUNK > 2 && three) return false;
    const int incA = wave_scan<OpAdd>(A);
    if (any_c) {                                        // (wave-uniform)
        const int run = st.ref_end + incA - A;          // reference end in front of the lane's first word
        const int eb1 = run + ab1, sa1 = eb1 + l1 + 1, eb2 = run + ab2, sa2 = eb2 + l2 + 1;
        const int incM = wave_scan<OpMax>(pick(any_c, INT32_MIN, sa2));
        const int before = __builtin_amdgcn_update_dpp(INT32_MIN, incM, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        const int s1 = max(before, st.cur_start);       // start of the candidate the lane's first cut closes
        const unsigned long long first = st.seen_cut ? 0ull : any_c & (0ull - any_c);       // the read's first cut: kept whatever its length
        const unsigned long long km1 = any_c & (first | __ballot(eb1 - s1 + 1 >= p.min_exon));
        const unsigned long long km2 = two & __ballot(eb2 - sa1 + 1 >= p.min_exon);
        const uint32_t in_front = __builtin_amdgcn_mbcnt_hi((uint32_t)(km1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km1, 0u))
                                + __builtin_amdgcn_mbcnt_hi((uint32_t)(km2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km2, 0u));
        const int at = (int)(st.n_kept + in_front);
        if (__builtin_amdgcn_inverse_ballot_w64(km1)) emit(at, s1, eb1);                 // (the mask is the branch's exec mask as it stands)
        if (__builtin_amdgcn_inverse_ballot_w64(km2)) emit(at + pick(km1, 0, 1), sa1, eb2);
        st.n_kept += (uint32_t)(__popcll(km1) + __popcll(km2));
        st.cur_start = __builtin_amdgcn_readlane(incM, WAVE - 1);            // (the starts behind the cuts do not decrease)
        st.seen_cut = true;
    }
    st.ref_end += __builtin_amdgcn_readlane(incA, WAVE - 1);
    return true;
}

// the round of `ch` = words [base, base + 64 W) of the read behind `cw`; with three cuts in a lane, the same words two per lane
template <int W = l2r::WCHUNK, typename Emit>
__device__ __forceinline__ void wave_chunk_walk(WaveWalk &st, const CigarWindow &cw, const WaveChunk &ch, uint32_t base, const DevParams &p, int lane, Emit &emit)
{
    if (wave_chunk_try<W>(st, ch, p, lane, emit)) return;
#pragma unroll 1
    for (uint32_t b = base; b < base + (uint32_t)(W * WAVE); b += 2u * (uint32_t)WAVE) {
        const WaveChunk part = wave_chunk_load(cw, b, lane, 2);
        wave_chunk_try<2>(st, part, p, lane, emit);
    }
}

// dynamic LDS of k_pass_a<true> for tiles of up to `rpt` reads
inline size_t pass_a_dynamic_lds(int rpt) { return (size_t)WALK_SLAB * rpt * 6 + (size_t)3 * WALK_OVF * 4 + (size_t)TILE_THREADS * 2; }

// first j with key_j > (tid, start): the value the reference's annotation cursor has for this read
__device__ __forceinline__ int cursor_value(const CursorDir &cd, int32_t tid, int32_t start)
{
    if (tid >= cd.n_tid) return cd.n_tx;                       // beyond every annotated chromosome
    const int32_t kb = cd.kb_base[tid], nb = cd.kb_base[tid + 1] - kb;
    if (nb <= 0) return (int)cd.dir[kb];
    const int c = min(max(start, 0) >> SITE_SHIFT, nb - 1);
    int lo = (int)cd.dir[kb + c], hi = (int)cd.dir[kb + c + 1];
    if (c == nb - 1) hi = (int)cd.dir[kb + nb];               // last bucket of the chromosome is open ended
    const int64_t q = pack_key(tid, start);
    if (hi - lo <= 8) {
        // a 512-bp bucket rarely holds more than a few transcripts: fetch them in one round trip and count the keys
        // that are not above q (the keys do not decrease)
        int64_t k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k[u] = lo + u < hi ? cd.key[lo + u] : INT64_MAX;
        int below = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) below += k[u] <= q ? 1 : 0;
        return lo + below;
    }
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cd.key[mid] > q) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Pass A.  j0_in != null: the cursor values were replayed on the host (unsorted input) and are only read here.
// WIDE: the input has long CIGARs (the host decides per upload), see walk_cigar.
#ifndef L2R_PASSA_WGS
#define L2R_PASSA_WGS 8
#endif
template <bool WIDE>
__global__ __launch_bounds__(TILE_THREADS, WIDE ? L2R_PASSA_WGS : 8)
void k_pass_a(int64_t n_reads, const int32_t *__restrict__ r_tid, const int32_t *__restrict__ r_pos,
              const int64_t *__restrict__ cig_off, const uint32_t *__restrict__ cig, CursorDir cd, SiteTabs tabs, DevParams p,
              const int32_t *__restrict__ j0_in, int32_t *__restrict__ j0_out, uint32_t *__restrict__ local_out,
              uint8_t *__restrict__ order_out, uint32_t *__restrict__ tile_sum, TileDesc *__restrict__ desc, uint32_t *__restrict__ redo_count,
              const TxHdr *__restrict__ hdr, TxHdr *__restrict__ win_hdr, const uint32_t *__restrict__ tile_first,
              int2 *__restrict__ walked /* WIDE: the tile's exons, LDS_EXON_CAP slots per tile */)
{
    __shared__ uint32_t s_wave[4];
    __shared__ int s_red[4][6];
    __shared__ int s_tid0;
    __shared__ uint32_t s_hist[WAVE];
    // WIDE only, dynamic LDS sized by the tile size (pass_a_dynamic_lds): small tiles keep the kernel's occupancy
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    const int slab_w = p.reads_per_tile;                                     // reads per exon row of the slab
    int32_t *const s_slab_s = reinterpret_cast<int32_t *>(s_dyn);            // exon-major: start ...
    int *const s_ovf = s_slab_s + WALK_SLAB * slab_w;                        // exons beyond the slab: {thread | k << 8, start, end}
    uint16_t *const s_slab_l = reinterpret_cast<uint16_t *>(s_ovf + 3 * WALK_OVF);      // ... and length (0: empty exon)
    uint16_t *const s_loc = s_slab_l + WALK_SLAB * slab_w;                   // in-tile exon offset of every thread's read
    __shared__ uint32_t s_ovf_n;
    __shared__ int4 s_rd[WIDE ? TILE_THREADS : 1];                             // WIDE: {exon count, read end, not handed over} per read of the tile
    if (WIDE) { if (threadIdx.x == 0) s_ovf_n = 0u; __syncthreads(); }
    if (threadIdx.x < WAVE) s_hist[threadIdx.x] = 0u;
    // tile = reads [tile_first[b], tile_first[b + 1]): at most reads_per_tile of them, of one chromosome when the input is sorted
    const uint32_t r0 = tile_first[blockIdx.x], n_act = tile_first[blockIdx.x + 1] - r0;
    const int64_t r = (int64_t)r0 + threadIdx.x;
    const bool active = threadIdx.x < n_act;
    if (blockIdx.x == 0 && threadIdx.x == 0) { redo_count[0] = 0u; redo_count[1] = 0u; redo_count[2] = 0u; }   // redo list and accepted-exon cursor: the kernels that fill them run after this one
    uint32_t n = 0;
    bool unwalked = false;
    int j0 = INT32_MAX, tid = 0, pos = 0, el = 0;
    CigarHead head;
    int n_cig_mine = 0;
    const uint32_t *cig_mine = cig;
    if (active) {
        pos = r_pos[r];
        tid = r_tid[r];
        if (!WIDE) {
            const int64_t c_a = cig_off[r], c_b = cig_off[r + 1];
            cig_mine = cig + c_a; n_cig_mine = (int)(c_b - c_a);
            head = load_cigar_head(cig_mine, n_cig_mine);                   // in flight during the cursor lookup
        }
        if (j0_in) j0 = j0_in[r];
        else { j0 = cursor_value(cd, tid, pos + 1); j0_out[r] = j0; }       // first exon always starts at pos + 1
        el = pos;
    }
    if (WIDE) {
        // long CIGARs: the classification kernel shall not read them again (they are 10-50 times the bytes of the exons they
        // describe), so the exons are handed over.  The reads of the tile are dealt to the four waves, ONE WAVE walks one read
        // at a time (walk_read_wave: coalesced loads of the op stream, prefix sums over the wave).  The place of an exon in the
        // tile is not known before the scan below: the exons wait in LDS (exon-major, start + 16-bit length) and leave afterwards.
        // A read with more than WALK_SLAB exons or an exon of 64 kb and more: the classification kernel walks that read itself
        // (its first slot says so).
        const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
        // every read's {first word relative to the tile's, number of ops, pos}: thread i brings read i's, the waves pick them up in LDS
        const int64_t c_tile = cig_off[r0];
        if (active) { const int64_t c_a = cig_off[r], c_b = cig_off[r + 1]; s_rd[threadIdx.x] = make_int4((int)(c_a - c_tile), (int)min((int64_t)0x7fffffff, c_b - c_a), pos, 0); }
        __syncthreads();
        const uint32_t *const tile_cig = cig + c_tile;
        constexpr uint32_t ROUND = (uint32_t)(WCHUNK * WAVE);
        uint32_t q = (uint32_t)wv;
        int4 meta = q < n_act ? s_rd[q] : make_int4(0, 0, 0, 0);
        CigarWindow cw = cigar_window(tile_cig, (uint32_t)meta.x, (uint32_t)meta.y);
        WaveChunk cur = wave_chunk_load(cw, 0u, lane);
        for (; q < n_act; q += TILE_THREADS / WAVE) {
            // the first round of the wave's next read is asked for before this read is walked
            const uint32_t qn = q + TILE_THREADS / WAVE;
            const int4 meta_n = qn < n_act ? s_rd[qn] : make_int4(0, 0, 0, 0);
            const CigarWindow cw_n = cigar_window(tile_cig, (uint32_t)meta_n.x, (uint32_t)meta_n.y);
            const WaveChunk nxt = wave_chunk_load(cw_n, 0u, lane);
            const uint32_t n_cig = (uint32_t)__builtin_amdgcn_readfirstlane(meta.y);
            bool unw = false;
            auto emit = [&](int k, int s, int e) {
                const uint32_t len = (uint32_t)(e - s + 1);
                if (k < WALK_SLAB) {
                    s_slab_s[k * slab_w + (int)q] = s; s_slab_l[k * slab_w + (int)q] = (uint16_t)len;
                    unw = unw | (len > 0xffffu);
                } else {
                    const uint32_t at = atomicAdd(&s_ovf_n, 1u);
                    if (at < (uint32_t)WALK_OVF) { s_ovf[3 * at] = (int)q | (k << 8); s_ovf[3 * at + 1] = s; s_ovf[3 * at + 2] = e; }
                    else unw = true;
                }
            };
            WaveWalk st{meta.z, meta.z + 1, 0u, false};
            wave_chunk_walk(st, cw, cur, 0u, p, lane, emit);
            for (uint32_t base = ROUND; base < n_cig; base += ROUND) {           // (reads beyond 512 ops: round by round)
                const WaveChunk more = wave_chunk_load(cw, base, lane);
                wave_chunk_walk(st, cw, more, base, p, lane, emit);
            }
            if (lane == 0) emit((int)st.n_kept, st.cur_start, st.ref_end);
            const bool unw_any = __any(unw);
            // (the read's own entry of s_rd is not read again: this wave was its only reader)
            if (lane == 0) s_rd[q] = make_int4((int)st.n_kept + 1, st.ref_end, unw_any ? 1 : 0, 0);
            meta = meta_n; cur = nxt; cw = cw_n;
        }
        __syncthreads();
        if (active) { const int4 v = s_rd[threadIdx.x]; n = (uint32_t)v.x; el = v.y; unwalked = v.z != 0; }
    } else if (active) {
        n = (uint32_t)walk_cigar_headed<WIDE>(head, cig_mine, n_cig_mine, pos, p, [&](int, int, int e) { el = e; });
    }
    if (threadIdx.x == 0) s_tid0 = active ? tid : INT32_MAX;          // (an empty launch has no first read: no chromosome matches)
    uint32_t total;
    const uint32_t local = block_exclusive_scan(n, s_wave, total);
    if (active) local_out[r] = local;
    const bool walked_tile = WIDE && total <= (uint32_t)LDS_EXON_CAP;
    if (walked_tile) {
        int2 *const tile_out = walked + (size_t)blockIdx.x * LDS_EXON_CAP;
        if (active) {
            int2 *const out = tile_out + local;
            s_loc[threadIdx.x] = (uint16_t)local;
            const int n_slab = min((int)n, WALK_SLAB);
            for (int k = 0; k < n_slab; ++k) {
                const int s0 = s_slab_s[k * slab_w + (int)threadIdx.x];
                out[k] = make_int2(s0, s0 + (int)s_slab_l[k * slab_w + (int)threadIdx.x] - 1);
            }
        }
        __syncthreads();
        const uint32_t n_ovf = min(s_ovf_n, (uint32_t)WALK_OVF);
        for (uint32_t i = threadIdx.x; i < n_ovf; i += TILE_THREADS) {
            const int who = s_ovf[3 * i] & 0xff, k = s_ovf[3 * i] >> 8;
            tile_out[(uint32_t)s_loc[who] + (uint32_t)k] = make_int2(s_ovf[3 * i + 1], s_ovf[3 * i + 2]);
        }
        if (active && unwalked) tile_out[local] = make_int2(WALK_SENTINEL, 0);      // (after the exons above: this slot wins)
    }
    {   // Order of the tile's reads by falling exon count (counting sort, ties in arrival order): the classification
        // kernel gives read order_out[slot] to thread `slot`, so the reads of a wave need about the same number of
        // rounds in its per-exon loops.  Results do not depend on the order: every read owns its output slots.
        const uint32_t bin = (uint32_t)(WAVE - 1) - min(n, (uint32_t)(WAVE - 1));      // inactive threads (n = 0) sort last
        const uint32_t rank = atomicAdd(&s_hist[bin], 1u);
        __syncthreads();
        if (threadIdx.x < WAVE) { const uint32_t c = s_hist[threadIdx.x]; s_hist[threadIdx.x] = wave_inclusive_scan(c) - c; }
        __syncthreads();
        const uint32_t slot = s_hist[bin] + rank;
        if (active) order_out[(int64_t)r0 + slot] = (uint8_t)threadIdx.x;
    }
    // the tile's chromosome is the one of its first read; reads on another one go to the generic kernel
    const int tid0 = s_tid0;
    int tb = 0, nb = 0;
    if (tid0 < tabs.n_tid) { tb = tabs.tid_base[tid0]; nb = tabs.tid_base[tid0 + 1] - tb; }
    int blo = INT32_MAX, bhi = -1;
    const bool mine = active && tid == tid0;
    if (mine && nb > 0) {             // bucket span of the read, clamped to the annotation's grid
        // (-d > 0: the probes look up to `dis` outside the read, probe_near)
        blo = min(max(pos + 1 - max(p.ss_dis, 0), 0) >> SITE_SHIFT, nb - 1);
        bhi = min(max(el + max(p.ss_dis, 0), 0) >> SITE_SHIFT, nb - 1);
    }
    {   // tile reductions: cursor range, bucket span, coordinate span
        const int a0 = wave_min(mine ? j0 : INT32_MAX), a1 = wave_max(mine ? j0 : -1), a3 = wave_min(blo), a4 = wave_max(bhi);
        const int a5 = wave_min(mine ? pos + 1 : INT32_MAX), a6 = wave_max(mine ? el : INT32_MIN);
        if ((threadIdx.x & (WAVE - 1)) == 0) {
            int *q = s_red[threadIdx.x >> 6];
            q[0] = a0; q[1] = a3; q[2] = a4; q[3] = a1; q[4] = a5; q[5] = a6;
        }
    }
    __syncthreads();
    if (threadIdx.x < WAVE) {             // wave 0 finishes the tile; everything below is wave-uniform but `lane`
        const int lane = (int)threadIdx.x;
        int jl = INT32_MAX, lo = INT32_MAX, hi = -1, jh = -1, tlo = INT32_MAX, thi = INT32_MIN;
        for (int w = 0; w < 4; ++w) {
            jl = min(jl, s_red[w][0]); lo = min(lo, s_red[w][1]); hi = max(hi, s_red[w][2]);
            jh = max(jh, s_red[w][3]); tlo = min(tlo, s_red[w][4]); thi = max(thi, s_red[w][5]);
        }
        TileDesc d;
        d.j_lo = jl == INT32_MAX ? 0 : jl; d.tid = tid0; d.b_off = 0; d.nb = 0; d.b0 = 0; d.nbk = 0;
        d.st_r0 = d.st_nk = d.en_r0 = d.en_nk = 0u; d.n_win = 0u;
        bool fast = total <= (uint32_t)LDS_EXON_CAP && p.ss_dis >= 0 && p.ss_dis <= DIS_MASK_MAX && !(p.ablate & 1);
        uint32_t why = fast ? 0u : (total > (uint32_t)LDS_EXON_CAP ? 1u : 7u);       // diagnostics: why a tile is not fast (flags bits 8..11)
        // dictionary slices of the tile's bucket span: four directory words, used after the window scan below (the
        // loads and the scan's header loads are in flight together)
        uint32_t sd_r0 = 0u, sd_r1 = 0u, ed_r0 = 0u, ed_r1 = 0u;
        const bool sliced = fast && hi >= 0 && hi - lo + 1 <= DIR_CAP;
        if (fast && hi >= 0 && !sliced) { fast = false; why = 2u; }
        if (sliced) {
            // START entries from the first one that reaches into the first bucket (full-length evidence scans them)
            sd_r0 = tabs.st.rdir[tb + lo]; sd_r1 = tabs.st.dir[tb + hi + 1];
            ed_r0 = tabs.en.dir[tb + lo]; ed_r1 = tabs.en.dir[tb + hi + 1];
        }
        // The tile's WINDOW: the transcripts from its smallest cursor value on that some read of the tile can overlap,
        // in file order, up to the first transcript every read of the tile lies before (src/update_gtf.c:799-800 ends
        // every sweep there; a sweep that starts behind it would leave the window, so then the tile is not fast).
        // Left out are transcripts that end before every read of the tile starts: :801 skips them for each read and they
        // can end no sweep (their start lies below every read end).  At most WIN_TX members = bits of a tile-frame mask.
        bool contig = true;
        if (fast && jl != INT32_MAX) {
            int first = -1, last = -1, after = INT32_MAX;
            uint32_t n_win = 0;
            for (int base = jl, trip = 0; ; ++trip) {
                const int j = base + lane;
                bool ov = false, aft = false;
                if (j < p.n_tx) {
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
                    if (rank < (uint32_t)WIN_TX) {          // the member's header, with its annotation index in the spare word
                        const int4 *hp = reinterpret_cast<const int4 *>(hdr + j);
                        int4 *wp = reinterpret_cast<int4 *>(win_hdr + (int64_t)blockIdx.x * WIN_TX + rank);
                        int4 h1 = hp[1];
                        h1.w = j;
                        wp[0] = hp[0]; wp[1] = h1; wp[2] = hp[2];
                    }
                }
                if (mo) {
                    if (first < 0) first = base + __ffsll((long long)mo) - 1;
                    last = base + 63 - __clzll((long long)mo);
                }
                n_win += (uint32_t)__popcll(mo);
                if (ma) { after = base + stop; break; }
                base += WAVE;
                if (base >= p.n_tx) break;
                if (n_win > (uint32_t)WIN_TX || trip == WIN_SCAN_TRIPS - 1) { fast = false; why = n_win > (uint32_t)WIN_TX ? 4u : 5u; break; }
            }
            if (fast && (n_win > (uint32_t)WIN_TX || jh > after)) { fast = false; why = n_win > (uint32_t)WIN_TX ? 4u : 6u; }
            if (fast) {
                d.n_win = n_win;
                if (n_win) { d.j_lo = first; contig = (uint32_t)(last - first + 1) == n_win; }
            }
        }
        if (sliced) {
            d.b_off = -lo; d.nb = nb; d.b0 = tb + lo; d.nbk = hi - lo + 1;
            d.st_r0 = sd_r0; d.st_nk = sd_r1 - sd_r0;
            d.en_r0 = ed_r0; d.en_nk = ed_r1 - ed_r0;
            if (fast && (d.st_nk > (uint32_t)KEY_CAP || d.en_nk > (uint32_t)KEY_CAP)) { fast = false; why = 3u; }
        }
        d.flags = (fast ? TD_FAST : 0u) | (contig ? TD_CONTIG : 0u) | (walked_tile ? TD_WALKED : 0u) | (why << 8);
        if (lane == 0) { tile_sum[blockIdx.x] = total; desc[blockIdx.x] = d; }
    }
}

// In-place exclusive scan of v[0..n) by ONE workgroup of 1024 threads, 16 consecutive elements per thread and round (four
// 16-byte loads in flight together: the kernel is a chain of load -> scan -> store round trips, so fewer, wider rounds);
// v[n] receives the sum (the array has n + 1 words) and so does *total.  blockIdx.x selects the array.
struct ScanJob { uint32_t *v; int64_t n; uint32_t *total; };
// lists: block 1 of a two-block launch (slab pipeline) does not scan: it lists the tiles of k_probe_slab_wide (TD_WIDE) and of
// k_probe_slab_chunked (slab_tile_is_chunked) from the descriptors k_walk_slab left -- in tile order, beside the scan of the tiles' exon
// counts, so the lists cost no launch and no tile appends to a shared counter (40 k appends to one address take 0.4 ms).
// cnt[0] / cnt[1] = entries, cnt[2] / cnt[3] = the two kernels' work cursors (cleared here).
struct TileLists { const uint32_t *flags0; uint32_t n_tiles, chunk_on; uint32_t *wide_list, *chunk_list, *cnt; };      // flags0: every tile's descriptor flags, densely (k_walk_slab)
struct ScanJobs { ScanJob job[2]; TileLists lists; };
// a tile k_probe_slab_chunked takes (l2r_chunk.hip.h): no window record (window beyond 63 members, dictionary slices beyond the
// one-window kernels' staging) or flagged by a one-window kernel that found a dictionary key in several entries
__device__ __forceinline__ bool slab_tile_is_chunked(uint32_t flags)
{
    const uint32_t why = (flags >> 8) & 7u;
    return (flags & TD_CHUNK) != 0u || ((flags & (TD_FAST | TD_WIDE)) == 0u && (why == 4u || why == 3u));
}
constexpr int SCAN_PER_THREAD = 16;

// (one workgroup of 1024 threads; s_wave[16] and s_carry are the caller's)
__device__ __forceinline__ void scan_block_u32(const ScanJob &job, uint32_t *s_wave, uint32_t &s_carry)
{
    uint32_t *__restrict__ v = job.v;
    const int64_t n = job.n;
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024 * SCAN_PER_THREAD) {
        const int64_t i = base + SCAN_PER_THREAD * (int64_t)threadIdx.x;
        uint32_t x[SCAN_PER_THREAD];
        if (i + SCAN_PER_THREAD <= n) {
#pragma unroll
            for (int q = 0; q < SCAN_PER_THREAD / 4; ++q) {
                const uint4 t = *reinterpret_cast<const uint4 *>(v + i + 4 * q);
                x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < SCAN_PER_THREAD; ++q) x[q] = i + q < n ? v[i + q] : 0u;
        }
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < SCAN_PER_THREAD; ++q) { const uint32_t t = x[q]; x[q] = mine; mine += t; }      // x: exclusive inside the thread
        const uint32_t inc = wave_inclusive_scan(mine);
        if (lane == WAVE - 1) s_wave[w] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint32_t t = s_wave[k]; if (k < w) wbase += t; tot += t; }
        const uint32_t carry = s_carry;
        const uint32_t e0 = carry + wbase + inc - mine;
        if (i + SCAN_PER_THREAD <= n) {
#pragma unroll
            for (int q = 0; q < SCAN_PER_THREAD / 4; ++q)
                *reinterpret_cast<uint4 *>(v + i + 4 * q) = make_uint4(e0 + x[4 * q], e0 + x[4 * q + 1], e0 + x[4 * q + 2], e0 + x[4 * q + 3]);
        } else {
#pragma unroll
            for (int q = 0; q < SCAN_PER_THREAD; ++q) if (i + q < n) v[i + q] = e0 + x[q];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) { v[n] = s_carry; *job.total = s_carry; }
}

__global__ __launch_bounds__(1024)
void k_scan_u32(ScanJobs jobs)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    if (blockIdx.x == 1 && jobs.lists.flags0) {
        // the tile lists, 16 tiles per thread and round like the scan below (four 16-byte loads in flight per thread): count, block
        // scan of the counts, write in tile order behind what the rounds before have listed
        __shared__ uint32_t s_w2[16];
        __shared__ uint32_t s_base[2];
        const TileLists L = jobs.lists;
        const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
        if (threadIdx.x == 0) { s_base[0] = 0u; s_base[1] = 0u; }
        __syncthreads();
        for (uint32_t base = 0; base < L.n_tiles; base += 1024u * (uint32_t)SCAN_PER_THREAD) {
            const uint32_t i = base + (uint32_t)SCAN_PER_THREAD * threadIdx.x;
            uint32_t f[SCAN_PER_THREAD];
            if (i + SCAN_PER_THREAD <= L.n_tiles) {
#pragma unroll
                for (int q = 0; q < SCAN_PER_THREAD / 4; ++q) {
                    const uint4 t = *reinterpret_cast<const uint4 *>(L.flags0 + i + 4 * q);
                    f[4 * q] = t.x; f[4 * q + 1] = t.y; f[4 * q + 2] = t.z; f[4 * q + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < SCAN_PER_THREAD; ++q) f[q] = i + q < L.n_tiles ? L.flags0[i + q] : TD_FAST;        // (behind the last tile: nobody's)
            }
            uint32_t nw = 0, nc = 0;
#pragma unroll
            for (int q = 0; q < SCAN_PER_THREAD; ++q) { nw += (f[q] & TD_WIDE) ? 1u : 0u; nc += (L.chunk_on && slab_tile_is_chunked(f[q])) ? 1u : 0u; }
            const uint32_t iw = wave_inclusive_scan(nw), ic = wave_inclusive_scan(nc);
            if (lane == WAVE - 1) { s_wave[w] = iw; s_w2[w] = ic; }
            __syncthreads();
            uint32_t bw = 0, bc = 0, tw_ = 0, tc_ = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) { if (k < w) { bw += s_wave[k]; bc += s_w2[k]; } tw_ += s_wave[k]; tc_ += s_w2[k]; }
            uint32_t aw = s_base[0] + bw + iw - nw, ac = s_base[1] + bc + ic - nc;
            if (nw | nc) {
#pragma unroll
                for (int q = 0; q < SCAN_PER_THREAD; ++q) {
                    if (f[q] & TD_WIDE) L.wide_list[aw++] = i + (uint32_t)q;
                    if (L.chunk_on && slab_tile_is_chunked(f[q])) L.chunk_list[ac++] = i + (uint32_t)q;
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) { s_base[0] += tw_; s_base[1] += tc_; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { L.cnt[0] = s_base[0]; L.cnt[1] = s_base[1]; L.cnt[2] = 0u; L.cnt[3] = 0u; }
        return;
    }
    scan_block_u32(jobs.job[blockIdx.x], s_wave, s_carry);
}

// Exclusive scan of n counts in SEGMENTS, one 256-thread workgroup per segment of SEG_COUNT counts, out of place: a workgroup adds up
// the counts in front of its segment itself (a few 16-byte loads per thread, all in flight), so the segments do not wait for each
// other -- for the 39 k tiles of 10 M reads that is ten workgroups and about one round trip, where one workgroup scanning in place
// (k_scan_u32) takes three dependent rounds (13 us).  in: n counts; out: n + 1 words (the sum last), never `in` itself: a workgroup
// reads the segments in front of its own while their workgroups write; total: the sum once more.  Worth it up to SEG_MAX counts.
struct SegScan { const uint32_t *in; uint32_t *out; uint32_t *total; int64_t n; };
constexpr int SEG_PER_THREAD = 16;
constexpr int SEG_COUNT = TILE_THREADS * SEG_PER_THREAD;
constexpr int64_t SEG_MAX = (int64_t)SEG_COUNT * 64;
__device__ __forceinline__ void scan_segment(const SegScan &job, uint32_t seg, uint32_t n_seg, uint32_t (*s_part)[TILE_THREADS / WAVE] /* [2][4] */)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const uint32_t *__restrict__ v = job.in;
    const uint32_t n = (uint32_t)job.n, seg0 = seg * (uint32_t)SEG_COUNT;
    const uint32_t i = seg0 + (uint32_t)SEG_PER_THREAD * threadIdx.x;
    uint32_t x[SEG_PER_THREAD];
    if (i + SEG_PER_THREAD <= n) {
#pragma unroll
        for (int q = 0; q < SEG_PER_THREAD / 4; ++q) {
            const uint4 t4 = *reinterpret_cast<const uint4 *>(v + i + 4 * q);
            x[4 * q] = t4.x; x[4 * q + 1] = t4.y; x[4 * q + 2] = t4.z; x[4 * q + 3] = t4.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < SEG_PER_THREAD; ++q) x[q] = i + q < n ? v[i + q] : 0u;
    }
    // everything in front of the segment (whole segments: 16 counts per thread each)
    uint32_t before = 0u;
#pragma unroll 2
    for (uint32_t sgm = 0; sgm < seg; ++sgm) {
        const uint4 *src = reinterpret_cast<const uint4 *>(v + sgm * (uint32_t)SEG_COUNT + (uint32_t)SEG_PER_THREAD * threadIdx.x);
        const uint4 a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
        before += (a0.x + a0.y + a0.z + a0.w) + (a1.x + a1.y + a1.z + a1.w) + (a2.x + a2.y + a2.z + a2.w) + (a3.x + a3.y + a3.z + a3.w);
    }
    uint32_t mine = 0u;
#pragma unroll
    for (int q = 0; q < SEG_PER_THREAD; ++q) { const uint32_t tq = x[q]; x[q] = mine; mine += tq; }      // x: exclusive inside the thread
    const uint32_t inc = wave_inclusive_scan(mine), bsum = wave_sum(before);
    if (lane == WAVE - 1) s_part[0][wv] = inc;
    if (lane == 0) s_part[1][wv] = bsum;
    __syncthreads();
    uint32_t wbase = 0u, tot = 0u, carry = 0u;
#pragma unroll
    for (int k = 0; k < TILE_THREADS / WAVE; ++k) { const uint32_t tk = s_part[0][k]; if (k < wv) wbase += tk; tot += tk; carry += s_part[1][k]; }
    uint32_t *__restrict__ out = job.out;
    const uint32_t e0 = carry + wbase + inc - mine;
    if (i + SEG_PER_THREAD <= n) {
#pragma unroll
        for (int q = 0; q < SEG_PER_THREAD / 4; ++q)
            *reinterpret_cast<uint4 *>(out + i + 4 * q) = make_uint4(e0 + x[4 * q], e0 + x[4 * q + 1], e0 + x[4 * q + 2], e0 + x[4 * q + 3]);
    } else {
#pragma unroll
        for (int q = 0; q < SEG_PER_THREAD; ++q) if (i + q < n) out[i + q] = e0 + x[q];
    }
    if (seg == n_seg - 1u && threadIdx.x == 0) { out[n] = carry + tot; *job.total = carry + tot; }
}
// two such scans in one launch: workgroups [0, n_seg) the first, [n_seg, 2 n_seg) the second
__global__ __launch_bounds__(TILE_THREADS)
void k_scan_segments(SegScan j0, SegScan j1, uint32_t n_seg)
{
    __shared__ uint32_t s_part[2][TILE_THREADS / WAVE];
    if (blockIdx.x < n_seg) scan_segment(j0, blockIdx.x, n_seg, s_part);
    else scan_segment(j1, blockIdx.x - n_seg, n_seg, s_part);
}

// ------------------------------------------------------------------ comparison rules

__device__ __forceinline__ bool near_eq(int a, int b, int dis) { return __builtin_abs(a - b) <= dis; }
__device__ __forceinline__ bool closed_overlap(int s1, int e1, int s2, int e2)
{   // src/update_gtf.c:91-95 exon_overlap
    return !(s1 > e2 || s2 > e1);
}

// src/update_gtf.c:80-89 exon_overlap_frac: int / double, rounded to float, compared as float (Q6)
__device__ __forceinline__ float overlap_frac(int s1, int e1, int s2, int e2)
{
    if (s1 > e2 || s2 > e1) return 0.0f;
    const int ov = min(e1, e2) - max(s1, s2) + 1;
    const int ml = min(e1 - s1 + 1, e2 - s2 + 1);
    return (float)((double)ov / ((double)ml + 0.0));
}

struct ReadState { bool lfull, rfull, lnoth, rnoth, known, ksite; };
struct ReadEnds { int s0, e0, sl, el; };     // first and last exon of the read

// src/update_gtf.c:629-681 check_full for one overlapping annotation transcript.
__device__ __forceinline__ void full_evidence(ReadState &st, int level, const ReadEnds &r, const TxHdr &a, const int2 *__restrict__ ax)
{
    if (st.lfull && st.rfull) return;
    if (level == 1) {
        if (!st.lfull && r.e0 == a.e0) st.lfull = true;
        if (!st.rfull && r.sl == a.sl) st.rfull = true;
    } else if (level == 2) {
        if (!st.lfull && closed_overlap(r.s0, r.e0, a.s0, a.e0)) st.lfull = true;
        if (!st.rfull && closed_overlap(r.sl, r.el, a.sl, a.el)) st.rfull = true;
    } else if (level == 3 || level == 4) {
        if (!st.lfull) {
            if (closed_overlap(r.s0, r.e0, a.s0, a.e0)) st.lfull = true;
            else if (st.lnoth)
                for (int k = 0; k < a.n; ++k) { const int2 x = ax[k]; if (closed_overlap(r.s0, r.e0, x.x, x.y)) { st.lnoth = false; break; } }
        }
        if (level == 3 && !st.rfull) {
            if (closed_overlap(r.sl, r.el, a.sl, a.el)) st.rfull = true;
            else if (st.rnoth)
                for (int k = 0; k < a.n; ++k) { const int2 x = ax[k]; if (closed_overlap(r.sl, r.el, x.x, x.y)) { st.rnoth = false; break; } }
        }
    }
}

// src/update_gtf.c:683-696 set_full
__device__ __forceinline__ bool full_decision(int level, bool lfull, bool lnoth, bool rfull, bool rnoth)
{
    if (level == 5) return true;
    if (level == 4) return lfull || lnoth;
    if (level == 3) return (lfull || lnoth) && (rfull || rnoth);
    return lfull && rfull;
}

// src/update_gtf.c:717-779 check_splice_site, all four double loops folded into one (i over annotation exons,
// j over read exons); clears are idempotent and the counters are plain sums, so the visiting order does not
// matter.  The acceptor comparison uses the read exon j start, j < n-1 (Q1, :746).
// returns 1 known, 2 has known site, 0 neither.
__device__ __forceinline__ int site_compare(const int *S, const int *E, uint8_t *F, int n, int r_start, int r_end,
                                            const TxHdr &a, const int2 *__restrict__ ax, int dis)
{
    const int lo = max(r_start, a.start), hi = min(r_end, a.end);
    int r_in = 0, same = 0;
    for (int j = 0; j + 1 < n; ++j) {
        const int e = E[j], s2 = S[j + 1];
        r_in += (e >= lo && e <= hi) + (s2 >= lo && s2 <= hi);
    }
    const int m = a.n;
    int2 cur = ax[0];
    for (int i = 0; i < m; ++i) {
        const bool has_next = i + 1 < m;
        const int2 nxt = has_next ? ax[i + 1] : cur;
        const bool don_in = has_next && cur.y >= lo && cur.y <= hi;
        const bool acc_in = has_next && nxt.x >= lo && nxt.x <= hi;
        for (int j = 0; j < n; ++j) {
            const int sj = S[j], ej = E[j];
            const bool end_eq = near_eq(cur.y, ej, dis);
            uint8_t clr = 0;
            if (end_eq && near_eq(cur.x, sj, dis)) clr |= F_EXON;
            if (has_next && j + 1 < n) {
                if (don_in && end_eq) { ++same; clr |= F_DON; }
                if (acc_in && near_eq(nxt.x, sj, dis)) { ++same; clr |= F_ACC; }
                if (end_eq && near_eq(nxt.x, S[j + 1], dis)) clr |= F_JUNC;
            }
            if (clr) F[j] &= (uint8_t)~clr;
        }
        cur = nxt;
    }
    if (2 * (n - 1) == r_in && r_in == same) return 1;
    return same > 0 ? 2 : 0;
}

__device__ __forceinline__ uint32_t finish_info(uint32_t info, int n, const DevParams &p)
{
    // routing of update_gtf.c:943-950 when there is no junction table
    if (p.n_sj == 0 && (info & (I_FULL | I_KNOWN | I_KSITE)) == (I_FULL | I_KSITE)) info |= I_ACCEPT;
    return info | ((uint32_t)n << 8);
}

// ------------------------------------------------------------------ generic classification (any -d, any annotation)

// src/update_gtf.c:792-835 check_with_anno_trans for one read with the literal loops.  (S, E, F) = the read's
// exons and flag bytes, in LDS or in HBM.
struct Verdict { uint32_t info; int ref; };

__device__ __forceinline__ Verdict sweep_literal(const int *S, const int *E, uint8_t *F, int n, int tid, bool rev, int j0,
                                                 const TxHdr *__restrict__ hdr, const int2 *__restrict__ anno_ex, const DevParams &p)
{
    for (int k = 0; k < n; ++k) F[k] = (k + 1 < n) ? (uint8_t)(F_EXON | F_DON | F_ACC | F_JUNC) : F_EXON;
    const ReadEnds re{S[0], E[0], S[n - 1], E[n - 1]};
    const int r_start = re.s0, r_end = re.el;
    ReadState st{false, false, true, true, false, false};
    int ref = -1, ref_rev = 0;
    for (int j = j0; j < p.n_tx; ++j) {
        const int4 *hp = reinterpret_cast<const int4 *>(hdr + j);
        const int4 h0 = hp[0];
        // src/update_gtf.c:786-790 comp_trans: <= (Q5)
        if (tid < h0.x || (tid == h0.x && r_end <= h0.y)) break;                 // :799-800
        if (h0.x < tid || (h0.x == tid && h0.z <= r_start)) continue;            // :801
        const int4 h1 = hp[1], h2 = hp[2];
        TxHdr a; a.tid = h0.x; a.start = h0.y; a.end = h0.z; a.ex_off = h0.w;
        a.n = h1.x; a.rev = h1.y; a.flags = h1.z; a.s0 = h2.x; a.e0 = h2.y; a.sl = h2.z; a.el = h2.w;
        const int2 *ax = anno_ex + a.ex_off;
        full_evidence(st, p.full_level, re, a, ax);
        int v = 0;
        if (n == 1 && a.n == 1) {
            if (overlap_frac(re.s0, re.e0, a.s0, a.e0) >= p.frac) { st.known = true; v = 1; }
        } else if (n > 1 && a.n > 1) {
            v = site_compare(S, E, F, n, r_start, r_end, a, ax, p.ss_dis);
            if (v == 1) st.known = true;
            if (v == 2) st.ksite = true;
        }
        if (v) { ref = j; ref_rev = a.rev; }
        if (v == 1) break;                                                        // :810,816
    }
    bool out_rev = rev;
    if (ref >= 0) out_rev = ref_rev != 0;               // :825-831 strand taken from the reference transcript
    uint32_t info = 0;
    if (st.known) info |= I_KNOWN;
    if (st.ksite) info |= I_KSITE;
    if (full_decision(p.full_level, st.lfull, st.lnoth, st.rfull, st.rnoth)) info |= I_FULL;
    if (out_rev) info |= I_REV;
    return Verdict{finish_info(info, n, p), ref};
}

// ONE WAVE per entry of the redo list: the reference's loops over annotation exons run across the 64 lanes
// (lane i takes annotation exons i, i + 64, ...), the loops over the read's exons run inside a lane; every branch
// of the sweep is wave-uniform because the whole wave works on one read.  Reads of up to GEN_CAP exons are held in
// LDS (flags as one word per exon, cleared with LDS atomics); longer ones take the one-thread sweep_literal.
constexpr int GEN_CAP = 128;
constexpr int GEN_WAVES = TILE_THREADS / WAVE;

__global__ __launch_bounds__(TILE_THREADS)
void k_classify_generic(const uint32_t *__restrict__ redo_count, const uint32_t *__restrict__ redo,
                        const int32_t *__restrict__ r_tid, const uint8_t *__restrict__ r_rev, const int32_t *__restrict__ j0_arr,
                        const TxHdr *__restrict__ hdr, const int2 *__restrict__ anno_ex, DevParams p,
                        const uint32_t *__restrict__ ex_off, const int32_t *__restrict__ ex_start, const int32_t *__restrict__ ex_end,
                        uint8_t *__restrict__ ex_flag, uint32_t *__restrict__ info_io, int32_t *__restrict__ ref_out,
                        uint32_t *__restrict__ tile_acc, uint32_t *__restrict__ tile_acc_ex, const uint32_t *__restrict__ tile_first, int n_tiles,
                        CursorDir cd /* used when j0_arr is null: the one-walk pipeline keeps no per-read cursor values */,
                        uint32_t *__restrict__ list_cnt /* slab pipelines: the entry counts of the list-driven probe kernels in front of this launch,
                                                           cleared here for the next run (its FIRST kernel appends to them); else null */)
{
    __shared__ int g_S[GEN_WAVES][GEN_CAP];
    __shared__ int g_E[GEN_WAVES][GEN_CAP];
    __shared__ uint32_t g_F[GEN_WAVES][GEN_CAP];
    __shared__ uint32_t g_cnt[GEN_WAVES][2];
    if (list_cnt && blockIdx.x == 0 && threadIdx.x == 0) { list_cnt[6] = list_cnt[0]; list_cnt[7] = list_cnt[1]; list_cnt[0] = 0u; list_cnt[1] = 0u; }
    const uint32_t cnt = *redo_count;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    int *S = g_S[wv], *E = g_E[wv];
    uint32_t *F = g_F[wv];
    for (uint32_t i = blockIdx.x * GEN_WAVES + wv; i < cnt; i += gridDim.x * GEN_WAVES) {
        const uint32_t r = redo[i];
        const int n = (int)(info_io[r] >> 8);
        const uint32_t off = ex_off[r];
        const int tid = r_tid[r];
        const int j0 = j0_arr ? j0_arr[r] : cursor_value(cd, tid, ex_start[off]);        // (the first exon starts at pos + 1)
        const bool rev = r_rev[r] != 0;
        Verdict v{0u, -1};
        if (n > GEN_CAP) {
            if (lane == 0) v = sweep_literal(ex_start + off, ex_end + off, ex_flag + off, n, tid, rev, j0, hdr, anno_ex, p);
        } else {
            for (int k = lane; k < n; k += WAVE) {
                S[k] = ex_start[off + (uint32_t)k]; E[k] = ex_end[off + (uint32_t)k];
                F[k] = (k + 1 < n) ? (uint32_t)(F_EXON | F_DON | F_ACC | F_JUNC) : (uint32_t)F_EXON;
            }
            const ReadEnds re{ex_start[off], ex_end[off], ex_start[off + (uint32_t)(n - 1)], ex_end[off + (uint32_t)(n - 1)]};
            const int r_start = re.s0, r_end = re.el, dis = p.ss_dis, level = p.full_level;
            bool lfull = false, rfull = false, lnoth = true, rnoth = true, known = false, ksite = false;
            int ref = -1, ref_rev = 0;
            // The sweep in batches of 64 transcripts: every lane fetches one header (three 16-byte words, all in flight
            // together), the stop (:799-800) and skip (:801) tests run on all 64 at once, and only the transcripts the
            // reference would compare are visited, their header words read out of the lanes' registers.  (One header
            // after the other cost a dependent round trip per transcript: a locus with 80 isoforms = 80 round trips.)
            bool done = false;
            for (int base = j0; base < p.n_tx && !done; base += WAVE) {
              const int jl = base + lane;
              int4 g0 = make_int4(INT32_MAX, 0, 0, 0), g1 = make_int4(0, 0, 0, 0), g2 = g1;     // (beyond the annotation: "the read lies before it")
              if (jl < p.n_tx) { const int4 *gp = reinterpret_cast<const int4 *>(hdr + jl); g0 = gp[0]; g1 = gp[1]; g2 = gp[2]; }
              // src/update_gtf.c:786-790 comp_trans: <= (Q5)
              const bool brk = tid < g0.x || (tid == g0.x && r_end <= g0.y);              // :799-800
              const bool skip = g0.x < tid || (g0.x == tid && g0.z <= r_start);            // :801
              const unsigned long long mb = __ballot(brk);
              unsigned long long mv = __ballot(!brk && !skip);
              if (mb) { mv &= (mb & (0ull - mb)) - 1ull; done = true; }                   // nothing at or behind the first stop
              while (mv) {
                const int src = __ffsll((long long)mv) - 1;
                mv &= mv - 1ull;
                const int j = base + src;
                const int4 h0 = make_int4(__shfl(g0.x, src, WAVE), __shfl(g0.y, src, WAVE), __shfl(g0.z, src, WAVE), __shfl(g0.w, src, WAVE));
                const int4 h1 = make_int4(__shfl(g1.x, src, WAVE), __shfl(g1.y, src, WAVE), 0, 0);
                const int4 h2 = make_int4(__shfl(g2.x, src, WAVE), __shfl(g2.y, src, WAVE), __shfl(g2.z, src, WAVE), __shfl(g2.w, src, WAVE));
                const int a_start = h0.y, a_end = h0.z, m = h1.x;
                const int2 *ax = anno_ex + h0.w;
                // ---- check_full :629-681
                if (!(lfull && rfull)) {
                    if (level == 1) {
                        if (re.e0 == h2.y) lfull = true;
                        if (re.sl == h2.z) rfull = true;
                    } else if (level == 2) {
                        if (closed_overlap(re.s0, re.e0, h2.x, h2.y)) lfull = true;
                        if (closed_overlap(re.sl, re.el, h2.z, h2.w)) rfull = true;
                    } else if (level == 3 || level == 4) {
                        bool need_l = false, need_r = false;
                        if (!lfull) { if (closed_overlap(re.s0, re.e0, h2.x, h2.y)) lfull = true; else need_l = lnoth; }
                        if (level == 3 && !rfull) { if (closed_overlap(re.sl, re.el, h2.z, h2.w)) rfull = true; else need_r = rnoth; }
                        if (need_l || need_r) {
                            bool hit_l = false, hit_r = false;
                            for (int k = lane; k < m; k += WAVE) {
                                const int2 x = ax[k];
                                hit_l = hit_l || closed_overlap(re.s0, re.e0, x.x, x.y);
                                hit_r = hit_r || closed_overlap(re.sl, re.el, x.x, x.y);
                            }
                            if (need_l && __any(hit_l)) lnoth = false;
                            if (need_r && __any(hit_r)) rnoth = false;
                        }
                    }
                }
                // ---- :806-820
                int vv = 0;
                if (n == 1 && m == 1) {
                    if (overlap_frac(re.s0, re.e0, h2.x, h2.y) >= p.frac) { known = true; vv = 1; }
                } else if (n > 1 && m > 1) {
                    // src/update_gtf.c:717-779 check_splice_site
                    const int lo = max(r_start, a_start), hi = min(r_end, a_end);
                    int r_in = 0, same = 0;
                    for (int q = lane; q + 1 < n; q += WAVE) {
                        const int e = E[q], s2 = S[q + 1];
                        r_in += (e >= lo && e <= hi) + (s2 >= lo && s2 <= hi);
                    }
                    // one lane per (annotation exon k, read exon q) pair: m x n pairs over the 64 lanes (a lane per annotation
                    // exon looping over the read's exons kept 9 lanes of 64 busy for a typical transcript)
                    const int pairs = m * n;
                    for (int idx = lane; idx < pairs; idx += WAVE) {
                        const int k = idx / n, q = idx - k * n;
                        const bool has_next = k + 1 < m;
                        const int2 cur = ax[k], nxt = has_next ? ax[k + 1] : cur;
                        const bool don_in = has_next && cur.y >= lo && cur.y <= hi;
                        const bool acc_in = has_next && nxt.x >= lo && nxt.x <= hi;
                        const int sq = S[q], eq = E[q];
                        const bool end_eq = near_eq(cur.y, eq, dis);
                        uint32_t clr = 0;
                        if (end_eq && near_eq(cur.x, sq, dis)) clr |= F_EXON;
                        if (has_next && q + 1 < n) {
                            if (don_in && end_eq) { ++same; clr |= F_DON; }
                            if (acc_in && near_eq(nxt.x, sq, dis)) { ++same; clr |= F_ACC; }        // Q1: read exon q's own start
                            if (end_eq && near_eq(nxt.x, S[q + 1], dis)) clr |= F_JUNC;
                        }
                        if (clr) atomicAnd(&F[q], ~clr);
                    }
                    // the two counters of check_splice_site are sums over the wave: LDS atomics instead of a shuffle tree
                    // (two dependent six-step ds_bpermute chains per candidate transcript otherwise)
                    if (lane == 0) { g_cnt[wv][0] = 0u; g_cnt[wv][1] = 0u; }
                    if (r_in) atomicAdd(&g_cnt[wv][0], (uint32_t)r_in);
                    if (same) atomicAdd(&g_cnt[wv][1], (uint32_t)same);
                    r_in = (int)g_cnt[wv][0]; same = (int)g_cnt[wv][1];
                    if (2 * (n - 1) == r_in && r_in == same) { vv = 1; known = true; }
                    else if (same > 0) { vv = 2; ksite = true; }
                }
                if (vv) { ref = j; ref_rev = h1.y; }
                if (vv == 1) { done = true; break; }                                       // :810,816
              }
            }
            bool out_rev = rev;
            if (ref >= 0) out_rev = ref_rev != 0;               // :825-831 strand taken from the reference transcript
            uint32_t info = 0;
            if (known) info |= I_KNOWN;
            if (ksite) info |= I_KSITE;
            if (full_decision(level, lfull, lnoth, rfull, rnoth)) info |= I_FULL;
            if (out_rev) info |= I_REV;
            v = Verdict{finish_info(info, n, p), ref};
            for (int k = lane; k < n; k += WAVE) ex_flag[off + (uint32_t)k] = (uint8_t)F[k];
        }
        if (lane == 0) {
            info_io[r] = v.info;
            ref_out[r] = v.ref;
            if ((v.info & I_ACCEPT) && (p.want & WANT_ACCEPTED)) {
                int lo = 0, hi = n_tiles;                  // the tile of read r: last t with tile_first[t] <= r
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tile_first[mid] <= r) lo = mid; else hi = mid; }
                const uint32_t t = (uint32_t)lo;
                atomicAdd(&tile_acc[t], 1u);
                atomicAdd(&tile_acc_ex[t], (uint32_t)n);
            }
        }
    }
}

// ------------------------------------------------------------------ fast classification (-d 0)
//
// With -d 0, check_splice_site only asks "is this read coordinate (pair) also a site of the transcript".
// The dictionaries answer that for ALL transcripts at once: a probe returns the set of transcripts that
// contain the site as a bit mask.  The tile re-bases those masks, while staging them in LDS, into its own frame:
// bit b = transcript j_lo + b of the tile's window.  Per read:
//     V'   = window transcripts the sequential sweep would visit and that overlap the read  (one pass over the window)
//     AND over the 2(n-1) probed sites of their masks = transcripts that contain every site  -> "known" candidates
//     OR  ...                                         = transcripts that share a site       -> "has known site"
//     the first known candidate j* in V' (file order) ends the sweep: V = V' up to j*
//     a novel_* flag is cleared iff its site's mask meets V: per site the first member in V' is kept (7 bits)
// Preconditions, checked per read / per visited transcript; anything else goes to the redo list:
//   * strictly increasing exon starts and ends and start <= end on both sides (then a value pairs with at most one
//     value of the other chain, pair count == common values, and equality implies that the site lies inside both
//     spans, i.e. inside the overlap window of check_splice_site);
//   * the read's sweep ends inside the window (WIN_TX transcripts from the tile's smallest cursor value);
//   * no dictionary entry of the tile's slices has members beyond 64 transcripts of its first.


struct AccRec { uint32_t read_lo, read_hi, info; int32_t ref_tx; };

struct FastArgs {
    int64_t n_reads;
    const int32_t *r_tid; const int32_t *r_pos; const uint8_t *r_rev; const int64_t *cig_off; const uint32_t *cig;
    const int2 *walked;          // long-CIGAR inputs: the exons pass A has walked (tile * LDS_EXON_CAP + in-tile offset), tiles with TD_WALKED
    const uint32_t *local; const uint8_t *order; const uint32_t *tile_base; const int32_t *j0; const TileDesc *desc;
    const TxHdr *win_hdr;        // per tile WIN_TX header copies, the annotation index in the spare word (pass A)
    const TxHdr *hdr; SiteDict st, en;
    uint32_t *ex_off; int32_t *ex_start; int32_t *ex_end; uint8_t *ex_flag; uint32_t *info; int32_t *ref_tx;
    uint32_t *tile_acc, *tile_acc_ex; uint32_t *redo_count, *redo;
    uint32_t *tile_chunk, *tile_rchunk;           // accepted list: first exon slot / first record slot of every tile's chunk
    unsigned long long *chunk_cursor;             // next free {record slot (high word), exon slot (low word)}
    int32_t *acc_start, *acc_end; uint8_t *acc_flag; AccRec *acc_rec; uint32_t *acc_ex_off; int64_t first_read;
    unsigned long long *stamps;
    DevParams p;
};

__device__ __forceinline__ uint32_t rebase_mask(uint32_t lo, uint32_t hi, int d)
{
    // 64-bit mask relative to tx_base -> 32 bits relative to j_lo, d = tx_base - j_lo
    const unsigned long long m = ((unsigned long long)hi << 32) | lo;
    if (d >= 0) return d < 32 ? (uint32_t)(m << d) : 0u;
    return -d < 64 ? (uint32_t)(m >> (-d)) : 0u;
}

// The same for a window whose members win[0 .. w_n) are not consecutive (rare: the loop is kept small on purpose, the
// staging code around it has little register room)
__device__ __forceinline__ uint32_t rebase_gaps(const int *win, int w_n, uint32_t lo, uint32_t hi, int tx_base)
{
    uint32_t out = 0u;
#pragma unroll 1
    for (int j = 0; j < w_n; ++j) {
        const uint32_t b = (uint32_t)(win[j] - tx_base);
        const uint32_t word = b < 32u ? lo : hi;
        out |= (b < 64u ? (word >> (b & 31u)) & 1u : 0u) << j;
    }
    return out;
}

// Staged dictionary entries are read as one 16-byte vector {k1, k2, pm, sm}: one ds_read_b128.
__device__ __forceinline__ v4i_t lds_entry(const v4i_t *ent, uint32_t i) { return ent[i]; }

// One probe of a staged slice: pm of the entry equal to (k1, k2), sm of any entry whose first key is k1.
// lo/hi = the bucket's entry range.  The first two entries are examined without a branch (buckets are 512 bp,
// most hold 0..2 entries); the caller runs probe_rest when some lane has a longer bucket.
__device__ __forceinline__ void probe2(const v4i_t q0, const v4i_t q1, uint32_t lo, uint32_t hi, int32_t k1, int32_t k2, uint32_t &pm, uint32_t &sm)
{
    const bool m0 = lo < hi && q0.x == k1, m1 = lo + 1u < hi && q1.x == k1;
    sm = m1 ? (uint32_t)q1.w : (m0 ? (uint32_t)q0.w : 0u);
    pm = (m1 && q1.y == k2) ? (uint32_t)q1.z : ((m0 && q0.y == k2) ? (uint32_t)q0.z : 0u);
}

__device__ __forceinline__ void probe_rest(const v4i_t *ent, uint32_t lo, uint32_t hi, int32_t k1, int32_t k2, uint32_t &pm, uint32_t &sm, uint32_t)
{
    for (uint32_t r = lo; r < hi; ++r) {
        const v4i_t q = lds_entry(ent, r);
        if (q.x == k1) { sm = (uint32_t)q.w; if (q.y == k2) pm = (uint32_t)q.z; }
    }
}

// -d > 0 (src/update_gtf.c:717-779 with dis > 0): a read site matches EVERY annotation site within `dis` of it, and identical_site_n
// counts the matching (annotation site, read site) PAIRS.  One probe of a staged slice with a tolerance: entries [lo, hi) are the
// buckets of k1 - dis .. k1 + dis (one or two of them: 2 dis + 1 <= 129 bases with dis <= DIS_MASK_MAX; the range arithmetic of
// near_range would take more -- lo = the first bucket's first entry, hi = the last bucket's end), sorted by (key 1, key 2).  The staged
// directories are bytes: a slice holds at most SLAB_KEY_CAP / KEY_CAP (< 256) entries whatever the tolerance adds to its span.
//   sm |= site members of every entry whose first key lies within dis of k1 AND inside the read's span [rs, re] -- an annotation site
//         counts only inside the overlap span (:732,742); inside the transcript's own span it always is (TX_COMPACT);
//   pm |= pair members of every entry with both keys within dis (the exon / junction flags know no overlap span, :753-768);
//   amb |= members that have TWO different sites within dis of this one read site: for them the pair count is not the number of
//         matched read sites, the masks cannot say whether the read is known -- such a read goes to the generic kernel (it takes a
//         transcript with two donors or two acceptors less than 2 dis + 1 bases apart).
static_assert(2 * DIS_MASK_MAX + 1 <= (1 << SITE_SHIFT) && KEY_CAP < 256, "a tolerance window spans two buckets at most; byte directories");
__device__ __forceinline__ void probe_near(const v4i_t *ent, uint32_t lo, uint32_t hi, int32_t k1, int32_t k2, int dis, int rs, int re,
                                           uint32_t &pm, uint32_t &sm, uint32_t &amb)
{
    pm = 0u; sm = 0u;
    int last = INT32_MIN;
    const uint32_t span = 2u * (uint32_t)dis, b1 = (uint32_t)k1 - (uint32_t)dis, b2 = (uint32_t)k2 - (uint32_t)dis;
    // (no branch per entry: selects; an entry outside the tolerance adds nothing, sorted or not)
    auto one = [&](const v4i_t q, bool on) {
        const bool in = on && (uint32_t)q.x - b1 <= span;
        const bool sp = in && q.x >= rs && q.x <= re;
        const uint32_t w = sp ? (uint32_t)q.w : 0u;
        amb |= q.x != last ? (sm & w) : 0u;
        last = sp ? q.x : last;
        sm |= w;
        pm |= (in && (uint32_t)q.y - b2 <= span) ? (uint32_t)q.z : 0u;
    };
    // the first NEAR_FIRST entries without a loop (most tolerance windows hold no more), the others only where some lane has them
    const v4i_t q0 = lds_entry(ent, lo), q1 = lds_entry(ent, lo + 1u);
    one(q0, lo < hi); one(q1, lo + 1u < hi);
    if (__any(hi > lo + 2u))
        for (uint32_t r = lo + 2u; r < hi; ++r) one(lds_entry(ent, r), true);
}
// the staged entries of the buckets of x - dis .. x + dis: [lo, hi)  (none: the empty bucket behind the staged ones)
__device__ __forceinline__ void near_range(const uint8_t *dir, int b_off, uint32_t none, bool live, int x, int dis, uint32_t &lo, uint32_t &hi)
{
    const uint32_t i0 = live ? min((uint32_t)((max(x - dis, 0) >> SITE_SHIFT) + b_off), none) : none;
    const uint32_t i1 = live ? min((uint32_t)(((x + dis) >> SITE_SHIFT) + b_off), none) : none;
    lo = dir[i0]; hi = dir[i1 + 1u];
}

// Transcripts (tile frame) that have an exon overlapping [s, e]: union of the exon masks of the START entries
// from the first one that reaches into the bucket of s up to the last one that starts in the bucket of e.
__device__ __forceinline__ uint32_t overlapping_exon_members(const uint8_t *rdir, const uint8_t *dir, const v4i_t *ent,
                                                             int b_off, int nb, int s, int e)
{
    const int bs = s >> SITE_SHIFT;
    if (bs >= nb) return 0u;                                 // beyond the last annotated site of the chromosome
    const int be = min(e >> SITE_SHIFT, nb - 1);
    uint32_t m = 0u;
    const uint32_t i1 = dir[be + b_off + 1];
    for (uint32_t i = rdir[bs + b_off]; i < i1; ++i) {
        const v4i_t q = ent[i];
        if (q.x <= e && q.y >= s) m |= (uint32_t)q.z;
    }
    return m;
}

// The kernel is PERSISTENT and software pipelined: a workgroup walks over tiles t, t + grid, ... and holds the
// raw inputs of its next tile in registers while it computes the current one, so no tile waits on HBM:
//     uniforms of tile t+G (descriptor, offsets)      issued at the top of tile t
//     vectors of tile t+G (CIGAR words, per-read fields, dictionary entries, directory words, headers)
//                                                     issued after tile t has staged its own dictionary
//     ... consumed (registers -> LDS) at the top of tile t+G.
constexpr int PF_CIG_VEC = 4;                              // 16-byte CIGAR vectors a thread holds for the next tile
constexpr int FAST_DIR_BYTES = (DIR_CAP + 2 + 3) & ~3;
constexpr int FAST_TAIL_WORDS = LDS_EXON_CAP / 2 + 2 * KEY_CAP * 4 + 3 * FAST_DIR_BYTES / 4;
constexpr int FAST_ALL_WORDS = 2 * LDS_EXON_CAP + FAST_TAIL_WORDS;

// base[idx] with a 32-bit BYTE offset, so that the load takes the form "uniform base + 32-bit lane offset" (one shift per
// lane instead of a 64-bit multiply-add); l2r_upload_reads keeps every array of a shard below 4 GB.
template <typename T>
__device__ __forceinline__ T ld32(const T *base, uint32_t idx)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)(idx * (uint32_t)sizeof(T)));
}

// The kernel's argument block is read through the kernarg segment pointer, re-fetched opaquely at every phase: a
// pointer is then loaded (one scalar load) where a phase needs it instead of occupying two SGPRs for the whole tile
// loop -- with ~25 arrays the alternative is dozens of SGPR spills and v_readlane reloads per tile.
typedef const __attribute__((address_space(4))) FastArgs *FastArgsK;
__device__ __forceinline__ FastArgsK fast_args()
{
    FastArgsK q = (FastArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}

struct TileUniforms {                                       // wave-uniform inputs of a tile ...
    TileDesc d;
    uint32_t base, total;
    uint32_t c0, c1;                                        // the tile's CIGAR words [c0, c1) (a shard has < 2^32 of them)
    uint32_t r0, n_act;                                     // the tile's reads [r0, r0 + n_act)
    int32_t src;                                            // ... and the thread's read of the tile (pass A's order), -1: none
};

struct TileVectors {                                        // per-thread raw inputs of a tile
    uint32_t local, nxt;
    uint32_t c_lo, c_hi;
    int32_t pos, j0, tid, rev;
    uint4 cg[PF_CIG_VEC];
    int4 xa, xb, xc, xd;        // threads < KEY_CAP: START entry (xa, xb) and END entry (xc, xd); threads >= KEY_CAP: header words (xa, xb, xc)
    uint32_t dd[3][2];
};

// (the three arrays come in as __restrict__ kernel parameters of their own, so that these are scalar loads that
//  wait at their first use, not vector loads that wait where they are issued)
__device__ __forceinline__ TileUniforms load_uniforms(const TileDesc *__restrict__ desc, const uint32_t *__restrict__ tile_base,
                                                     const int64_t *__restrict__ cig_off, const uint8_t *__restrict__ order,
                                                     const uint32_t *__restrict__ tile_first, uint32_t t)
{
    // 32-bit indices throughout: a shard has fewer than 2^32 reads and CIGAR words (l2r_upload_reads checks)
    TileUniforms u;
    const uint32_t r0 = tile_first[t], r1 = tile_first[t + 1u];
    u.r0 = r0; u.n_act = r1 - r0;
    u.d = desc[t];
    u.base = tile_base[t]; u.total = tile_base[t + 1u] - u.base;
    u.c0 = (uint32_t)cig_off[r0]; u.c1 = (uint32_t)cig_off[r1];
    // Thread -> slot of pass A's order.  A workgroup's wave w runs on SIMD w of its CU: the slots are rotated by one
    // wave per tile (a persistent workgroup sees tiles b, b + 1024, ...), which spreads the waves with the long reads
    // -- and, for tiles of fewer than 256 reads, the only waves that hold reads at all -- over the four SIMDs.
    const uint32_t slot = (threadIdx.x - (((t + (t >> 10)) & 3u) << 6)) & (uint32_t)(TILE_THREADS - 1);
    u.src = slot < r1 - r0 ? (int32_t)ld32(order, r0 + slot) : -1;
    return u;
}

// CIGAR staging geometry: the copy starts at the 16-byte boundary below c0 (the array is padded at its end)
__device__ __forceinline__ int cigar_vectors(const TileUniforms &u) { return (int)((u.c1 - (u.c0 & ~3u) + 3u) >> 2); }
// LDS words left for the CIGAR of a tile behind its S and E arrays (layout in k_classify_fast)
__device__ __forceinline__ int cigar_room(const TileUniforms &u)
{
    const uint32_t lay = u.total <= (uint32_t)LDS_EXON_CAP ? u.total : 0u;
    return FAST_ALL_WORDS - (int)(2u * ((lay + 3u) & ~3u));
}
__device__ __forceinline__ bool cigar_staged(const TileUniforms &u, int region_words)
{
    const int n4 = cigar_vectors(u);
    return n4 <= PF_CIG_VEC * TILE_THREADS && 4 * n4 <= region_words;
}

__device__ __forceinline__ TileVectors load_vectors(FastArgsK a, uint32_t t, const TileUniforms &u, int region_words, bool want_cigar)
{
    TileVectors v;
    // every pointer this needs, fetched from the argument block in one go (left to the compiler, each one is loaded
    // inside the branch that uses it and waited for there: a chain of scalar-cache round trips per tile)
    const uint32_t *const p_local = a->local; const int64_t *const p_cig_off = a->cig_off; const uint32_t *const p_cig = a->cig;
    const int32_t *const p_pos = a->r_pos, *const p_tid = a->r_tid, *const p_j0 = a->j0; const uint8_t *const p_rev = a->r_rev;
    const TxHdr *const p_win = a->win_hdr; const SiteEnt *const p_st = a->st.ent, *const p_en = a->en.ent;
    const uint32_t *const p_sd = a->st.dir, *const p_ed = a->en.dir, *const p_sr = a->st.rdir;
    asm volatile("" :: "s"(p_local), "s"(p_cig_off), "s"(p_cig), "s"(p_pos), "s"(p_tid), "s"(p_j0), "s"(p_rev), "s"(p_win),
                 "s"(p_st), "s"(p_en), "s"(p_sd), "s"(p_ed), "s"(p_sr));
    const uint32_t r = u.r0 + (uint32_t)u.src;
    const bool active = u.src >= 0;
    v.local = 0; v.nxt = 0; v.c_lo = 0; v.c_hi = 0; v.pos = 0; v.j0 = 0; v.tid = 0; v.rev = 0;
    if (active) {
        v.local = ld32(p_local, r);
        const bool last = (uint32_t)u.src + 1u == u.n_act;
        v.nxt = last ? u.total : ld32(p_local, r + 1u);
        v.c_lo = (uint32_t)ld32(p_cig_off, r); v.c_hi = (uint32_t)ld32(p_cig_off, r + 1u);
        v.pos = ld32(p_pos, r); v.tid = ld32(p_tid, r); v.j0 = ld32(p_j0, r); v.rev = ld32(p_rev, r);
    }
    const bool staged = want_cigar && cigar_staged(u, region_words);
    const int n4 = cigar_vectors(u);
    const uint4 *src = reinterpret_cast<const uint4 *>(p_cig + (u.c0 & ~3u));
#pragma unroll
    for (int q = 0; q < PF_CIG_VEC; ++q) {
        const int i = q * TILE_THREADS + (int)threadIdx.x;
        v.cg[q] = make_uint4(0u, 0u, 0u, 0u);
        if (staged && i < n4) v.cg[q] = ld32(src, (uint32_t)i);
    }
    const TileDesc &d = u.d;
    const bool fast = (d.flags & TD_FAST) != 0;
    const int w_n = fast ? (int)d.n_win : 0;
    // One load site per register group, whatever the thread stages (a START and an END entry, or a window header): with a
    // load per role the compiler merges the roles through register copies -- and waits for the loads right here.
    v.xa = v.xb = v.xc = v.xd = make_int4(0, 0, 0, 0);
    const bool hdr_role = (int)threadIdx.x >= KEY_CAP;
    const int4 *const hp = reinterpret_cast<const int4 *>(p_win + (t * (uint32_t)WIN_TX + (threadIdx.x - (uint32_t)KEY_CAP)));
    const int4 *const qs = reinterpret_cast<const int4 *>(p_st + d.st_r0 + threadIdx.x), *const qe = reinterpret_cast<const int4 *>(p_en + d.en_r0 + threadIdx.x);
    const int4 *const pa = hdr_role ? hp : qs, *const pc = hdr_role ? hp + 2 : qe;          // (the header's third word; xd is not used then)
    const bool va = hdr_role ? (int)threadIdx.x - KEY_CAP < w_n : (fast && threadIdx.x < d.st_nk);
    const bool vc = hdr_role ? (int)threadIdx.x - KEY_CAP < w_n : (fast && threadIdx.x < d.en_nk);
    if (va) { v.xa = pa[0]; v.xb = pa[1]; }
    if (vc) { v.xc = pc[0]; if (!hdr_role) v.xd = pc[1]; }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = (int)threadIdx.x + q * TILE_THREADS;
        v.dd[0][q] = v.dd[1][q] = v.dd[2][q] = 0u;
        if (fast && d.nbk > 0 && i <= d.nbk) { const uint32_t b = (uint32_t)(d.b0 + i); v.dd[0][q] = ld32(p_sd, b); v.dd[1][q] = ld32(p_ed, b); v.dd[2][q] = ld32(p_sr, b); }
    }
    return v;
}

// Makes every prefetched register of `v` a (no-op) operand: the compiler has to wait for those loads here.  Placed in
// front of the tile's write-out: the loads are long done by then, the wait is free -- and without it the first use of
// `v` at the top of the next tile waits with vmcnt(small), which on gfx9 (one in-order counter for loads and stores)
// also waits for the write-out's stores, issued moments before, to be acknowledged.
__device__ __forceinline__ void settle_vectors(const TileVectors &v)
{
    asm volatile("" :: "v"(v.local), "v"(v.nxt), "v"(v.c_lo), "v"(v.c_hi), "v"(v.pos), "v"(v.j0), "v"(v.tid), "v"(v.rev),
                 "v"(v.dd[0][0]), "v"(v.dd[0][1]), "v"(v.dd[1][0]), "v"(v.dd[1][1]), "v"(v.dd[2][0]), "v"(v.dd[2][1]));
    asm volatile("" :: "v"(v.cg[0].x), "v"(v.cg[0].y), "v"(v.cg[0].z), "v"(v.cg[0].w), "v"(v.cg[1].x), "v"(v.cg[1].y), "v"(v.cg[1].z), "v"(v.cg[1].w),
                 "v"(v.cg[2].x), "v"(v.cg[2].y), "v"(v.cg[2].z), "v"(v.cg[2].w), "v"(v.cg[3].x), "v"(v.cg[3].y), "v"(v.cg[3].z), "v"(v.cg[3].w));
    asm volatile("" :: "v"(v.xa.x), "v"(v.xa.y), "v"(v.xa.z), "v"(v.xa.w), "v"(v.xb.x), "v"(v.xb.y), "v"(v.xb.z), "v"(v.xb.w),
                 "v"(v.xc.x), "v"(v.xc.y), "v"(v.xc.z), "v"(v.xc.w), "v"(v.xd.x), "v"(v.xd.y), "v"(v.xd.z), "v"(v.xd.w));
}

// ---- the three classification phases of a tile (all lanes of a wave call them together) -------------------------

struct TileLds {                 // the tile's LDS image (layout in k_classify_fast)
    int *S, *E; uint16_t *W;
    const v4i_t *ent0, *ent1; const uint8_t *dir0, *dir1, *rdir;
    const int4 *hk, *hx;
    const int *win;              // window member -> annotation index
};
// m = m << 1 | predicate, as ONE compare and ONE add-with-carry (m + m + carry): the member passes collect their per-member predicates
// highest member first -- the select-and-or form costs three vector instructions per predicate on 32-bit masks, five on 64-bit ones.
__device__ __forceinline__ void shift_in_le(uint32_t &m, int a, int b)          // m = m << 1 | (a <= b)
{
    asm volatile("v_cmp_le_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void shift_in_eq(uint32_t &m, int a, int b)          // m = m << 1 | (a == b)
{
    asm volatile("v_cmp_eq_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void shift_in_le2(uint32_t &m, int a, int b, int c, int d)      // m = m << 1 | (a <= b && c <= d)
{
    unsigned long long t;
    asm volatile("v_cmp_le_i32 vcc, %2, %3\n\tv_cmp_le_i32 %1, %4, %5\n\ts_and_b64 vcc, vcc, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                 : "+v"(m), "=&s"(t) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc", "scc");
}
struct VisitMasks { uint32_t vpre, lmask, rmask, k1mask; bool redo; };
struct SiteMasks { uint32_t kand, kor, dm_first, am_last; uint32_t amb; };      // amb (-d > 0 only): members with TWO sites within the tolerance of one read site (probe_near)

// V' = window transcripts j >= j0 up to the first one the read lies before (src/update_gtf.c:799-800), minus the ones
// that lie before the read (:801); terminal-exon masks of check_full (:629-681); single-exon known candidates (:806-811).
// One pass over the window with a wave-uniform member index (header words broadcast from LDS) that only BUILDS per-read
// bit masks -- "the read lies before member j", "member j lies before the read", "first / last exons overlap" -- with no
// per-member branch; the sweep's stop (:799) and the cursor (:801-802) are applied to the masks afterwards:
//     V' = ~before & (members below the first "read lies before") & (members from the cursor value on).
// tilemask[0] / [1]: the tile's members with one exon / without TX_COMPACT (set where the headers are staged).
template <int LEVEL>
__device__ __forceinline__ VisitMasks visit_window(const TileLds &L, const TileDesc &d, int w_n, bool work, uint32_t n, int j0,
                                                   const ReadEnds &re, const uint32_t *tilemask)
{
    VisitMasks m{0u, 0u, 0u, 0u, false};
    // (a wave without a read to classify -- the upper waves of a tile of few reads -- has nothing to visit: the pass below would
    //  still cost it a dozen instructions per member)
    if (!__any(work)) return m;
    uint32_t m_aft = 0u, m_bef = 0u;
    for (int j = 0; j < w_n; ++j) {
        const int4 hk = L.hk[j];
        const uint32_t bit = 1u << j;
        m_aft |= re.el <= hk.x ? bit : 0u;                                   // comp_trans <= (Q5): the read lies before the member
        m_bef |= hk.y <= re.s0 ? bit : 0u;                                   // the member lies before the read
        if (LEVEL >= 1 && LEVEL <= 4) {
            const int4 hx = L.hx[j];
            if (LEVEL == 1) {
                m.lmask |= re.e0 == hx.y ? bit : 0u;
                m.rmask |= re.sl == hx.z ? bit : 0u;
            } else {
                m.lmask |= closed_overlap(re.s0, re.e0, hx.x, hx.y) ? bit : 0u;
                if (LEVEL != 4) m.rmask |= closed_overlap(re.sl, re.el, hx.z, hx.w) ? bit : 0u;
            }
        }
    }
    int jrel0 = j0 - d.j_lo;                       // first member the read's sweep reaches
    if (!(d.flags & TD_CONTIG)) {
        jrel0 = 0;
        for (int j = 0; j < w_n; ++j) jrel0 += L.win[j] < j0 ? 1 : 0;
    }
    // members the sweep reaches: index >= jrel0 (which may be negative or beyond the window)
    const uint32_t reach = jrel0 <= 0 ? 0xffffffffu : (jrel0 >= 32 ? 0u : ~((1u << jrel0) - 1u));
    const uint32_t stop = m_aft & reach;                                     // the sweep ends at the lowest of these (:799-800)
    const uint32_t below = (stop & (0u - stop)) - 1u;                        // all ones when there is none
    m.vpre = work ? (~m_bef & below & reach & (w_n >= 32 ? 0xffffffffu : ((1u << w_n) - 1u))) : 0u;
    m.lmask &= m.vpre; m.rmask &= m.vpre;
    // single-exon members only count against single-exon reads (:806-811); a member without TX_COMPACT needs the literal loops
    const uint32_t single = tilemask[0];
    if (n == 1) {
        uint32_t c = m.vpre & single;
        while (c) {
            const int j = __ffs((int)c) - 1;
            c &= c - 1u;
            const int4 hx = L.hx[j];
            if (overlap_frac(re.s0, re.e0, hx.x, hx.y) >= fast_args()->p.frac) m.k1mask |= 1u << j;
        }
    } else if (m.vpre & tilemask[1] & ~single) m.redo = true;
    return m;
}

// index of the lowest set bit, 63 when there is none
__device__ __forceinline__ uint32_t first_member(uint32_t x)
{
    return (uint32_t)(__ffs((int)x) - 1) & 63u;                         // v_ffbl_b32 (-1 for 0), masked to 63
}

// x != 0 as 0 / 1 in one VALU instruction (the compiler turns min(x, 1) into compare + select + a literal move)
__device__ __forceinline__ uint32_t nonzero(uint32_t x)
{
    uint32_t r;
    asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(x));
    return r;
}

// -d > 0: the loop of map_exons (below) with every probe looking at the entries within the tolerance of its coordinate (probe_near); [rs, re] = the
// read's span.  (A loop of its own: a wave-uniform branch inside the exact loop cost k_classify_fast 5 % on config 5's shard.)
__device__ __forceinline__ SiteMasks map_exons_near(const TileLds &L, const TileDesc &d, bool mapping, uint32_t local, uint32_t n, uint32_t vpre, int dis, int rs, int re)
{
    SiteMasks m{0xffffffffu, 0u, 0u, 0u, 0u};
    const int *S = L.S + local, *E = L.E + local;
    uint16_t *W = L.W + local;
    int s = 0, e = 0;
    if (mapping) { s = S[0]; e = E[0]; }
    const uint32_t none = (uint32_t)d.nbk + 1u;
    const int k_max = wave_max(mapping ? (int)n : 0);
    for (int k = 0; k < k_max; ++k) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        const uint32_t inext = junc ? local + (uint32_t)k + 1u : 0u;
        const int s2 = L.S[inext], e2 = L.E[inext];
        uint32_t xm, am, jm, dm, ls, hs, le, he;
        near_range(L.dir0, d.b_off, none, live, s, dis, ls, hs); near_range(L.dir1, d.b_off, none, junc, e, dis, le, he);
        probe_near(L.ent0, ls, hs, s, e, dis, rs, re, xm, am, m.amb);
        probe_near(L.ent1, le, he, e, s2, dis, rs, re, jm, dm, m.amb);
        const uint32_t amj = junc ? am : 0u;
        uint32_t word = first_member(xm & vpre);
        word |= first_member(jm & vpre) << 6;
        word |= nonzero(dm & vpre) << 12;
        word |= nonzero(amj & vpre) << 13;
        m.kand &= junc ? (am & dm) : 0xffffffffu;     // Q1: the acceptor probed with exon k is ITS OWN start, k < n-1
        m.kor |= amj | dm;
        // (with a tolerance a shared donor / acceptor does not say that the exons overlap: dm_first / am_last stay empty, the
        //  full-length evidence asks the START slice)
        if (live) W[k] = (uint16_t)word;
        s = s2; e = e2;
    }
    return m;
}

// One START and one END probe per exon; the wave runs as many rounds as its longest read has exons.  Per round
// {next exon, both bucket ranges} are read together, then the first entries of both buckets.  Leaves per exon in W:
// first member of V' with the exon / the junction (6 bits each, 63: none), "donor / acceptor is in V'" (bits 12, 13).
__device__ __forceinline__ SiteMasks map_exons(const TileLds &L, const TileDesc &d, bool mapping, uint32_t local, uint32_t n, uint32_t vpre)
{
    SiteMasks m{0xffffffffu, 0u, 0u, 0u, 0u};
    const int *S = L.S + local, *E = L.E + local;
    uint16_t *W = L.W + local;
    int s = 0, e = 0;
    if (mapping) { s = S[0]; e = E[0]; }
    const uint32_t none = (uint32_t)d.nbk + 1u;         // a bucket behind the staged ones: the staging leaves it empty
    // (scalar trip count: the wave's longest mapped read)
    const int k_max = wave_max(mapping ? (int)n : 0);
    for (int k = 0; k < k_max; ++k) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        const uint32_t inext = junc ? local + (uint32_t)k + 1u : 0u;
        // (unsigned: a coordinate below the staged span wraps to a huge index and lands on `none` as well)
        const uint32_t is = live ? min((uint32_t)((s >> SITE_SHIFT) + d.b_off), none) : none;
        const uint32_t ie = junc ? min((uint32_t)((e >> SITE_SHIFT) + d.b_off), none) : none;
        const int s2 = L.S[inext], e2 = L.E[inext];
        const uint32_t ls = L.dir0[is], hs = L.dir0[is + 1u], le = L.dir1[ie], he = L.dir1[ie + 1u];
        // START buckets mostly hold one exon, END buckets often several junctions of one donor.  (An index up to
        // KEY_CAP + 1 reads on into the arrays behind the slice; such an entry is never inside [ls, hs).)
        const v4i_t qs0 = lds_entry(L.ent0, ls);
        const v4i_t qe0 = lds_entry(L.ent1, le), qe1 = lds_entry(L.ent1, le + 1u);
        uint32_t xm, am, jm, dm;
        {   const bool m0 = ls < hs && qs0.x == s;
            am = m0 ? (uint32_t)qs0.w : 0u; xm = (m0 && qs0.y == e) ? (uint32_t)qs0.z : 0u; }
        probe2(qe0, qe1, le, he, e, s2, jm, dm);
        if (__any(hs > ls + 1u || he > le + 2u)) { probe_rest(L.ent0, ls + 1u, hs, s, e, xm, am, 0u); probe_rest(L.ent1, le + 2u, he, e, s2, jm, dm, 0u); }
        // without a junction jm = dm = 0 (empty bucket); the acceptor of the last exon is not a probed site (Q1)
        const uint32_t amj = junc ? am : 0u;
        uint32_t word = first_member(xm & vpre);
        word |= first_member(jm & vpre) << 6;
        word |= nonzero(dm & vpre) << 12;
        word |= nonzero(amj & vpre) << 13;
        m.kand &= junc ? (am & dm) : 0xffffffffu;     // Q1: the acceptor probed with exon k is ITS OWN start, k < n-1
        m.kor |= amj | dm;
        if (k == 0) m.dm_first = dm;
        m.am_last = (live && !junc) ? am : m.am_last;  // transcripts in which the last exon's start begins a later exon
        if (live) W[k] = (uint16_t)word;
        s = s2; e = e2;
    }
    return m;
}

// Known / known site / reference transcript / full-length / flag bytes of one read from its masks.
// getw(k) = the work word map_exons left for exon k, emit(k, flag byte) receives the exons' flags (the classic kernel keeps both in its
// 16-bit W array, the slab kernel in the upper bits of its staged positions).
template <int LEVEL, typename GetW, typename Emit>
__device__ __forceinline__ Verdict decide(const TileLds &L, const TileDesc &d, uint32_t n, const ReadEnds &re,
                                          const VisitMasks &vm, const SiteMasks &sm, bool rev_in, GetW getw, Emit emit)
{
    // ---- first known transcript in visiting order
    int jstar = -1;
    if (n > 1) {
        // every probed site is in the transcript; known also needs every read site inside the overlap span:
        // donors e_0..e_{n-2} and acceptors s_1..s_{n-1} increase, so e_0 >= a.start and s_{n-1} <= a.end suffice
        uint32_t c = sm.kand & vm.vpre;
        while (c) {
            const int j = __ffs((int)c) - 1;
            c &= c - 1u;
            const int4 hk = L.hk[j];
            if (hk.x <= re.e0 && re.sl <= hk.y) { jstar = j; break; }
        }
    } else if (vm.k1mask) jstar = __ffs((int)vm.k1mask) - 1;
    const bool known = jstar >= 0;
    const uint32_t V = known ? (vm.vpre & ((2u << jstar) - 1u)) : vm.vpre;
    const uint32_t ks = (n > 1) ? (sm.kor & V) : 0u;
    const bool ksite = (ks & ~(known ? (1u << jstar) : 0u)) != 0u;
    int jref = -1;
    if (n > 1) { if (ks) jref = 31 - __clz((int)ks); }
    else jref = jstar;
    // ---- full-length evidence (:629-681) over V.  lfull: the first exon of a member of V overlaps the read's first
    // exon.  lnoth stays set unless SOME exon of a member of V overlaps it; that is certain when the read's first donor
    // is a donor of a member (the exon that ends there), else the START slice decides.
    bool lfull = false, rfull = false, lnoth = true, rnoth = true;
    if (LEVEL >= 1 && LEVEL <= 4) { lfull = (vm.lmask & V) != 0u; rfull = (vm.rmask & V) != 0u; }
    if (LEVEL == 3 || LEVEL == 4) {
        if (!lfull) {
            if (sm.dm_first & V) lnoth = false;
            else if (V) lnoth = (overlapping_exon_members(L.rdir, L.dir0, L.ent0, d.b_off, d.nb, re.s0, re.e0) & V) == 0u;
        }
        if (LEVEL == 3 && !rfull) {
            if (sm.am_last & V) rnoth = false;
            else if (V) rnoth = (overlapping_exon_members(L.rdir, L.dir0, L.ent0, d.b_off, d.nb, re.sl, re.el) & V) == 0u;
        }
    }
    // ---- flags: an exon / junction is no longer novel iff the first member of V' that has it comes no later than
    // j*; a known read has every probed donor and acceptor in transcript j*
    const uint32_t lim = known ? (uint32_t)jstar : 62u;
    const uint32_t site_bits = known ? 0u : (uint32_t)(F_DON | F_ACC);         // bits 12, 13 of a work word: the site is in V' (F_DON = 2, F_ACC = 4)
    if (n > 1) {
        for (int k = 0; k < (int)n; ++k) {
            const uint32_t w = getw(k);
            uint32_t f = ((w & 63u) > lim ? (uint32_t)F_EXON : 0u) | (((w >> 6) & 63u) > lim ? (uint32_t)F_JUNC : 0u) | ((~w >> 11) & site_bits);
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
    // routing of update_gtf.c:943-950 when there is no junction table (finish_info)
    if (fast_args()->p.n_sj == 0 && (info & (I_FULL | I_KNOWN | I_KSITE)) == (I_FULL | I_KSITE)) info |= I_ACCEPT;
    return Verdict{info | (n << 8), ref};
}

template <int LEVEL, bool WIDE>
__global__ __launch_bounds__(TILE_THREADS, 4)
void k_classify_fast(FastArgs kernarg_block /* read through fast_args() */, int64_t n_tiles, const TileDesc *__restrict__ u_desc, const uint32_t *__restrict__ u_tile_base,
                     const int64_t *__restrict__ u_cig_off, const uint8_t *__restrict__ u_order, const uint32_t *__restrict__ u_tile_first)
{
    // One LDS array per workgroup, laid out per tile (total = the tile's exon count <= LDS_EXON_CAP):
    //   [0, total)            S   exon starts            [total, 2 total)   E   exon ends
    //   tail = 2 total rounded up to 4 words, at least TAIL_WORDS long, used twice:
    //     phase 1: the tile's CIGAR words (when they fit; else the walk reads HBM)
    //     then:    W     16 bits per exon: first member of V' with the exon / with the junction (6 bits each),
    //                    "donor / acceptor not in V'" (1 bit each); later the flag byte
    //              ent   dictionary entries {k1, k2, pm, sm} (START, END)
    //              dir   entry index relative to the slice per staged bucket, 8 bits (START, END, START reach-back)
    constexpr int DIR_BYTES = FAST_DIR_BYTES, W_WORDS = LDS_EXON_CAP / 2, ALL_WORDS = FAST_ALL_WORDS;
    __shared__ __attribute__((aligned(16))) uint32_t s_all[ALL_WORDS];
    __shared__ __attribute__((aligned(16))) int4 s_hk[WIN_TX];     // {start, end, n, flags | rev << 8} on the tile's chromosome
    __shared__ __attribute__((aligned(16))) int4 s_hx[WIN_TX];     // {s0, e0, sl, el}
    __shared__ int s_win[WIN_TX];                                   // window member -> annotation index
    __shared__ uint32_t s_cnt[4][3];
    __shared__ int s_wide;                                          // some staged entry has members beyond its 64-bit masks
    __shared__ uint16_t s_nat[TILE_THREADS];                        // per read of the tile in READ order: exon count, bit 15 = accepted
    __shared__ uint32_t s_chunk[2];
    __shared__ uint32_t s_tilemask[2];                              // window members with one exon / without TX_COMPACT

    (void)kernarg_block;
    const bool stamping = fast_args()->stamps != nullptr;
    unsigned long long t_prev = stamping ? __builtin_readcyclecounter() : 0ull;
#define L2R_STAMP(i) do { if (stamping && threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
        atomicAdd(&fast_args()->stamps[(blockIdx.x & 1023u) * 8u + (i)], t_ - t_prev); t_prev = t_; } } while (0)
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;

    uint32_t t = blockIdx.x;
    if ((int64_t)t >= n_tiles) return;
    TileUniforms u = load_uniforms(u_desc, u_tile_base, u_cig_off, u_order, u_tile_first, t);
    TileVectors v = load_vectors(fast_args(), t, u, cigar_room(u), !WIDE);
    settle_vectors(v);               // (first tile only: every later one is settled in front of its predecessor's write-out)

    for (; (int64_t)t < n_tiles; t += gridDim.x) {
        const uint32_t t_next = t + gridDim.x;
        const bool has_next = (int64_t)t_next < n_tiles;
        TileUniforms u_next = u;
        if (has_next) u_next = load_uniforms(u_desc, u_tile_base, u_cig_off, u_order, u_tile_first, t_next);

        const uint32_t r = u.r0 + (uint32_t)u.src;
        const bool active = u.src >= 0;
        const TileDesc d = u.d;
        const uint32_t base = u.base, tile_total = u.total;
        const bool fast = (d.flags & TD_FAST) != 0;
        const bool in_lds = tile_total <= (uint32_t)LDS_EXON_CAP;
        const uint32_t lay_total = in_lds ? tile_total : 0u;
        const uint32_t lay4 = (lay_total + 3u) & ~3u;                     // E starts on a 16-byte boundary as well (vector write-out)
        int *const s_S = reinterpret_cast<int *>(s_all), *const s_E = s_S + lay4;
        const int tail_off = (int)(2u * lay4), tail_words = ALL_WORDS - tail_off;
        uint32_t *const s_cig = s_all + tail_off;
        uint16_t *const s_W = reinterpret_cast<uint16_t *>(s_all + tail_off);
        v4i_t *const s_ent0 = reinterpret_cast<v4i_t *>(s_all + tail_off + W_WORDS), *const s_ent1 = s_ent0 + KEY_CAP;
        uint8_t *const s_dir0 = reinterpret_cast<uint8_t *>(s_all + tail_off + W_WORDS + 2 * KEY_CAP * 4);
        uint8_t *const s_dir1 = s_dir0 + DIR_BYTES, *const s_rdir = s_dir1 + DIR_BYTES;
        const uint32_t local = v.local, n = active ? v.nxt - v.local : 0u;
        const int32_t pos = v.pos, j0 = v.j0, tid = v.tid;
        const int w_n = fast ? (int)d.n_win : 0;                         // transcripts in the window

        // ---- phase 0: the tile's CIGAR words, registers -> LDS
        const bool staged = !WIDE && cigar_staged(u, tail_words);          // (long CIGARs are not read here at all: pass A leaves the exons)
        if (staged) {
            const int n4 = cigar_vectors(u);
            uint4 *dst = reinterpret_cast<uint4 *>(s_cig);
#pragma unroll
            for (int q = 0; q < PF_CIG_VEC; ++q) { const int i = q * TILE_THREADS + (int)threadIdx.x; if (i < n4) dst[i] = v.cg[q]; }
        }
        if (threadIdx.x == 0) { s_wide = 0; s_tilemask[0] = 0u; s_tilemask[1] = 0u; }
        __syncthreads();
        L2R_STAMP(0);

        // ---- phase 1: CIGAR -> exons
        ReadEnds re{0, 0, 0, 0};
        bool sane = true;
        const bool bulk = WIDE && in_lds && (d.flags & TD_WALKED) != 0u;      // (tile-uniform) pass A has left the exons
        if (bulk) {
            const int2 *const src = fast_args()->walked + (size_t)t * LDS_EXON_CAP;
            for (uint32_t i = threadIdx.x; i < tile_total; i += TILE_THREADS) { const int2 x = src[i]; s_S[i] = x.x; s_E[i] = x.y; }
            __syncthreads();
        }
        if (active) {
            const FastArgsK a = fast_args();
            DevParams p;                                    // the three thresholds the walk reads
            p.min_exon = a->p.min_exon; p.min_intron = a->p.min_intron; p.max_delet = a->p.max_delet;
            const int n_cig = (int)(v.c_hi - v.c_lo);
            if (in_lds) {
                // every exon non-empty <=> starts and ends strictly increasing and start <= end (an exon starts after
                // the previous one ends)
                auto emit = [&](int k, int s, int e) {
                    s_S[local + k] = s; s_E[local + k] = e;
                    sane = sane & (s <= e);
                    re.sl = s; re.el = e;
                };
                if (bulk && s_S[local] != WALK_SENTINEL) {
                    for (int k = 0; k < (int)n; ++k) sane = sane & (s_S[local + k] <= s_E[local + k]);
                    re.sl = s_S[local + n - 1u]; re.el = s_E[local + n - 1u];
                } else if (staged) walk_cigar<false>(s_cig + (v.c_lo - (u.c0 & ~3u)), n_cig, pos, p, emit);
                else walk_cigar<WIDE>(a->cig + v.c_lo, n_cig, pos, p, emit);
                re.s0 = s_S[local]; re.e0 = s_E[local];
            } else {
                int32_t *const xs = a->ex_start, *const xe = a->ex_end;
                walk_cigar<WIDE>(a->cig + v.c_lo, n_cig, pos, p, [&](int k, int s, int e) {
                    xs[base + local + k] = s; xe[base + local + k] = e;
                });
            }
        }
        __syncthreads();                 // every walk is done: the CIGAR region is free
        L2R_STAMP(7);
        // ---- stage the dictionary slices and the transcript window, re-based to the tile
        int my_wide = 0;
        if (fast) {
            if ((int)threadIdx.x >= KEY_CAP) {
                const int j = (int)threadIdx.x - KEY_CAP;
                if (j < w_n) {
                    // coordinates on another chromosome become -inf (before every read) / +inf (after every read)
                    int st = v.xa.y, en = v.xa.z;
                    if (v.xa.x < d.tid) { st = INT32_MIN; en = INT32_MIN; }
                    else if (v.xa.x > d.tid) { st = INT32_MAX; en = INT32_MAX; }
                    s_hk[j] = make_int4(st, en, v.xb.x, (v.xb.z & 0xff) | (v.xb.y << 8));
                    s_hx[j] = v.xc;
                    s_win[j] = v.xb.w;
                }
            }
            if (wv == TILE_THREADS / WAVE - 1) {                    // (wave-uniform) the header lanes are the upper half of the last wave
                const int j = (int)threadIdx.x - KEY_CAP;
                const bool mine = j >= 0 && j < w_n;
                const unsigned long long b1 = __ballot(mine && v.xb.x == 1), b2 = __ballot(mine && !((v.xb.z & 0xff) & TX_COMPACT));
                if (lane == WAVE / 2) { s_tilemask[0] = (uint32_t)(b1 >> 32); s_tilemask[1] = (uint32_t)(b2 >> 32); }
            }
            if (!(d.flags & TD_CONTIG)) __syncthreads();         // (tile-uniform) the entries below need the member list
            if ((int)threadIdx.x < KEY_CAP) {
                const bool has_st = threadIdx.x < d.st_nk, has_en = threadIdx.x < d.en_nk;
                v4i_t e0, e1;
                e0.x = v.xa.x; e0.y = v.xa.y; e1.x = v.xc.x; e1.y = v.xc.y;
                if (d.flags & TD_CONTIG) {           // consecutive window: a shift moves a mask into the tile frame
                    e0.z = (int)rebase_mask((uint32_t)v.xb.x, (uint32_t)v.xb.y, v.xa.z - d.j_lo);
                    e0.w = (int)rebase_mask((uint32_t)v.xb.z, (uint32_t)v.xb.w, v.xa.z - d.j_lo);
                    e1.z = (int)rebase_mask((uint32_t)v.xd.x, (uint32_t)v.xd.y, v.xc.z - d.j_lo);
                    e1.w = (int)rebase_mask((uint32_t)v.xd.z, (uint32_t)v.xd.w, v.xc.z - d.j_lo);
                } else {                             // window with gaps: member by member
                    e0.z = (int)rebase_gaps(s_win, w_n, (uint32_t)v.xb.x, (uint32_t)v.xb.y, v.xa.z);
                    e0.w = (int)rebase_gaps(s_win, w_n, (uint32_t)v.xb.z, (uint32_t)v.xb.w, v.xa.z);
                    e1.z = (int)rebase_gaps(s_win, w_n, (uint32_t)v.xd.x, (uint32_t)v.xd.y, v.xc.z);
                    e1.w = (int)rebase_gaps(s_win, w_n, (uint32_t)v.xd.z, (uint32_t)v.xd.w, v.xc.z);
                }
                if (has_st) { s_ent0[threadIdx.x] = e0; if (v.xa.w & SE_WIDE) my_wide = 1; }
                if (has_en) { s_ent1[threadIdx.x] = e1; if (v.xc.w & SE_WIDE) my_wide = 1; }
            }
            if (d.nbk > 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = (int)threadIdx.x + q * TILE_THREADS;
                    if (i <= d.nbk) {
                        s_dir0[i] = (uint8_t)(v.dd[0][q] - d.st_r0); s_dir1[i] = (uint8_t)(v.dd[1][q] - d.en_r0);
                        s_rdir[i] = (uint8_t)(v.dd[2][q] - d.st_r0);
                    }
                }
            }
            // two empty buckets behind the staged ones: where probes of coordinates outside the span land (map_exons)
            if (threadIdx.x < 3u && (threadIdx.x > 0u || d.nbk == 0)) {
                s_dir0[d.nbk + (int)threadIdx.x] = (uint8_t)d.st_nk; s_dir1[d.nbk + (int)threadIdx.x] = (uint8_t)d.en_nk;
            }
        }
        const bool rev_in = v.rev != 0;
        if (my_wide) s_wide = 1;
        __syncthreads();
        const int any_wide = s_wide;             // (cleared at the top of the next tile, two barriers from here)
        // ---- the next tile's vectors start their trip now; they are not needed before the top of the next round
        if (has_next) v = load_vectors(fast_args(), t_next, u_next, cigar_room(u_next), !WIDE);
        L2R_STAMP(1);

        // ---- phase 2: classification
        uint32_t info = n << 8; int ref = -1;
        bool redo = active && (!fast || !in_lds || any_wide != 0 || tid != d.tid || (n > 1 && !sane));
        const bool work = active && !redo;
        const TileLds L{s_S, s_E, s_W, s_ent0, s_ent1, s_dir0, s_dir1, s_rdir, s_hk, s_hx, s_win};
        const VisitMasks vm = visit_window<LEVEL>(L, d, w_n, work, n, j0, re, s_tilemask);
        redo = redo || vm.redo;
        L2R_STAMP(2);
        const int o_dis = fast_args()->p.ss_dis;
        const SiteMasks sm = o_dis > 0 ? map_exons_near(L, d, work && !redo && n > 1, local, n, vm.vpre, o_dis, re.s0, re.el)
                                       : map_exons(L, d, work && !redo && n > 1, local, n, vm.vpre);
        // (-d > 0: a visited member with two sites within the tolerance of one read site -- its pair count is the generic kernel's)
        if (work && !redo && (sm.amb & vm.vpre) != 0u) redo = true;
        L2R_STAMP(3);
        if (work && !redo) {
            uint16_t *const Wr = s_W + local;
            const Verdict vd = decide<LEVEL>(L, d, n, re, vm, sm, rev_in, [&](int k) { return (uint32_t)Wr[k]; }, [&](int k, uint32_t f) { Wr[k] = (uint16_t)f; });
            info = vd.info; ref = vd.ref;
        } else if (active && in_lds) {
            for (int k = 0; k < (int)n; ++k) s_W[local + k] = (uint16_t)0;
        }
        L2R_STAMP(4);

        // ---- phase 3: redo list, accepted counts, coalesced write-out of the tile
        redo = redo && active;
        if (stamping && redo) {          // diagnostics: why reads leave the fast path
            const int why = !fast ? 0 : !in_lds ? 1 : any_wide ? 2 : tid != d.tid ? 3 : (n > 1 && !sane) ? 4 : 5;
            atomicAdd(&fast_args()->stamps[8192 + why], 1ull);
        }
        const FastArgsK ao = fast_args();
        // the output pointers of the rest of the tile, fetched from the argument block in one go (see load_vectors)
        uint32_t *const o_redo_count = ao->redo_count, *const o_redo = ao->redo, *const o_tile_acc = ao->tile_acc, *const o_tile_acc_ex = ao->tile_acc_ex;
        uint32_t *const o_tile_chunk = ao->tile_chunk, *const o_tile_rchunk = ao->tile_rchunk, *const o_ex_off = ao->ex_off, *const o_info = ao->info, *const o_acc_ex_off = ao->acc_ex_off;
        unsigned long long *const o_cursor = ao->chunk_cursor; AccRec *const o_acc_rec = ao->acc_rec; const int64_t o_first_read = ao->first_read;
        int32_t *const o_ex_start = ao->ex_start, *const o_ex_end = ao->ex_end, *const o_ref = ao->ref_tx, *const o_acc_start = ao->acc_start, *const o_acc_end = ao->acc_end;
        uint8_t *const o_ex_flag = ao->ex_flag, *const o_acc_flag = ao->acc_flag;
        const int32_t o_n_sj = ao->p.n_sj, o_ablate = ao->p.ablate;
        const bool want_acc = (ao->p.want & WANT_ACCEPTED) != 0;
        asm volatile("" :: "s"(o_redo_count), "s"(o_redo), "s"(o_tile_acc), "s"(o_tile_acc_ex), "s"(o_tile_chunk), "s"(o_cursor), "s"(o_ex_off), "s"(o_info),
                     "s"(o_ex_start), "s"(o_ex_end), "s"(o_ref), "s"(o_acc_start), "s"(o_acc_end), "s"(o_ex_flag), "s"(o_acc_flag), "s"(o_n_sj), "s"(o_ablate),
                     "s"(o_tile_rchunk), "s"(o_acc_ex_off), "s"(o_acc_rec), "s"(o_first_read));
        {
            const unsigned long long m = __ballot(redo);
            if (m) {
                uint32_t at = 0;
                if (lane == 0) at = atomicAdd(o_redo_count, (uint32_t)__popcll(m));
                at = __shfl(at, 0, WAVE);
                if (redo) o_redo[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)r;
            }
            if (want_acc) {                  // (uniform) the accepted list is compacted only when somebody asked for it
                const bool acc = (info & I_ACCEPT) != 0;
                const uint32_t ca = (uint32_t)__popcll(__ballot(acc)), cx = wave_sum(acc ? n : 0u);
                if (lane == 0) { s_cnt[wv][0] = ca; s_cnt[wv][1] = cx; s_cnt[wv][2] = (uint32_t)__popcll(m); }
                if (active) s_nat[u.src] = (uint16_t)(min(n, 0x7fffu) | (acc ? 0x8000u : 0u));
            }
        }
        __syncthreads();
        L2R_STAMP(5);
        settle_vectors(v);                           // (unconditional: the compiler cannot tell that `has_next` guards both)
        asm volatile("" :: "v"(u_next.src));
        // ---- accepted exons of the tile, compacted in read order into a chunk of the accepted arrays (chunks are handed
        // out by an atomic cursor: their order is arbitrary, tile_chunk says where a tile's chunk starts).  Only when
        // every verdict of the tile is final here: no read on the redo list, no junction table (k_validate_sj decides).
        const uint32_t ca_t = s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0], cx_t = s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1];
        const bool fused = want_acc && in_lds && (s_cnt[0][2] + s_cnt[1][2] + s_cnt[2][2] + s_cnt[3][2]) == 0u && o_n_sj == 0 && !(o_ablate & 2);
        unsigned long long chunk = 0ull;               // {first record slot, first exon slot} of the tile's chunk
        if (want_acc && threadIdx.x == 0) {
            o_tile_acc[t] = fused ? 0u : ca_t;          // what k_gather_accepted has to place: reads ...
            o_tile_acc_ex[t] = fused ? 0u : cx_t;       // ... and exons
            if (!fused) o_tile_chunk[t] = CHUNK_DEFERRED;
            else if (ca_t) chunk = atomicAdd(o_cursor, ((unsigned long long)ca_t << 32) | cx_t);      // (answer needed after the write-out below)
        }
        uint16_t *const s_map = reinterpret_cast<uint16_t *>(s_ent0);           // the dictionary slices are dead by now
        uint32_t *const s_rk = reinterpret_cast<uint32_t *>(s_map + LDS_EXON_CAP);   // per read of the tile: rank among the accepted | exon offset << 16
        if (fused && ca_t) {
            // thread i takes read i of the tile (reads are spread over the threads in pass A's order): exclusive sums
            // of {all exons, accepted exons} over the reads before it, both below 2^16, packed in one word; accepted reads before it
            const uint32_t n_act = u.n_act;
            const uint32_t w16 = threadIdx.x < n_act ? (uint32_t)s_nat[threadIdx.x] : 0u;
            const uint32_t nn = w16 & 0x7fffu, pk = nn | ((w16 >> 15) ? nn << 16 : 0u);
            const unsigned long long am = __ballot((w16 >> 15) != 0u);
            uint32_t before = wave_inclusive_scan(pk) - pk;
            uint32_t rank = (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
            for (int k = 0; k < wv; ++k) {
                const uint32_t idx = (uint32_t)(k * WAVE + lane);
                const uint32_t z = idx < n_act ? (uint32_t)s_nat[idx] : 0u, zn = z & 0x7fffu;
                before += wave_sum(zn | ((z >> 15) ? zn << 16 : 0u));
                rank += (uint32_t)__popcll(__ballot((z >> 15) != 0u));
            }
            if (w16 >> 15) {
                const uint32_t from = before & 0xffffu, to = before >> 16;
                s_rk[threadIdx.x] = rank | (to << 16);
                for (uint32_t k = 0; k < nn; ++k) s_map[to + k] = (uint16_t)(from + k);
            }
        }
        if (in_lds) {
            for (uint32_t i = threadIdx.x; i < tile_total; i += TILE_THREADS) {
                o_ex_start[base + i] = s_S[i];
                o_ex_end[base + i] = s_E[i];
                o_ex_flag[base + i] = (uint8_t)s_W[i];
            }
        }
        if (active) {
            o_ex_off[r] = base + local;
            o_info[r] = info;
            o_ref[r] = ref;
        }
        if (fused) {
            if (threadIdx.x == 0) {
                s_chunk[0] = (uint32_t)chunk; s_chunk[1] = (uint32_t)(chunk >> 32);
                o_tile_chunk[t] = (uint32_t)chunk; o_tile_rchunk[t] = (uint32_t)(chunk >> 32);
            }
            if (ca_t) {
                __syncthreads();
                const uint32_t to = s_chunk[0], to_r = s_chunk[1];
                if (info & I_ACCEPT) {          // the record of the thread's own read
                    const uint32_t rk = s_rk[u.src];
                    const uint32_t slot = to_r + (rk & 0xffffu);
                    const uint64_t gidx = (uint64_t)(o_first_read + (int64_t)r);
                    AccRec a; a.read_lo = (uint32_t)gidx; a.read_hi = (uint32_t)(gidx >> 32); a.info = info; a.ref_tx = ref;
                    o_acc_rec[slot] = a;
                    o_acc_ex_off[slot] = to + (rk >> 16);
                }
                uint32_t q = threadIdx.x < cx_t ? (uint32_t)s_map[threadIdx.x] : 0u;
                for (uint32_t i = threadIdx.x; i < cx_t; i += TILE_THREADS) {
                    const uint32_t i_next = i + TILE_THREADS;                    // its map entry travels while this one is copied
                    const uint32_t q_next = i_next < cx_t ? (uint32_t)s_map[i_next] : 0u;
                    o_acc_start[to + i] = s_S[q];
                    o_acc_end[to + i] = s_E[q];
                    o_acc_flag[to + i] = (uint8_t)s_W[q];
                    q = q_next;
                }
            }
        }
        __syncthreads();                 // the tile's LDS image has been written out: the next tile may overwrite it
        L2R_STAMP(6);
        u = u_next;
    }
#undef L2R_STAMP
}

// ------------------------------------------------------------------ buffer sizing at upload
// Number of CIGAR words that can end an exon whatever the thresholds are (N or D): reads + this bounds the exons of a shard.
__global__ __launch_bounds__(TILE_THREADS)
void k_count_cut_ops(const uint32_t *__restrict__ cig, int64_t n_words, unsigned long long *__restrict__ total)
{
    unsigned long long mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t op = cig[i] & 15u;
        mine += (op == 2u || op == 3u) ? 1u : 0u;
    }
    const uint32_t w = wave_sum((uint32_t)mine);          // (a thread sees < 2^32 words)
    if ((threadIdx.x & (WAVE - 1)) == 0 && w) atomicAdd(total, (unsigned long long)w);
}

// ------------------------------------------------------------------ short-read junction support

__device__ __forceinline__ int first_key_above(const int64_t *__restrict__ key, int n, int64_t q)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key[mid] > q) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Directories over the junction table (built by l2r_set_junctions): `cur` = the cursor directory over the prefix-max keys of (tid, acc)
// (the same structure the annotation cursor uses: cursor_value), and a directory over 512-bp buckets of the donors:
// ddir[dbase[tid] + (x >> 9)] = first row >= (tid, x << 9).  A lookup is one directory read and a few rows instead of a binary search
// over the whole table (22 dependent loads for a STAR table of 4 M rows: k_validate_sj took 1.7 ms for the 7.4 M candidates of config 3).
struct SjDir { CursorDir cur; const uint32_t *ddir; const int32_t *dbase; int32_t d_ntid; const int4 *row; };      // row[i] = {don, acc, uniq, multi}: one 16-byte load per row
// where a chromosome's rows are: first bucket, bucket count, first row behind them (rows are sorted by tid: every row from the
// chromosome's first one up to `end` is its own)
struct SjTid { int tid, db, nb, end; };
__device__ __forceinline__ SjTid sj_tid_rows(const SjDir &sd, int tid, int n_sj)
{
    if (tid >= sd.d_ntid) return SjTid{tid, 0, 0, n_sj};
    const int32_t db = sd.dbase[tid], nb = sd.dbase[tid + 1] - db;
    return SjTid{tid, db, nb, (int)sd.ddir[db + nb]};
}

// first row >= (tid, want) of the table (rows sorted by (tid, don, acc))
__device__ __forceinline__ int sj_first_row(const SjDir &sd, const SjTid &ti, int want)
{
    if (ti.nb <= 0) return ti.end;
    const int b = max(want, 0) >> SITE_SHIFT;
    if (b >= ti.nb) return ti.end;                             // behind the chromosome's last donor: the next chromosome's first row
    int lo = (int)sd.ddir[ti.db + b], hi = min((int)sd.ddir[ti.db + b + 1], ti.end);
    if (hi - lo > 16) {                                        // a crowded bucket: lower bound inside it
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sd.row[mid].x < want) lo = mid + 1; else hi = mid;
        }
        return lo;
    }
    // (a bucket's rows are the chromosome's: only the donor is compared; the first two rows travel together)
    if (lo + 1 < hi) { const int d0 = sd.row[lo].x, d1 = sd.row[lo + 1].x; if (d0 >= want) return lo; if (d1 >= want) return lo + 1; lo += 2; }
    while (lo < hi && sd.row[lo].x < want) ++lo;
    return lo;
}

// src/update_gtf.c:589-603 check_short_sj1 with the linear scan from the cursor row replaced by a lower-bound on
// (tid, don): rows below don-dis cannot match, and the reference stops at the first row with don >= acc (intron end).
__device__ __forceinline__ bool junction_supported(const SjTid &ti, int don, int acc, int from, const DevParams &p, const SjDir &sd)
{
    int lo = from;
    const int want = don - p.ss_dis;
    // (a degenerate intron with acc < don, or a negative -d, keeps the literal linear scan:
    //  only then could a skipped row have triggered the reference's early "don >= acc" stop)
    if (!(acc < don || p.ss_dis < 0)) lo = max(from, sj_first_row(sd, ti, want));      // first row >= (tid, want) at or after `from`
    // (rows from `from` on are on the read's chromosome or behind it: the first row of a later chromosome ends the scan, :594)
    for (int i = lo; i < ti.end; ++i) {
        const int4 q = sd.row[i];
        const int d = q.x;
        if (d >= acc) return false;
        if (p.ss_dis >= 0 && d - don > p.ss_dis) return false;   // sorted by don: nothing further can match
        if (near_eq(d, don, p.ss_dis) && near_eq(q.y, acc, p.ss_dis)) {
            const int c = p.use_multi ? q.z + q.w : q.z;
            if (c >= p.min_sj_cnt) return true;
        }
    }
    return false;
}

// src/update_gtf.c:698-709 check_with_short_sj + :609-627 check_short_sj for every read that reaches it (full, not known, has a
// known site).  One workgroup per 256 consecutive reads, in three steps:
//   1. thread i = read i: is it a candidate, its cursor row (:613-614) and the Q7 test on it; every exon position of the block learns
//      its read (an LDS map over the block's exon run: the results are in read order, so the run is contiguous);
//   2. the block's exon POSITIONS across the threads: flag bytes, exon ends and next starts are read coalesced, and the table lookups
//      of the novel junctions of candidates are spread over all lanes -- one thread per read left most lanes idle behind the reads
//      with the most novel junctions, each lookup a chain of dependent loads (0.86 ms for the 7.4 M candidates of config 3);
//   3. thread i again: the read's verdict from what its junctions left in LDS.
// A read whose exons do not lie in the mapped part of the run (a block of very long reads) checks its junctions itself, as before.
constexpr int SJ_MAP_CAP = 6144;              // exon positions of a block that are mapped to their reads
__global__ __launch_bounds__(TILE_THREADS)
void k_validate_sj(int64_t n_reads, const int32_t *__restrict__ r_tid, const uint32_t *__restrict__ ex_off,
                   const int32_t *__restrict__ ex_start, const int32_t *__restrict__ ex_end, uint8_t *__restrict__ ex_flag,
                   const int64_t *__restrict__ sj_key, const int32_t *__restrict__ sj_cursor,
                   const int32_t *__restrict__ sj_tid, const int32_t *__restrict__ sj_don, const int32_t *__restrict__ sj_acc,
                   const int32_t *__restrict__ sj_uniq, const int32_t *__restrict__ sj_multi, DevParams p,
                   uint32_t *__restrict__ info_io, SjDir sd)
{
    __shared__ uint8_t s_owner[SJ_MAP_CAP];
    __shared__ int s_from[TILE_THREADS];               // the read's cursor row, -1: its junctions are not looked up here
    __shared__ int s_tid[TILE_THREADS];
    __shared__ uint32_t s_bad[TILE_THREADS];
    __shared__ uint32_t s_base, s_end, s_owned;
    const int64_t r = (int64_t)blockIdx.x * TILE_THREADS + threadIdx.x;
    const bool have = r < n_reads;
    uint32_t info = have ? info_io[r] : 0u;
    // (a read k_tile has checked on its tile's LDS image carries I_SJCHK already: l2r_tile.hip.h)
    const bool cand = have && (info & (I_FULL | I_KNOWN | I_KSITE | I_SJCHK)) == (I_FULL | I_KSITE);
    const int n = (int)(info >> 8), tid = have ? r_tid[r] : 0;
    const uint32_t off = have ? ex_off[r] : 0u;
    // the rows of the block's chromosome (its first read's: wave-uniform loads), a read of another one looks its own up
    const SjTid ti0 = sj_tid_rows(sd, r_tid[(int64_t)blockIdx.x * TILE_THREADS], p.n_sj);
    if (threadIdx.x == 0) { s_base = off; s_owned = 0u; }
    // (the block's last read closes the run)
    if (have && (threadIdx.x == TILE_THREADS - 1 || r == n_reads - 1)) s_end = off + (uint32_t)n;
    int from = -1;
    bool ok0 = false;
    if (cand) {
        const int r_start = ex_start[off], r_end = ex_end[off + (uint32_t)(n - 1)];
        from = sj_cursor ? sj_cursor[r] : cursor_value(sd.cur, tid, r_start);      // (first row whose prefix-max key is above (tid, start): update_gtf.c:613-614)
        if (from < p.n_sj) {
            // Q7: cursor row beyond the read -> unsupported, no unreliable flag  (rows from the cursor on are on the read's chromosome or behind it)
            const SjTid ti = tid == ti0.tid ? ti0 : sj_tid_rows(sd, tid, p.n_sj);
            ok0 = !(from >= ti.end || sd.row[from].x >= r_end);
        }
    }
    s_bad[threadIdx.x] = 0u;
    s_tid[threadIdx.x] = tid;
    __syncthreads();
    const uint32_t base = s_base, total = min(s_end - base, (uint32_t)SJ_MAP_CAP);
    // is the read's run inside the mapped positions?  (read order: it is, unless the block has more than SJ_MAP_CAP exons)
    const bool mapped = have && off >= base && off - base + (uint32_t)n <= total;
    s_from[threadIdx.x] = (cand && ok0 && mapped) ? from : -1;
    if (mapped) for (int k = 0; k < n; ++k) s_owner[off - base + (uint32_t)k] = (uint8_t)threadIdx.x;
    // The mapped reads are a prefix of the block's reads (read order: behind the first one that does not fit, none does), so the positions
    // up to the end of the last mapped read all have an owner; the ones behind it -- of the read that straddles position SJ_MAP_CAP, in
    // a block with more exons than that -- hold whatever an earlier workgroup left there and are not looked at here (their reads take
    // the per-read loop below).
    if (mapped && n > 0) atomicMax(&s_owned, off - base + (uint32_t)n);
    __syncthreads();
    const uint32_t owned = s_owned;
    for (uint32_t q = threadIdx.x; q < owned; q += (uint32_t)TILE_THREADS) {
        const uint8_t f = ex_flag[base + q];
        if (!(f & F_JUNC)) continue;
        const uint32_t who = s_owner[q];
        const int fr = s_from[who];
        if (fr < 0) continue;
        // (a junction flag only stands on an exon that is not its read's last: position q + 1 is the same read's)
        const int wt = s_tid[who];
        const SjTid ti = wt == ti0.tid ? ti0 : sj_tid_rows(sd, wt, p.n_sj);
        if (!junction_supported(ti, ex_end[base + q] + 1, ex_start[base + q + 1u] - 1, fr, p, sd)) {
            ex_flag[base + q] = f | F_UNREL;
            s_bad[who] = 1u;
        }
    }
    __syncthreads();
    if (!cand) return;
    bool ok = ok0;
    if (ok0 && !mapped) {
        const SjTid ti = tid == ti0.tid ? ti0 : sj_tid_rows(sd, tid, p.n_sj);
        for (int j = 0; j + 1 < n; ++j) {
            const uint8_t f = ex_flag[off + (uint32_t)j];
            if ((f & F_JUNC) &&
                !junction_supported(ti, ex_end[off + (uint32_t)j] + 1, ex_start[off + (uint32_t)(j + 1)] - 1, from, p, sd)) {
                ex_flag[off + (uint32_t)j] = f | F_UNREL;
                ok = false;
            }
        }
    } else if (ok0) ok = s_bad[threadIdx.x] == 0u;
    info |= I_SJCHK;
    if (ok) info |= I_SJPASS; else info |= I_UNREL;
    if (ok || p.split_trans) info |= I_ACCEPT;
    info_io[r] = info;
}

// ------------------------------------------------------------------ compaction of accepted reads
// Tiles are the classification tiles (reads_per_tile records, one workgroup).

// (one WAVE per tile, four tiles per workgroup: 39 k workgroups that find nothing to do cost 13 us of dispatch alone)
__global__ __launch_bounds__(TILE_THREADS)
void k_count_accepted(const uint32_t *__restrict__ tile_first, const uint32_t *__restrict__ info, const uint32_t *__restrict__ tile_chunk,
                      uint32_t *__restrict__ tile_reads, uint32_t *__restrict__ tile_exons, uint32_t n_tiles)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * (uint32_t)(TILE_THREADS / WAVE) + (uint32_t)wv;
    if (t >= n_tiles) return;
    // (a tile whose classification kernel has written its chunk itself has left zeros here: nothing of it is k_gather_accepted's)
    if (tile_chunk[t] != CHUNK_DEFERRED) return;          // (wave-uniform)
    const uint32_t r0 = tile_first[t], n_act = tile_first[t + 1] - r0;
    uint32_t cnt = 0u, ex = 0u;
    for (uint32_t i = (uint32_t)lane; i < n_act; i += (uint32_t)WAVE) {
        const uint32_t w = info[r0 + i];
        if (w & I_ACCEPT) { ++cnt; ex += w >> 8; }
    }
    cnt = wave_sum(cnt); ex = wave_sum(ex);
    if (lane == 0) { tile_reads[t] = cnt; tile_exons[t] = ex; }
}


// The accepted list: records {read index, info, ref_tx} + the offset of the read's exons, and the exon arrays themselves.
// Both are made of one chunk per tile -- the reads of a chunk in read order, the chunks in the order they were handed
// out -- so a consumer that needs read order sorts the chunks by their first record (l2r_download_accepted does).
// Most chunks are written by k_classify_fast from its LDS image (tile_rchunk / tile_chunk = first record / exon slot,
// taken from an atomic cursor).  A tile it marked CHUNK_DEFERRED (a read on the redo list, or a junction table:
// acceptance is decided after it) gets its chunk here, behind the cursor's final value, in tile order among the
// deferred ones (tile_reads / tile_exons = exclusive scans of their counts): every accepted read writes, for each of
// its exons, the source position into an LDS map at the exon's compacted slot; the tile then copies slot by slot, so
// the stores are contiguous and the loads run over contiguous pieces.
constexpr uint32_t MAP_DIRECT = 0xffffu;      // map entry of an exon that its read has copied itself

constexpr int GATHER_TILES = 4;               // tiles per workgroup, in turn (most are not deferred: 39 k workgroups that leave at once cost 13 us of dispatch)
__global__ __launch_bounds__(TILE_THREADS)
void k_gather_accepted(const uint32_t *__restrict__ tile_first, int64_t first_read, const uint32_t *__restrict__ info, const int32_t *__restrict__ ref_tx,
                       const uint32_t *__restrict__ ex_off, const int32_t *__restrict__ ex_start, const int32_t *__restrict__ ex_end,
                       const uint8_t *__restrict__ ex_flag, const uint32_t *__restrict__ tile_reads, const uint32_t *__restrict__ tile_exons,
                       uint32_t *__restrict__ tile_chunk, uint32_t *__restrict__ tile_rchunk, const uint32_t *__restrict__ chunk_cursor /* {exons, records} */,
                       AccRec *__restrict__ rec, uint32_t *__restrict__ acc_ex_off, int32_t *__restrict__ acc_start,
                       int32_t *__restrict__ acc_end, uint8_t *__restrict__ acc_flag, uint32_t n_tiles)
{
    __shared__ uint32_t s_wcnt[4], s_wex[4];
    __shared__ uint16_t s_map[LDS_EXON_CAP];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    for (uint32_t t = blockIdx.x * (uint32_t)GATHER_TILES; t < min((blockIdx.x + 1u) * (uint32_t)GATHER_TILES, n_tiles); ++t) {
        if (tile_chunk[t] != CHUNK_DEFERRED) continue;          // (workgroup-uniform)
        const uint32_t r0 = tile_first[t], n_act = tile_first[t + 1] - r0;
        const int64_t r = (int64_t)r0 + threadIdx.x;
        const bool active = threadIdx.x < n_act;
        const uint32_t w = active ? info[r] : 0u;
        const bool acc = (w & I_ACCEPT) != 0;
        const unsigned long long m = __ballot(acc);
        const uint32_t rank_w = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t nex = acc ? (w >> 8) : 0u;
        const uint32_t inc = wave_inclusive_scan(nex);
        if (lane == WAVE - 1) { s_wcnt[wv] = (uint32_t)__popcll(m); s_wex[wv] = inc; }
        const uint32_t ebase0 = chunk_cursor[0] + tile_exons[t], cbase0 = chunk_cursor[1] + tile_reads[t];
        __syncthreads();
        const uint32_t e_tot = s_wex[0] + s_wex[1] + s_wex[2] + s_wex[3];        // accepted exons of the tile
        if (threadIdx.x == 0) { tile_chunk[t] = ebase0; tile_rchunk[t] = cbase0; }
        uint32_t cb = 0, eb = 0;
        for (int k = 0; k < wv; ++k) { cb += s_wcnt[k]; eb += s_wex[k]; }
        const uint32_t src0 = n_act ? ex_off[r0] : 0u;                            // first exon of the tile
        const uint32_t e_loc = eb + inc - nex;                                    // tile-local compacted exon offset
        const bool mapped = e_tot <= (uint32_t)LDS_EXON_CAP;
        if (acc) {
            const uint32_t src = ex_off[r];
            const uint32_t slot = cbase0 + cb + rank_w;
            const uint64_t gidx = (uint64_t)(first_read + r);
            AccRec a; a.read_lo = (uint32_t)gidx; a.read_hi = (uint32_t)(gidx >> 32); a.info = w; a.ref_tx = ref_tx[r];
            rec[slot] = a;
            acc_ex_off[slot] = ebase0 + e_loc;
            if (mapped && src >= src0 && (uint64_t)(src - src0) + (uint64_t)nex < MAP_DIRECT) {
                for (uint32_t k = 0; k < nex; ++k) s_map[e_loc + k] = (uint16_t)(src - src0 + k);
            } else {
                for (uint32_t k = 0; k < nex; ++k) {
                    if (mapped) s_map[e_loc + k] = (uint16_t)MAP_DIRECT;
                    acc_start[ebase0 + e_loc + k] = ex_start[src + k];
                    acc_end[ebase0 + e_loc + k] = ex_end[src + k];
                    acc_flag[ebase0 + e_loc + k] = ex_flag[src + k];
                }
            }
        }
        __syncthreads();
        if (mapped) {
            for (uint32_t i = threadIdx.x; i < e_tot; i += TILE_THREADS) {
                const uint32_t q = s_map[i];
                if (q == MAP_DIRECT) continue;
                const uint32_t sidx = src0 + q;
                acc_start[ebase0 + i] = ex_start[sidx];
                acc_end[ebase0 + i] = ex_end[sidx];
                acc_flag[ebase0 + i] = ex_flag[sidx];
            }
        }
        __syncthreads();                 // (the next tile of this workgroup overwrites the map and the wave counts)
    }
}

}  // namespace l2r
