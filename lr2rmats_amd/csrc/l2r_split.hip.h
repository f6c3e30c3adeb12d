// l2r_split.hip.h -- the one-walk pipeline cut in two so that the expensive half runs at FULL occupancy (gfx950).
//
// Measured on MI355X (profiles/r02/ubench_valu_issue.txt): one wave issues a VALU instruction every ~7.5 cycles
// whatever its ILP, and a SIMD's issue rate keeps growing up to 8 resident waves (one instruction per 2.5 cycles at 4
// waves, per 1.4 cycles at 8).  k_classify_fast / k_fused keep a tile's exons (10 bytes each) in LDS for the whole
// tile: 40 KB per workgroup = 4 waves per SIMD, and they are issue bound.  Here the exons live in HBM between the
// two halves:
//
//   k_order   (l2r_fused.hip.h)  slot ranges and lane order of a tile from the CIGAR lengths
//   k_walk    per tile: CIGAR words in registers, ONE walk, exons through LDS (coalesced) into the tile's chunk of the
//             result arrays (atomic cursor), ex_off / exon count per read; one wave turns the tile's span into the
//             descriptor and transcript window, which go to HBM (1.2 KB per tile)
//   k_probe   per tile: window and dictionary slices into LDS (13 KB -> 8 workgroups per CU, <= 64 VGPRs); every lane
//             streams its read's exons back (they are L2 / Infinity-Cache warm), window pass, probes, verdicts with the
//             device functions of the classic kernel; flag bytes leave coalesced through LDS
//
// HBM traffic: CIGAR once, exons written once and read once (64 + 64 bytes per read against 60 for a second CIGAR walk).
#pragma once
#include "l2r_fused.hip.h"

namespace l2r {

constexpr int SPLIT_EXON_CAP = 2880;                     // k_walk: exon slots of a tile in LDS; k_probe: flag words of a tile in LDS
constexpr int SPLIT_TW_VECS = (int)(sizeof(TileWin) / 16);
static_assert(sizeof(TileWin) % 16 == 0, "TileWin is copied in 16-byte pieces");
constexpr uint32_t I_PRE_INSANE = I_UNREL;               // k_walk -> k_probe, in info[]: the read's exons are not strictly increasing
constexpr uint32_t I_PRE_DIRECT = I_SJCHK;               // ... the read could not be kept in LDS (k_probe sends it to the generic kernel)

struct SplitArgs {
    FusedArgs g;
    TileWin *tw;                                         // per tile: descriptor + window (k_walk -> k_probe)
};
typedef const __attribute__((address_space(4))) SplitArgs *SplitArgsK;
__device__ __forceinline__ SplitArgsK split_args()
{
    SplitArgsK q = (SplitArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}

__global__ __launch_bounds__(TILE_THREADS, 5)
void k_walk(SplitArgs kernarg_block, const uint32_t *__restrict__ u_tile_first, const uint8_t *__restrict__ u_order,
            const int32_t *__restrict__ u_tid, const int32_t *__restrict__ u_pos, const uint32_t *__restrict__ u_tile_ub)
{
    __shared__ __attribute__((aligned(16))) int s_S[SPLIT_EXON_CAP];
    __shared__ __attribute__((aligned(16))) int s_E[SPLIT_EXON_CAP];
    __shared__ __attribute__((aligned(16))) uint16_t s_map[SPLIT_EXON_CAP];     // output slot -> LDS slot
    __shared__ __attribute__((aligned(16))) TileWin s_tw;
    __shared__ __attribute__((aligned(16))) uint32_t s_nx[TILE_THREADS];        // per read, READ order: exon count, then exact exon offset inside the tile
    __shared__ int s_wmax[4];
    __shared__ uint32_t s_base[2];
    (void)kernarg_block;
    const SplitArgsK sa = split_args();
    const FusedArgsK a = fused_args();
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x;
    FusedTile T;
    T.r0 = u_tile_first[t]; T.n_act = u_tile_first[t + 1u] - T.r0;
    T.tid0 = T.n_act ? u_tid[T.r0] : 0; T.pos0 = T.n_act ? u_pos[T.r0] : 0;
    T.in_lds = u_tile_ub[t] <= (uint32_t)SPLIT_EXON_CAP;
    const int32_t src = threadIdx.x < T.n_act ? (int32_t)ld32(u_order, T.r0 + threadIdx.x) : -1;
    const FusedRead v = fused_load_read(a, T, src);
    const bool active = src >= 0, in_lds = T.in_lds;
    const uint32_t r = T.r0 + (uint32_t)max(src, 0);
    // ---- the ONE walk, CIGAR words out of registers; the wave stops where its longest CIGAR ends
    DevParams p;
    p.min_exon = a->f.p.min_exon; p.min_intron = a->f.p.min_intron; p.max_delet = a->f.p.max_delet;
    const uint32_t room = exon_bound(v.n_cig, p.min_exon);
    const int c_max = wave_max(active ? (int)min(v.n_cig, (uint32_t)FUSED_HEAD) : 0);
    uint32_t n = 0u;
    bool sane = true, over = false;
    int el = INT32_MIN;
    if (active) {
        WalkState w{v.pos + 1, v.pos, 0};
        auto emit = [&](int k, int s, int e) {
            if (in_lds && (uint32_t)k < room) { s_S[v.lub + (uint32_t)k] = s; s_E[v.lub + (uint32_t)k] = e; }
            else over = true;
            sane = sane & (s <= e);
            el = e;
        };
#pragma unroll
        for (int q = 0; q < FUSED_HEAD_VEC; ++q) {
            if (4 * q < c_max) {         // (wave-uniform)
                walk_step(w, v.cg[4 * q], p, emit); walk_step(w, v.cg[4 * q + 1], p, emit);
                walk_step(w, v.cg[4 * q + 2], p, emit); walk_step(w, v.cg[4 * q + 3], p, emit);
            }
        }
        if (v.n_cig > (uint32_t)FUSED_HEAD) walk_ops<false>(w, a->f.cig + v.c_lo, FUSED_HEAD, (int)v.n_cig, p, emit);
        emit(w.n, w.start, w.end);
        n = (uint32_t)w.n + 1u;
        s_nx[src] = n;
    }
    {
        const int m = wave_max((active && v.tid == T.tid0) ? el : INT32_MIN);
        if (lane == 0) s_wmax[wv] = m;
    }
    __syncthreads();
    // ---- wave 0: exact exon offsets in read order and the tile's chunk; last wave: the tile's descriptor and window
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
    } else if (wv == TILE_THREADS / WAVE - 1) {
        make_descriptor(a, lane, T.tid0, T.pos0 + 1, max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])), in_lds, &s_tw);
    }
    __syncthreads();
    const uint32_t base = s_base[0], tile_total = s_base[1];
    if (active) {
        const uint32_t loc = s_nx[src];
        a->f.ex_off[r] = base + loc;
        a->f.info[r] = (n << 8) | (sane ? 0u : I_PRE_INSANE) | ((!in_lds || over) ? I_PRE_DIRECT : 0u);
        if (in_lds && !over) for (uint32_t k = 0; k < n; ++k) s_map[loc + k] = (uint16_t)(v.lub + k);
        else {
            // (rare) a tile beyond the LDS capacity, or a read beyond its bound: walked again, lane by lane to HBM
            int32_t *const xs = a->f.ex_start, *const xe = a->f.ex_end;
            WalkState w{v.pos + 1, v.pos, 0};
            auto put = [&](int k, int s, int e) { xs[base + loc + (uint32_t)k] = s; xe[base + loc + (uint32_t)k] = e; };
            walk_ops<false>(w, a->f.cig + v.c_lo, 0, (int)v.n_cig, p, put);
            put(w.n, w.start, w.end);
            if (in_lds) for (uint32_t k = 0; k < n; ++k) s_map[loc + k] = (uint16_t)0xffffu;         // (skipped by the copy below)
        }
    }
    if ((int)threadIdx.x < SPLIT_TW_VECS)
        reinterpret_cast<int4 *>(sa->tw + t)[threadIdx.x] = reinterpret_cast<const int4 *>(&s_tw)[threadIdx.x];
    __syncthreads();
    if (in_lds) {
        int32_t *const xs = a->f.ex_start, *const xe = a->f.ex_end;
        for (uint32_t i = threadIdx.x; i < tile_total; i += TILE_THREADS) {
            const uint32_t q = s_map[i];
            if (q == 0xffffu) continue;
            xs[base + i] = s_S[q];
            xe[base + i] = s_E[q];
        }
    }
}

// One START and one END probe per exon as map_exons (l2r_kernels.hip.h), the read's exons streamed from HBM (the lane's
// own run of the result arrays, three exons in flight) instead of LDS.
__device__ __forceinline__ SiteMasks map_exons_stream(const TileLds &L, const TileDesc &d, bool mapping, const int32_t *__restrict__ xs,
                                                      const int32_t *__restrict__ xe, uint32_t off, uint32_t loc, uint32_t n, uint32_t vpre)
{
    SiteMasks m{0xffffffffu, 0u, 0u, 0u};
    uint16_t *W = L.W + loc;
    const uint32_t last = mapping ? n - 1u : 0u;
    int s = 0, e = 0, s1 = 0, e1 = 0, s2n = 0, e2n = 0;
    if (mapping) {
        s = xs[off]; e = xe[off];
        const uint32_t i1 = off + min(1u, last), i2 = off + min(2u, last);
        s1 = xs[i1]; e1 = xe[i1]; s2n = xs[i2]; e2n = xe[i2];
    }
    const uint32_t none = (uint32_t)d.nbk + 1u;         // a bucket behind the staged ones: the staging leaves it empty
    const int k_max = wave_max(mapping ? (int)n : 0);
    for (int k = 0; k < k_max; ++k) {
        const bool live = mapping && k < (int)n, junc = mapping && k + 1 < (int)n;
        int s3 = 0, e3 = 0;
        if (mapping) { const uint32_t i3 = off + min((uint32_t)k + 3u, last); s3 = xs[i3]; e3 = xe[i3]; }     // in flight during this round
        const uint32_t is = live ? min((uint32_t)((s >> SITE_SHIFT) + d.b_off), none) : none;
        const uint32_t ie = junc ? min((uint32_t)((e >> SITE_SHIFT) + d.b_off), none) : none;
        const int s2 = s1;
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
        s = s1; e = e1; s1 = s2n; e1 = e2n; s2n = s3; e2n = e3;
    }
    return m;
}

template <int LEVEL>
__global__ __launch_bounds__(TILE_THREADS, 8)
void k_probe(SplitArgs kernarg_block, const uint32_t *__restrict__ u_tile_first, const uint8_t *__restrict__ u_order,
             const uint32_t *__restrict__ u_tile_start, const uint32_t *__restrict__ u_tile_total)
{
    constexpr int DIR_BYTES = FAST_DIR_BYTES;
    __shared__ __attribute__((aligned(16))) uint16_t s_W[SPLIT_EXON_CAP];
    __shared__ __attribute__((aligned(16))) v4i_t s_ent[2 * FUSED_KEY_CAP];
    __shared__ __attribute__((aligned(16))) uint8_t s_dir[3 * DIR_BYTES];
    __shared__ __attribute__((aligned(16))) TileWin s_tw;
    __shared__ int s_wide;
    (void)kernarg_block;
    const SplitArgsK sa = split_args();
    const FusedArgsK a = fused_args();
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t t = blockIdx.x;
    const uint32_t r0 = u_tile_first[t], n_act = u_tile_first[t + 1u] - r0;
    const uint32_t base = u_tile_start[t], tile_total = u_tile_total[t];
    v4i_t *const s_ent0 = s_ent, *const s_ent1 = s_ent + FUSED_KEY_CAP;
    uint8_t *const s_dir0 = s_dir, *const s_dir1 = s_dir + DIR_BYTES, *const s_rdir = s_dir + 2 * DIR_BYTES;
    if ((int)threadIdx.x < SPLIT_TW_VECS)
        reinterpret_cast<int4 *>(&s_tw)[threadIdx.x] = reinterpret_cast<const int4 *>(sa->tw + t)[threadIdx.x];
    if (threadIdx.x == 0) s_wide = 0;
    // ---- the thread's read: where its exons are, how many, first and last exon
    const int32_t src = threadIdx.x < n_act ? (int32_t)ld32(u_order, r0 + threadIdx.x) : -1;
    const bool active = src >= 0;
    const uint32_t r = r0 + (uint32_t)max(src, 0);
    uint32_t off = base, pre = 0u;
    int32_t tid = 0; bool rev_in = false;
    if (active) { off = ld32(a->f.ex_off, r); pre = ld32(a->f.info, r); tid = ld32(a->f.r_tid, r); rev_in = ld32(a->f.r_rev, r) != 0; }
    const uint32_t n = pre >> 8, loc = off - base;
    ReadEnds re{0, 0, 0, 0};
    const int32_t *const xs = a->f.ex_start, *const xe = a->f.ex_end;
    if (active && n) { re.s0 = xs[off]; re.e0 = xe[off]; re.sl = xs[off + n - 1u]; re.el = xe[off + n - 1u]; }
    __syncthreads();
    const TileDesc d = s_tw.d;
    const bool in_lds = tile_total <= (uint32_t)SPLIT_EXON_CAP;
    const bool fast = (d.flags & TD_FAST) != 0 && in_lds;
    const int w_n = fast ? (int)d.n_win : 0;
    // ---- stage the dictionary slices, re-based to the tile's window
    int my_wide = 0;
    if (fast) {
        const FusedDict dv = fused_load_dict(a, d);
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
                e0.z = (int)rebase_gaps(s_tw.win, w_n, (uint32_t)dv.xb.x, (uint32_t)dv.xb.y, dv.xa.z);
                e0.w = (int)rebase_gaps(s_tw.win, w_n, (uint32_t)dv.xb.z, (uint32_t)dv.xb.w, dv.xa.z);
                e1.z = (int)rebase_gaps(s_tw.win, w_n, (uint32_t)dv.xd.x, (uint32_t)dv.xd.y, dv.xc.z);
                e1.w = (int)rebase_gaps(s_tw.win, w_n, (uint32_t)dv.xd.z, (uint32_t)dv.xd.w, dv.xc.z);
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
    __syncthreads();
    const int any_wide = s_wide;
    // ---- classification (device functions of the classic kernel; W = 16 bits per exon at the read's exact offset)
    uint32_t info = n << 8; int ref = -1;
    bool redo = active && (!fast || (pre & I_PRE_DIRECT) != 0u || any_wide != 0 || tid != d.tid || (n > 1 && (pre & I_PRE_INSANE) != 0u));
    const bool work = active && !redo;
    const TileLds L{nullptr, nullptr, s_W, s_ent0, s_ent1, s_dir0, s_dir1, s_rdir, s_tw.hk, s_tw.hx, s_tw.win};
    const VisitMasks vm = visit_window<LEVEL>(L, d, w_n, work, n, d.j_lo, re, s_tw.mask);
    redo = redo || vm.redo;
    const SiteMasks sm = map_exons_stream(L, d, work && !redo && n > 1, xs, xe, off, loc, n, vm.vpre);
    if (work && !redo) {
        const Verdict vd = decide<LEVEL>(L, d, loc, n, re, vm, sm, rev_in);
        info = vd.info; ref = vd.ref;
    } else if (active && in_lds) {
        for (uint32_t k = 0; k < n; ++k) s_W[loc + k] = (uint16_t)0;
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
    __syncthreads();
    uint8_t *const xf = a->f.ex_flag;
    if (in_lds) for (uint32_t i = threadIdx.x; i < tile_total; i += TILE_THREADS) xf[base + i] = (uint8_t)s_W[i];
    else for (uint32_t i = threadIdx.x; i < tile_total; i += TILE_THREADS) xf[base + i] = 0;
}

}  // namespace l2r
