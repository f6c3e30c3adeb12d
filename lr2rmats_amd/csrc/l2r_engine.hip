// l2r_engine.hip -- C-ABI implementation (include/lr2rmats_hip.h) over the gfx950
// kernels of l2r_kernels.hip.h.  One context = one GPU = one HIP stream.
// There is deliberately no CPU path in this file: every entry point that does
// work needs a HIP device and fails with a message otherwise.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/lr2rmats_hip.h"
#include "l2r_kernels.hip.h"
#include "l2r_window.hip.h"
#include "l2r_slab.hip.h"
#include "l2r_chunk.hip.h"
#include "l2r_tchunk.hip.h"
#include "l2r_filter.hip.h"

using namespace l2r;

static thread_local char g_err[1024] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(-2, "[%s] %s: %s", __func__, #expr, hipGetErrorString(e_));   \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;     // elements
    int ensure(size_t n)
    {
        if (n <= cap && p) return 0;
        // (contents are never carried over: every user refills the buffer.)  The old buffer is released first -- at 288 GB
        // shards both would not always fit -- so a failed grow leaves an EMPTY buffer (p = null, cap = 0), never a stale one.
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = n ? n : 1;
        T *q = nullptr;
        const hipError_t e = hipMalloc((void **)&q, want * sizeof(T));
        if (e != hipSuccess) return fail(-2, "hipMalloc(%zu bytes): %s", want * sizeof(T), hipGetErrorString(e));
        p = q; cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct l2r_ctx {
    int device = 0;
    int fast_grid = 0;
    bool wide_cigar = false;                // long CIGARs: the HBM walks fetch 16 words per lane and round (l2r_upload_reads decides)
    int n_cu = 256, wg_per_cu = 4;          // persistent grid of k_classify_fast (L2R_WG_PER_CU overrides)
    int ablate = 0;                         // diagnostics, L2R_ABLATE (read once, at l2r_create)
    int64_t seg_max = SEG_MAX;              // tiles up to which the segmented scans are used (l2r_kernels.hip.h); L2R_SEG_MAX
    int want_pipeline = 2;                  // L2R_PIPELINE: classic (0: l2r_kernels.hip.h, two walks), slab (1: l2r_slab.hip.h, one walk, two kernels), tile (2, default: l2r_tile.hip.h, one kernel per tile where the input allows it, else slab)
    bool many_exon_reads = false;           // the upload's sample: more than 0.5 % of the reads have more exons than a slab has rows
    bool slab_ok = false;                   // the current upload can run the slab pipeline: coordinate-sorted records, short CIGARs, its slab layout fits
    bool slab = false;                      // ... and the last launch did (the parameters have a say: launch_all)
    bool lists_heavy = false;                           // ... or most tiles went to them (an isoform-rich annotation): k_tile would only walk for them, which k_walk_slab does faster -- later runs take the slab pipeline
    bool redo_empty = false;                            // ... and nothing to the generic kernel either: every read had its junction check in k_tile, k_validate_sj has nothing to do
    bool wide_direct = true;                            // L2R_WIDE_DIRECT=0: the exact 64-bit-mask tiles keep the slab form and k_probe_slab_wide (k_tile's WIDE instance, l2r_tile.hip.h)
    bool fb_empty = false;                              // ... and k_tile left no tile in slab form for k_probe_slab (list_cnt[4])
    bool chunk_direct = true;                           // L2R_CHUNK_DIRECT=0: the exact tiles of the chunked kernel keep the slab form and k_probe_slab_chunked (k_tile_chunk, l2r_tchunk.hip.h)
    bool chunk_rest_empty = false;                      // ... and k_tile_chunk declined none, nobody appended late (list_cnt[9], [8]): k_probe_slab_chunked has nothing to do
    uint32_t n_late_tiles = 0;                          // ... tiles a one-window kernel handed to the chunked kernel late (list_cnt[8]): the grid of k_tile_chunk's second launch
    uint32_t n_chunk_tiles = 0;                         // ... entries of chunk_list a completed run has left: k_tile_chunk's grid
    bool wide_rest_empty = false;                       // ... and none of them kept the slab form (list_cnt[5]): k_probe_slab_wide has nothing to do
    uint32_t n_wide_tiles = 0;                          // ... entries of wide_list a completed run has left (l2r_sync): the WIDE instance's grid
    bool prev_run_tile = false;                         // the last launch of this upload took the tile path (else the list counters are whatever a slab run left: cleared before the next tile run)
    uint32_t lc_flip = 0;                               // ... and which of the two blocks of list counters (SlabArgs::list_cnt / list_cnt_next)
    uint32_t lb_flip = 0;                               // which of the two lb_sup arrays the next run of the tile path uses (l2r_slab.hip.h SlabArgs::lb_sup)
    bool lists_known = false, lists_empty = false;      // one-kernel tile path: a completed run of these inputs and parameters left nothing to k_probe_slab / _wide / _chunked (l2r_sync looks): their launches are skipped until something changes
    bool tile = false;                      // ... with the one-kernel tile path (l2r_tile.hip.h: short CIGARs, -e >= 1)
    DevBuf<unsigned long long> lb_tile, lb_blk, lb_sup;     // one-kernel tile path: the tiles' exon counts on their way to the later tiles' first slots
    DevBuf<uint32_t> fb_list;                               //                       the tiles it leaves to k_probe_slab
    DevBuf<uint16_t> sum_nn;                                //                       the records' N operations (l2r_reads::cig_summary) for k_tile_index<true>
    bool env_tile_anyway = false, env_launch_all = false;   // L2R_TILE_ANYWAY / L2R_LAUNCH_ALL (diagnostics), read once at l2r_create
    int64_t inexact_tiles = -1;                             //                       tiles that are not exact under the parameters now set (-1: not counted yet; drop_graph forgets it)
    bool tile_starved = false;                              //                       a tile of k_tile has waited in vain for the counts in front of it (or the device cannot hold the workgroups its look-back needs): the slab pipeline from then on
    int64_t n_lb_fallback = 0;                              //                       ... runs that were done again on the slab pipeline for that reason (l2r_debug_counters)
    bool have_index = false;                                //                       the current upload has its tile index (slot records, op statistics)
    bool pipeline_forced = false;                           // L2R_PIPELINE is set: the pipeline it names also for single-run uploads (tests, A/B runs)
    bool one_shot_upload = false;                           //                       ONE run will follow the upload (l2r_classify, l2r_hint_single_run): see l2r_classify
    float index_ms = 0.0f;                                  //                       GPU time of the last upload's k_tile_index (l2r_upload_index_ms)
    DevBuf<SlotRec> slot_rec;                               //                       the upload's slot records (k_tile_index)
    DevBuf<TileStat> sup_stat;                              //                       ... summed up per super-block of 1024 tiles
    DevBuf<TileStat> tile_stat; std::vector<TileStat> h_tile_stat;      //                 the upload's index of the tiles' CIGAR operations (k_tile_index)
    DevBuf<uint32_t> tile_sbase, s_pre, s_loc, s_pl, cig_off32, tile_rec, tile_total, tile_xbase, tile_span;    // (tile_span: 16-byte TileSpan records, l2r_slab.hip.h)
    DevBuf<int32_t> dense_start, dense_end;                 // slab pipeline: the outliers' dense area
    DevBuf<uint32_t> slab_row;                              //                the exon rows between its kernels (one word per exon)
    DevBuf<TileWin> tw;
    DevBuf<TileWin64> tw64; DevBuf<uint32_t> wide_list, chunk_list, list_cnt, tile_flags;     // tiles with 33 .. 63 window members (l2r_wide.hip.h)
    DevBuf<unsigned long long> ovf_cursor;
    std::string anno_cache_dir;             // L2R_ANNO_CACHE / l2r_set_annotation_cache: where the annotation tables are kept between runs
    int anno_cache_state = 0;               // last l2r_set_annotation: 0 no cache, 1 built + stored, 2 read from the cache
    unsigned want = L2R_WANT_RESULTS | L2R_WANT_ACCEPTED;      // l2r_set_outputs
    hipStream_t stream = nullptr;
    // one-kernel tile path: the instances of k_tile that take the isoform-rich tiles (WIDE, CHUNK) need nothing of the plain instance --
    // their tiles' first slots are known since k_describe_scan -- and run BESIDE it on streams of their own (forked behind
    // k_describe_scan, joined in front of the list kernels): a few thousand long-lived workgroups fill in where the plain instance's
    // short ones leave CUs, instead of costing a launch each with a tail of its own.  L2R_SIDE=0: one behind the other on `stream`.
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    bool side_on = true;
    l2r_params prm;
    // annotation
    int64_t n_tx = 0, n_anno_exon = 0;
    DevBuf<TxHdr> hdr;
    DevBuf<int2> anno_ex;
    DevBuf<int64_t> anno_key;
    DevBuf<SiteEnt> sk_st, sk_en;                   // site dictionaries (l2r_kernels.hip.h): START / END entries
    DevBuf<uint32_t> sd_st, sd_en, sr_st;           // bucket directories, reach-back directory of START
    DevBuf<int32_t> tid_base; int32_t n_tid_dir = 0;
    DevBuf<uint32_t> key_dir; DevBuf<int32_t> kb_base; int32_t n_tid_key = 0;   // cursor directory
    DevBuf<int32_t> j0;
    int64_t n_compact = 0, n_wide = 0;
    std::vector<int64_t> h_anno_key_raw;    // per transcript (tid,end) key, NOT prefix-maxed (unsorted-input cursor)
    std::vector<int64_t> h_anno_key_pm;     // ... and its running maximum (what anno_key holds on the device)
    // junctions
    int64_t n_sj = 0;
    DevBuf<int32_t> sj_tid, sj_don, sj_acc, sj_uniq, sj_multi;
    DevBuf<int64_t> sj_key;
    DevBuf<int32_t> sj_cbase, sj_dbase; DevBuf<uint32_t> sj_cdir, sj_ddir; DevBuf<int4> sj_row; int32_t sj_ntid = 0;      // SjDir (l2r_kernels.hip.h)
    std::vector<int64_t> h_sj_key_raw;      // per row (tid,acc) key
    std::vector<int64_t> h_sj_key_pm;       // ... and its running maximum
    // reads
    int64_t n_reads = 0, n_cigar = 0, first_read = 0;
    bool sorted = true;
    int reads_per_tile = TILE_THREADS;
    DevBuf<int32_t> r_tid, r_pos;
    DevBuf<uint8_t> r_rev;
    DevBuf<int64_t> cig_off;
    DevBuf<uint32_t> cig;
    std::vector<int32_t> h_tid, h_pos;      // kept for unsorted input, and with a junction table (cursor carry, below)
    // One input may arrive as SEVERAL uploads (a single-GPU run too large for one shard: l2r_upload_reads with
    // first_read_index == the number of records uploaded so far).  The two sequential cursors of check_trans()
    // (src/update_gtf.c:938 last_anno_i, last_sj_i) are then carried from upload to upload, so that unsorted input
    // sees the same history as one sequential pass.
    struct Stream {
        bool valid = false, sorted = true;
        int64_t next = 0;                   // first_read_index a continuation must have
        int64_t last_key = INT64_MIN;       // (tid, pos) of the last record so far (sortedness across uploads)
        int64_t anno_cur = 0, sj_cur = 0;   // cursor values after everything uploaded (and, for sj_cur, classified) so far
        int64_t anno_cur_start = 0, sj_cur_start = 0;   // ... and at the start of the current upload
        bool sj_pending = false;            // the current upload used the device's prefix cursor: sj_cur is brought up to date at the next upload
    } seq;
    DevBuf<int32_t> win_start, sj_cursor;   // only for unsorted input
    bool have_win = false;
    // work + results
    int64_t n_tiles = 0, n_tiles256 = 0;
    DevBuf<uint32_t> local, tile_base, ex_off, info, tile_acc, tile_acc_ex, tile_acc_at, tile_acc_ex_at, tile_chunk, tile_rchunk, totals;  // totals[0]=exons [1]=accepted [2]=accepted exons [3]=redo count [4],[5]=chunk cursor of the accepted list (one 64-bit word: exon slot low, record slot high)
    DevBuf<uint32_t> redo;                  // reads the fast kernel hands to the generic one
    DevBuf<uint8_t> order;                  // per tile: reads by falling exon count (pass A)
    DevBuf<TileDesc> desc;
    DevBuf<int2> walked;               // long-CIGAR inputs: exons walked by pass A (n_cigar + n_reads slots)
    DevBuf<uint32_t> tile_first;       // first read of every tile (+ one closing entry)
    DevBuf<TxHdr> win_hdr;             // WIN_TX window headers per tile (pass A)
    DevBuf<int32_t> ex_start, ex_end, ref_tx;
    DevBuf<uint8_t> ex_flag;
    int64_t ex_cap = 0;
    DevBuf<AccRec> acc_rec;
    DevBuf<uint32_t> acc_ex_off;
    DevBuf<int32_t> acc_start, acc_end;
    DevBuf<uint8_t> acc_flag;
    bool ran = false;
    hipGraphExec_t graph = nullptr;         // the launch sequence of l2r_run, captured once per (inputs, parameters)
    bool graph_valid = false;
    bool check_stages = false;              // L2R_CHECK
    DevBuf<unsigned long long> stamps;      // diagnostics, L2R_STAMPS=1
    uint32_t h_totals[3] = {0, 0, 0};
    bool totals_valid = false;
};

static void drop_graph(l2r_ctx *c)
{
    c->lists_known = false; c->lists_empty = false; c->redo_empty = false; c->lists_heavy = false;     // (called wherever inputs, parameters or outputs change)
    c->inexact_tiles = -1;
    if (c->graph) { (void)hipGraphExecDestroy(c->graph); c->graph = nullptr; }
    c->graph_valid = false;
}

static DevParams dev_params(const l2r_ctx *c)
{
    DevParams p;
    p.min_exon = c->prm.min_exon; p.min_intron = c->prm.min_intron; p.max_delet = c->prm.max_delet;
    p.ss_dis = c->prm.ss_dis; p.full_level = c->prm.full_level; p.use_multi = c->prm.use_multi;
    p.min_sj_cnt = c->prm.min_sj_cnt; p.split_trans = c->prm.split_trans; p.frac = c->prm.single_exon_ovlp_frac;
    p.n_tx = (int32_t)c->n_tx; p.n_sj = (int32_t)c->n_sj; p.reads_per_tile = c->reads_per_tile;
    p.ablate = c->ablate; p.want = (int32_t)c->want;
    return p;
}

static inline int64_t host_key(int32_t tid, int32_t x) { return ((int64_t)(tid + 1) << 32) | (uint32_t)x; }

extern "C" {

int l2r_abi_version(void) { return L2R_ABI_VERSION; }
const char *l2r_last_error(void) { return g_err; }

int l2r_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

l2r_ctx *l2r_create(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { fail(-1, "[l2r_create] no HIP device available (%s); this library has no CPU path", hipGetErrorString(e)); return nullptr; }
    if (device < 0 || device >= n) { fail(-1, "[l2r_create] device %d out of range (have %d)", device, n); return nullptr; }
    if ((e = hipSetDevice(device)) != hipSuccess) { fail(-2, "[l2r_create] hipSetDevice: %s", hipGetErrorString(e)); return nullptr; }
    l2r_ctx *c = new l2r_ctx();
    c->device = device;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        fail(-2, "[l2r_create] hipStreamCreate: %s", hipGetErrorString(e)); delete c; return nullptr;
    }
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);      // (lowest, highest)
    const char *sp = getenv("L2R_SIDE_PRIO");
    // (the side streams carry the long-lived workgroups: highest priority, so that they are on their way early and the plain instance's short ones
    //  fill in -- measured on cfg3_gencode: lowest 0.665, default 0.660, highest 0.658 ms: the dispatcher hardly cares.  L2R_SIDE_PRIO: 0 default, < 0 lowest)
    const int side_prio = !sp ? prio_hi : (atoi(sp) == 0 ? 0 : (atoi(sp) > 0 ? prio_hi : prio_lo));
    for (int k = 0; k < 2; ++k)
        if (((e = hipStreamCreateWithPriority(&c->side[k], hipStreamNonBlocking, side_prio)) != hipSuccess &&
             ((void)hipGetLastError(), e = hipStreamCreateWithFlags(&c->side[k], hipStreamNonBlocking)) != hipSuccess) ||      // (a runtime without stream priorities: a plain stream does)
            (e = hipEventCreateWithFlags(&c->ev_join[k], hipEventDisableTiming)) != hipSuccess) {
            fail(-2, "[l2r_create] side stream: %s", hipGetErrorString(e)); l2r_destroy(c); return nullptr;
        }
    if ((e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess) { fail(-2, "[l2r_create] hipEventCreate: %s", hipGetErrorString(e)); l2r_destroy(c); return nullptr; }
    // src/update_gtf.c:24-35 defaults
    c->prm = l2r_params{3, 3, 50, 0, 0x7fffffff, 5, 0, 0, 1, 0, 0.80f};
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
        const char *e = getenv("L2R_WG_PER_CU");
        if (e && atoi(e) > 0) c->wg_per_cu = atoi(e);
        e = getenv("L2R_FAST_GRID");
        if (e && atoi(e) > 0) c->fast_grid = atoi(e);
        e = getenv("L2R_ABLATE");
        c->ablate = e ? atoi(e) : 0;
        c->check_stages = getenv("L2R_CHECK") != nullptr;
        e = getenv("L2R_SEG_MAX");                       // diagnostics / tests: tiles beyond which the scans take k_scan_u32 in a launch of its own
        if (e && atoll(e) >= 0) c->seg_max = atoll(e);
        e = getenv("L2R_ANNO_CACHE");
        if (e && *e) c->anno_cache_dir = e;
        e = getenv("L2R_WIDE_DIRECT");
        if (e) c->wide_direct = atoi(e) != 0;
        c->env_tile_anyway = getenv("L2R_TILE_ANYWAY") != nullptr; c->env_launch_all = getenv("L2R_LAUNCH_ALL") != nullptr;
        e = getenv("L2R_CHUNK_DIRECT");
        if (e) c->chunk_direct = atoi(e) != 0;
        e = getenv("L2R_SIDE");
        if (e) c->side_on = atoi(e) != 0;
        e = getenv("L2R_PIPELINE");
        if (e) { c->want_pipeline = !strcmp(e, "classic") ? 0 : !strcmp(e, "slab") ? 1 : 2; c->pipeline_forced = true; }
        {   // k_tile's look-back: a tile may wait for one whose workgroup comes up to 8 * TILE_GROUP - 1 block indices later (fused_tile), so
            // that many workgroups + 1 have to be resident together -- guaranteed nowhere; checked here (small or partitioned devices, CU
            // masks): a device that cannot hold them takes the slab pipeline from the start
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tile<3, false, false, false>, TILE_THREADS, 0) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
            if ((int64_t)per_cu * c->n_cu < 8 * (int64_t)TILE_GROUP + 1 || getenv("L2R_TILE_STARVED")) c->tile_starved = true;
        }
    }
    return c;
}

void l2r_destroy(l2r_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    c->hdr.release(); c->anno_ex.release(); c->anno_key.release();
    c->sk_st.release(); c->sk_en.release(); c->sd_st.release(); c->sd_en.release(); c->sr_st.release(); c->tid_base.release();
    c->key_dir.release(); c->kb_base.release(); c->j0.release();
    c->sj_tid.release(); c->sj_don.release(); c->sj_acc.release(); c->sj_uniq.release(); c->sj_multi.release(); c->sj_key.release(); c->sj_cbase.release(); c->sj_cdir.release(); c->sj_dbase.release(); c->sj_ddir.release(); c->sj_row.release();
    c->r_tid.release(); c->r_pos.release(); c->r_rev.release(); c->cig_off.release(); c->cig.release();
    c->win_start.release(); c->sj_cursor.release();
    c->local.release(); c->order.release(); c->redo.release(); c->desc.release(); c->win_hdr.release(); c->tile_first.release(); c->walked.release(); c->stamps.release(); c->tile_base.release(); c->ex_off.release(); c->info.release(); c->tile_acc.release(); c->tile_acc_ex.release(); c->tile_acc_at.release(); c->tile_acc_ex_at.release(); c->tile_chunk.release(); c->tile_rchunk.release(); c->totals.release();
    c->ex_start.release(); c->ex_end.release(); c->ref_tx.release(); c->ex_flag.release();
    c->tile_total.release(); c->tile_xbase.release(); c->tile_rec.release(); c->cig_off32.release(); c->s_pl.release(); c->tile_span.release(); c->tile_sbase.release(); c->ovf_cursor.release(); c->tw64.release(); c->wide_list.release(); c->chunk_list.release(); c->list_cnt.release(); c->tile_flags.release();
    c->lb_tile.release(); c->lb_blk.release(); c->lb_sup.release(); c->fb_list.release(); c->tile_stat.release(); c->sup_stat.release(); c->slot_rec.release(); c->sum_nn.release();
    c->slab_row.release(); c->dense_start.release(); c->dense_end.release(); c->s_pre.release(); c->s_loc.release(); c->tw.release();
    c->acc_rec.release(); c->acc_ex_off.release(); c->acc_start.release(); c->acc_end.release(); c->acc_flag.release();
    drop_graph(c);
    for (int k = 0; k < 2; ++k) { if (c->side[k]) (void)hipStreamDestroy(c->side[k]); if (c->ev_join[k]) (void)hipEventDestroy(c->ev_join[k]); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void *l2r_stream(l2r_ctx *c) { return c ? (void *)c->stream : nullptr; }
float l2r_upload_index_ms(l2r_ctx *c) { return c ? c->index_ms : 0.0f; }
int l2r_hint_single_run(l2r_ctx *c, int on) { if (!c) return fail(-1, "[l2r_hint_single_run] null context"); c->one_shot_upload = on != 0; return 0; }

int l2r_set_outputs(l2r_ctx *c, unsigned want)
{
    if (!c) return fail(-1, "[l2r_set_outputs] null context");
    if (!(want & (L2R_WANT_RESULTS | L2R_WANT_ACCEPTED)) || (want & ~(unsigned)(L2R_WANT_RESULTS | L2R_WANT_ACCEPTED)))
        return fail(-1, "[l2r_set_outputs] want = %u: expected L2R_WANT_RESULTS and/or L2R_WANT_ACCEPTED", want);
    // The per-read result arrays are always produced on the device (the junction check and the redo list read them);
    // what the flag saves is the compaction of the accepted list (kernel work, 16 + 9n bytes per accepted read).
    c->want = want;
    c->ran = false; drop_graph(c);
    return 0;
}

int l2r_set_params(l2r_ctx *c, const l2r_params *prm)
{
    if (!c || !prm) return fail(-1, "[l2r_set_params] null argument");
    c->prm = *prm;
    c->ran = false; drop_graph(c);
    return 0;
}

// ---- site dictionaries -------------------------------------------------------------------------
// A site of a kind with the transcript (file order) that has it.  Sorting by (tid, k1, k2, tx) groups the
// members of every distinct site.
struct SiteTx {
    int32_t tid, k1, k2, tx;
    bool operator<(const SiteTx &o) const
    {
        if (tid != o.tid) return tid < o.tid;
        if (k1 != o.k1) return k1 < o.k1;
        if (k2 != o.k2) return k2 < o.k2;
        return tx < o.tx;
    }
};

// Everything l2r_set_annotation derives from the annotation, on the host: what is uploaded, and what the cache file holds.
struct AnnoTables {
    std::vector<TxHdr> hdr;
    std::vector<int64_t> key, key_raw;                 // cursor keys: prefix maximum / per transcript
    std::vector<int2> ex;
    std::vector<SiteEnt> st_ent, en_ent;               // START / END dictionaries
    std::vector<uint32_t> st_dir, st_rdir, en_dir;     // their bucket directories (START: + reach-back directory)
    std::vector<int32_t> tid_base, kb_base;
    std::vector<uint32_t> key_dir;
    int64_t n_wide = 0, n_compact = 0;
    int32_t n_tid_dir = 0, n_tid_key = 0;
};

// Entries of one dictionary + its bucket directory.  `pairs`: sorted (tid, k1, k2, tx) rows of the pair kind
// (exons for START, junctions for END); `singles`: sorted (tid, k1, 0, tx) rows of the single kind (acceptors /
// donors).  An entry's masks are relative to its tx_base and say 64 transcripts; a key whose members (of either kind) lie further
// apart gets SEVERAL entries in a row -- parts, same (k1, k2), rising tx_base, each flagged SE_WIDE: the 32- and 64-bit mask kernels
// leave a tile with such an entry to k_probe_slab_chunked (slab pipeline; it ORs the parts) or to the generic kernel (classic).
static void build_dict(const std::vector<SiteTx> &pairs, const std::vector<SiteTx> &singles, const std::vector<int32_t> &tid_base,
                       std::vector<SiteEnt> &ent, std::vector<uint32_t> &dir, std::vector<uint32_t> *rdir_out, int64_t &n_wide)
{
    const size_t nb = (size_t)tid_base.back();
    dir.assign(nb + 1, 0);                                // dir[b] = number of entries whose bucket id is < b
    ent.clear();
    ent.reserve(pairs.size());
    std::vector<uint32_t> parts;                          // entries per distinct pair (reach-back directory below)
    parts.reserve(pairs.size());
    size_t si = 0;                                        // walks `singles` in step (both sorted by (tid, k1))
    for (size_t i = 0; i < pairs.size();) {
        size_t j = i;
        while (j < pairs.size() && pairs[j].tid == pairs[i].tid && pairs[j].k1 == pairs[i].k1 && pairs[j].k2 == pairs[i].k2) ++j;
        // members of the single kind with the same (tid, k1)
        while (si < singles.size() && (singles[si].tid < pairs[i].tid || (singles[si].tid == pairs[i].tid && singles[si].k1 < pairs[i].k1))) ++si;
        size_t sj = si;
        while (sj < singles.size() && singles[sj].tid == pairs[i].tid && singles[sj].k1 == pairs[i].k1) ++sj;
        const size_t first_ent = ent.size();
        size_t pk = i, sk = si;                            // (members of both kinds rise with tx)
        while (pk < j || sk < sj) {
            int32_t lo = INT32_MAX;
            if (pk < j) lo = std::min(lo, pairs[pk].tx);
            if (sk < sj) lo = std::min(lo, singles[sk].tx);
            SiteEnt e;
            memset(&e, 0, sizeof e);
            e.k1 = pairs[i].k1; e.k2 = pairs[i].k2; e.tx_base = lo;
            for (; pk < j && (int64_t)pairs[pk].tx - lo < 64; ++pk) { const int off = pairs[pk].tx - lo; e.pm[off >> 5] |= 1u << (off & 31); }
            for (; sk < sj && (int64_t)singles[sk].tx - lo < 64; ++sk) { const int off = singles[sk].tx - lo; e.sm[off >> 5] |= 1u << (off & 31); }
            ent.push_back(e);
        }
        const uint32_t np = (uint32_t)(ent.size() - first_ent);
        if (np > 1) { for (size_t q = first_ent; q < ent.size(); ++q) ent[q].flags |= SE_WIDE; ++n_wide; }
        parts.push_back(np);
        const size_t b = (size_t)tid_base[(size_t)pairs[i].tid] + (size_t)(pairs[i].k1 >> SITE_SHIFT);
        dir[b + 1] += np;
        i = j;                                             // `si` stays: the next pair may share (tid, k1)
    }
    for (size_t b = 0; b < nb; ++b) dir[b + 1] += dir[b];
    if (rdir_out) {
        // reach-back directory (START: k1 = exon start, k2 = exon end): first entry whose exon reaches into the bucket
        std::vector<uint32_t> &rdir = *rdir_out;
        rdir = dir;
        size_t i = 0, u = 0;
        for (size_t q = 0; q < pairs.size(); ++u) {       // entries [i, i + parts[u]) <-> the u-th distinct pair
            size_t j = q;
            while (j < pairs.size() && pairs[j].tid == pairs[q].tid && pairs[j].k1 == pairs[q].k1 && pairs[j].k2 == pairs[q].k2) ++j;
            const size_t t0 = (size_t)tid_base[(size_t)pairs[q].tid], nbt = (size_t)tid_base[(size_t)pairs[q].tid + 1] - t0;
            const size_t sb = (size_t)(pairs[q].k1 >> SITE_SHIFT);
            size_t eb = pairs[q].k2 < 0 ? sb : (size_t)(pairs[q].k2 >> SITE_SHIFT);
            if (eb >= nbt) eb = nbt - 1;
            for (size_t b = sb + 1; b <= eb; ++b) if (rdir[t0 + b] > (uint32_t)i) rdir[t0 + b] = (uint32_t)i;
            i += parts[u]; q = j;
        }
    }
}

// Directory over non-decreasing 64-bit keys host_key(tid, x): dir[base[tid] + c] = first j with key_j >= (tid, c << 9), one closing word
// (first j with key_j >= (n_tid, 0)); mx[tid] = the largest x of the chromosome (-1: none, no buckets).  Used for the annotation cursor,
// the junction cursor and the junction table's donors (cursor_value / sj_first_row in l2r_kernels.hip.h).
static int build_key_dir(const int64_t *key, int64_t n, int32_t n_tid, const std::vector<int64_t> &mx, std::vector<int32_t> &base, std::vector<uint32_t> &dir)
{
    base.assign((size_t)n_tid + 1, 0);
    int64_t acc = 0;
    for (int32_t t = 0; t < n_tid; ++t) { base[(size_t)t] = (int32_t)acc; acc += mx[(size_t)t] < 0 ? 0 : (mx[(size_t)t] >> SITE_SHIFT) + 1; }
    if (acc > 0x7ffffff0LL) return -1;
    base[(size_t)n_tid] = (int32_t)acc;
    dir.assign((size_t)acc + 1, 0);
    size_t j = 0;
    for (int32_t t = 0; t < n_tid; ++t) {
        const int32_t nbk = base[(size_t)t + 1] - base[(size_t)t];
        for (int32_t cb = 0; cb < nbk; ++cb) {
            const int64_t q = host_key(t, cb << SITE_SHIFT);
            while (j < (size_t)n && key[j] < q) ++j;
            dir[(size_t)base[(size_t)t] + (size_t)cb] = (uint32_t)j;
        }
    }
    {   // closing word: first j with key >= (n_tid, 0)
        const int64_t q = host_key(n_tid, 0);
        while (j < (size_t)n && key[j] < q) ++j;
        dir[(size_t)acc] = (uint32_t)j;
    }
    // (words of chromosomes without buckets: they share the next chromosome's first word)
    return 0;
}

static int build_tables(const l2r_annotation *a, AnnoTables &o)
{
    const int64_t T = a->n_tx;
    std::vector<TxHdr> &h = o.hdr;
    std::vector<int64_t> &key = o.key;
    h.assign((size_t)T, TxHdr());
    key.assign((size_t)T, 0);
    o.key_raw.assign((size_t)T, 0);
    int64_t run = INT64_MIN;
    // headers, cursor keys, and the (site, transcript) rows of every kind
    std::vector<SiteTx> kd, ka, kx, kj;
    kd.reserve((size_t)a->n_exon); ka.reserve((size_t)a->n_exon); kx.reserve((size_t)a->n_exon); kj.reserve((size_t)a->n_exon);
    int64_t n_compact = 0;
    for (int64_t i = 0; i < T; ++i) {
        const int64_t lo = a->tx_ex_off[i], hi = a->tx_ex_off[i + 1];
        if (lo < 0 || hi < lo || hi > a->n_exon) return fail(-1, "[l2r_set_annotation] bad exon offsets at transcript %lld", (long long)i);
        if (hi == lo) return fail(-1, "[l2r_set_annotation] transcript %lld has no exon", (long long)i);
        TxHdr &t = h[(size_t)i];
        memset(&t, 0, sizeof t);
        t.tid = a->tx_tid[i]; t.start = a->tx_start[i]; t.end = a->tx_end[i]; t.ex_off = (int32_t)lo;
        t.n = (int32_t)(hi - lo); t.rev = a->tx_rev[i] ? 1 : 0;
        t.s0 = a->ex_start[lo]; t.e0 = a->ex_end[lo]; t.sl = a->ex_start[hi - 1]; t.el = a->ex_end[hi - 1];
        bool mono = true, sane = a->ex_start[lo] <= a->ex_end[lo];
        for (int64_t k = lo + 1; k < hi; ++k) {
            if (!(a->ex_start[k] > a->ex_start[k - 1] && a->ex_end[k] > a->ex_end[k - 1])) mono = false;
            if (a->ex_start[k] > a->ex_end[k]) sane = false;
        }
        int flags = mono ? TX_MONO : 0;
        // the "equal value => inside both spans" argument of the fast kernel holds for these
        if (mono && sane && t.tid >= 0 && t.n >= 2 && t.start == t.s0 && t.end == t.el) { flags |= TX_COMPACT; ++n_compact; }
        t.flags = flags;
        if (t.tid >= 0 && t.n >= 2) for (int64_t k = lo; k < hi; ++k) {
            if (a->ex_start[k] < 0 || a->ex_end[k] < 0) return fail(-1, "[l2r_set_annotation] negative exon coordinate");
            kx.push_back(SiteTx{t.tid, a->ex_start[k], a->ex_end[k], (int32_t)i});
            if (k + 1 < hi) { kd.push_back(SiteTx{t.tid, a->ex_end[k], 0, (int32_t)i}); kj.push_back(SiteTx{t.tid, a->ex_end[k], a->ex_start[k + 1], (int32_t)i}); }
            if (k > lo) ka.push_back(SiteTx{t.tid, a->ex_start[k], 0, (int32_t)i});
        }
        // "annotation before read": tid smaller, or same tid and end <= read start (update_gtf.c:786-790).
        // The sequential cursor equals the longest prefix that is entirely before the read = first index
        // whose running maximum of (tid,end) exceeds (read.tid, read.start)  (SURVEY.md 3.3).
        const int64_t k = host_key(t.tid, t.end);
        o.key_raw[(size_t)i] = k;
        if (k > run) run = k;
        key[(size_t)i] = run;
    }
    {   // the four row sets are independent: one thread each
        std::thread t1([&] { std::sort(kd.begin(), kd.end()); }), t2([&] { std::sort(ka.begin(), ka.end()); }), t3([&] { std::sort(kx.begin(), kx.end()); });
        std::sort(kj.begin(), kj.end());
        t1.join(); t2.join(); t3.join();
    }
    o.n_compact = n_compact;
    {   // one bucket grid for the four kinds: per tid, enough 512-bp buckets for its largest site coordinate
        int32_t n_tid = 0;
        for (const auto *v : {&kd, &ka, &kx, &kj}) if (!v->empty()) n_tid = std::max(n_tid, v->back().tid + 1);
        std::vector<int64_t> mx((size_t)n_tid, -1);
        for (const auto *v : {&kd, &ka, &kx, &kj}) for (const SiteTx &k : *v) mx[(size_t)k.tid] = std::max<int64_t>(mx[(size_t)k.tid], k.k1);
        // ... and for the last exon END of every chromosome: the full-length evidence looks for exons that reach into
        // the bucket of a read's terminal exon (rdir), so the grid has to cover exon ends, not only the probe keys
        for (const SiteTx &k : kx) mx[(size_t)k.tid] = std::max<int64_t>(mx[(size_t)k.tid], k.k2);
        std::vector<int32_t> &tb = o.tid_base;
        tb.assign((size_t)n_tid + 1, 0);
        int64_t acc = 0;
        for (int32_t t = 0; t < n_tid; ++t) { tb[(size_t)t] = (int32_t)acc; acc += mx[(size_t)t] < 0 ? 0 : (mx[(size_t)t] >> SITE_SHIFT) + 1; }
        if (acc > 0x7ffffff0LL) return fail(-1, "[l2r_set_annotation] site directory too large");
        tb[(size_t)n_tid] = (int32_t)acc;
        // START: exons + the transcripts in which their start is an acceptor; END: junctions + donors
        o.n_wide = 0;
        build_dict(kx, ka, tb, o.st_ent, o.st_dir, &o.st_rdir, o.n_wide);
        build_dict(kj, kd, tb, o.en_ent, o.en_dir, nullptr, o.n_wide);
        o.n_tid_dir = n_tid;
    }
    {   // cursor directory over the prefix-max keys: dir[kb_base[tid] + c] = first j with key_j >= (tid, c << 9)
        int32_t n_tid = 0;
        for (int64_t i = 0; i < T; ++i) n_tid = std::max(n_tid, h[(size_t)i].tid + 1);
        std::vector<int64_t> mxe((size_t)n_tid, -1);
        for (int64_t i = 0; i < T; ++i) if (h[(size_t)i].tid >= 0) mxe[(size_t)h[(size_t)i].tid] = std::max<int64_t>(mxe[(size_t)h[(size_t)i].tid], std::max(h[(size_t)i].end, 0));
        if (build_key_dir(key.data(), T, n_tid, mxe, o.kb_base, o.key_dir)) return fail(-1, "[l2r_set_annotation] cursor directory too large");
        // words of chromosomes without buckets (no transcript): they share the next chromosome's first word
        o.n_tid_key = n_tid;
    }
    o.ex.resize((size_t)a->n_exon);
    for (int64_t k = 0; k < a->n_exon; ++k) o.ex[(size_t)k] = make_int2(a->ex_start[k], a->ex_end[k]);
    return 0;
}

// ---- the tables on disk (L2R_ANNO_CACHE=<directory>, or l2r_set_annotation_cache): one file per annotation, named by a
// hash of the arrays l2r_set_annotation was given + the layout constants; a hit replaces the sorts and the dictionary build
// by one read.  The file is written to a temporary name and renamed, so a reader never sees a partial file; anything that
// does not match (length, magic, hash, sizes) is ignored and rebuilt.
static uint64_t mix64(uint64_t h, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 0x9E3779B97F4A7C15ULL; h ^= h >> 29; }
    uint64_t w = 0;
    if (i < n) { memcpy(&w, b + i, n - i); h = (h ^ w) * 0x9E3779B97F4A7C15ULL; h ^= h >> 29; }
    return (h ^ n) * 0xD6E8FEB86659FD93ULL;
}

static uint64_t annotation_hash(const l2r_annotation *a)
{
    const uint64_t consts[] = {0x4c3252414e4e4f32ULL /* format */, (uint64_t)SITE_SHIFT, sizeof(TxHdr), sizeof(SiteEnt), (uint64_t)a->n_tx, (uint64_t)a->n_exon};
    uint64_t h = mix64(0x1234567ULL, consts, sizeof consts);
    const size_t T = (size_t)a->n_tx, X = (size_t)a->n_exon;
    h = mix64(h, a->tx_tid, T * 4); h = mix64(h, a->tx_start, T * 4); h = mix64(h, a->tx_end, T * 4); h = mix64(h, a->tx_rev, T);
    h = mix64(h, a->tx_ex_off, (T + 1) * 8); h = mix64(h, a->ex_start, X * 4); h = mix64(h, a->ex_end, X * 4);
    return h;
}

struct CacheHead { char magic[8]; uint64_t hash; uint64_t n[12]; int64_t n_wide, n_compact; int32_t n_tid_dir, n_tid_key; uint64_t sum; };

static uint64_t tables_sum(const AnnoTables &t)
{
    uint64_t h = 0x7ab1e5;
    h = mix64(h, t.hdr.data(), t.hdr.size() * sizeof(TxHdr)); h = mix64(h, t.key.data(), t.key.size() * 8); h = mix64(h, t.key_raw.data(), t.key_raw.size() * 8);
    h = mix64(h, t.ex.data(), t.ex.size() * sizeof(int2)); h = mix64(h, t.st_ent.data(), t.st_ent.size() * sizeof(SiteEnt));
    h = mix64(h, t.en_ent.data(), t.en_ent.size() * sizeof(SiteEnt)); h = mix64(h, t.st_dir.data(), t.st_dir.size() * 4);
    h = mix64(h, t.st_rdir.data(), t.st_rdir.size() * 4); h = mix64(h, t.en_dir.data(), t.en_dir.size() * 4);
    h = mix64(h, t.tid_base.data(), t.tid_base.size() * 4); h = mix64(h, t.kb_base.data(), t.kb_base.size() * 4);
    return mix64(h, t.key_dir.data(), t.key_dir.size() * 4);
}

extern "C++" {
template <typename T> static bool put_vec(FILE *f, const std::vector<T> &v) { return v.empty() || fwrite(v.data(), sizeof(T), v.size(), f) == v.size(); }
template <typename T> static bool get_vec(FILE *f, std::vector<T> &v, uint64_t n) { v.resize((size_t)n); return n == 0 || fread(v.data(), sizeof(T), (size_t)n, f) == (size_t)n; }
}

static std::string cache_path(const std::string &dir, uint64_t hash)
{
    char name[64];
    snprintf(name, sizeof name, "/l2r_anno_%016llx.tables", (unsigned long long)hash);
    return dir + name;
}

static bool cache_load(const std::string &path, uint64_t hash, int64_t n_tx, int64_t n_exon, AnnoTables &t)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    CacheHead hd;
    bool ok = fread(&hd, sizeof hd, 1, f) == 1 && memcmp(hd.magic, "L2RANNO3", 8) == 0 && hd.hash == hash &&
              hd.n[0] == (uint64_t)n_tx && hd.n[1] == (uint64_t)n_tx && hd.n[2] == (uint64_t)n_tx && hd.n[3] == (uint64_t)n_exon;
    if (ok) {
        // the lengths must add up to the file's length before anything is allocated from them
        const uint64_t sz[12] = {sizeof(TxHdr), 8, 8, sizeof(int2), sizeof(SiteEnt), sizeof(SiteEnt), 4, 4, 4, 4, 4, 4};
        uint64_t want = sizeof hd;
        for (int k = 0; k < 12; ++k) { if (hd.n[k] > 0x7ffffff0ULL) ok = false; want += hd.n[k] * sz[k]; }
        fseek(f, 0, SEEK_END);
        ok = ok && (uint64_t)ftell(f) == want;
        fseek(f, (long)sizeof hd, SEEK_SET);
    }
    ok = ok && get_vec(f, t.hdr, hd.n[0]) && get_vec(f, t.key, hd.n[1]) && get_vec(f, t.key_raw, hd.n[2]) && get_vec(f, t.ex, hd.n[3]) &&
         get_vec(f, t.st_ent, hd.n[4]) && get_vec(f, t.en_ent, hd.n[5]) && get_vec(f, t.st_dir, hd.n[6]) && get_vec(f, t.st_rdir, hd.n[7]) &&
         get_vec(f, t.en_dir, hd.n[8]) && get_vec(f, t.tid_base, hd.n[9]) && get_vec(f, t.kb_base, hd.n[10]) && get_vec(f, t.key_dir, hd.n[11]);
    fclose(f);
    if (ok) { t.n_wide = hd.n_wide; t.n_compact = hd.n_compact; t.n_tid_dir = hd.n_tid_dir; t.n_tid_key = hd.n_tid_key; }
    // the payload is what was written, and the directories index the entries: nothing else may reach the kernels
    ok = ok && tables_sum(t) == hd.sum;
    ok = ok && t.tid_base.size() == (size_t)t.n_tid_dir + 1 && t.kb_base.size() == (size_t)t.n_tid_key + 1 &&
         t.st_dir.size() == t.st_rdir.size() && t.st_dir.size() == t.en_dir.size() && !t.st_dir.empty() &&
         t.st_dir.back() == t.st_ent.size() && t.en_dir.back() == t.en_ent.size() &&
         (int64_t)t.st_dir.size() == (int64_t)t.tid_base.back() + 1 && (int64_t)t.key_dir.size() == (int64_t)t.kb_base.back() + 1;
    return ok;
}

static void cache_store(const std::string &path, uint64_t hash, const AnnoTables &t)
{
    char tmp_suffix[48];
    snprintf(tmp_suffix, sizeof tmp_suffix, ".tmp.%ld", (long)getpid());
    const std::string tmp = path + tmp_suffix;
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;                                        // a cache that cannot be written is no error
    CacheHead hd;
    memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, "L2RANNO3", 8); hd.hash = hash;
    const uint64_t n[12] = {t.hdr.size(), t.key.size(), t.key_raw.size(), t.ex.size(), t.st_ent.size(), t.en_ent.size(), t.st_dir.size(),
                            t.st_rdir.size(), t.en_dir.size(), t.tid_base.size(), t.kb_base.size(), t.key_dir.size()};
    memcpy(hd.n, n, sizeof n);
    hd.n_wide = t.n_wide; hd.n_compact = t.n_compact; hd.n_tid_dir = t.n_tid_dir; hd.n_tid_key = t.n_tid_key;
    hd.sum = tables_sum(t);
    bool ok = fwrite(&hd, sizeof hd, 1, f) == 1 && put_vec(f, t.hdr) && put_vec(f, t.key) && put_vec(f, t.key_raw) && put_vec(f, t.ex) &&
              put_vec(f, t.st_ent) && put_vec(f, t.en_ent) && put_vec(f, t.st_dir) && put_vec(f, t.st_rdir) && put_vec(f, t.en_dir) &&
              put_vec(f, t.tid_base) && put_vec(f, t.kb_base) && put_vec(f, t.key_dir);
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
}

extern "C++" {
template <typename T> static int put_dev(l2r_ctx *c, DevBuf<T> &b, const std::vector<T> &v)
{
    if (b.ensure(v.size() ? v.size() : 1)) return -2;
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return 0;
}
}

int l2r_set_annotation_cache(l2r_ctx *c, const char *dir)
{
    if (!c) return fail(-1, "[l2r_set_annotation_cache] null context");
    c->anno_cache_dir = dir ? dir : "";
    return 0;
}

int l2r_annotation_cache_state(l2r_ctx *c) { return c ? c->anno_cache_state : -1; }

int l2r_set_annotation(l2r_ctx *c, const l2r_annotation *a)
{
    if (!c || !a) return fail(-1, "[l2r_set_annotation] null argument");
    if (a->n_tx < 0 || a->n_tx > 0x7ffffff0LL || a->n_exon < 0 || a->n_exon > 0x7ffffff0LL) return fail(-1, "[l2r_set_annotation] size out of range");
    HIP_TRY(hipSetDevice(c->device));
    AnnoTables t;
    bool hit = false;
    std::string path;
    c->anno_cache_state = 0;
    if (!c->anno_cache_dir.empty()) {
        const uint64_t hash = annotation_hash(a);
        path = cache_path(c->anno_cache_dir, hash);
        hit = cache_load(path, hash, a->n_tx, a->n_exon, t);
        if (!hit) {
            t = AnnoTables();
            int rc = build_tables(a, t);
            if (rc) return rc;
            (void)mkdir(c->anno_cache_dir.c_str(), 0777);
            cache_store(path, hash, t);
        }
        c->anno_cache_state = hit ? 2 : 1;
    } else {
        int rc = build_tables(a, t);
        if (rc) return rc;
    }
    int rc = 0;
    if ((rc = put_dev(c, c->sk_st, t.st_ent)) || (rc = put_dev(c, c->sd_st, t.st_dir)) || (rc = put_dev(c, c->sr_st, t.st_rdir)) ||
        (rc = put_dev(c, c->sk_en, t.en_ent)) || (rc = put_dev(c, c->sd_en, t.en_dir)) || (rc = put_dev(c, c->tid_base, t.tid_base)) ||
        (rc = put_dev(c, c->key_dir, t.key_dir)) || (rc = put_dev(c, c->kb_base, t.kb_base)) || (rc = put_dev(c, c->hdr, t.hdr)) ||
        (rc = put_dev(c, c->anno_key, t.key)) || (rc = put_dev(c, c->anno_ex, t.ex))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));              // (t is a local)
    c->h_anno_key_raw.swap(t.key_raw);
    c->h_anno_key_pm.swap(t.key);
    c->seq = l2r_ctx::Stream();
    c->n_compact = t.n_compact; c->n_wide = t.n_wide; c->n_tid_dir = t.n_tid_dir; c->n_tid_key = t.n_tid_key;
    c->n_tx = a->n_tx; c->n_anno_exon = a->n_exon;
    c->have_win = false; c->ran = false; drop_graph(c);
    return 0;
}

int l2r_set_junctions(l2r_ctx *c, const l2r_junctions *s)
{
    if (!c) return fail(-1, "[l2r_set_junctions] null context");
    HIP_TRY(hipSetDevice(c->device));
    c->ran = false; drop_graph(c);
    if (!s || s->n == 0) { c->n_sj = 0; c->h_sj_key_raw.clear(); c->h_sj_key_pm.clear(); c->seq = l2r_ctx::Stream(); return 0; }
    if (s->n < 0 || s->n > 0x7ffffff0LL) return fail(-1, "[l2r_set_junctions] size out of range");
    const int64_t n = s->n;
    std::vector<int64_t> key((size_t)n);
    c->h_sj_key_raw.assign((size_t)n, 0);
    int64_t run = INT64_MIN;
    for (int64_t i = 0; i < n; ++i) {
        if (i && (s->tid[i] < s->tid[i - 1] || (s->tid[i] == s->tid[i - 1] && (s->don[i] < s->don[i - 1] ||
            (s->don[i] == s->don[i - 1] && s->acc[i] < s->acc[i - 1])))))
            return fail(-1, "[l2r_set_junctions] rows are not sorted by (tid, don, acc) at row %lld", (long long)i);
        // "row before read": tid smaller, or same tid and acc <= read start (update_gtf.c:613)
        const int64_t k = host_key(s->tid[i], s->acc[i]);
        c->h_sj_key_raw[(size_t)i] = k;
        if (k > run) run = k;
        key[(size_t)i] = run;
    }
    // directories: the junction cursor (prefix-max keys of (tid, acc)) and the donors (rows are sorted by (tid, don, acc)): SjDir
    std::vector<int32_t> cbase, dbase; std::vector<uint32_t> cdir, ddir;
    int32_t sj_ntid = 0;
    {
        for (int64_t i = 0; i < n; ++i) sj_ntid = std::max(sj_ntid, s->tid[i] + 1);
        std::vector<int64_t> mxa((size_t)sj_ntid, -1), mxd((size_t)sj_ntid, -1), dkey((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            dkey[(size_t)i] = host_key(s->tid[i], std::max(s->don[i], 0));
            if (s->tid[i] < 0) continue;
            mxa[(size_t)s->tid[i]] = std::max<int64_t>(mxa[(size_t)s->tid[i]], std::max(s->acc[i], 0));
            mxd[(size_t)s->tid[i]] = std::max<int64_t>(mxd[(size_t)s->tid[i]], std::max(s->don[i], 0));
        }
        if (build_key_dir(key.data(), n, sj_ntid, mxa, cbase, cdir) || build_key_dir(dkey.data(), n, sj_ntid, mxd, dbase, ddir))
            return fail(-1, "[l2r_set_junctions] junction directories too large");
    }
    if (c->sj_tid.ensure((size_t)n) || c->sj_don.ensure((size_t)n) || c->sj_acc.ensure((size_t)n) ||
        c->sj_uniq.ensure((size_t)n) || c->sj_multi.ensure((size_t)n) || c->sj_key.ensure((size_t)n) ||
        c->sj_cbase.ensure(cbase.size()) || c->sj_cdir.ensure(cdir.size()) || c->sj_dbase.ensure(dbase.size()) || c->sj_ddir.ensure(ddir.size()) || c->sj_row.ensure((size_t)n)) return -2;
    std::vector<int4> rows((size_t)n);
    for (int64_t i = 0; i < n; ++i) rows[(size_t)i] = make_int4(s->don[i], s->acc[i], s->uniq_c[i], s->multi_c[i]);
    HIP_TRY(hipMemcpyAsync(c->sj_row.p, rows.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->sj_cbase.p, cbase.data(), cbase.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->sj_cdir.p, cdir.data(), cdir.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->sj_dbase.p, dbase.data(), dbase.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->sj_ddir.p, ddir.data(), ddir.size() * 4, hipMemcpyHostToDevice, c->stream));
    c->sj_ntid = sj_ntid;
    const size_t b = (size_t)n * 4;
    HIP_TRY(hipMemcpyAsync(c->sj_tid.p, s->tid, b, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->sj_don.p, s->don, b, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->sj_acc.p, s->acc, b, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->sj_uniq.p, s->uniq_c, b, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->sj_multi.p, s->multi_c, b, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->sj_key.p, key.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n_sj = n;
    c->h_sj_key_pm = key;
    c->seq = l2r_ctx::Stream();
    return 0;
}

static int prepare_unsorted_windows(l2r_ctx *c);
static int finish_stream_sj_cursor(l2r_ctx *c);

int l2r_upload_reads(l2r_ctx *c, const l2r_reads *r)
{
    if (!c || !r) return fail(-1, "[l2r_upload_reads] null argument");
    if (r->n_reads < 0 || r->n_cigar < 0) return fail(-1, "[l2r_upload_reads] negative size");
    // exon offsets are 32 bit on the device: n_exon(read) <= n_cigar(read) + 1
    if ((uint64_t)r->n_cigar + (uint64_t)r->n_reads >= 0xfffffff0ULL)
        return fail(-1, "[l2r_upload_reads] shard too large for 32-bit exon offsets (%lld ops + %lld reads); split it", (long long)r->n_cigar, (long long)r->n_reads);
    HIP_TRY(hipSetDevice(c->device));
    const int64_t N = r->n_reads;
    if (N && (r->cig_off[0] != 0 || r->cig_off[N] != r->n_cigar)) return fail(-1, "[l2r_upload_reads] cig_off does not span n_cigar");
    bool sorted = true;
    for (int64_t i = 0; i < N; ++i) {
        if (r->tid[i] < 0) return fail(-1, "[l2r_upload_reads] record %lld has no reference (unmapped); the reference aborts on it (bam2gtf.c:100)", (long long)i);
        if (r->cig_off[i + 1] < r->cig_off[i]) return fail(-1, "[l2r_upload_reads] cig_off not monotone at %lld", (long long)i);
        if (i && (r->tid[i] < r->tid[i - 1] || (r->tid[i] == r->tid[i - 1] && r->pos[i] < r->pos[i - 1]))) sorted = false;
    }
    {   // is this upload the continuation of the previous one?  (see l2r_ctx::Stream)
        l2r_ctx::Stream &st = c->seq;
        const bool cont = st.valid && r->first_read_index > 0 && r->first_read_index == st.next;
        if (cont) { int rc = finish_stream_sj_cursor(c); if (rc) return rc; }      // (needs the previous upload's results: before they are overwritten)
        else st = l2r_ctx::Stream();
        if (N && st.sorted) {
            const int64_t first = ((int64_t)r->tid[0] << 32) | (uint32_t)r->pos[0];
            if (!sorted || first < st.last_key) st.sorted = false;            // from here on the cursors depend on the history
        }
        sorted = st.sorted;
        if (N) st.last_key = ((int64_t)r->tid[N - 1] << 32) | (uint32_t)r->pos[N - 1];
        st.anno_cur_start = st.anno_cur; st.sj_cur_start = st.sj_cur;
        st.next = r->first_read_index + N; st.valid = true;
        st.sj_pending = false;
    }
    c->sorted = sorted; c->have_win = false;
    if (!sorted || c->n_sj > 0) { c->h_tid.assign(r->tid, r->tid + N); c->h_pos.assign(r->pos, r->pos + N); }
    else { c->h_tid.clear(); c->h_pos.clear(); }

    // tile size: keep the expected exons of a tile inside the LDS staging area.
    // Estimate exons/read from a sample of the CIGARs (ops that can start an exon).
    int rpt = TILE_THREADS;
    c->many_exon_reads = false;
    if (N) {
        const int64_t sample = N < 4096 ? N : 4096;
        const int64_t step = N / sample;
        double cuts = 0;
        int64_t many = 0;                                  // sampled reads with more exons than a slab has rows
        for (int64_t s = 0; s < sample; ++s) {
            const int64_t i = s * step;
            int64_t mine = 0;
            for (int64_t k = r->cig_off[i]; k < r->cig_off[i + 1]; ++k) {
                const uint32_t op = r->cig[k] & 15u; const int len = (int)(r->cig[k] >> 4);
                mine += (op == 3u && len >= c->prm.min_intron) || (op == 2u && len > c->prm.max_delet);
            }
            cuts += (double)mine; many += mine + 1 > (int64_t)SLAB_ROWS;
        }
        // (k_walk_slab_long hands a read beyond SLAB_ROWS exons to the generic kernel, the classic kernels keep it on the mask path: an
        //  input where such reads are more than a rarity stays with them)
        c->many_exon_reads = many * 200 > sample;
        const double est = cuts / (double)sample + 1.0;
        // (long CIGARs on the slab pipeline: the probe kernels stage SLAB_POS_CAP positions per tile)
        const bool slab_long = c->want_pipeline > 0 && sorted && (double)r->n_cigar / (double)N > 32.0 && !getenv("L2R_NO_SLAB_LONG") && !c->many_exon_reads;
        while (rpt > 32 && est * rpt * 1.25 > (double)(slab_long ? SLAB_POS_CAP : LDS_EXON_CAP)) rpt >>= 1;
        // ... and keep the genomic span of a tile inside the staged bucket directory (DIR_CAP buckets of 512 bp):
        // sparse input (few reads per locus) makes 256 consecutive reads span many genes, and a tile that does not
        // fit goes to the generic kernel read by read (~30x the cost).  Sample windows of the sorted input, take for
        // every candidate size the share of windows that would not fit, and pick the cheapest size.
        // (The slab pipeline's tiles are cut by span one by one, below: a sparse stretch makes ITS tiles small, not every tile of the
        //  upload -- an annotation with a few isoform-rich loci and long sparse stretches got 128-read tiles throughout, twice the tiles.)
        const bool span_cut_tiles = c->want_pipeline > 0 && sorted && ((double)r->n_cigar / (double)N <= 32.0 || slab_long);
        if (sorted && N >= 2 * TILE_THREADS && !span_cut_tiles) {
            const int64_t n_win = std::min<int64_t>(N / TILE_THREADS, 384);
            const int64_t wstep = (N / TILE_THREADS) / n_win;
            const int64_t limit = (int64_t)(DIR_CAP - 8) << SITE_SHIFT;
            int64_t bad[4] = {0, 0, 0, 0};                 // sizes 256, 128, 64, 32
            for (int64_t w = 0; w < n_win; ++w) {
                const int64_t i0 = w * wstep * TILE_THREADS;
                int64_t hi = 0;
                int size_idx = 3, next_mark = 32;
                for (int64_t q = 0; q < TILE_THREADS && i0 + q < N; ++q) {
                    const int64_t i = i0 + q;
                    if (r->tid[i] != r->tid[i0]) break;
                    int64_t end = r->pos[i];
                    for (int64_t k = r->cig_off[i]; k < r->cig_off[i + 1]; ++k) if ((0x18du >> (r->cig[k] & 15u)) & 1u) end += r->cig[k] >> 4;
                    hi = std::max(hi, end - r->pos[i0]);
                    if (q + 1 == next_mark) {              // the first 32 / 64 / 128 / 256 reads of the window
                        if (hi > limit) { for (int z = 0; z <= size_idx; ++z) bad[z]++; break; }   // this size and every larger one
                        --size_idx; next_mark <<= 1;
                    }
                }
            }
            double best = 1e300; int best_rpt = rpt;
            for (int z = 0; z < 4; ++z) {
                const int cand = TILE_THREADS >> z;
                if (cand > rpt) continue;
                const double f = (double)bad[z] / (double)n_win;
                const double cost = (1.0 - f) * (z == 0 ? 1.0 : z == 1 ? 1.6 : z == 2 ? 2.6 : 4.5) + 30.0 * f;
                if (cost < best - 1e-9) { best = cost; best_rpt = cand; }
            }
            rpt = best_rpt;
        }
    }
    c->reads_per_tile = rpt;
    c->wide_cigar = N > 0 && (double)r->n_cigar / (double)N > 32.0;
    // Tiles: runs of up to rpt consecutive reads; for sorted input a tile also ends where the chromosome changes, so
    // that every read of a tile can use the tile's dictionary slices (unsorted input: plain runs, the reads that are
    // not on the chromosome of their tile's first read take the generic kernel).
    std::vector<uint32_t> tile_first;
    tile_first.reserve((size_t)(N / rpt + 64));
    // (slab pipeline: a tile's exons are staged by position in LDS on their way out, l2r_slab.hip.h SLAB_POS_CAP: a tile also ends
    //  where the exon bounds of its reads -- from the CIGAR lengths -- would exceed that, so no read of it is left outside)
    const bool slab_long_tiles = c->want_pipeline > 0 && sorted && c->wide_cigar && !getenv("L2R_NO_SLAB_LONG") && !c->many_exon_reads;      // (k_walk_slab_long: tiles of rpt reads, cut by span like the slab's)
    const bool slab_tiles = c->want_pipeline > 0 && sorted && !c->wide_cigar;
    uint64_t pos_sum = 0;
    for (int64_t i = 0, start = 0; i <= N; ++i) {
        if (i == N) { if (i > start) tile_first.push_back((uint32_t)start); break; }
        const uint64_t need = slab_tiles ? (uint64_t)slab_rows_of((uint32_t)std::min<int64_t>(r->cig_off[i + 1] - r->cig_off[i], 0x7ffffff0)) : 0u;
        // (tiles of sorted records also end where the reads would begin 2^17 bases apart: a slab tile's exons are kept relative to its
        //  first base, and any tile's dictionary slices cover 196 kb -- sparse stretches give small tiles instead of tiles for the
        //  generic kernel; the classic pipeline, whose tiles own 24 KB of hand-over buffer each, keeps at least 8 reads per tile)
        if (i - start == rpt || (sorted && r->tid[i] != r->tid[start]) || (i > start && pos_sum + need > (uint64_t)TILE_POS_CAP) ||
            (sorted && i > start && (int64_t)r->pos[i] - (int64_t)r->pos[start] >= (int64_t)SLAB_TILE_SPAN && (slab_tiles || slab_long_tiles || i - start >= 8))) { tile_first.push_back((uint32_t)start); start = i; pos_sum = 0; }
        pos_sum += need;
    }
    c->n_tiles = (int64_t)tile_first.size();
    tile_first.push_back((uint32_t)N);
    if (tile_first.size() < 2) tile_first.push_back((uint32_t)N);        // (an empty launch still runs one workgroup)
    c->n_tiles256 = (N + TILE_THREADS - 1) / TILE_THREADS;
    if (c->tile_first.ensure(tile_first.size())) return -2;
    HIP_TRY(hipMemcpyAsync(c->tile_first.p, tile_first.data(), tile_first.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));                              // (tile_first is a local)

    if (c->r_tid.ensure((size_t)N) || c->r_pos.ensure((size_t)N) || c->r_rev.ensure((size_t)N) ||
        c->cig_off.ensure((size_t)N + 1) || c->cig.ensure((size_t)r->n_cigar + 8)) return -2;     // + 8: the kernels read whole 16-byte vectors (pass A: two per lane)
    if (N) {
        HIP_TRY(hipMemcpyAsync(c->r_tid.p, r->tid, (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->r_pos.p, r->pos, (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->r_rev.p, r->rev, (size_t)N, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->cig_off.p, r->cig_off, (size_t)(N + 1) * 8, hipMemcpyHostToDevice, c->stream));
        if (r->n_cigar) HIP_TRY(hipMemcpyAsync(c->cig.p, r->cig, (size_t)r->n_cigar * 4, hipMemcpyHostToDevice, c->stream));
    }
    // work buffers.  n_exon(read) <= ops(read) + 1, so n_cigar + n_reads bounds the exon arrays; for long CIGARs (hundreds of
    // M/I/D ops per exon) that bound is 10-50 times too generous, so the ops that can end an exon at all (N, D) are
    // counted on the device (one pass over the words that were just uploaded; sizing only, nothing of it is kept).
    size_t exb = (size_t)r->n_cigar + (size_t)N;
    if (c->wide_cigar) {
        unsigned long long *d_cnt = reinterpret_cast<unsigned long long *>(c->totals.p ? c->totals.p : nullptr);
        if (!d_cnt) { if (c->totals.ensure(8)) return -2; d_cnt = reinterpret_cast<unsigned long long *>(c->totals.p); }
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, c->stream));
        hipLaunchKernelGGL(k_count_cut_ops, dim3(4096), dim3(TILE_THREADS), 0, c->stream, (const uint32_t *)c->cig.p, (int64_t)r->n_cigar, d_cnt);
        unsigned long long cuts = 0;
        HIP_TRY(hipMemcpyAsync(&cuts, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        exb = (size_t)cuts + (size_t)N;
    }
    if (c->j0.ensure((size_t)N) || c->local.ensure((size_t)N + 1) || c->ex_off.ensure((size_t)N) || c->info.ensure((size_t)N) || c->ref_tx.ensure((size_t)N) ||
        c->redo.ensure((size_t)N) || c->order.ensure((size_t)N + TILE_THREADS) || c->desc.ensure((size_t)c->n_tiles) || c->win_hdr.ensure((size_t)c->n_tiles * WIN_TX) ||
        c->tile_base.ensure((size_t)c->n_tiles + 1) || c->tile_acc.ensure((size_t)c->n_tiles + 1) || c->tile_acc_ex.ensure((size_t)c->n_tiles + 1) ||
        c->tile_acc_at.ensure((size_t)c->n_tiles + 2) || c->tile_acc_ex_at.ensure((size_t)c->n_tiles + 2) ||
        c->totals.ensure(8) || c->tile_chunk.ensure((size_t)c->n_tiles + 1) || c->tile_rchunk.ensure((size_t)c->n_tiles + 1) || c->ex_start.ensure(exb) || c->ex_end.ensure(exb) || c->ex_flag.ensure(exb) ||
        c->acc_rec.ensure((size_t)N) || c->acc_ex_off.ensure((size_t)N) ||
        c->acc_start.ensure(exb) || c->acc_end.ensure(exb) || c->acc_flag.ensure(exb) || (c->wide_cigar && c->walked.ensure((size_t)(c->n_tiles + 1) * LDS_EXON_CAP))) return -2;
    c->ex_cap = (int64_t)exb;
    c->slab_ok = false; c->slab = false;
    if (c->want_pipeline > 0 && sorted && (!c->wide_cigar || (!getenv("L2R_NO_SLAB_LONG") && !c->many_exon_reads))) {
        // the slab layout (l2r_slab.hip.h): per tile as many rows of 256 elements as its longest read can have exons (bound from
        // the CIGAR lengths); reads beyond SLAB_ROWS rows are outliers and get a run of the dense area
        const size_t T = (size_t)c->n_tiles;
        std::vector<uint32_t> sbase(T + 1, 0u);
        uint64_t total = 0, ovf = 0;
        for (size_t t = 0; t < T; ++t) {
            uint32_t m = 1;
            if (c->wide_cigar) {
                // long CIGARs (k_walk_slab_long): the CIGAR length says nothing about the exons -- every tile has SLAB_ROWS rows, and
                // the dense area has room for every exon of the shard (exb: reads + the operations that can cut)
                m = (uint32_t)SLAB_ROWS;
            } else
            for (uint32_t i = tile_first[t]; i < tile_first[t + 1]; ++i) {
                const uint64_t cc = (uint64_t)(r->cig_off[i + 1] - r->cig_off[i]);
                const uint64_t rw = (cc + 3u) >> 1;
                // (room in the dense area for EVERY read: besides the long CIGARs a read with an exon of 64 kb or more ends up
                //  there, which only the walk finds out)
                ovf += cc + 1;
                if (rw <= (uint64_t)SLAB_ROWS) m = std::max<uint32_t>(m, (uint32_t)rw);
            }
            sbase[t] = (uint32_t)total; total += (uint64_t)m * SLAB_STRIDE;
            if (total >= 0x7ffffff0ULL || ovf >= 0x7ffffff0ULL) break;
        }
        if (c->wide_cigar) ovf = (uint64_t)exb;
        if (total < 0x7ffffff0ULL && ovf < 0x7ffffff0ULL) {
            sbase[T] = (uint32_t)total;                     // (rows of tile t = (sbase[t + 1] - sbase[t]) / 256)
            c->slab_ok = true;
            if (c->tw64.ensure(T + 1) || c->wide_list.ensure(2 * (T + 1)) || c->chunk_list.ensure(2 * (T + 1)) || c->list_cnt.ensure(32) || c->tile_flags.ensure(T + 8) ||
                c->lb_tile.ensure(T + 64) || c->lb_blk.ensure(T / LB_BLK + 64) || c->lb_sup.ensure(2 * ((T >> LB_SUP_SHIFT) + 64)) || c->fb_list.ensure(T + 1) || c->tile_stat.ensure(T + 1) || c->sup_stat.ensure((T >> LB_SUP_SHIFT) + 2) ||
                (!c->wide_cigar && c->slot_rec.ensure((T + 1) * TILE_THREADS))) return -2;
            HIP_TRY(hipMemsetAsync(c->lb_sup.p, 0, 2 * ((T >> LB_SUP_SHIFT) + 64) * 8, c->stream)); c->lb_flip = 0;      // (two arrays taking turns; from then on each is cleared by the run in front of its own)      // (an isoform-rich annotation makes EVERY tile wide: 2.4 KB each)
            HIP_TRY(hipMemsetAsync(c->list_cnt.p, 0, 128, c->stream)); c->lc_flip = 0; c->prev_run_tile = false;
            if (c->tile_sbase.ensure(T + 1) || c->ovf_cursor.ensure(1) || c->tw.ensure(T + 1) || c->tile_total.ensure(T + 2) || c->tile_xbase.ensure(T + 2) || c->tile_span.ensure(12 * (T + 1)) ||
                c->s_pre.ensure((size_t)N + 1) || c->s_loc.ensure((size_t)N + 1) ||
                c->slab_row.ensure((size_t)total + 4) ||
                c->dense_start.ensure((size_t)ovf + 1) || c->dense_end.ensure((size_t)ovf + 1)) return -2;
            HIP_TRY(hipMemsetAsync(c->ovf_cursor.p, 0, 8, c->stream));
            HIP_TRY(hipMemcpyAsync(c->tile_sbase.p, sbase.data(), (T + 1) * 4, hipMemcpyHostToDevice, c->stream));
            // the tiles' records for k_walk_slab (TileRec: reads, slab, chromosome and first base of the tile in one place) and the
            // records' CIGAR offsets in 32 bits (the walk reads 4 bytes per record instead of 8 at a stride of 8)
            std::vector<TileRec> rec(T ? T : 1);
            for (size_t t = 0; t < T; ++t) {
                TileRec &q = rec[t];
                q.r0 = tile_first[t]; q.n_act = tile_first[t + 1] - tile_first[t]; q.sbase = sbase[t]; q.rows = (sbase[t + 1] - sbase[t]) >> 8;
                q.tid0 = q.n_act ? r->tid[q.r0] : 0; q.lo = (q.n_act ? r->pos[q.r0] : 0) + 1; q.pad[0] = q.pad[1] = 0u;
            }
            std::vector<uint32_t> off32((size_t)N + 1);
            for (int64_t i = 0; i <= N; ++i) off32[(size_t)i] = (uint32_t)r->cig_off[i];
            if (c->tile_rec.ensure(8 * (T + 1)) || c->cig_off32.ensure((size_t)N + 2) || c->s_pl.ensure((size_t)N + 1)) return -2;
            HIP_TRY(hipMemcpyAsync(c->tile_rec.p, rec.data(), rec.size() * sizeof(TileRec), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->cig_off32.p, off32.data(), off32.size() * 4, hipMemcpyHostToDevice, c->stream));
            // every tile's last base (the largest read end: CIGAR lengths only, no parameter has a say) into its record: the one-kernel
            // tile path makes the tiles' windows from it in front of the walk (l2r_tile.hip.h)
            // ... and an index of its CIGAR operations from which a run knows the tile's exon count unless a threshold is borderline in it
            c->h_tile_stat.assign(T, TileStat{0, INT32_MAX, 0, INT32_MAX});
            c->index_ms = 0.0f; c->have_index = false;
            if (T && !c->wide_cigar && c->want_pipeline >= 2 && !(c->one_shot_upload && !c->env_tile_anyway && !c->pipeline_forced)) {
                c->have_index = true;
                struct Ev { hipEvent_t a = nullptr, b = nullptr; ~Ev() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev;
                HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b));
                const unsigned gi = (unsigned)std::min<size_t>(T, 8192);
                if (r->cig_summary) {
                    // the reader's per-record summaries: the tiles' statistics and last bases on the host (no parameter has a say in them), the
                    // records' N operations as one 16-bit column for the kernel -- which then touches no CIGAR
                    std::vector<uint16_t> nn((size_t)N);
                    for (size_t t = 0; t < T; ++t) {
                        TileStat st{0, INT32_MAX, 0, INT32_MAX};
                        int64_t hi = INT32_MIN; uint32_t tot_x = 0u; bool many = false;
                        for (uint32_t i = tile_first[t]; i < tile_first[t + 1]; ++i) {
                            const uint32_t *q = r->cig_summary + 3 * (size_t)i;
                            const uint32_t n_n = q[1] & 0xffffu, mn = q[1] >> 16, md = q[2] & 0xffffu, ms = q[2] >> 16;
                            nn[i] = (uint16_t)n_n;
                            st.n_ops_n += (int32_t)n_n; st.min_n = std::min(st.min_n, (int32_t)mn); st.min_seg = std::min(st.min_seg, (int32_t)ms);
                            st.max_d = std::max(st.max_d, md == 0xffffu ? INT32_MAX : (int32_t)md);      // (65535: that long or longer)
                            hi = std::max<int64_t>(hi, (int64_t)r->pos[i] + (int64_t)q[0]);
                            tot_x += n_n + 1u; many = many || n_n + 1u >= 255u;
                        }
                        // (a read of 255 exons or more, places a slot record cannot say: never an exact tile -- as k_tile_index<false> rules)
                        if (many || tot_x >= SLOT_LOC_LIMIT) st.min_seg = INT32_MIN;
                        c->h_tile_stat[t] = st;
                        rec[t].pad[0] = (uint32_t)std::min<int64_t>(std::max<int64_t>(hi, INT32_MIN), INT32_MAX);
                    }
                    if (c->sum_nn.ensure((size_t)N + 1)) return -2;
                    HIP_TRY(hipMemcpyAsync(c->sum_nn.p, nn.data(), (size_t)N * 2, hipMemcpyHostToDevice, c->stream));
                    HIP_TRY(hipMemcpyAsync(c->tile_rec.p, rec.data(), rec.size() * sizeof(TileRec), hipMemcpyHostToDevice, c->stream));
                    HIP_TRY(hipMemcpyAsync(c->tile_stat.p, c->h_tile_stat.data(), T * sizeof(TileStat), hipMemcpyHostToDevice, c->stream));
                    HIP_TRY(hipEventRecord(ev.a, c->stream));
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_index<true>), dim3(gi), dim3(TILE_THREADS), 0, c->stream, (TileRec *)c->tile_rec.p, c->tile_stat.p, c->slot_rec.p, (uint32_t)T,
                                       (const uint32_t *)c->cig_off32.p, (const int32_t *)c->r_pos.p, (const uint8_t *)c->r_rev.p, (const uint32_t *)c->cig.p, (const uint16_t *)c->sum_nn.p);
                    HIP_TRY(hipEventRecord(ev.b, c->stream));
                    HIP_TRY(hipStreamSynchronize(c->stream));       // (nn, rec)
                } else {
                    HIP_TRY(hipEventRecord(ev.a, c->stream));
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_index<false>), dim3(gi), dim3(TILE_THREADS), 0, c->stream, (TileRec *)c->tile_rec.p, c->tile_stat.p, c->slot_rec.p, (uint32_t)T,
                                       (const uint32_t *)c->cig_off32.p, (const int32_t *)c->r_pos.p, (const uint8_t *)c->r_rev.p, (const uint32_t *)c->cig.p, (const uint16_t *)nullptr);
                    HIP_TRY(hipEventRecord(ev.b, c->stream));
                    HIP_TRY(hipMemcpyAsync(c->h_tile_stat.data(), c->tile_stat.p, T * sizeof(TileStat), hipMemcpyDeviceToHost, c->stream));
                }
                HIP_TRY(hipEventSynchronize(ev.b));
                HIP_TRY(hipEventElapsedTime(&c->index_ms, ev.a, ev.b));
            }
            HIP_TRY(hipStreamSynchronize(c->stream));       // (locals)
            {   // the index once more per super-block (l2r_slab.hip.h SlabArgs::sup_stat)
                std::vector<TileStat> sup((T >> LB_SUP_SHIFT) + 1, TileStat{0, INT32_MAX, 0, INT32_MAX});
                for (size_t t = 0; t < T; ++t) {
                    TileStat &q = sup[t >> LB_SUP_SHIFT]; const TileStat &st = c->h_tile_stat[t];
                    q.n_ops_n += st.n_ops_n + (int32_t)rec[t].n_act; q.min_n = std::min(q.min_n, st.min_n); q.max_d = std::max(q.max_d, st.max_d); q.min_seg = std::min(q.min_seg, st.min_seg);
                }
                HIP_TRY(hipMemcpyAsync(c->sup_stat.p, sup.data(), sup.size() * sizeof(TileStat), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (getenv("L2R_STAMPS") && !c->stamps.p) {
        if (c->stamps.ensure(1024 * 8 + 16)) return -2;
        HIP_TRY(hipMemsetAsync(c->stamps.p, 0, (1024 * 8 + 16) * 8, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->n_reads = N; c->n_cigar = r->n_cigar; c->first_read = r->first_read_index;
    c->ran = false; c->totals_valid = false; drop_graph(c);
    if (sorted) {
        // the annotation cursor after a sorted prefix is the prefix function of its last record
        if (N) {
            const int64_t q = host_key(r->tid[N - 1], r->pos[N - 1] + 1);
            const int64_t at = std::upper_bound(c->h_anno_key_pm.begin(), c->h_anno_key_pm.end(), q) - c->h_anno_key_pm.begin();
            c->seq.anno_cur = std::max(c->seq.anno_cur, at);
        }
        c->seq.sj_pending = c->n_sj > 0;
    } else {
        int rc = prepare_unsorted_windows(c);            // replays the cursor now, so that the next upload can continue it
        if (rc) return rc;
    }
    return 0;
}

// Brings Stream::sj_cur up to date after an upload whose junction cursor ran on the device (sorted so far): the
// reference's last_sj_i only moves for reads that reach check_short_sj (src/update_gtf.c:947,613-614), so it is the
// prefix function of the LAST such read -- known only once that upload has been classified.
static int finish_stream_sj_cursor(l2r_ctx *c)
{
    l2r_ctx::Stream &st = c->seq;
    if (!st.sj_pending || !c->ran || c->n_sj == 0 || c->n_reads == 0) { st.sj_pending = false; return 0; }
    const int64_t N = c->n_reads;
    std::vector<uint32_t> info((size_t)N);
    HIP_TRY(hipMemcpyAsync(info.data(), c->info.p, (size_t)N * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int64_t i = N - 1; i >= 0; --i) {
        if ((info[(size_t)i] & (I_FULL | I_KNOWN | I_KSITE)) != (I_FULL | I_KSITE)) continue;
        const int64_t q = host_key(c->h_tid[(size_t)i], c->h_pos[(size_t)i] + 1);
        const int64_t at = std::upper_bound(c->h_sj_key_pm.begin(), c->h_sj_key_pm.end(), q) - c->h_sj_key_pm.begin();
        st.sj_cur = std::max(st.sj_cur, at);
        break;
    }
    st.sj_pending = false;
    return 0;
}

// Unsorted input: the annotation cursor is history dependent (update_gtf.c:801-802).  Replay it on the
// host (O(N + T)); the per-read work still runs on the GPU.
static int prepare_unsorted_windows(l2r_ctx *c)
{
    if (c->sorted || c->have_win) return 0;
    const int64_t N = c->n_reads, T = c->n_tx;
    std::vector<int32_t> w((size_t)N);
    int64_t cur = c->seq.anno_cur_start;              // 0 unless this upload continues an earlier one
    for (int64_t i = 0; i < N; ++i) {
        const int64_t q = host_key(c->h_tid[(size_t)i], c->h_pos[(size_t)i] + 1);
        while (cur < T && c->h_anno_key_raw[(size_t)cur] <= q) ++cur;
        w[(size_t)i] = (int32_t)cur;
    }
    c->seq.anno_cur = cur;
    if (c->win_start.ensure((size_t)N)) return -2;
    if (N) HIP_TRY(hipMemcpyAsync(c->win_start.p, w.data(), (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_win = true;
    return 0;
}

// Unsorted input with a junction table: the junction cursor only moves for reads that reach the check
// (update_gtf.c:947), so it needs the classification first.
static int prepare_unsorted_sj_cursor(l2r_ctx *c)
{
    const int64_t N = c->n_reads, S = c->n_sj;
    std::vector<uint32_t> info((size_t)N);
    if (N) HIP_TRY(hipMemcpyAsync(info.data(), c->info.p, (size_t)N * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<int32_t> cur_v((size_t)N, 0);
    int64_t cur = c->seq.sj_cur_start;                // 0 unless this upload continues an earlier one
    for (int64_t i = 0; i < N; ++i) {
        if ((info[(size_t)i] & (I_FULL | I_KNOWN | I_KSITE)) != (I_FULL | I_KSITE)) continue;
        const int64_t q = host_key(c->h_tid[(size_t)i], c->h_pos[(size_t)i] + 1);
        while (cur < S && c->h_sj_key_raw[(size_t)cur] <= q) ++cur;
        cur_v[(size_t)i] = (int32_t)cur;
    }
    c->seq.sj_cur = cur;
    if (c->sj_cursor.ensure((size_t)N)) return -2;
    if (N) HIP_TRY(hipMemcpyAsync(c->sj_cursor.p, cur_v.data(), (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

enum { ST_PASS_A = 0, ST_SCAN1, ST_FAST, ST_GENERIC, ST_SJ, ST_SCAN2, ST_GATHER, ST_N };

#define launch_fast_level(L, fa, grid, s) if (c->wide_cigar) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_classify_fast<L, true>), dim3(grid), dim3(TILE_THREADS), 0, s, fa, c->n_tiles, (const TileDesc *)c->desc.p, (const uint32_t *)c->tile_base.p, (const int64_t *)c->cig_off.p, (const uint8_t *)c->order.p, (const uint32_t *)c->tile_first.p); \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_classify_fast<L, false>), dim3(grid), dim3(TILE_THREADS), 0, s, fa, c->n_tiles, (const TileDesc *)c->desc.p, (const uint32_t *)c->tile_base.p, (const int64_t *)c->cig_off.p, (const uint8_t *)c->order.p, (const uint32_t *)c->tile_first.p)

static int launch_all(l2r_ctx *c, hipEvent_t *ev /* ST_N + 1 events or null */)
{
    const DevParams p = dev_params(c);
    const int64_t N = c->n_reads;
    hipStream_t s = c->stream;
    const unsigned gt = (unsigned)(c->n_tiles ? c->n_tiles : 1), g256 = (unsigned)(c->n_tiles256 ? c->n_tiles256 : 1);
    const int32_t *j0 = c->sorted ? (const int32_t *)c->j0.p : (const int32_t *)c->win_start.p;
    // L2R_CHECK=1 (diagnostics): wait for the device at every stage boundary, so that a kernel fault is reported with the stage it
    // happened in instead of at the next synchronisation of the caller
    static const char *const stage_name[ST_N + 1] = {"(start)", "pass A / order", "scan / walk", "classification", "generic", "junction check", "accepted scan", "accepted gather"};
#define MARK(i) do { if (ev) HIP_TRY(hipEventRecord(ev[i], s)); \
        if (c->check_stages) { const hipError_t e_ = hipStreamSynchronize(s); \
            if (e_ != hipSuccess) return fail(-2, "[launch_all] device error behind stage \"%s\": %s", stage_name[(i) <= ST_N ? (i) : 0], hipGetErrorString(e_)); } } while (0)
    MARK(ST_PASS_A);
    const CursorDir cd{c->anno_key.p, c->key_dir.p, c->kb_base.p, c->n_tid_key, (int32_t)c->n_tx};
    const SiteTabs tabs{{c->sk_st.p, c->sd_st.p, c->sr_st.p}, {c->sk_en.p, c->sd_en.p, nullptr}, c->tid_base.p, c->n_tid_dir};
    FastArgs fa;
    fa.n_reads = N; fa.r_tid = c->r_tid.p; fa.r_pos = c->r_pos.p; fa.r_rev = c->r_rev.p; fa.cig_off = c->cig_off.p; fa.cig = c->cig.p;
    fa.walked = c->walked.p; fa.local = c->local.p; fa.order = c->order.p; fa.tile_base = c->tile_base.p; fa.j0 = j0; fa.desc = c->desc.p; fa.win_hdr = c->win_hdr.p;
    fa.hdr = c->hdr.p; fa.st = tabs.st; fa.en = tabs.en;
    fa.ex_off = c->ex_off.p; fa.ex_start = c->ex_start.p; fa.ex_end = c->ex_end.p; fa.ex_flag = c->ex_flag.p; fa.info = c->info.p; fa.ref_tx = c->ref_tx.p;
    fa.tile_acc = c->tile_acc.p; fa.tile_acc_ex = c->tile_acc_ex.p; fa.redo_count = c->totals.p + 3; fa.redo = c->redo.p;
    fa.tile_chunk = c->tile_chunk.p; fa.tile_rchunk = c->tile_rchunk.p; fa.chunk_cursor = (unsigned long long *)(c->totals.p + 4);
    fa.acc_start = c->acc_start.p; fa.acc_end = c->acc_end.p; fa.acc_flag = c->acc_flag.p; fa.acc_rec = (AccRec *)c->acc_rec.p; fa.acc_ex_off = c->acc_ex_off.p; fa.first_read = c->first_read;
    fa.stamps = c->stamps.p; fa.p = p;
    // persistent grid: a few workgroups per CU walk over the tiles (l2r_kernels.hip.h)
    unsigned gp = (unsigned)std::min<int64_t>(c->n_tiles ? c->n_tiles : 1, (int64_t)c->n_cu * c->wg_per_cu);
    if (c->fast_grid > 0) gp = (unsigned)std::min<int64_t>(gp, c->fast_grid);           // L2R_FAST_GRID: tests force many tiles per workgroup
    // the slab pipeline wants the straight-line walk: thresholds that fit a CIGAR word (else: the classic kernels)
    c->slab = c->slab_ok && p.min_intron >= 0 && p.min_intron < (1 << 28) && p.max_delet >= -1 && p.max_delet < (1 << 28) - 1 &&
              (!c->wide_cigar || p.min_exon >= 1);           // (k_walk_slab_long has no -e < 1 form: the classic kernels take that)
    // the one-kernel tile path: short CIGARs whose exon counts the CIGAR lengths bound (-e >= 1)
    c->tile = c->slab && c->want_pipeline >= 2 && !c->wide_cigar && p.min_exon >= 1 && (c->have_index || c->n_tiles == 0) && !c->tile_starved;
    if (c->tile) {
        // A tile whose exon count the first kernel cannot derive from the upload's index (a threshold is borderline in it) publishes it
        // from k_tile, and every later tile's write-out waits for it: fine for a few, a convoy for many (measured: 3 x the kernel when
        // every tile does) -- such a run takes the two-kernel path.
        if (c->inexact_tiles < 0) {     // (depends on the upload and the parameters alone: counted once, forgotten with them -- drop_graph)
            int64_t inexact = 0;
            for (const TileStat &st : c->h_tile_stat) inexact += tile_exact(st, p.min_exon, p.min_intron, p.max_delet) ? 0 : 1;
            c->inexact_tiles = inexact;
        }
        if (c->inexact_tiles * 50 > c->n_tiles + 800 && !c->env_tile_anyway) c->tile = false;
        // (measured: cfg3_iso40 -- every tile wide or chunked -- 1.34 ms on this path against 1.26 on the slab pipeline)
        if (c->lists_known && c->lists_heavy && !c->env_tile_anyway) c->tile = false;
    }
    if (c->slab) {
        // ---- two light kernels at full occupancy: the walk (exons into the tiles' slabs, read-order places, descriptors), a scan of
        //      the tiles' exon counts, then the probes, which write the read-order results (l2r_slab.hip.h).  Every launch does all
        //      of it: nothing is kept from an earlier run of the same upload.
        bool skip_lists = false;
        SlabArgs sa;
        sa.g.f = fa; sa.g.cd = cd; sa.g.tid_base = c->tid_base.p; sa.g.n_tid_dir = c->n_tid_dir; sa.g.tile_total = c->tile_total.p;
        sa.tile_sbase = c->tile_sbase.p; sa.slab_row = c->slab_row.p;
        sa.dense_start = c->dense_start.p; sa.dense_end = c->dense_end.p; sa.ovf_cursor = c->ovf_cursor.p;
        sa.pl = c->s_pl.p; sa.pre_x = c->s_pre.p; sa.loc_x = c->s_loc.p; sa.cig_off32 = c->cig_off32.p; sa.tw = c->tw.p; sa.span = (TileSpan *)c->tile_span.p;
        sa.n_tiles = (uint32_t)c->n_tiles;
        const unsigned gx = 8u * (unsigned)std::max<int64_t>((c->n_tiles + 7) / 8, 1);      // (l2r_slab.hip.h xcd_tile; an empty upload still launches)
        sa.tw64 = (c->ablate & 4) ? nullptr : c->tw64.p;
        sa.chunk_on = (c->ablate & 32) ? 0u : 1u;          // (L2R_ABLATE bit 2: no 64-member windows, bit 5: no chunked windows)
        sa.wide_list = c->wide_list.p; sa.chunk_list = c->chunk_list.p; sa.list_cnt = c->list_cnt.p; sa.list_cnt_next = nullptr; sa.tile_flags = c->tile_flags.p;
        {   const size_t sup_words = (size_t)(c->n_tiles >> LB_SUP_SHIFT) + 64;
            sa.lb_sup = c->lb_sup.p ? c->lb_sup.p + (c->lb_flip ? sup_words : 0) : nullptr;
            sa.lb_sup_next = c->lb_sup.p ? c->lb_sup.p + (c->lb_flip ? 0 : sup_words) : nullptr;
            sa.n_sup = (uint32_t)(c->n_tiles >> LB_SUP_SHIFT) + 1u; }
        sa.lb_tile = c->lb_tile.p; sa.lb_blk = c->lb_blk.p; sa.lb_err = c->totals.p + 6; sa.fb_list = c->fb_list.p; sa.exon_total = c->totals.p + 0; sa.tile_stat = c->tile_stat.p; sa.sup_stat = c->sup_stat.p;
        sa.sj = SjDir{CursorDir{c->sj_key.p, c->sj_cdir.p, c->sj_cbase.p, c->sj_ntid, (int32_t)c->n_sj}, c->sj_ddir.p, c->sj_dbase.p, c->sj_ntid, c->sj_row.p};
        sa.has_wide_keys = c->n_wide > 0 ? 1u : 0u;
        sa.wide_direct_on = (c->tile && c->wide_direct && c->tw64.p && !(c->ablate & 4)) ? 1u : 0u;
        sa.chunk_direct_on = (c->tile && c->chunk_direct && sa.chunk_on) ? 1u : 0u;
        // (with the accepted list wanted and no junction table to decide later, the tiles leave their accepted chunks themselves)
        const bool probe_acc = (c->want & L2R_WANT_ACCEPTED) && (c->n_sj == 0 || c->tile);      // (k_tile decides acceptance itself, junction table or not)
#define launch_probe_k(L, A, D, LIST, G) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_probe_slab<L, A, D, LIST>), dim3(G), dim3(TILE_THREADS), 0, s, sa, (const TileSpan *)c->tile_span.p, \
            (const TileWin *)c->tw.p, (const uint32_t *)c->tile_xbase.p, (const uint32_t *)c->fb_list.p)
        // (the tiles k_tile left in slab form leave no accepted chunks themselves: they stay k_gather_accepted's -- half the instantiations)
#define launch_probe_level(L, LIST, G) do { if (p.ss_dis > 0) { if (probe_acc && !LIST) launch_probe_k(L, !LIST, true, LIST, G); else launch_probe_k(L, false, true, LIST, G); } \
                                   else { if (probe_acc && !LIST) launch_probe_k(L, !LIST, false, LIST, G); else launch_probe_k(L, false, false, LIST, G); } } while (0)
#define launch_probe(LIST, G) do { switch (p.full_level) { \
        case 1: launch_probe_level(1, LIST, G); break; case 2: launch_probe_level(2, LIST, G); break; case 3: launch_probe_level(3, LIST, G); break; \
        case 4: launch_probe_level(4, LIST, G); break; case 5: launch_probe_level(5, LIST, G); break; default: launch_probe_level(0, LIST, G); break; } } while (0)
        if (c->tile) {
            if (!c->prev_run_tile) { HIP_TRY(hipMemsetAsync(c->list_cnt.p, 0, 128, s)); c->lc_flip = 0; }
            sa.list_cnt = c->list_cnt.p + 16 * (c->lc_flip & 1u); sa.list_cnt_next = c->list_cnt.p + 16 * ((c->lc_flip & 1u) ^ 1u);
            // ---- ONE kernel per tile (l2r_tile.hip.h): the descriptors first (spans from the upload), then walk + probes + write-out in
            //      one workgroup; k_probe_slab behind it for the few tiles that kept the slab form (none on most inputs)
            const DescribeScan job{c->tile_total.p, c->tile_xbase.p, c->totals.p + 0, c->n_tiles};
            const unsigned gd = (unsigned)std::max<int64_t>((c->n_tiles + DESCRIBE_TILES - 1) / DESCRIBE_TILES, 1);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_describe_scan<true>), dim3(gd), dim3(TILE_THREADS), 0, s, sa, job, 0u, (const TileRec *)c->tile_rec.p);
            MARK(ST_SCAN1);
            // (the three list-driven kernels behind k_tile: not launched once a completed run of the same inputs and parameters has shown
            //  their lists empty -- what ends up on them does not depend on anything else)
            skip_lists = c->lists_known && c->lists_empty && !c->env_launch_all;
            const unsigned gl = (unsigned)std::min<int64_t>(c->n_tiles ? c->n_tiles : 1, (int64_t)c->n_cu * 2);
            // (the WIDE instance beside the plain one, on a stream of its own: see l2r_ctx::side; with per-stage events or L2R_CHECK one behind the other)
            const bool wide_launch = !skip_lists && sa.wide_direct_on;
            const bool beside = c->side_on && !ev && !c->check_stages;
            hipStream_t sw = s;
            if (wide_launch && beside) { sw = c->side[0]; HIP_TRY(hipEventRecord(c->ev_fork, s)); HIP_TRY(hipStreamWaitEvent(sw, c->ev_fork, 0)); }
            if (wide_launch) {
                // the exact 64-bit-mask tiles straight from their CIGARs: k_tile's WIDE instance, a workgroup per entry of wide_list (the
                // list's length is known to the host once a run has completed: until then a grid for every tile, most of which leave at once)
                const unsigned gwd = c->lists_known ? std::max(c->n_wide_tiles, 1u) : (unsigned)std::max<int64_t>(c->n_tiles, 1);
#define launch_tw_k(L, D) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile<L, false, D, true>), dim3(gwd), dim3(TILE_THREADS), 0, sw, sa, (const TileRec *)c->tile_rec.p, (const TileWin *)c->tw.p, (const TileStat *)c->tile_stat.p, (const SlotRec *)c->slot_rec.p, c->tile_xbase.p)
#define launch_tw_level(L) do { if (p.ss_dis > 0) launch_tw_k(L, true); else launch_tw_k(L, false); } while (0)
                switch (p.full_level) {
                case 1: launch_tw_level(1); break;
                case 2: launch_tw_level(2); break;
                case 3: launch_tw_level(3); break;
                case 4: launch_tw_level(4); break;
                case 5: launch_tw_level(5); break;
                default: launch_tw_level(0); break;
                }
#undef launch_tw_level
#undef launch_tw_k
            }
            if (wide_launch && beside) HIP_TRY(hipEventRecord(c->ev_join[0], sw));
            // ... and the exact tiles of the chunked kernel: k_tile_chunk (l2r_tchunk.hip.h), a workgroup per entry of chunk_list
            const bool chunk_launch = !skip_lists && sa.chunk_direct_on && !(c->lists_known && c->n_chunk_tiles == 0u);
            hipStream_t sc = s;
            if (chunk_launch && beside) { sc = c->side[1]; if (!wide_launch) HIP_TRY(hipEventRecord(c->ev_fork, s)); HIP_TRY(hipStreamWaitEvent(sc, c->ev_fork, 0)); }
            auto launch_tchunk = [&](hipStream_t sc, unsigned gcd, uint32_t late) {
#define launch_tc_level(L) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_chunk<L>), dim3(gcd), dim3(TILE_THREADS), 0, sc, sa, (const TileRec *)c->tile_rec.p, (const TileWin *)c->tw.p, (const TileStat *)c->tile_stat.p, (const SlotRec *)c->slot_rec.p, c->tile_xbase.p, late)
                switch (p.full_level) {
                case 1: launch_tc_level(1); break;
                case 2: launch_tc_level(2); break;
                case 3: launch_tc_level(3); break;
                case 4: launch_tc_level(4); break;
                case 5: launch_tc_level(5); break;
                default: launch_tc_level(0); break;
                }
#undef launch_tc_level
            };
            const unsigned gcd = c->lists_known ? std::max(c->n_chunk_tiles, 1u) : (unsigned)std::max<int64_t>(c->n_tiles, 1);
            if (chunk_launch && beside) { launch_tchunk(sc, gcd, 0u); HIP_TRY(hipEventRecord(c->ev_join[1], sc)); }
            const unsigned gf = fused_grid(c->n_tiles);
#define launch_tile_k(L, A, D) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile<L, A, D>), dim3(gf), dim3(TILE_THREADS), 0, s, sa, (const TileRec *)c->tile_rec.p, (const TileWin *)c->tw.p, (const TileStat *)c->tile_stat.p, (const SlotRec *)c->slot_rec.p, c->tile_xbase.p)
#define launch_tile_level(L) do { if (p.ss_dis > 0) { if (probe_acc) launch_tile_k(L, true, true); else launch_tile_k(L, false, true); } \
                                  else { if (probe_acc) launch_tile_k(L, true, false); else launch_tile_k(L, false, false); } } while (0)
            switch (p.full_level) {
            case 1: launch_tile_level(1); break;
            case 2: launch_tile_level(2); break;
            case 3: launch_tile_level(3); break;
            case 4: launch_tile_level(4); break;
            case 5: launch_tile_level(5); break;
            default: launch_tile_level(0); break;
            }
#undef launch_tile_level
#undef launch_tile_k
            MARK(ST_FAST);
            if (chunk_launch && !beside) launch_tchunk(s, gcd, 0u);
            if (wide_launch && beside) HIP_TRY(hipStreamWaitEvent(s, c->ev_join[0], 0));
            if (chunk_launch && beside) HIP_TRY(hipStreamWaitEvent(s, c->ev_join[1], 0));
            if (!skip_lists && !(c->lists_known && c->fb_empty && !c->env_launch_all)) launch_probe(true, gl);
        } else {
        if (c->wide_cigar)
            hipLaunchKernelGGL(k_walk_slab_long, dim3(gx), dim3(TILE_THREADS), pass_a_dynamic_lds(c->reads_per_tile), s, sa, (const TileRec *)c->tile_rec.p);
        else if (p.min_exon >= 1)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_walk_slab<false>), dim3(gx), dim3(TILE_THREADS), 0, s, sa, (const TileRec *)c->tile_rec.p);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_walk_slab<true>), dim3(gx), dim3(TILE_THREADS), 0, s, sa, (const TileRec *)c->tile_rec.p);
        MARK(ST_SCAN1);
        {   // the tiles' exon counts -> their first slots in the read-order result arrays (tile_xbase; the sum = the exon count): the
            // first workgroups of the launch, a segment each; the tiles' descriptors and windows, sixteen lanes per tile, and the lists
            // of the 64-bit-mask and the chunked kernel: the workgroups behind them (l2r_slab.hip.h)
            const DescribeScan job{c->tile_total.p, c->tile_xbase.p, c->totals.p + 0, c->n_tiles};
            unsigned n_scan = (unsigned)std::max<int64_t>((c->n_tiles + DESCRIBE_SEG - 1) / DESCRIBE_SEG, 1);
            if (c->n_tiles > c->seg_max) {
                // (very large shards: one workgroup scans, in a launch of its own)
                HIP_TRY(hipMemcpyAsync(c->tile_xbase.p, c->tile_total.p, (size_t)c->n_tiles * 4, hipMemcpyDeviceToDevice, s));
                ScanJobs jobs = {}; jobs.job[0] = ScanJob{c->tile_xbase.p, c->n_tiles, c->totals.p + 0}; jobs.job[1] = jobs.job[0];
                hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, s, jobs);
                n_scan = 0;
            }
            const unsigned gd = n_scan + (unsigned)std::max<int64_t>((c->n_tiles + DESCRIBE_TILES - 1) / DESCRIBE_TILES, 1);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_describe_scan<false>), dim3(gd), dim3(TILE_THREADS), 0, s, sa, job, (uint32_t)n_scan, (const TileRec *)nullptr);
        }
        MARK(ST_FAST);
        launch_probe(false, gx);
        }
#undef launch_probe
#undef launch_probe_level
#undef launch_probe_k
        if (!skip_lists && !(c->tile && c->lists_known && c->wide_rest_empty && !c->env_launch_all))
        {   // the tiles with 33 .. 63 window members (none on most inputs: the grid finds an empty list and leaves)
            const WideArgs wa{c->tw64.p};
            const unsigned gw = (unsigned)std::min<int64_t>(c->n_tiles ? c->n_tiles : 1, (int64_t)c->n_cu * 5);
#define launch_wide_level(L) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_probe_slab_wide<L>), dim3(gw), dim3(TILE_THREADS), 0, s, sa, wa, (const uint32_t *)c->tile_first.p, \
                (const int32_t *)c->r_pos.p, (const uint32_t *)c->tile_sbase.p, (const TileWin *)c->tw.p, (const uint32_t *)c->tile_xbase.p, (const TileStat *)(c->tile ? c->tile_stat.p : nullptr))
            switch (p.full_level) {
            case 1: launch_wide_level(1); break;
            case 2: launch_wide_level(2); break;
            case 3: launch_wide_level(3); break;
            case 4: launch_wide_level(4); break;
            case 5: launch_wide_level(5); break;
            default: launch_wide_level(0); break;
            }
#undef launch_wide_level
        }
        if (c->tile && sa.chunk_direct_on && !skip_lists && !(c->lists_known && c->n_late_tiles == 0u && !c->env_launch_all)) {
            // the tiles a one-window kernel handed on late (a key in several entries): k_tile_chunk once more, over that list
            const unsigned gl2 = c->lists_known ? std::max(c->n_late_tiles, 1u) : (unsigned)std::max<int64_t>(c->n_tiles, 1);      // (every entry needs its workgroup: k_probe_slab_chunked skips what this launch takes)
            auto launch_late = [&]() {
#define launch_tcl_level(L) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_chunk<L>), dim3(gl2), dim3(TILE_THREADS), 0, s, sa, (const TileRec *)c->tile_rec.p, (const TileWin *)c->tw.p, (const TileStat *)c->tile_stat.p, (const SlotRec *)c->slot_rec.p, c->tile_xbase.p, 1u)
                switch (p.full_level) {
                case 1: launch_tcl_level(1); break;
                case 2: launch_tcl_level(2); break;
                case 3: launch_tcl_level(3); break;
                case 4: launch_tcl_level(4); break;
                case 5: launch_tcl_level(5); break;
                default: launch_tcl_level(0); break;
                }
#undef launch_tcl_level
            };
            launch_late();
        }
        if (sa.chunk_on && !skip_lists && !(c->tile && c->lists_known && c->chunk_rest_empty && !c->env_launch_all)) {   // the tiles without a window record, or with a dictionary key in several entries (none on most inputs)
            const unsigned gc = (unsigned)std::min<int64_t>(c->n_tiles ? c->n_tiles : 1, (int64_t)c->n_cu * 4);
#define launch_chunk_level(L) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_probe_slab_chunked<L>), dim3(gc), dim3(TILE_THREADS), 0, s, sa, (const uint32_t *)c->tile_first.p, \
                (const int32_t *)c->r_pos.p, (const uint32_t *)c->tile_sbase.p, (const TileWin *)c->tw.p, (const uint32_t *)c->tile_xbase.p)
            switch (p.full_level) {
            case 1: launch_chunk_level(1); break;
            case 2: launch_chunk_level(2); break;
            case 3: launch_chunk_level(3); break;
            case 4: launch_chunk_level(4); break;
            case 5: launch_chunk_level(5); break;
            default: launch_chunk_level(0); break;
            }
#undef launch_chunk_level
        }
        // (accepted list: k_describe_scan marks every tile CHUNK_DEFERRED, k_probe_slab<., true> takes that back for the tiles whose
        //  chunk it has written itself; k_count_accepted / k_gather_accepted place the rest)
    } else {
    // sorted input: the cursor value of every read is computed on the device; unsorted input: it was replayed on the host
    if (c->wide_cigar)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass_a<true>), dim3(gt), dim3(TILE_THREADS), pass_a_dynamic_lds(c->reads_per_tile), s, N, c->r_tid.p, c->r_pos.p, c->cig_off.p, c->cig.p, cd, tabs, p,
                           (c->sorted ? (const int32_t *)nullptr : (const int32_t *)c->win_start.p), c->j0.p, c->local.p, c->order.p, c->tile_base.p, c->desc.p,
                           c->totals.p + 3, (const TxHdr *)c->hdr.p, c->win_hdr.p, (const uint32_t *)c->tile_first.p, c->walked.p);
    else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass_a<false>), dim3(gt), dim3(TILE_THREADS), 0, s, N, c->r_tid.p, c->r_pos.p, c->cig_off.p, c->cig.p, cd, tabs, p,
                       (c->sorted ? (const int32_t *)nullptr : (const int32_t *)c->win_start.p), c->j0.p, c->local.p, c->order.p, c->tile_base.p, c->desc.p,
                       c->totals.p + 3, (const TxHdr *)c->hdr.p, c->win_hdr.p, (const uint32_t *)c->tile_first.p, c->walked.p);
    MARK(ST_SCAN1);
    {
        ScanJobs jobs = {}; jobs.job[0] = ScanJob{c->tile_base.p, c->n_tiles, c->totals.p + 0}; jobs.job[1] = jobs.job[0];
        hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, s, jobs);
    }
    MARK(ST_FAST);
    {
        switch (p.full_level) {
        case 1: launch_fast_level(1, fa, gp, s); break;
        case 2: launch_fast_level(2, fa, gp, s); break;
        case 3: launch_fast_level(3, fa, gp, s); break;
        case 4: launch_fast_level(4, fa, gp, s); break;
        case 5: launch_fast_level(5, fa, gp, s); break;
        default: launch_fast_level(0, fa, gp, s); break;      // src/update_gtf.c:629-696: no evidence is gathered, full = lfull && rfull = 0
        }
    }
    }
    MARK(ST_GENERIC);
    // (one-kernel tile path: a completed run of the same inputs and parameters has left nothing on the redo list and nothing to the
    //  list-driven kernels -- no read is left for the generic kernel, and what it used to clear for the next run is cleared in front)
    // (... or nothing on the redo list: the list counters need no launch for their clearing, they take turns)
    const bool nothing_left = c->tile && c->lists_known && c->redo_empty && (c->lists_empty || c->n_sj == 0) && !c->env_launch_all;
    if (c->tile) { c->lb_flip ^= 1u; c->lc_flip ^= 1u; }
    c->prev_run_tile = c->tile;
    if (!nothing_left) {
        const unsigned gg = (unsigned)std::min<int64_t>(c->n_tiles ? c->n_tiles * 4 : 1, 4096);      // one wave per listed read, grid-stride
        hipLaunchKernelGGL(k_classify_generic, dim3(gg), dim3(TILE_THREADS), 0, s, c->totals.p + 3, c->redo.p, c->r_tid.p, c->r_rev.p,
                           (c->slab ? (const int32_t *)nullptr : j0),
                           c->hdr.p, c->anno_ex.p, p, c->ex_off.p, c->ex_start.p, c->ex_end.p, c->ex_flag.p, c->info.p, c->ref_tx.p,
                           c->tile_acc.p, c->tile_acc_ex.p, (const uint32_t *)c->tile_first.p, (int)c->n_tiles, cd, (uint32_t *)nullptr);
    }
    MARK(ST_SJ);
    // (one-kernel tile path: k_tile has checked every read whose verdict it made; with nothing on the redo list and nothing left to the
    //  list-driven kernels -- seen by a completed run of the same inputs and parameters -- no read is left for this launch)
    const bool sj_all_in_tile = nothing_left;
    if (c->n_sj > 0 && !sj_all_in_tile) {
        if (!c->sorted) { int rc = prepare_unsorted_sj_cursor(c); if (rc) return rc; }
        hipLaunchKernelGGL(k_validate_sj, dim3(g256), dim3(TILE_THREADS), 0, s, N, c->r_tid.p, c->ex_off.p, c->ex_start.p, c->ex_end.p, c->ex_flag.p,
                           c->sj_key.p, (c->sorted ? (const int32_t *)nullptr : c->sj_cursor.p), c->sj_tid.p, c->sj_don.p, c->sj_acc.p,
                           c->sj_uniq.p, c->sj_multi.p, p, c->info.p,
                           SjDir{CursorDir{c->sj_key.p, c->sj_cdir.p, c->sj_cbase.p, c->sj_ntid, (int32_t)c->n_sj}, c->sj_ddir.p, c->sj_dbase.p, c->sj_ntid, c->sj_row.p});
    }
    if ((c->n_sj > 0 || c->slab) && (c->want & L2R_WANT_ACCEPTED)) {
        // acceptance is decided by the junction check (and the slab pipeline counts nothing itself): count per tile
            hipLaunchKernelGGL(k_count_accepted, dim3((gt + 3u) / 4u), dim3(TILE_THREADS), 0, s, (const uint32_t *)c->tile_first.p, c->info.p, (const uint32_t *)c->tile_chunk.p, c->tile_acc.p, c->tile_acc_ex.p, (uint32_t)c->n_tiles);
    }
    MARK(ST_SCAN2);
    if (c->want & L2R_WANT_ACCEPTED) {
        // the deferred tiles' counts -> their places behind the fused chunks (tile_acc_at / tile_acc_ex_at; the sums = totals[1], [2])
        if (c->n_tiles <= c->seg_max) {
            const unsigned n_seg = (unsigned)std::max<int64_t>((c->n_tiles + SEG_COUNT - 1) / SEG_COUNT, 1);
            hipLaunchKernelGGL(k_scan_segments, dim3(2u * n_seg), dim3(TILE_THREADS), 0, s, SegScan{c->tile_acc.p, c->tile_acc_at.p, c->totals.p + 1, c->n_tiles},
                               SegScan{c->tile_acc_ex.p, c->tile_acc_ex_at.p, c->totals.p + 2, c->n_tiles}, (uint32_t)n_seg);
        } else {
            HIP_TRY(hipMemcpyAsync(c->tile_acc_at.p, c->tile_acc.p, (size_t)c->n_tiles * 4, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(c->tile_acc_ex_at.p, c->tile_acc_ex.p, (size_t)c->n_tiles * 4, hipMemcpyDeviceToDevice, s));
            ScanJobs jobs = {}; jobs.job[0] = ScanJob{c->tile_acc_at.p, c->n_tiles, c->totals.p + 1}; jobs.job[1] = ScanJob{c->tile_acc_ex_at.p, c->n_tiles, c->totals.p + 2};
            hipLaunchKernelGGL(k_scan_u32, dim3(2), dim3(1024), 0, s, jobs);
        }
    }
    MARK(ST_GATHER);
    if (c->want & L2R_WANT_ACCEPTED)
    hipLaunchKernelGGL(k_gather_accepted, dim3((gt + GATHER_TILES - 1) / GATHER_TILES), dim3(TILE_THREADS), 0, s, (const uint32_t *)c->tile_first.p, c->first_read, c->info.p, c->ref_tx.p, c->ex_off.p,
                       c->ex_start.p, c->ex_end.p, c->ex_flag.p, c->tile_acc_at.p, c->tile_acc_ex_at.p, c->tile_chunk.p, c->tile_rchunk.p, c->totals.p + 4,
                       c->acc_rec.p, c->acc_ex_off.p, c->acc_start.p, c->acc_end.p, c->acc_flag.p, (uint32_t)c->n_tiles);
    MARK(ST_N);
#undef MARK
    HIP_TRY(hipGetLastError());
    return 0;
}

/* diagnostics: L2R_STAMPS=1 makes k_classify_fast accumulate per-phase cycles; this prints and clears them */
int l2r_debug_stamps(l2r_ctx *c, unsigned long long *out, int n)
{
    if (!c || !out) return fail(-1, "[l2r_debug_stamps] null argument");
    if (!c->stamps.p) { for (int i = 0; i < n; ++i) out[i] = 0; return 0; }
    std::vector<unsigned long long> h(1024 * 8 + 16);
    HIP_TRY(hipMemcpyAsync(h.data(), c->stamps.p, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemsetAsync(c->stamps.p, 0, h.size() * 8, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i) out[i] = 0;
    for (size_t k = 0; k < 1024 * 8; ++k) if ((int)(k & 7) < n) out[k & 7] += h[k];
    for (int i = 8; i < n && i < 16; ++i) out[i] = h[1024 * 8 + (i - 8)];      /* redo reasons: not fast, not in LDS, wide, other tid, not sane, window/compact */
    return 0;
}

/* diagnostics (L2R_STAMPS=1, one-kernel tile path): per tile four words -- the 100 MHz clock at its start << 3 | its XCD, the clock at the
   publication of its exon count, at the begin and at the end of its wait for the counts of the tiles in front */
int l2r_debug_tile_times(l2r_ctx *c, uint32_t *out, int64_t n_tiles)
{
    if (!c || !out) return fail(-1, "[l2r_debug_tile_times] null argument");
    if (!c->tile || !c->ran || n_tiles > c->n_tiles) return fail(-1, "[l2r_debug_tile_times] no run of the one-kernel tile path to report");
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t *src[4] = {c->tile_total.p, c->tile_acc.p, c->tile_acc_ex.p, c->tile_flags.p};
    std::vector<uint32_t> h((size_t)n_tiles);
    for (int k = 0; k < 4; ++k) {
        HIP_TRY(hipMemcpyAsync(h.data(), src[k], (size_t)n_tiles * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int64_t i = 0; i < n_tiles; ++i) out[4 * i + k] = h[(size_t)i];
    }
    return 0;
}

/* diagnostics: [0] reads the last run sent to the generic kernel, [1] dictionary entries flagged wide,
   [2] compact transcripts, [3] tiles, [4..11] tiles by the reason they are not fast, [12] tiles of k_probe_slab_wide */
int l2r_debug_counters(l2r_ctx *c, long long *out, int n)
{
    if (!c || !out || n < 4) return fail(-1, "[l2r_debug_counters] bad argument");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t redo = 0;
    if (c->totals.p) { HIP_TRY(hipMemcpyAsync(&redo, c->totals.p + 3, 4, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); }
    out[0] = redo; out[1] = c->n_wide; out[2] = c->n_compact; out[3] = c->n_tiles;
    if (n >= 12) for (int k = 0; k < 8; ++k) out[4 + k] = 0;
    if (n >= 13) out[12] = 0;
    if (n >= 14) out[13] = c->n_lb_fallback;              // runs done again on the slab pipeline because k_tile's look-back starved
    if (n >= 16) {                                        // one-kernel tile path, last run: entries of chunk_list k_tile_chunk declined, tiles handed to the chunked kernel late
        out[14] = 0; out[15] = 0;
        if (c->tile && c->ran && c->list_cnt.p) {
            uint32_t lc[16];
            HIP_TRY(hipMemcpyAsync(lc, c->list_cnt.p + 16 * ((c->lc_flip & 1u) ^ 1u), sizeof lc, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            out[14] = lc[9]; out[15] = lc[8];
        }
    }
    if (n >= 12 && c->slab && c->ran && c->tw.p && c->n_tiles > 0) {     // slab pipeline: the descriptors k_walk_slab made (flags as the probe kernels left them)
        std::vector<TileWin> w((size_t)c->n_tiles);
        HIP_TRY(hipMemcpyAsync(w.data(), c->tw.p, w.size() * sizeof(TileWin), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (n >= 24) for (int k = 16; k < 24; ++k) out[k] = 0;
        for (const TileWin &t : w) {
            if (n >= 24) {                                  // tiles of the chunked kernel by the END entries of their dictionary slices (<= 256, 512, 768, 1024, more), START entries beyond 128 / 256, all of them
                const uint32_t why = (t.d.flags >> 8) & 7u;
                if ((t.d.flags & TD_CHUNK) || (!(t.d.flags & (TD_FAST | TD_WIDE)) && (why == 4u || why == 3u))) {
                    out[16 + (t.d.en_nk <= 256u ? 0 : t.d.en_nk <= 512u ? 1 : t.d.en_nk <= 768u ? 2 : t.d.en_nk <= 1024u ? 3 : 4)]++;
                    if (t.d.st_nk > 128u) out[21]++;
                    if (t.d.st_nk > 256u) out[22]++;
                    out[23]++;
                }
            }
            out[4 + ((t.d.flags >> 8) & 7u)]++;
            if (n >= 13 && (t.d.flags & TD_WIDE) && !(t.d.flags & TD_CHUNK)) out[12]++;     // out[12]: tiles k_probe_slab_wide classified (33 .. 63 window members)
        }
    } else if (n >= 12 && !c->slab && c->desc.p && c->n_tiles > 0) {     // out[4 + k]: tiles that are not fast for reason k (k_pass_a), k = 0: fast
        std::vector<TileDesc> d((size_t)c->n_tiles);
        HIP_TRY(hipMemcpyAsync(d.data(), c->desc.p, d.size() * sizeof(TileDesc), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (const TileDesc &t : d) out[4 + ((t.flags >> 8) & 7u)]++;
    }
    return 0;
}

/* Which kernel(s) stand behind stage_ms[stage] of l2r_timing for the inputs and parameters now set (the pipeline is
   chosen per launch: slab for coordinate-sorted records with short CIGARs, classic otherwise). */
const char *l2r_stage_kernel(l2r_ctx *c, int stage)
{
    if (!c || stage < 0 || stage >= L2R_N_STAGES) return "";
    // (the first word is the kernel's name as a profile lists it; both scans are launches of k_scan_u32)
    static const char *const classic[L2R_N_STAGES] = {"k_pass_a", "k_scan_u32 (tile sums)", "k_classify_fast", "k_classify_generic",
                                                      "k_validate_sj", "k_scan_accepted (k_scan_u32 of the accepted counts)", "k_gather_accepted", ""};
    if (stage >= 3 || !c->slab) return classic[stage];
    if (c->tile) return stage == 0 ? "k_describe_scan (tile descriptors + tile lists; first kernel of the run)" : stage == 1 ? "k_tile (walk + probes + write-out, one workgroup per tile)" : "k_probe_slab (tiles k_tile left in slab form) (+ k_tile_chunk + k_probe_slab_wide + k_probe_slab_chunked)";
    return stage == 0 ? (c->wide_cigar ? "k_walk_slab_long" : "k_walk_slab") : stage == 1 ? "k_describe_scan (tile descriptors + scan of the exon counts + tile lists)" : "k_probe_slab (+ k_probe_slab_wide + k_probe_slab_chunked)";
}

int l2r_run(l2r_ctx *c)
{
    if (!c) return fail(-1, "[l2r_run] null context");
    HIP_TRY(hipSetDevice(c->device));
    int rc = prepare_unsorted_windows(c);
    if (rc) return rc;
    // L2R_GRAPH=1: the launch sequence (7 kernels, no host round trip) is captured into a hipGraph on first use and
    // replayed, one submission per pass instead of seven.  Off by default: measured on MI355X / ROCm 7.2 the replay is
    // 2-4 % slower than the seven direct launches (config 2: 0.093 vs 0.089 ms, config 3: 1.42 vs 1.40 ms per pass).
    // Not for unsorted input with a junction table (its cursor replay syncs).
    // Not for the one-kernel tile path either (two launches: nothing to gain, and its launch arguments change run by run -- the
    // super-block words take turns, launches are dropped once a run has shown them empty).
    const bool graphable = !(c->n_sj > 0 && !c->sorted) && (c->want_pipeline < 2 || (c->ran && !c->tile)) && getenv("L2R_GRAPH") != nullptr;
    if (graphable && !c->graph_valid) {
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            rc = launch_all(c, nullptr);
            const hipError_t e = hipStreamEndCapture(c->stream, &g);
            if (rc == 0 && e == hipSuccess && g && hipGraphInstantiate(&c->graph, g, nullptr, nullptr, 0) == hipSuccess) c->graph_valid = true;
            if (g) (void)hipGraphDestroy(g);
            if (!c->graph_valid) { (void)hipGetLastError(); c->graph = nullptr; }
        }
    }
    if (graphable && c->graph_valid) HIP_TRY(hipGraphLaunch(c->graph, c->stream));
    else { rc = launch_all(c, nullptr); if (rc) return rc; }
    c->ran = true; c->totals_valid = false;
    return 0;
}

int l2r_sync(l2r_ctx *c)
{
    if (!c) return fail(-1, "[l2r_sync] null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->ran && c->tile && c->totals.p) {
        // a tile of k_tile waited in vain for the exon counts in front of it (its poll limit ended every wait: the run is complete but its
        // result slots are not to be trusted): the SAME resident upload once more on the slab pipeline, which has no such wait -- and no
        // later run of this context takes the tile path again (the cause is the device's occupancy, not this input)
        uint32_t lb = 0u;
        HIP_TRY(hipMemcpyAsync(&lb, c->totals.p + 6, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (lb != 0u) {
            c->tile_starved = true; c->n_lb_fallback++;
            c->lists_known = false; c->totals_valid = false;
            int rc = launch_all(c, nullptr);
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(c->stream));
            snprintf(g_err, sizeof g_err, "[l2r_sync] note: k_tile's look-back starved; the run was done again on the slab pipeline (this context keeps to it)");
        }
    }
    if (c->ran && c->tile && !c->lists_known && c->list_cnt.p) {
        // what the run left on the lists of the kernels behind k_tile (k_classify_generic keeps the counts of the 64-bit-mask and the
        // chunked kernel's lists in words 6, 7 when it clears them; word 4: k_probe_slab's)
        uint32_t lc[16];
        HIP_TRY(hipMemcpyAsync(lc, c->list_cnt.p + 16 * ((c->lc_flip & 1u) ^ 1u), sizeof lc, hipMemcpyDeviceToHost, c->stream));      // (the block of the run that has just ended: launch_all has flipped already)
        lc[6] = lc[0]; lc[7] = lc[1];                              // (entries of wide_list / chunk_list: nobody clears them behind their readers any more)
        HIP_TRY(hipStreamSynchronize(c->stream));
        uint32_t redo_n = 1u;
        HIP_TRY(hipMemcpyAsync(&redo_n, c->totals.p + 3, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        // (lc[8]: tiles a one-window kernel handed to the chunked kernel late; lc[9]: entries of chunk_list k_tile_chunk declined)
        c->lists_empty = lc[4] == 0u && lc[6] == 0u && lc[7] == 0u && lc[8] == 0u;
        const bool tchunk = c->chunk_direct && !(c->ablate & 32);
        // (what is left to k_probe_slab_chunked: with k_tile_chunk the entries it declined in its two launches, else every entry)
        const unsigned long long chunk_rest = tchunk ? (unsigned long long)lc[9] + lc[10] : (unsigned long long)lc[8] + lc[7];
        c->n_late_tiles = lc[8];
        // (with k_tile's WIDE instance / k_tile_chunk the tiles they take are no burden of the tile path: what counts is what keeps the slab form)
        c->lists_heavy = 2ull * ((unsigned long long)lc[4] + (c->wide_direct ? lc[5] : lc[6]) + chunk_rest) > (unsigned long long)c->n_tiles;
        c->n_wide_tiles = lc[6]; c->wide_rest_empty = lc[5] == 0u; c->fb_empty = lc[4] == 0u;
        c->n_chunk_tiles = lc[7]; c->chunk_rest_empty = chunk_rest == 0ull;
        c->redo_empty = redo_n == 0u;
        c->lists_known = true;
    }
    return 0;
}

static int fetch_totals(l2r_ctx *c)
{
    if (!c->ran) return fail(-1, "no completed run on this context");
    if (c->totals_valid) return 0;
    uint32_t dev[8];
    HIP_TRY(hipMemcpyAsync(dev, c->totals.p, sizeof dev, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // accepted exons = the chunks the classification kernel placed itself (cursor) + the ones k_gather_accepted placed
    if (c->tile && dev[6] != 0u) return fail(-2, "[l2r] k_tile: a tile waited in vain for the exon counts of the tiles in front of it and l2r_sync was not called behind the run (it does the run again on the slab pipeline)");
    c->h_totals[0] = dev[0]; c->h_totals[1] = dev[1] + dev[5]; c->h_totals[2] = dev[2] + dev[4];
    if (!(c->want & L2R_WANT_ACCEPTED)) c->h_totals[1] = c->h_totals[2] = 0;
    c->totals_valid = true;
    return 0;
}

int l2r_run_timed(l2r_ctx *c, int iters, l2r_timing *out)
{
    if (!c || !out || iters <= 0) return fail(-1, "[l2r_run_timed] bad argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = prepare_unsorted_windows(c);
    if (rc) return rc;
    memset(out, 0, sizeof *out);
    // every event lives in one holder that releases them on every way out of this function
    struct Events {
        hipEvent_t e[ST_N + 3]; int n = 0;
        ~Events() { for (int i = 0; i < n; ++i) (void)hipEventDestroy(e[i]); }
    } evs;
    for (int i = 0; i < ST_N + 3; ++i) { HIP_TRY(hipEventCreate(&evs.e[i])); evs.n = i + 1; }
    hipEvent_t t0 = evs.e[ST_N + 1], t1 = evs.e[ST_N + 2], *ev = evs.e;
    // pass A: whole pipeline, back to back
    HIP_TRY(hipEventRecord(t0, c->stream));
    for (int it = 0; it < iters; ++it) { rc = launch_all(c, nullptr); if (rc) return rc; }
    HIP_TRY(hipEventRecord(t1, c->stream));
    HIP_TRY(hipEventSynchronize(t1));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
    out->total_ms = ms / (float)iters;
    // pass B: per-stage events
    for (int it = 0; it < iters; ++it) {
        rc = launch_all(c, ev); if (rc) return rc;
        HIP_TRY(hipEventSynchronize(ev[ST_N]));
        for (int i = 0; i < ST_N; ++i) { float d = 0; HIP_TRY(hipEventElapsedTime(&d, ev[i], ev[i + 1])); out->stage_ms[i] += d / (float)iters; }
    }
    out->iters = iters;
    c->ran = true; c->totals_valid = false;
    return 0;
}

int l2r_result_sizes(l2r_ctx *c, int64_t *n_reads, int64_t *n_exons, int64_t *n_acc, int64_t *n_acc_ex)
{
    if (!c) return fail(-1, "[l2r_result_sizes] null context");
    HIP_TRY(hipSetDevice(c->device));
    int rc = fetch_totals(c);
    if (rc) return rc;
    if (n_reads) *n_reads = c->n_reads;
    if (n_exons) *n_exons = c->h_totals[0];
    if (n_acc) *n_acc = c->h_totals[1];
    if (n_acc_ex) *n_acc_ex = c->h_totals[2];
    return 0;
}

int l2r_download(l2r_ctx *c, l2r_result *res)
{
    if (!c || !res) return fail(-1, "[l2r_download] null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = fetch_totals(c);
    if (rc) return rc;
    const int64_t N = c->n_reads, X = c->h_totals[0];
    if (res->n_reads < N || res->ex_cap < X) return fail(-3, "[l2r_download] buffers too small: need %lld reads, %lld exons", (long long)N, (long long)X);
    // both pipelines leave the results in read order: exon k of read i at ex_off[i] + k, offsets = the running sum of the exon counts
    std::vector<uint32_t> off((size_t)N);
    if (N) {
        HIP_TRY(hipMemcpyAsync(off.data(), c->ex_off.p, (size_t)N * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(res->info, c->info.p, (size_t)N * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(res->ref_tx, c->ref_tx.p, (size_t)N * 4, hipMemcpyDeviceToHost, c->stream));
    }
    if (X) {
        HIP_TRY(hipMemcpyAsync(res->ex_start, c->ex_start.p, (size_t)X * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(res->ex_end, c->ex_end.p, (size_t)X * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(res->ex_flag, c->ex_flag.p, (size_t)X, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    {
        int64_t at = 0;
        for (int64_t i = 0; i < N; ++i) {
            if ((int64_t)off[(size_t)i] != at) return fail(-5, "[l2r_download] exon offsets are not the running sum of the exon counts at read %lld (%u, expected %lld)", (long long)i, off[(size_t)i], (long long)at);
            res->ex_off[i] = at; at += (int64_t)(res->info[i] >> 8);
        }
        if (at != X) return fail(-5, "[l2r_download] exon counts (%lld) do not add up to the exon total (%lld)", (long long)at, (long long)X);
    }
    res->ex_off[N] = X;
    res->n_reads = N; res->n_exons = X;
    return 0;
}

// The accepted list of one engine as it lies in HBM (per-tile chunks in the order they were handed out, see k_gather_accepted) -> read
// order, appended to `a` at record at_r / exon at: the chunks are told apart by the first-record slots the tiles left (starts), each is
// in read order inside, and the exons are laid out record by record, so that ex_off is the running sum.
static int order_accepted(const AccRec *rec, const uint32_t *off, std::vector<uint32_t> starts, const int32_t *xs, const int32_t *xe, const uint8_t *xf,
                          int64_t M, int64_t X, l2r_accepted *a, int64_t &at_r, int64_t &at)
{
    if (!M) return 0;
    const int64_t r_end = at_r + M, x_base = at;
    // (a tile without accepted reads leaves the slot of some other chunk, 0 or M: duplicates and M drop out)
    starts.push_back(0u);
    std::sort(starts.begin(), starts.end());
    starts.erase(std::unique(starts.begin(), starts.end()), starts.end());
    while (!starts.empty() && (int64_t)starts.back() >= M) starts.pop_back();
    std::vector<std::pair<uint64_t, std::pair<uint32_t, uint32_t>>> chunks;        // first read index -> [from, to)
    for (size_t k = 0; k < starts.size(); ++k) {
        const uint32_t from = starts[k], to = k + 1 < starts.size() ? starts[k + 1] : (uint32_t)M;
        chunks.push_back({((uint64_t)rec[from].read_hi << 32) | rec[from].read_lo, {from, to}});
    }
    std::sort(chunks.begin(), chunks.end());
    for (const auto &ch : chunks) {
        for (uint32_t i = ch.second.first; i < ch.second.second; ++i) {
            const int64_t n = (int64_t)(rec[i].info >> 8), from = off[i];
            if (from + n > X || at - x_base + n > X || at_r >= r_end) return fail(-5, "[l2r_download_accepted] inconsistent accepted list");
            memcpy(&a->rec[at_r], &rec[i], sizeof(AccRec));
            a->ex_off[at_r] = at;
            memcpy(a->ex_start + at, xs + from, (size_t)n * 4);
            memcpy(a->ex_end + at, xe + from, (size_t)n * 4);
            memcpy(a->ex_flag + at, xf + from, (size_t)n);
            at += n; ++at_r;
        }
    }
    if (at_r != r_end) return fail(-5, "[l2r_download_accepted] inconsistent accepted list (%lld of %lld records)", (long long)(at_r - (r_end - M)), (long long)M);
    return 0;
}

int l2r_download_accepted(l2r_ctx *c, l2r_accepted *a)
{
    if (!c || !a) return fail(-1, "[l2r_download_accepted] null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (!(c->want & L2R_WANT_ACCEPTED)) return fail(-1, "[l2r_download_accepted] the accepted list was not requested (l2r_set_outputs)");
    int rc = fetch_totals(c);
    if (rc) return rc;
    const int64_t M = c->h_totals[1], X = c->h_totals[2];
    if (a->n_reads < M || a->ex_cap < X) return fail(-3, "[l2r_download_accepted] buffers too small: need %lld records, %lld exons", (long long)M, (long long)X);
    // On the device the list is a sequence of per-tile chunks in the order they were handed out (see k_gather_accepted):
    // the chunks are put into read order here (each is in read order inside; they are told apart by the first-record
    // slots the tiles left in tile_rchunk) and the exons laid out record by record, so that ex_off is the running sum.
    std::vector<AccRec> rec((size_t)M);
    std::vector<uint32_t> off((size_t)M), starts((size_t)c->n_tiles);
    std::vector<int32_t> xs((size_t)X), xe((size_t)X);
    std::vector<uint8_t> xf((size_t)X);
    if (M) {
        HIP_TRY(hipMemcpyAsync(rec.data(), c->acc_rec.p, (size_t)M * sizeof(AccRec), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(off.data(), c->acc_ex_off.p, (size_t)M * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(starts.data(), c->tile_rchunk.p, (size_t)c->n_tiles * 4, hipMemcpyDeviceToHost, c->stream));
    }
    if (X) {
        HIP_TRY(hipMemcpyAsync(xs.data(), c->acc_start.p, (size_t)X * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(xe.data(), c->acc_end.p, (size_t)X * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(xf.data(), c->acc_flag.p, (size_t)X, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    int64_t at_r = 0, at = 0;
    rc = order_accepted(rec.data(), off.data(), starts, xs.data(), xe.data(), xf.data(), M, X, a, at_r, at);
    if (rc) return rc;
    a->ex_off[M] = at;
    a->n_reads = M; a->n_exons = X;
    return 0;
}

int l2r_device_view_get(l2r_ctx *c, l2r_device_view *v)
{
    if (!c || !v) return fail(-1, "[l2r_device_view_get] null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = fetch_totals(c);
    if (rc) return rc;
    v->n_reads = c->n_reads; v->n_exons = c->h_totals[0]; v->n_accepted = c->h_totals[1]; v->n_accepted_exons = c->h_totals[2];
    v->ex_off = nullptr; v->ex_start = nullptr; v->ex_end = nullptr; v->ex_flag = nullptr;
    if (c->want & L2R_WANT_RESULTS) { v->ex_off = c->ex_off.p; v->ex_start = c->ex_start.p; v->ex_end = c->ex_end.p; v->ex_flag = c->ex_flag.p; }      // (read order, as the kernels left them; fetch_totals has waited for the stream)
    v->info = c->info.p; v->ref_tx = c->ref_tx.p;
    v->acc_rec = (const l2r_accepted_read *)c->acc_rec.p; v->acc_ex_off = c->acc_ex_off.p;
    v->acc_ex_start = c->acc_start.p; v->acc_ex_end = c->acc_end.p; v->acc_ex_flag = c->acc_flag.p;
    return 0;
}

int l2r_classify(l2r_ctx *c, const l2r_reads *reads, l2r_result *res)
{
    if (!c) return fail(-1, "[l2r_classify] null context");
    // ONE run follows this upload: by total GPU time the two-kernel (slab) pipeline wins that case -- measured on 10 M reads 0.62 ms against
    // tile index + first run of the one-kernel path 0.68 ms (0.80 ms where the engine has to walk the CIGARs for the index itself); the
    // one-kernel path pays off from the second run of an upload on (0.50 ms a run).  So this upload makes no tile index and its run takes
    // the slab pipeline (L2R_PIPELINE=tile or L2R_TILE_ANYWAY=1: index + tile path all the same).
    const bool was = c->one_shot_upload;
    c->one_shot_upload = true;
    int rc = l2r_upload_reads(c, reads);
    c->one_shot_upload = was;
    if (rc) return rc;
    if ((rc = l2r_run(c))) return rc;
    if ((rc = l2r_sync(c))) return rc;
    return l2r_download(c, res);
}

// ---------------------------------------------------------------------------------------------- filter
extern "C++" {
template <typename T> static int to_dev(l2r_ctx *c, DevBuf<T> &b, const T *src, size_t n)
{
    if (b.ensure(n ? n : 1)) return -2;
    if (n) HIP_TRY(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return 0;
}
}

int l2r_filter_score(l2r_ctx *c, const l2r_filter_records *r, const l2r_filter_params *prm, const l2r_filter_spans *rm,
                     uint8_t *drop, int32_t *score, int32_t *intron_n)
{
    if (!c || !r || !prm || !drop || !score || !intron_n) return fail(-1, "[l2r_filter_score] null argument");
    if (r->n < 0 || r->n_cigar < 0) return fail(-1, "[l2r_filter_score] negative size");
    HIP_TRY(hipSetDevice(c->device));
    const size_t N = (size_t)r->n;
    if (N && (r->cig_off[0] != 0 || r->cig_off[N] != r->n_cigar)) return fail(-1, "[l2r_filter_score] cig_off does not span the CIGAR array");
    // the -r transcripts a record of every chromosome can meet (l2r_filter.hip.h FilterSpans), on the host
    std::vector<int64_t> off; std::vector<int32_t> st, pe;
    int32_t n_tid = 0;
    if (rm && rm->n > 0) {
        for (int64_t j = 0; j < rm->n; ++j) n_tid = std::max(n_tid, rm->tid[j] + 1);
        // first_gt[v]: first transcript (file order) with tid > v = where remove_overlap() stops for a record of tid v
        std::vector<int64_t> first_gt((size_t)n_tid, rm->n);
        int32_t seen = -1;                                 // largest tid so far
        for (int64_t j = 0; j < rm->n; ++j) if (rm->tid[j] > seen) { for (int32_t v = std::max(seen, 0); v < rm->tid[j]; ++v) first_gt[(size_t)v] = std::min(first_gt[(size_t)v], j); seen = rm->tid[j]; }
        std::vector<std::vector<std::pair<int32_t, int32_t>>> per((size_t)n_tid);
        for (int64_t j = 0; j < rm->n; ++j) { const int32_t t = rm->tid[j]; if (t >= 0 && j < first_gt[(size_t)t]) per[(size_t)t].push_back({rm->start[j], rm->end[j]}); }
        off.assign((size_t)n_tid + 1, 0);
        for (int32_t t = 0; t < n_tid; ++t) {
            auto &v = per[(size_t)t];
            std::sort(v.begin(), v.end());
            int32_t run = INT32_MIN;
            for (auto &se : v) { run = std::max(run, se.second); st.push_back(se.first); pe.push_back(run); }
            off[(size_t)t + 1] = (int64_t)st.size();
        }
    }
    DevBuf<uint16_t> d_flag; DevBuf<int32_t> d_tid, d_pos, d_lq, d_nm, d_score, d_in, d_st, d_pe; DevBuf<int64_t> d_off, d_soff; DevBuf<uint32_t> d_cig; DevBuf<uint8_t> d_drop;
    int rc = 0;
    if ((rc = to_dev(c, d_flag, r->flag, N)) || (rc = to_dev(c, d_tid, r->tid, N)) || (rc = to_dev(c, d_pos, r->pos, N)) || (rc = to_dev(c, d_lq, r->l_qseq, N)) ||
        (rc = to_dev(c, d_nm, r->nm, N)) || (rc = to_dev(c, d_off, r->cig_off, N + 1)) || (rc = to_dev(c, d_cig, r->cig, (size_t)r->n_cigar)) ||
        (rc = to_dev(c, d_soff, off.data(), off.size())) || (rc = to_dev(c, d_st, st.data(), st.size())) || (rc = to_dev(c, d_pe, pe.data(), pe.size())) ||
        d_drop.ensure(N ? N : 1) || d_score.ensure(N ? N : 1) || d_in.ensure(N ? N : 1)) { rc = rc ? rc : -2; }
    if (!rc && N) {
        const FilterPrm fp{prm->cov_rate, prm->map_qual, prm->sec_rat, prm->min_intron_n};
        const FilterSpans sp{d_soff.p, d_st.p, d_pe.p, n_tid};
        hipLaunchKernelGGL(k_filter_score, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, (int64_t)N, (const uint16_t *)d_flag.p, (const int32_t *)d_tid.p,
                           (const int32_t *)d_pos.p, (const int32_t *)d_lq.p, (const int32_t *)d_nm.p, (const int64_t *)d_off.p, (const uint32_t *)d_cig.p, fp, sp,
                           d_drop.p, d_score.p, d_in.p);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(drop, d_drop.p, N, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(score, d_score.p, N * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(intron_n, d_in.p, N * 4, hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) rc = fail(-2, "[l2r_filter_score] %s", hipGetErrorString(e));
    }
    hipError_t e = hipStreamSynchronize(c->stream);          // (the host vectors and the DevBufs above are locals)
    if (!rc && e != hipSuccess) rc = fail(-2, "[l2r_filter_score] %s", hipGetErrorString(e));
    d_flag.release(); d_tid.release(); d_pos.release(); d_lq.release(); d_nm.release(); d_score.release(); d_in.release(); d_st.release(); d_pe.release();
    d_off.release(); d_soff.release(); d_cig.release(); d_drop.release();
    return rc;
}

int l2r_filter_select(l2r_ctx *c, int64_t n_groups, const int64_t *group_off, const int32_t *score, const int32_t *intron_n,
                      const l2r_filter_params *prm, int64_t *winner)
{
    if (!c || !prm || n_groups < 0 || (n_groups && (!group_off || !score || !intron_n || !winner))) return fail(-1, "[l2r_filter_select] bad argument");
    if (n_groups == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    const size_t G = (size_t)n_groups, R = (size_t)group_off[G];
    for (size_t g = 0; g < G; ++g) if (group_off[g + 1] <= group_off[g]) return fail(-1, "[l2r_filter_select] group %lld is empty", (long long)g);
    DevBuf<int64_t> d_off, d_win; DevBuf<int32_t> d_score, d_in;
    int rc = 0;
    if ((rc = to_dev(c, d_off, group_off, G + 1)) || (rc = to_dev(c, d_score, score, R)) || (rc = to_dev(c, d_in, intron_n, R)) || d_win.ensure(G)) rc = rc ? rc : -2;
    if (!rc) {
        const FilterPrm fp{prm->cov_rate, prm->map_qual, prm->sec_rat, prm->min_intron_n};
        hipLaunchKernelGGL(k_filter_select, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, c->stream, (int64_t)G, (const int64_t *)d_off.p, (const int32_t *)d_score.p,
                           (const int32_t *)d_in.p, fp, d_win.p);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(winner, d_win.p, G * 8, hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) rc = fail(-2, "[l2r_filter_select] %s", hipGetErrorString(e));
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (!rc && e != hipSuccess) rc = fail(-2, "[l2r_filter_select] %s", hipGetErrorString(e));
    d_off.release(); d_win.release(); d_score.release(); d_in.release();
    return rc;
}

}  // extern "C"

static_assert(sizeof(AccRec) == sizeof(l2r_accepted_read), "accepted record layout");
static_assert(sizeof(TxHdr) == 48, "TxHdr must be three int4");
static_assert(sizeof(SiteEnt) == 32 && sizeof(TileDesc) == 48, "dictionary entry / tile descriptor layout");

#include "l2r_xchg.hip.h"
