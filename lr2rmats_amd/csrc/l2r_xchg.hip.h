// l2r_xchg.hip.h -- the gathered route's exchange in the engine library: the per-read results of the ranks of one node, left in HBM by
// their engines, travel over RCCL (xGMI) to rank 0, which hands them to the order-dependent host tail as ONE read-order result set.
// (The reference merges in read order -- src/update_gtf.c:946-960 -- and with -s and a junction table its split pieces are compared
// across chromosomes, Q2: that input cannot be cut into independent shards, so its shards' results are gathered instead.)
// One process per GPU; the caller forks / launches the ranks and carries the 128-byte id from rank 0 to the others
// (host/cmds.c: through memory shared before the fork; lr2rmats_amd/dist.py does the same job through torch.distributed).
// Included at the end of l2r_engine.hip (it needs l2r_ctx).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

// RCCL is resolved when the exchange is first used, not when libl2r_hip.so is loaded: a one-GPU run, bench.py and the Python binding
// need no librccl at all, and a process that holds one already (torch ships its own) keeps using THAT copy -- two RCCLs in one process
// is asking for trouble.  Order: a copy that is loaded already (RTLD_NOLOAD), then $L2R_RCCL_LIB, $ROCM_PATH/lib/librccl.so, the loader's path.
struct RcclApi {
    void *h = nullptr;
    decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&::ncclCommInitRank) CommInitRank = nullptr;
    decltype(&::ncclCommDestroy) CommDestroy = nullptr;
    decltype(&::ncclCommCount) CommCount = nullptr;
    decltype(&::ncclCommUserRank) CommUserRank = nullptr;
    decltype(&::ncclCommCuDevice) CommCuDevice = nullptr;
    decltype(&::ncclAllGather) AllGather = nullptr;
    decltype(&::ncclSend) Send = nullptr;
    decltype(&::ncclRecv) Recv = nullptr;
    decltype(&::ncclGroupStart) GroupStart = nullptr;
    decltype(&::ncclGroupEnd) GroupEnd = nullptr;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
};
static RcclApi g_rccl;
static const char *rccl_load()                              // nullptr: ready; else what went wrong
{
    if (g_rccl.h) return nullptr;
    static char why[512];
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so"}) if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (!h) { const char *e = getenv("L2R_RCCL_LIB"); if (e && *e) h = dlopen(e, RTLD_NOW | RTLD_LOCAL); }
    if (!h) {
        const char *rp = getenv("ROCM_PATH");
        char path[1024];
        snprintf(path, sizeof path, "%s/lib/librccl.so", (rp && *rp) ? rp : "/opt/rocm");
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    }
    for (const char *name : {"librccl.so.1", "librccl.so"}) if (!h) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h) { snprintf(why, sizeof why, "librccl.so not found (L2R_RCCL_LIB, $ROCM_PATH/lib, loader path): %s", dlerror()); return why; }
    RcclApi a; a.h = h;
#define L2R_RCCL_SYM(field, sym) do { a.field = (decltype(a.field))dlsym(h, #sym); if (!a.field) { snprintf(why, sizeof why, "%s is missing from librccl", #sym); return why; } } while (0)
    L2R_RCCL_SYM(GetUniqueId, ncclGetUniqueId); L2R_RCCL_SYM(CommInitRank, ncclCommInitRank); L2R_RCCL_SYM(CommDestroy, ncclCommDestroy);
    L2R_RCCL_SYM(CommCount, ncclCommCount); L2R_RCCL_SYM(CommUserRank, ncclCommUserRank); L2R_RCCL_SYM(CommCuDevice, ncclCommCuDevice);
    L2R_RCCL_SYM(AllGather, ncclAllGather); L2R_RCCL_SYM(Send, ncclSend); L2R_RCCL_SYM(Recv, ncclRecv);
    L2R_RCCL_SYM(GroupStart, ncclGroupStart); L2R_RCCL_SYM(GroupEnd, ncclGroupEnd); L2R_RCCL_SYM(GetErrorString, ncclGetErrorString);
#undef L2R_RCCL_SYM
    g_rccl = a;
    return nullptr;
}
// (the calls below go through the table)
#define ncclGetUniqueId g_rccl.GetUniqueId
#define ncclCommInitRank g_rccl.CommInitRank
#define ncclCommDestroy g_rccl.CommDestroy
#define ncclCommCount g_rccl.CommCount
#define ncclCommUserRank g_rccl.CommUserRank
#define ncclCommCuDevice g_rccl.CommCuDevice
#define ncclAllGather g_rccl.AllGather
#define ncclSend g_rccl.Send
#define ncclRecv g_rccl.Recv
#define ncclGroupStart g_rccl.GroupStart
#define ncclGroupEnd g_rccl.GroupEnd
#define ncclGetErrorString g_rccl.GetErrorString

struct l2r_xchg {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    l2r_ctx *ctx = nullptr;
};

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return fail(-3, "[%s] %s: %s", __func__, #expr, ncclGetErrorString(r_));  \
    } while (0)

extern "C" {

int l2r_xchg_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int l2r_xchg_unique_id(void *id_out)
{
    if (!id_out) return fail(-1, "[l2r_xchg_unique_id] null argument");
    if (const char *why = rccl_load()) return fail(-3, "[l2r_xchg_unique_id] %s", why);
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return 0;
}

l2r_xchg *l2r_xchg_create(l2r_ctx *c, int rank, int world, const void *id_in)
{
    if (!c || !id_in || world < 1 || rank < 0 || rank >= world) { fail(-1, "[l2r_xchg_create] bad argument"); return nullptr; }
    if (hipSetDevice(c->device) != hipSuccess) { fail(-2, "[l2r_xchg_create] hipSetDevice failed"); return nullptr; }
    if (const char *why = rccl_load()) { fail(-3, "[l2r_xchg_create] %s", why); return nullptr; }
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    l2r_xchg *x = new l2r_xchg();
    x->rank = rank; x->world = world; x->ctx = c;
    const ncclResult_t r = ncclCommInitRank(&x->comm, world, id, rank);
    if (r != ncclSuccess) { fail(-3, "[l2r_xchg_create] ncclCommInitRank (rank %d of %d): %s", rank, world, ncclGetErrorString(r)); delete x; return nullptr; }
    {   // what the communicator holds, said once per rank (the first run on several physical GPUs proves by it that RCCL had them all)
        int n = -1, me = -1, dev = -1; char bus[32] = "?";
        (void)ncclCommCount(x->comm, &n); (void)ncclCommUserRank(x->comm, &me); (void)ncclCommCuDevice(x->comm, &dev);
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, c->device) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof bus, "?"); }
        fprintf(stderr, "[l2r_xchg_create] RCCL communicator: rank %d of %d, device %d (bus %s)\n", me, n, dev, bus);
    }
    return x;
}

// Every rank says whether it can go on (0) before any of them enters a send / receive group: a rank that left alone -- rank 0 with
// buffers too small -- would leave its peers blocked in ncclSend.  Returns the first non-zero status of the world (on every rank).
static int xchg_agree(l2r_xchg *x, hipStream_t s, long long status, const char *who)
{
    const int W = x->world;
    DevBuf<long long> d;
    if (d.ensure((size_t)W + 1)) return -2;
    HIP_TRY(hipMemcpyAsync(d.p + W, &status, 8, hipMemcpyHostToDevice, s));
    NCCL_TRY(ncclAllGather(d.p + W, d.p, 1, ncclInt64, x->comm, s));
    std::vector<long long> all((size_t)W);
    HIP_TRY(hipMemcpyAsync(all.data(), d.p, (size_t)W * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    d.release();
    for (int k = 0; k < W; ++k)
        if (all[(size_t)k] != 0) {
            if (k == x->rank) return (int)status;              // (its own message stands)
            return fail((int)all[(size_t)k], "[%s] rank %d cannot take part (status %lld): nothing was exchanged", who, k, all[(size_t)k]);
        }
    return 0;
}

void l2r_xchg_destroy(l2r_xchg *x)
{
    if (!x) return;
    if (x->comm && g_rccl.h) (void)ncclCommDestroy(x->comm);
    delete x;
}

/* Every rank calls this behind its l2r_run + l2r_sync.  The ranks' {reads, exons} are all-gathered; then every array of the per-read
 * results goes from the engines' HBM to rank 0's (ncclSend / ncclRecv inside one group: a gatherv, each rank's part at the place its
 * shard has in read order), and rank 0 copies them to the host: res (rank 0 only; capacities as for l2r_download) receives the
 * results of ALL ranks as one read-order set -- ex_off[0 .. total reads] with global exon offsets.  counts_out: world x 2 (any rank). */
int l2r_xchg_gather_results(l2r_xchg *x, l2r_result *res, int64_t *counts_out)
{
    if (!x || !x->ctx) return fail(-1, "[l2r_xchg_gather_results] null argument");
    l2r_ctx *c = x->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int rc = fetch_totals(c);
    if (rc) return rc;
    if (!(c->want & L2R_WANT_RESULTS)) return fail(-1, "[l2r_xchg_gather_results] the run did not produce the per-read results (l2r_set_outputs)");
    const int W = x->world;
    hipStream_t s = c->stream;
    // ---- sizes
    DevBuf<long long> d_cnt;
    if (d_cnt.ensure((size_t)2 * (W + 1))) return -2;
    const long long mine[2] = {(long long)c->n_reads, (long long)c->h_totals[0]};
    HIP_TRY(hipMemcpyAsync(d_cnt.p + 2 * W, mine, sizeof mine, hipMemcpyHostToDevice, s));
    NCCL_TRY(ncclAllGather(d_cnt.p + 2 * W, d_cnt.p, 2, ncclInt64, x->comm, s));
    std::vector<long long> cnt((size_t)2 * W);
    HIP_TRY(hipMemcpyAsync(cnt.data(), d_cnt.p, cnt.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    d_cnt.release();
    std::vector<long long> r_at((size_t)W + 1, 0), x_at((size_t)W + 1, 0);
    for (int k = 0; k < W; ++k) { r_at[k + 1] = r_at[k] + cnt[2 * k]; x_at[k + 1] = x_at[k] + cnt[2 * k + 1]; }
    if (counts_out) for (int k = 0; k < 2 * W; ++k) counts_out[k] = cnt[k];
    const long long R = r_at[W], X = x_at[W];
    // ---- the arrays: {device pointer of this rank, bytes per element, per-read (0) or per-exon (1)}
    struct Part { const void *src; size_t width; int per_exon; };
    const Part parts[6] = {{c->ex_off.p, 4, 0}, {c->info.p, 4, 0}, {c->ref_tx.p, 4, 0}, {c->ex_start.p, 4, 1}, {c->ex_end.p, 4, 1}, {c->ex_flag.p, 1, 1}};
    DevBuf<uint8_t> g[6];
    long long status = 0;
    if (x->rank == 0) {
        if (!res) status = fail(-1, "[l2r_xchg_gather_results] rank 0 needs a result to fill");
        else if (res->n_reads < R || res->ex_cap < X) status = fail(-4, "[l2r_xchg_gather_results] result buffers too small (%lld reads, %lld exons)", R, X);
        else for (int a = 0; a < 6 && !status; ++a) if (g[a].ensure((size_t)(parts[a].per_exon ? X : R) * parts[a].width + 16)) status = -2;
    }
    if ((rc = xchg_agree(x, s, status, "l2r_xchg_gather_results"))) { for (int a = 0; a < 6; ++a) g[a].release(); return rc; }
    NCCL_TRY(ncclGroupStart());
    for (int a = 0; a < 6; ++a) {
        const std::vector<long long> &at = parts[a].per_exon ? x_at : r_at;
        if (x->rank == 0) {
            for (int k = 1; k < W; ++k) {
                const size_t n = (size_t)(at[k + 1] - at[k]) * parts[a].width;
                if (n) NCCL_TRY(ncclRecv(g[a].p + (size_t)at[k] * parts[a].width, n, ncclUint8, k, x->comm, s));
            }
        } else {
            const size_t n = (size_t)(at[x->rank + 1] - at[x->rank]) * parts[a].width;
            if (n) NCCL_TRY(ncclSend(parts[a].src, n, ncclUint8, 0, x->comm, s));
        }
    }
    NCCL_TRY(ncclGroupEnd());
    if (x->rank == 0) {
        // (rank 0's own part: the head of every array)
        for (int a = 0; a < 6; ++a) {
            const size_t n = (size_t)((parts[a].per_exon ? x_at[1] : r_at[1])) * parts[a].width;
            if (n) HIP_TRY(hipMemcpyAsync(g[a].p, parts[a].src, n, hipMemcpyDeviceToDevice, s));
        }
        std::vector<uint32_t> off((size_t)R + 1);
        HIP_TRY(hipMemcpyAsync(off.data(), g[0].p, (size_t)R * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(res->info, g[1].p, (size_t)R * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(res->ref_tx, g[2].p, (size_t)R * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(res->ex_start, g[3].p, (size_t)X * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(res->ex_end, g[4].p, (size_t)X * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(res->ex_flag, g[5].p, (size_t)X, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        // shard-local exclusive offsets -> global ones
        for (int k = 0; k < W; ++k)
            for (long long i = r_at[k]; i < r_at[k + 1]; ++i) res->ex_off[i] = (int64_t)off[(size_t)i] + x_at[k];
        res->ex_off[R] = X;
        res->n_reads = R; res->n_exons = X;
        // what the totals say against what the reads say (l2r_download checks the same)
        for (long long i = 0; i < R; ++i)
            if (res->ex_off[i] + (int64_t)(res->info[i] >> 8) != res->ex_off[i + 1]) return fail(-5, "[l2r_xchg_gather_results] exon counts do not add up at read %lld", i);
    } else HIP_TRY(hipStreamSynchronize(s));
    for (int a = 0; a < 6; ++a) g[a].release();
    return 0;
}

/* The accepted-novel records alone (all that the order-dependent merge needs when no output wants every read): every
 * rank calls this behind its l2r_run + l2r_sync with L2R_WANT_ACCEPTED set.  The ranks' {records, exons, tiles} are all-gathered; the
 * five arrays of every rank's list and its tiles' first-record slots go from HBM to rank 0 (ncclSend / ncclRecv in one group), which
 * puts each rank's chunks into read order (order_accepted) one rank behind the other: acc (rank 0 only; capacities for the records /
 * exons of ALL ranks) = the accepted reads of the whole input in read order, read indices global (first_read_index of the uploads).
 * counts_out: world x 2 {records, exons} (any rank; may be NULL). */
int l2r_xchg_gather_accepted(l2r_xchg *x, l2r_accepted *acc, int64_t *counts_out)
{
    if (!x || !x->ctx) return fail(-1, "[l2r_xchg_gather_accepted] null argument");
    l2r_ctx *c = x->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int rc = fetch_totals(c);
    if (rc) return rc;
    if (!(c->want & L2R_WANT_ACCEPTED)) return fail(-1, "[l2r_xchg_gather_accepted] the accepted list was not requested (l2r_set_outputs)");
    const int W = x->world;
    hipStream_t s = c->stream;
    DevBuf<long long> d_cnt;
    if (d_cnt.ensure((size_t)3 * (W + 1))) return -2;
    const long long mine[3] = {(long long)c->h_totals[1], (long long)c->h_totals[2], (long long)c->n_tiles};
    HIP_TRY(hipMemcpyAsync(d_cnt.p + 3 * W, mine, sizeof mine, hipMemcpyHostToDevice, s));
    NCCL_TRY(ncclAllGather(d_cnt.p + 3 * W, d_cnt.p, 3, ncclInt64, x->comm, s));
    std::vector<long long> cnt((size_t)3 * W);
    HIP_TRY(hipMemcpyAsync(cnt.data(), d_cnt.p, cnt.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    d_cnt.release();
    std::vector<long long> m_at((size_t)W + 1, 0), x_at((size_t)W + 1, 0), t_at((size_t)W + 1, 0);
    for (int k = 0; k < W; ++k) { m_at[k + 1] = m_at[k] + cnt[3 * k]; x_at[k + 1] = x_at[k] + cnt[3 * k + 1]; t_at[k + 1] = t_at[k] + cnt[3 * k + 2]; }
    if (counts_out) for (int k = 0; k < W; ++k) { counts_out[2 * k] = cnt[3 * k]; counts_out[2 * k + 1] = cnt[3 * k + 1]; }
    const long long M = m_at[W], X = x_at[W], T = t_at[W];
    // {device pointer of this rank, bytes per element, counted by records (0) / exons (1) / tiles (2)}
    struct Part { const void *src; size_t width; int by; };
    const Part parts[6] = {{c->acc_rec.p, sizeof(AccRec), 0}, {c->acc_ex_off.p, 4, 0}, {c->tile_rchunk.p, 4, 2}, {c->acc_start.p, 4, 1}, {c->acc_end.p, 4, 1}, {c->acc_flag.p, 1, 1}};
    auto at_of = [&](int by) -> const std::vector<long long> & { return by == 0 ? m_at : (by == 1 ? x_at : t_at); };
    DevBuf<uint8_t> g[6];
    long long status = 0;
    if (x->rank == 0) {
        if (!acc) status = fail(-1, "[l2r_xchg_gather_accepted] rank 0 needs a list to fill");
        else if (acc->n_reads < M || acc->ex_cap < X) status = fail(-4, "[l2r_xchg_gather_accepted] buffers too small (%lld records, %lld exons)", M, X);
        else for (int a = 0; a < 6 && !status; ++a) if (g[a].ensure((size_t)at_of(parts[a].by)[W] * parts[a].width + 16)) status = -2;
    }
    if ((rc = xchg_agree(x, s, status, "l2r_xchg_gather_accepted"))) { for (int a = 0; a < 6; ++a) g[a].release(); return rc; }
    NCCL_TRY(ncclGroupStart());
    for (int a = 0; a < 6; ++a) {
        const std::vector<long long> &at = at_of(parts[a].by);
        // (a rank without accepted reads has not made the arrays: its tiles' slots do not matter either)
        if (x->rank == 0) {
            for (int k = 1; k < W; ++k) {
                const size_t n = cnt[3 * k] ? (size_t)(at[k + 1] - at[k]) * parts[a].width : 0;
                if (n) NCCL_TRY(ncclRecv(g[a].p + (size_t)at[k] * parts[a].width, n, ncclUint8, k, x->comm, s));
            }
        } else {
            const size_t n = cnt[3 * x->rank] ? (size_t)(at[x->rank + 1] - at[x->rank]) * parts[a].width : 0;
            if (n) NCCL_TRY(ncclSend(parts[a].src, n, ncclUint8, 0, x->comm, s));
        }
    }
    NCCL_TRY(ncclGroupEnd());
    if (x->rank == 0) {
        for (int a = 0; a < 6; ++a) {
            const size_t n = cnt[0] ? (size_t)at_of(parts[a].by)[1] * parts[a].width : 0;
            if (n) HIP_TRY(hipMemcpyAsync(g[a].p, parts[a].src, n, hipMemcpyDeviceToDevice, s));
        }
        std::vector<AccRec> rec((size_t)M);
        std::vector<uint32_t> off((size_t)M), starts((size_t)T);
        std::vector<int32_t> xs((size_t)X), xe((size_t)X);
        std::vector<uint8_t> xf((size_t)X);
        if (M) {
            HIP_TRY(hipMemcpyAsync(rec.data(), g[0].p, (size_t)M * sizeof(AccRec), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(off.data(), g[1].p, (size_t)M * 4, hipMemcpyDeviceToHost, s));
            if (T) HIP_TRY(hipMemcpyAsync(starts.data(), g[2].p, (size_t)T * 4, hipMemcpyDeviceToHost, s));
        }
        if (X) {
            HIP_TRY(hipMemcpyAsync(xs.data(), g[3].p, (size_t)X * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(xe.data(), g[4].p, (size_t)X * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(xf.data(), g[5].p, (size_t)X, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipStreamSynchronize(s));
        int64_t at_r = 0, at = 0;
        for (int k = 0; k < W; ++k) {
            const long long Mk = cnt[3 * k], Xk = cnt[3 * k + 1];
            if (!Mk) continue;
            std::vector<uint32_t> st(starts.begin() + t_at[k], starts.begin() + t_at[k + 1]);
            rc = order_accepted(rec.data() + m_at[k], off.data() + m_at[k], st, xs.data() + x_at[k], xe.data() + x_at[k], xf.data() + x_at[k], Mk, Xk, acc, at_r, at);
            if (rc) return rc;
        }
        acc->ex_off[M] = at;
        acc->n_reads = M; acc->n_exons = X;
        // the shards follow each other in read order: so do the global read indices
        for (long long i = 1; i < M; ++i) {
            const uint64_t a0 = ((uint64_t)acc->rec[i - 1].read_hi << 32) | acc->rec[i - 1].read_lo, a1 = ((uint64_t)acc->rec[i].read_hi << 32) | acc->rec[i].read_lo;
            if (a1 <= a0) return fail(-5, "[l2r_xchg_gather_accepted] records out of read order at %lld (do the uploads say their first_read_index?)", i);
        }
    } else HIP_TRY(hipStreamSynchronize(s));
    for (int a = 0; a < 6; ++a) g[a].release();
    return 0;
}

}  // extern "C"
