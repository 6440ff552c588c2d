// fzp_comm.hip -- the ONE exchange step of the multi-GPU path behind the C-ABI: all-gather of rid_to_phase records over RCCL.
//
// Reference: get_rid_to_phase_all (falcon_unzip/unzip.py:303-314) concatenates every rid_to_phase.<ctg> in sorted-path
// order.  Here every rank holds the fixed 16-byte records of its contigs; one ncclAllGather of the counts, one of the
// payload padded to the largest count, then every rank orders the records by (contig index, pread id) = the
// order of rid_to_phase.all.  RCCL is loaded at run time (librccl.so.1, the ROCm one): the library itself carries no link
// dependency on it, and a host in any language reaches the collective through these four calls.  falcon_unzip_amd/dist.py
// does the same over torch.distributed (gloo in CPU tests).
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <thread>

#include "fzp_common.h"

namespace {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclUint8 = 1, ncclUint64 = 5 };   // nccl.h ncclDataType_t
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string load_error, path;
};
Rccl &rccl_state() { static Rccl r; return r; }
std::string rccl_why() { return rccl_state().load_error.empty() ? std::string("not found") : rccl_state().load_error; }
Rccl *rccl() {
    Rccl &r = rccl_state();
    static bool tried = false;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!tried) {
        tried = true;
        const char *names[] = {getenv("FZP_RCCL_LIB"), "librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n) continue;
            r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
            const char *e = dlerror();                       // read once: a second dlerror() returns NULL
            r.load_error = e ? e : "not found";
        }
        if (r.h) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
            r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
            r.CommCount = (decltype(r.CommCount))dlsym(r.h, "ncclCommCount");
            r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.h, "ncclCommUserRank");
            r.GetVersion = (decltype(r.GetVersion))dlsym(r.h, "ncclGetVersion");
            // WHICH library answered: a process that imported torch already has torch's own librccl.so mapped under the same soname, and dlopen hands that one back
            Dl_info di;
            if (r.AllGather && dladdr((void *)r.AllGather, &di) && di.dli_fname) r.path = di.dli_fname;
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) { dlclose(r.h); r.h = nullptr; r.load_error = "a needed nccl* symbol is missing"; }
        }
    }
    return r.h ? &r : nullptr;
}
int rccl_missing(const char *who) {
    Rccl *R = rccl();
    if (R) return FZP_OK;
    fzp_set_error("%s: RCCL (librccl.so.1) could not be loaded: %s", who, rccl_why().c_str());
    return FZP_ENODEVICE;
}
// how long a rank waits inside the communicator's creation or the collective before it gives up (FZP_COMM_TIMEOUT_S, default 600; 0 = for ever): a rank stuck there
// would otherwise hang the node, and the blind N-GPU run could not say why
int comm_timeout_s() {
    if (const char *e = getenv("FZP_COMM_TIMEOUT_S")) { const long v = atol(e); if (v >= 0) return (int)v; }
    return 600;
}
int nccl_fail(Rccl *R, const char *what, ncclResult_t rc) {
    fzp_set_error("%s: %s", what, R && R->GetErrorString ? R->GetErrorString(rc) : "RCCL error");
    return FZP_EDEVICE;
}
}  // namespace

// which RCCL this process bound (path as dladdr sees ncclAllGather, version as ncclGetVersion says): for the bench line of a run nobody watches
extern "C" int fzp_comm_library(char *path, size_t cap, int *version) {
    FZP_TRY(rccl_missing("fzp_comm_library"));
    Rccl *R = rccl();
    if (path && cap) { snprintf(path, cap, "%s", R->path.c_str()); }
    if (version) { *version = 0; if (R->GetVersion) (void)R->GetVersion(version); }
    return FZP_OK;
}

struct fzp_comm {
    fzp_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool dead = false;      // a collective of this communicator ran into FZP_COMM_TIMEOUT_S: it may still be in flight -- the communicator is not used again, its buffers are not given back
};

extern "C" int fzp_comm_unique_id(char id[FZP_COMM_ID_BYTES]) {
    FZP_TRY(rccl_missing("fzp_comm_unique_id"));
    Rccl *R = rccl();
    ncclUniqueId u;
    ncclResult_t rc = R->GetUniqueId(&u);
    if (rc) return nccl_fail(R, "ncclGetUniqueId", rc);
    memcpy(id, u.internal, sizeof u.internal);
    return FZP_OK;
}

extern "C" int fzp_comm_create(fzp_ctx *ctx, int rank, int world, const char id[FZP_COMM_ID_BYTES], fzp_comm **out) {
    if (!ctx || !out || !id || world < 1 || rank < 0 || rank >= world) { fzp_set_error("fzp_comm_create: bad arguments"); return FZP_EINVAL; }
    *out = nullptr;
    FZP_TRY(rccl_missing("fzp_comm_create"));
    Rccl *R = rccl();
    FZP_TRY(fzp_bind(ctx));
    ncclUniqueId u;
    memcpy(u.internal, id, sizeof u.internal);
    fzp_comm *c = new fzp_comm();
    c->ctx = ctx; c->rank = rank; c->world = world;
    // ncclCommInitRank is collective: it returns when every rank has called it.  It runs on a helper thread so that a rank whose peers never arrive can say so and
    // leave (the caller falls back to another gather or fails the job) instead of hanging the node; the helper is abandoned where it is.
    struct Shared { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t rc = 0; ncclComm_t comm = nullptr; };
    auto sh = std::make_shared<Shared>();
    const int device = ctx->device;
    std::thread([sh, R, world, u, rank, device]() {
        (void)hipSetDevice(device);
        ncclComm_t cm = nullptr;
        const ncclResult_t r2 = R->CommInitRank(&cm, world, u, rank);
        { std::lock_guard<std::mutex> lk(sh->mu); sh->rc = r2; sh->comm = cm; sh->done = true; }
        sh->cv.notify_all();
    }).detach();
    {
        std::unique_lock<std::mutex> lk(sh->mu);
        const int limit = comm_timeout_s();
        if (limit > 0) {
            if (!sh->cv.wait_for(lk, std::chrono::seconds(limit), [&] { return sh->done; })) {
                delete c;
                fzp_set_error("ncclCommInitRank (rank %d of %d, RCCL %s) did not return within %d s (FZP_COMM_TIMEOUT_S): a peer never arrived", rank, world, R->path.c_str(), limit);
                return FZP_EDEVICE;
            }
        } else sh->cv.wait(lk, [&] { return sh->done; });
        c->comm = sh->comm;
        if (sh->rc) { const ncclResult_t r2 = sh->rc; delete c; return nccl_fail(R, "ncclCommInitRank", r2); }
    }
    *out = c;
    return FZP_OK;
}

// rank and size as the COMMUNICATOR reports them (ncclCommUserRank / ncclCommCount), not as the caller passed them in: what bench.py
// prints as `rccl_ranks`
extern "C" int fzp_comm_ranks(fzp_comm *c, int *rank, int *world) {
    if (!c || !rank || !world) { fzp_set_error("fzp_comm_ranks: bad arguments"); return FZP_EINVAL; }
    Rccl *R = rccl();
    *rank = c->rank; *world = c->world;
    if (R && R->CommCount && R->CommUserRank && c->comm) {
        ncclResult_t rc = R->CommCount(c->comm, world);
        if (rc) return nccl_fail(R, "ncclCommCount", rc);
        rc = R->CommUserRank(c->comm, rank);
        if (rc) return nccl_fail(R, "ncclCommUserRank", rc);
    }
    return FZP_OK;
}

extern "C" void fzp_comm_destroy(fzp_comm *c) {
    if (!c) return;
    Rccl *R = rccl();
    if (R && c->comm && !c->dead) { (void)fzp_bind(c->ctx); (void)hipStreamSynchronize(c->ctx->stream); (void)R->CommDestroy(c->comm); }      // (a dead one is left where it is: its collective may never end)
    delete c;
}

// hipStreamSynchronize with a deadline: a collective a peer never joins would keep the stream busy for ever
static int wait_stream(hipStream_t st, const char *what) {
    const int limit = comm_timeout_s();
    if (limit <= 0) { FZP_HIP(hipStreamSynchronize(st)); return FZP_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return FZP_OK;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); fzp_set_error("%s: %s", what, hipGetErrorString(e)); return FZP_EDEVICE; }
        (void)hipGetLastError();
        const auto waited = std::chrono::steady_clock::now() - t0;
        if (waited > std::chrono::seconds(limit)) { fzp_set_error("%s did not finish within %d s (FZP_COMM_TIMEOUT_S): a peer never joined the collective", what, limit); return FZP_EDEVICE; }
        // (r6: a gather of a few megabytes is through in tens of microseconds -- look again at once for the first 300 us, then every 50; ADVICE r5: naps of 200 us from the
        //  first look on added their length to every gather of every step)
        if (waited > std::chrono::microseconds(300)) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
extern "C" int fzp_allgather_rid_to_phase(fzp_comm *c, const fzp_r2p *local, int64_t n_local, fzp_r2p **all, int64_t *n_all) {
    if (!c || !all || !n_all || n_local < 0 || (n_local && !local)) { fzp_set_error("fzp_allgather_rid_to_phase: bad arguments"); return FZP_EINVAL; }
    Rccl *R = rccl();
    if (!R) { fzp_set_error("RCCL could not be loaded"); return FZP_ENODEVICE; }
    if (c->dead) { fzp_set_error("fzp_allgather_rid_to_phase: an earlier collective of this communicator timed out; it is not used again"); return FZP_EDEVICE; }
    fzp_ctx *ctx = c->ctx;
    // what a timed-out collective may still be writing is never handed back to the pool: the buffers are dropped where they are, the communicator is marked
    auto give_up = [&](auto &...bufs) { c->dead = true; ((bufs.p = nullptr, bufs.n = 0), ...); };
    FZP_TRY(fzp_bind(ctx));
    hipStream_t st = ctx->stream;
    const int W = c->world;
    // 1. counts
    DevBuf<uint64_t> d_cnt, d_cnts;
    FZP_TRY(d_cnt.alloc(1)); FZP_TRY(d_cnts.alloc((size_t)W));
    const uint64_t mine = (uint64_t)n_local;
    FZP_HIP(hipMemcpyAsync(d_cnt.p, &mine, 8, hipMemcpyHostToDevice, st));
    ncclResult_t rc = R->AllGather(d_cnt.p, d_cnts.p, 1, ncclUint64, c->comm, st);
    if (rc) return nccl_fail(R, "ncclAllGather(counts)", rc);
    std::vector<uint64_t> cnts((size_t)W);
    FZP_TRY(d_cnts.download(cnts.data(), (size_t)W, st));
    { const int wrc = wait_stream(st, "ncclAllGather(counts)"); if (wrc != FZP_OK) { give_up(d_cnt, d_cnts); return wrc; } }
    uint64_t mx = 1, total = 0;
    for (auto v : cnts) { mx = std::max(mx, v); total += v; }
    // 2. payload, padded to the largest shard
    DevBuf<fzp_r2p> d_loc, d_all;
    FZP_TRY(d_loc.alloc((size_t)mx)); FZP_TRY(d_all.alloc((size_t)mx * (size_t)W));
    FZP_HIP(hipMemsetAsync(d_loc.p, 0, (size_t)mx * sizeof(fzp_r2p), st));
    if (n_local) FZP_HIP(hipMemcpyAsync(d_loc.p, local, (size_t)n_local * sizeof(fzp_r2p), hipMemcpyHostToDevice, st));
    rc = R->AllGather(d_loc.p, d_all.p, (size_t)mx * sizeof(fzp_r2p), ncclUint8, c->comm, st);
    if (rc) return nccl_fail(R, "ncclAllGather(records)", rc);
    fzp_r2p *out = (fzp_r2p *)malloc((total ? total : 1) * sizeof(fzp_r2p));
    if (!out) return FZP_ENOMEM;
    size_t at = 0;
    for (int r = 0; r < W; r++) {
        if (cnts[(size_t)r] && hipMemcpyAsync(out + at, d_all.p + (size_t)r * mx, (size_t)cnts[(size_t)r] * sizeof(fzp_r2p), hipMemcpyDeviceToHost, st) != hipSuccess) { free(out); fzp_set_error("D2H copy failed"); return FZP_EDEVICE; }
        at += (size_t)cnts[(size_t)r];
    }
    { const int wrc = wait_stream(st, "ncclAllGather(records)"); if (wrc != FZP_OK) { give_up(d_cnt, d_cnts, d_loc, d_all); return wrc; } }      // (`out` too: copies into it may still be queued)
    // 3. the order of rid_to_phase.all: sorted per-contig paths (unzip.py:306-307) = contig index, then pread id
    // (r6: a rank's records arrive in that order, and ranks that hold consecutive contig ranges -- the usual deal -- arrive in order as a whole: look first, then merge the
    // ranks' runs, and sort only what is left.  Eight ranks x 40 000 records: std::sort took ~10 ms of every step on every rank.)
    {
        const auto less = [](const fzp_r2p &a, const fzp_r2p &b) { return a.ctg != b.ctg ? a.ctg < b.ctg : a.arid < b.arid; };
        if (!std::is_sorted(out, out + total, less)) {
            std::vector<size_t> cut{0};
            for (int r = 0; r < W; r++) cut.push_back(cut.back() + (size_t)cnts[(size_t)r]);
            bool runs_sorted = true;
            for (int r = 0; r < W && runs_sorted; r++) runs_sorted = std::is_sorted(out + cut[(size_t)r], out + cut[(size_t)r + 1], less);
            if (runs_sorted) {
                for (size_t width = 1; width < (size_t)W; width *= 2)
                    for (size_t r = 0; r + width < (size_t)W; r += 2 * width)
                        std::inplace_merge(out + cut[r], out + cut[r + width], out + cut[std::min(r + 2 * width, (size_t)W)], less);
            } else std::sort(out, out + total, less);
        }
    }
    *all = out;
    *n_all = (int64_t)total;
    return FZP_OK;
}
