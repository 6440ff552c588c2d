// fzp_host.hip -- host side of libfzphase.so: context, errors, profiling, the SAM record parser
// (phasing.py:42-75), the text serializers (the reference's `print >>f` statements) and
// get_phasing_readmap (phasing_readmap.py:8-51, pure host bookkeeping).
#include <algorithm>
#include <chrono>
#include <sched.h>
#include <thread>
#include <unordered_map>

#include "fzp_common.h"

static thread_local char g_err[1024] = "";

void fzp_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *fzp_last_error(void) { return g_err; }
extern "C" const char *fzp_version(void) { return "fzphase 0.1.0 (gfx950)"; }
extern "C" void fzp_free(void *p) { free(p); }

// ---------------------------------------------------------------- caching device allocator
struct DevPool {
    std::mutex mu;
    int device = 0;
    std::multimap<size_t, void *> free_blocks;      // bucket size -> block
};
namespace {
struct LiveBlock { std::shared_ptr<DevPool> pool; size_t bucket; };
std::mutex g_live_mu;
std::unordered_map<void *, LiveBlock> g_live;        // every block handed out, whichever pool it belongs to
std::mutex g_def_mu;
std::map<int, std::shared_ptr<DevPool>> g_default;   // per device: for threads that never bound a ctx
thread_local std::shared_ptr<DevPool> t_pool;
std::mutex g_pools_mu;
std::vector<std::weak_ptr<DevPool>> g_pools;         // every pool ever made (contexts, lanes, defaults): the out-of-memory path trims ALL pools of the device
std::shared_ptr<DevPool> new_pool(int dev) {
    auto p = std::make_shared<DevPool>();
    p->device = dev;
    std::lock_guard<std::mutex> lk(g_pools_mu);
    size_t keep = 0;
    for (auto &w : g_pools) if (!w.expired()) g_pools[keep++] = w;
    g_pools.resize(keep);
    g_pools.push_back(p);
    return p;
}
size_t bucket_of(size_t bytes) {
    if (bytes < 256) bytes = 256;
    size_t p2 = 256;
    while (p2 < bytes) p2 <<= 1;
    if (p2 <= (1u << 20)) return p2;                 // powers of two up to 1 MiB
    size_t half = p2 >> 1;                           // above: 1/8-octave steps (<= 12.5 % slack)
    for (int k = 1; k <= 8; k++) { size_t b = half + (half >> 3) * k; if (b >= bytes) return b; }
    return p2;
}
std::shared_ptr<DevPool> current_pool() {
    if (t_pool) return t_pool;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_def_mu);
    auto &p = g_default[dev];
    if (!p) p = new_pool(dev);
    return p;
}
void trim_pool(DevPool &P) {
    std::multimap<size_t, void *> drop;
    { std::lock_guard<std::mutex> lk(P.mu); drop.swap(P.free_blocks); }
    for (auto &kv : drop) (void)hipFree(kv.second);           // hipFree waits for the device: a cached block that is still read by a queued kernel is safe to drop
}
// idle sibling pools (the lane contexts of fzp_phase_contigs, a second engine) may hold most of HBM in their caches: give all of it back
void trim_device_pools(int dev) {
    std::vector<std::shared_ptr<DevPool>> all;
    {
        std::lock_guard<std::mutex> lk(g_pools_mu);
        for (auto &w : g_pools) if (auto p = w.lock()) if (p->device == dev) all.push_back(p);
    }
    for (auto &p : all) trim_pool(*p);
}
}  // namespace
// threads this process may really use: the affinity mask and the cgroup CPU quota, not the machine's thread count (a container that shows 256 hardware threads
// behind a 16-CPU quota is throttled to a crawl by 256 busy threads)
int usable_cores() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t cs;
    if (sched_getaffinity(0, sizeof cs, &cs) == 0) n = std::min(n, std::max(1, CPU_COUNT(&cs)));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota> <period>" or "max <period>"
        char q[64]; long long per = 0;
        if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) n = std::min<long long>(n, std::max<long long>(1, (atoll(q) + per - 1) / per));
        fclose(f);
    } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
        long long quota = -1, per = 0;
        if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
        fclose(f1);
        if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%lld", &per) != 1) per = 0; fclose(f2); }
        if (quota > 0 && per > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + per - 1) / per));
    }
    return n;
}

// ... and this rank's share of them: the ranks of a node (LOCAL_WORLD_SIZE, as torch.distributed.run sets it) divide the usable cores among themselves.  Sizes the
// host thread pools: eight ranks behind a 16-CPU quota get two workers each instead of sixteen that take turns.
int cores_per_rank() {
    int ranks = 1;
    if (const char *w = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(w));
    return std::max(1, usable_cores() / ranks);
}

// ---------------------------------------------------------------- fzp_fetch (fzp_common.h)
namespace {
constexpr int FETCH_MAX = 8;                                       // pieces per fetch: 8 x 64 words of payload, the sequence number behind them
struct FetchArgs { const uint32_t *src[FETCH_MAX]; int n[FETCH_MAX]; };
__global__ void __launch_bounds__(64) k_fetch_post(FetchArgs a, volatile uint32_t *slot, uint64_t seq) {
    int base = 0;
#pragma unroll
    for (int k = 0; k < FETCH_MAX; k++) {
        if ((int)threadIdx.x < a.n[k]) slot[base + (int)threadIdx.x] = a.src[k][threadIdx.x];
        base += a.n[k];
    }
    __threadfence_system();                                        // the payload is out before the number that says so
    if (threadIdx.x == 0) *(volatile uint64_t *)(slot + 64 * FETCH_MAX) = seq;
}
}  // namespace
// How the calling thread waits (r5).  The count it asks for is usually microseconds away (the stages of a step follow each other), so it looks at the mapped word for a
// short while -- a look costs nothing and answers in ~1 us -- and then stops burning its core: it hands the wait to the runtime (hipStreamSynchronize), which under the
// device's blocking-sync scheduling (fzp_ctx_create: hipDeviceScheduleBlockingSync) sleeps on the queue's completion signal.  The spin budget: FZP_FETCH_SPIN_US, default
// 60 us (tools/ubench/fetch_latency.hip: a dependent kernel answers in 12); the step's long waits -- the DP behind the plan's counts, K2 behind the record counts -- are
// what a launch thread used to spin through, 22 ms of a 22 ms step.  One fetching thread per ctx (a ctx is used by one host thread at a time: fzphase.h).
static inline void cpu_relax() {
#if !defined(__HIP_DEVICE_COMPILE__)      // (host code in a .hip file is parsed by the device pass too; the x86 builtin exists on the host side only)
    __builtin_ia32_pause();
#endif
}
static int64_t fetch_spin_ns() {
    static const int64_t v = [] { const char *e = getenv("FZP_FETCH_SPIN_US"); const long g = e ? atol(e) : 60; return (int64_t)(g < 0 ? 0 : g) * 1000; }();
    return v;
}
int fzp_fetch(fzp_ctx *ctx, hipStream_t st, const fzp_fetch_piece *pieces, int n_pieces) {
    if (!ctx || !pieces || n_pieces < 1 || n_pieces > FETCH_MAX) { fzp_set_error("fzp_fetch: bad arguments"); return FZP_EINVAL; }
    FetchArgs a;
    for (int k = 0; k < FETCH_MAX; k++) {
        a.src[k] = k < n_pieces ? (const uint32_t *)pieces[k].dev : nullptr;
        a.n[k] = k < n_pieces ? (int)(pieces[k].bytes / 4) : 0;
        if (k < n_pieces && ((pieces[k].bytes & 3) || pieces[k].bytes > 256 || !pieces[k].dev || !pieces[k].host)) { fzp_set_error("fzp_fetch: bad piece"); return FZP_EINVAL; }
    }
    if (!ctx->fetch_slot) {
        void *p = nullptr;
        // coherent + mapped, whatever HIP_HOST_COHERENT says: the device's system-scope stores must be visible to the spinning host thread as they land
        if (hipHostMalloc(&p, 4096, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); fzp_set_error("fzp_fetch: pinned allocation failed"); return FZP_ENOMEM; }
        memset(p, 0, 4096);
        ctx->fetch_slot = (uint32_t *)p;
    }
    const uint64_t seq = ++ctx->fetch_seq;
    hipLaunchKernelGGL(k_fetch_post, dim3(1), dim3(64), 0, st, a, (volatile uint32_t *)ctx->fetch_slot, seq);
    if (hipGetLastError() != hipSuccess) { fzp_set_error("fzp_fetch: launch failed"); return FZP_EDEVICE; }
    volatile uint64_t *sq = (volatile uint64_t *)(ctx->fetch_slot + 64 * FETCH_MAX);
    const int64_t budget = fetch_spin_ns();
    const auto t0 = std::chrono::steady_clock::now();
    bool have = false;
    for (uint32_t spins = 0;; spins++) {
        if (__atomic_load_n((const uint64_t *)sq, __ATOMIC_ACQUIRE) == seq) { have = true; break; }
        cpu_relax();
        if ((spins & 63) == 63 && std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() > budget) break;
    }
    if (!have) {
        // the long wait: the runtime's (a sleep under blocking-sync scheduling); a dead stream comes back as its error instead of an endless spin
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) { (void)hipGetLastError(); fzp_set_error("fzp_fetch: %s", hipGetErrorString(e)); return FZP_EDEVICE; }
        if (__atomic_load_n((const uint64_t *)sq, __ATOMIC_ACQUIRE) != seq) { fzp_set_error("fzp_fetch: the stream is idle and the words never arrived"); return FZP_EDEVICE; }
    }
    int base = 0;
    for (int k = 0; k < n_pieces; k++) { memcpy(pieces[k].host, ctx->fetch_slot + base, pieces[k].bytes); base += a.n[k]; }
    return FZP_OK;
}

// ---------------------------------------------------------------- fzp_fill (fzp_common.h)
// Several regions set to a 32-bit word each by ONE launch.  A stage that clears six counters arrays used to put twelve runtime fills on the stream (hipMemsetAsync splits an
// odd-sized region in two), 4-5 us each and each a launch the host thread pays for; regions whose size or address is not a multiple of four bytes still go to the runtime.
namespace {
constexpr int FILL_MAX = 12, FILL_TILE = 2048;      // words per workgroup
struct FillArgs { uint32_t *p[FILL_MAX]; uint32_t n[FILL_MAX]; uint32_t v[FILL_MAX]; uint32_t first[FILL_MAX + 1]; int k; };
__global__ void __launch_bounds__(256) k_fill_regions(FillArgs a) {
    int r = 0;
    while (r + 1 < a.k && blockIdx.x >= a.first[r + 1]) r++;
    uint32_t *p = a.p[r];
    const uint32_t n = a.n[r], v = a.v[r];
    uint32_t i = (blockIdx.x - a.first[r]) * FILL_TILE + threadIdx.x;
#pragma unroll
    for (int t = 0; t < FILL_TILE / 256; t++, i += 256) if (i < n) p[i] = v;
}
}  // namespace
int fzp_fill(fzp_ctx *ctx, hipStream_t st, const fzp_fill_piece *pieces, int n_pieces) {
    if (!ctx || !pieces || n_pieces < 0) { fzp_set_error("fzp_fill: bad arguments"); return FZP_EINVAL; }
    FillArgs a;
    memset(&a, 0, sizeof a);
    auto flush = [&]() -> int {
        if (a.k == 0) return FZP_OK;
        hipLaunchKernelGGL(k_fill_regions, dim3(a.first[a.k]), dim3(256), 0, st, a);
        if (hipGetLastError() != hipSuccess) { fzp_set_error("fzp_fill: launch failed"); return FZP_EDEVICE; }
        memset(&a, 0, sizeof a);
        return FZP_OK;
    };
    for (int k = 0; k < n_pieces; k++) {
        const fzp_fill_piece &f = pieces[k];
        if (f.bytes == 0) continue;
        if (!f.dev) { fzp_set_error("fzp_fill: null region"); return FZP_EINVAL; }
        const uint32_t b0 = f.word & 0xff;
        if ((f.bytes & 3) || ((uintptr_t)f.dev & 3) || f.bytes / 4 > 0xffffffffull) {      // the runtime's fill: byte-valued words only
            if (f.word != b0 * 0x01010101u) { fzp_set_error("fzp_fill: a region of odd size or address takes a byte value"); return FZP_EINVAL; }
            FZP_HIP(hipMemsetAsync(f.dev, (int)b0, f.bytes, st));
            continue;
        }
        const uint32_t words = (uint32_t)(f.bytes / 4), blocks = (words + FILL_TILE - 1) / FILL_TILE;
        a.p[a.k] = (uint32_t *)f.dev; a.n[a.k] = words; a.v[a.k] = f.word;
        a.first[a.k + 1] = a.first[a.k] + blocks;
        if (++a.k == FILL_MAX) FZP_TRY(flush());
    }
    return flush();
}

int fzp_read_back(fzp_ctx *ctx, hipStream_t st, void *host, const void *dev, size_t bytes) {
    if (bytes == 0) return FZP_OK;
    if (bytes <= 256 && !(bytes & 3)) return fzp_fetch(ctx, st, host, dev, bytes);
    FZP_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
    FZP_HIP(hipStreamSynchronize(st));
    return FZP_OK;
}

int fzp_bind(fzp_ctx *ctx) {
    FZP_HIP(hipSetDevice(ctx->device));
    t_pool = ctx->pool;
    return FZP_OK;
}
void *fzp_dev_alloc(size_t bytes) {
    std::shared_ptr<DevPool> P = current_pool();
    const size_t b = bucket_of(bytes);
    void *p = nullptr;
    {
        std::lock_guard<std::mutex> lk(P->mu);
        auto it = P->free_blocks.find(b);
        if (it != P->free_blocks.end()) { p = it->second; P->free_blocks.erase(it); }
    }
    if (!p) {
        if (hipMalloc(&p, b) != hipSuccess) {
            (void)hipGetLastError();
            trim_pool(*P);                             // retry with this pool's cache emptied ...
            if (hipMalloc(&p, b) != hipSuccess) {
                (void)hipGetLastError();
                trim_device_pools(P->device);          // ... then with every pool of the device emptied (free lists only; live blocks stay)
                if (hipMalloc(&p, b) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            }
        }
    }
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live[p] = LiveBlock{P, b};
    return p;
}
void fzp_dev_free(void *p) {
    if (!p) return;
    LiveBlock lb;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        auto it = g_live.find(p);
        if (it == g_live.end()) { (void)hipFree(p); return; }
        lb = it->second;
        g_live.erase(it);
    }
    std::lock_guard<std::mutex> lk(lb.pool->mu);
    lb.pool->free_blocks.emplace(lb.bucket, p);
}
void fzp_dev_trim() { trim_pool(*current_pool()); }

void *fzp_pinned_acquire(fzp_ctx *ctx, size_t bytes, size_t *cap) {
    if (bytes < 4096) bytes = 4096;
    {
        std::lock_guard<std::mutex> lk(ctx->pin_mu);
        size_t best = (size_t)-1;
        // best fit -- but not a block more than four times the size asked for: a 100 MB text block that settles in the 1.5 GB block of a group's FASTA files sends the next group's
        // files to hipHostMalloc (hundreds of milliseconds per GB), and the pool takes calls to converge (r6: the second call of configs[4] from files 0.69 s, the third 0.19)
        const size_t roof = bytes * 4 + (8u << 20);
        for (size_t i = 0; i < ctx->pin_free.size(); i++)
            if (ctx->pin_free[i].second >= bytes && ctx->pin_free[i].second <= roof && (best == (size_t)-1 || ctx->pin_free[i].second < ctx->pin_free[best].second)) best = i;
        if (best != (size_t)-1) {
            auto blk = ctx->pin_free[best];
            ctx->pin_free.erase(ctx->pin_free.begin() + (long)best);
            ctx->pin_live.push_back(blk);
            if (cap) *cap = blk.second;
            return blk.first;
        }
    }
    void *p = nullptr;
    const size_t want = bytes + bytes / 4 + (1 << 20);
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        std::vector<std::pair<void *, size_t>> drop;
        { std::lock_guard<std::mutex> lk(ctx->pin_mu); drop.swap(ctx->pin_free); }
        for (auto &d : drop) (void)hipHostFree(d.first);
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    std::lock_guard<std::mutex> lk(ctx->pin_mu);
    ctx->pin_live.push_back({p, want});
    if (cap) *cap = want;
    return p;
}
void fzp_pinned_release(fzp_ctx *ctx, void *p) {
    if (!p || !ctx) return;
    std::lock_guard<std::mutex> lk(ctx->pin_mu);
    for (size_t i = 0; i < ctx->pin_live.size(); i++)
        if (ctx->pin_live[i].first == p) {
            ctx->pin_free.push_back(ctx->pin_live[i]);
            ctx->pin_live.erase(ctx->pin_live.begin() + (long)i);
            return;
        }
}

// ---------------------------------------------------------------- big host -> device uploads
// The staged variant copies the caller's bytes into pinned blocks with a few threads (each with two 8 MiB blocks and its own stream)
// while earlier chunks are in flight; it is kept as an option for platforms whose pageable path is slow.
int fzp_upload_segments(fzp_ctx *ctx, void *dst_dev, const std::vector<const void *> &src, const std::vector<size_t> &dst_off, const std::vector<size_t> &len, hipStream_t st) {
    constexpr size_t CHUNK = 8u << 20;
    size_t total = 0;
    for (size_t v : len) total += v;
    // Default: hand the caller's (pageable) bytes to hipMemcpyAsync -- on this platform the runtime's own staging moves them at
    // 40-45 GB/s, under both the ROCm 7.2 runtime and the 7.0 one torch ships (profiles/r2_h2d_staging_ubench.txt, tools/e2e_sweep.py).
    // FZP_UPLOAD_MODE=staged: the library's own pinned, threaded staging below (21 ms vs 15 ms for the bench's 700 MB under 7.0, 48 ms
    // under 7.2); =register: page-lock the caller's bytes in place first.
    const char *mode = getenv("FZP_UPLOAD_MODE");
    if (!mode || !strcmp(mode, "direct") || total < (4u << 20)) {
        for (size_t k = 0; k < src.size(); k++)
            if (len[k]) FZP_HIP(hipMemcpyAsync((char *)dst_dev + dst_off[k], src[k], len[k], hipMemcpyHostToDevice, st));
        FZP_HIP(hipStreamSynchronize(st));
        return FZP_OK;
    }
    if (!strcmp(mode, "register")) {
        std::vector<void *> reg;
        int rc = FZP_OK;
        for (size_t k = 0; k < src.size() && rc == FZP_OK; k++) {
            if (!len[k]) continue;
            const uintptr_t a = (uintptr_t)src[k] & ~(uintptr_t)4095, e = ((uintptr_t)src[k] + len[k] + 4095) & ~(uintptr_t)4095;
            if (hipHostRegister((void *)a, e - a, hipHostRegisterDefault) == hipSuccess) reg.push_back((void *)a); else (void)hipGetLastError();
            if (hipMemcpyAsync((char *)dst_dev + dst_off[k], src[k], len[k], hipMemcpyHostToDevice, st) != hipSuccess) rc = FZP_EDEVICE;
        }
        if (hipStreamSynchronize(st) != hipSuccess) rc = FZP_EDEVICE;
        for (void *q : reg) (void)hipHostUnregister(q);
        if (rc) fzp_set_error("upload: a registered copy failed");
        return rc;
    }
    struct Piece { const char *s; size_t d, n; };
    std::vector<Piece> pieces;
    for (size_t k = 0; k < src.size(); k++)
        for (size_t o = 0; o < len[k]; o += CHUNK) pieces.push_back({(const char *)src[k] + o, dst_off[k] + o, std::min(CHUNK, len[k] - o)});
    int T = (int)std::min<size_t>(4, std::max<size_t>(1, std::thread::hardware_concurrency() / 2));
    if (const char *e = getenv("FZP_UPLOAD_THREADS")) { int g = atoi(e); if (g > 0 && g <= 16) T = g; }
    T = (int)std::min<size_t>((size_t)T, pieces.size());
    std::vector<int> rcs((size_t)T, FZP_OK);
    std::vector<void *> blocks((size_t)T * 2, nullptr);
    for (auto &b : blocks) { b = fzp_pinned_acquire(ctx, CHUNK, nullptr); if (!b) { for (auto q : blocks) if (q) fzp_pinned_release(ctx, q); fzp_set_error("pinned staging allocation failed"); return FZP_ENOMEM; } }
    const int device = ctx->device;
    auto work = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) { rcs[(size_t)t] = FZP_EDEVICE; return; }
        hipStream_t s2 = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) { rcs[(size_t)t] = FZP_EDEVICE; }
        bool used[2] = {false, false};
        int k = 0;
        for (size_t pi = (size_t)t; pi < pieces.size() && rcs[(size_t)t] == FZP_OK; pi += (size_t)T, k ^= 1) {
            char *blk = (char *)blocks[(size_t)t * 2 + (size_t)k];
            if (used[k] && hipEventSynchronize(ev[k]) != hipSuccess) { rcs[(size_t)t] = FZP_EDEVICE; break; }
            memcpy(blk, pieces[pi].s, pieces[pi].n);
            if (hipMemcpyAsync((char *)dst_dev + pieces[pi].d, blk, pieces[pi].n, hipMemcpyHostToDevice, s2) != hipSuccess || hipEventRecord(ev[k], s2) != hipSuccess) { rcs[(size_t)t] = FZP_EDEVICE; break; }
            used[k] = true;
        }
        if (s2 && hipStreamSynchronize(s2) != hipSuccess) rcs[(size_t)t] = FZP_EDEVICE;
        for (auto e : ev) if (e) (void)hipEventDestroy(e);
        if (s2) (void)hipStreamDestroy(s2);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    for (auto q : blocks) fzp_pinned_release(ctx, q);
    (void)hipSetDevice(ctx->device);
    for (int rc : rcs) if (rc != FZP_OK) { fzp_set_error("upload: a staged copy failed"); return rc; }
    return FZP_OK;
}

// ---------------------------------------------------------------- context
static int g_sched_rc = -1;      // what hipSetDeviceFlags answered the last time a context asked for its scheduling mode (-1: nobody asked yet)
// 0 = the runtime accepted the scheduling flag (hipSuccess), otherwise its error code; -1 = never asked
extern "C" int fzp_sched_status(void) { return g_sched_rc; }

extern "C" int fzp_ctx_create(int device_id, unsigned flags, fzp_ctx **out) {
    (void)flags;
    if (!out) return FZP_EINVAL;
    *out = nullptr;
    // (before the runtime reads its settings, i.e. before this process's first HIP call -- a no-op otherwise, and never over the user's own value)  The runtime hands every
    // queued command a completion signal from a pool of 64 and WAITS on the CPU when the pool wraps; a step queues several hundred commands ahead of the device, and the
    // waits cost a rank 7-9 ms of system time per step on the runtime's completion thread (r5: tools/runs/who_is_busy.py, host_cpu_ab2.sh)
    (void)setenv("ROC_SIGNAL_POOL_SIZE", "4096", 0);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        fzp_set_error("no HIP device available (%s); libfzphase has no CPU fallback",
                      e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return FZP_ENODEVICE;
    }
    if (device_id < 0 || device_id >= n) {
        fzp_set_error("device_id %d out of range (have %d)", device_id, n);
        return FZP_EINVAL;
    }
    FZP_HIP(hipSetDevice(device_id));
    {   // How this runtime waits is a property of the DEVICE (hipSetDeviceFlags), not of the event or the call: under the default (hipDeviceScheduleAuto) every
        // hipStreamSynchronize / hipEventSynchronize spins for as long as the GPU works -- a core per rank gone, and eight ranks share a 16-CPU quota (unzip.py:255,342-350:
        // the reference gives a phasing job its core budget too).  Blocking sync: a short look, then the thread sleeps on the completion signal.  FZP_SCHED=auto|spin|yield|blocking.
        const char *m = getenv("FZP_SCHED");
        const unsigned f = !m || !strcmp(m, "blocking") ? hipDeviceScheduleBlockingSync : !strcmp(m, "spin") ? hipDeviceScheduleSpin : !strcmp(m, "yield") ? hipDeviceScheduleYield : hipDeviceScheduleAuto;
        g_sched_rc = (int)hipSetDeviceFlags(f);
        if (g_sched_rc != (int)hipSuccess) (void)hipGetLastError();      // (a runtime that refuses keeps its default: slower to yield, not wrong)
    }
    hipDeviceProp_t prop;
    FZP_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fzp_set_error("device %d is %s; libfzphase carries gfx950 code objects only", device_id, prop.gcnArchName);
        return FZP_ENODEVICE;
    }
    fzp_ctx *c = new fzp_ctx();
    c->device = device_id;
    c->pool = new_pool(device_id);
    c->n_cu = prop.multiProcessorCount;
    t_pool = c->pool;
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se == hipSuccess) se = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
    if (se == hipSuccess) se = hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking);
    if (se != hipSuccess) {
        fzp_set_error("hipStreamCreate: %s", hipGetErrorString(se));
        delete c;
        return FZP_EDEVICE;
    }
    *out = c;
    return FZP_OK;
}

extern "C" void fzp_ctx_destroy(fzp_ctx *ctx) {
    if (!ctx) return;
    for (auto l : ctx->lanes) fzp_ctx_destroy(l);
    ctx->lanes.clear();
    fzp_writer_destroy(ctx);                         // joins outstanding background file writes
    (void)fzp_bind(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    for (auto &p : ctx->pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (auto &b : ctx->scan_tmp) b.release();
    if (ctx->fetch_slot) (void)hipHostFree(ctx->fetch_slot);
    for (auto &pb : ctx->pin_free) (void)hipHostFree(pb.first);
    for (auto &pb : ctx->pin_live) (void)hipHostFree(pb.first);      // (views handed out die with the ctx, as documented)
    if (ctx->ev_pf) (void)hipEventDestroy(ctx->ev_pf);
    if (ctx->ev_pf_done) (void)hipEventDestroy(ctx->ev_pf_done);
    if (ctx->ev_late) (void)hipEventDestroy(ctx->ev_late);
    trim_pool(*ctx->pool);
    t_pool.reset();
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    (void)hipStreamDestroy(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int fzp_mem_info(fzp_ctx *ctx, size_t *free_bytes, size_t *total_bytes) {
    if (!ctx || !free_bytes || !total_bytes) return FZP_EINVAL;
    FZP_HIP(hipSetDevice(ctx->device));
    FZP_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return FZP_OK;
}

extern "C" int fzp_ctx_synchronize(fzp_ctx *ctx) {
    FZP_HIP(hipStreamSynchronize(ctx->stream));
    FZP_HIP(hipStreamSynchronize(ctx->stream2));
    FZP_HIP(hipStreamSynchronize(ctx->stream3));
    return FZP_OK;
}

// ---------------------------------------------------------------- profiling
static hipEvent_t get_event(fzp_ctx *c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
ProfScope::ProfScope(fzp_ctx *ctx, const char *name, hipStream_t stream) : c(ctx), on(ctx->prof), st(stream ? stream : ctx->stream) {
    // level 2: the DP stage only.  A bracket is two event records: ~5 us of the stream and ~40 us of CPU each (calling thread + the runtime's completion thread) -- sixty
    // of them per step cost the step 1.3 ms and the rank 6 ms of CPU (r5, FZP_BENCH_NO_PROF A/B); the roofline needs exactly one
    if (on && ctx->prof_level == 2 && strcmp(name, "k1_sw") != 0) on = false;
    if (!on) return;
    ev.name = name;
    ev.a = get_event(c);
    ev.b = get_event(c);
    (void)hipEventRecord(ev.a, st);
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(ev.b, st);
    c->pending.push_back(ev);
}
int fzp_prof_flush(fzp_ctx *ctx) {
    if (ctx->pending.empty()) return FZP_OK;
    FZP_HIP(hipStreamSynchronize(ctx->stream));
    FZP_HIP(hipStreamSynchronize(ctx->stream2));
    for (auto &p : ctx->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &e = ctx->prof_tab[p.name];
            e.ms += ms;
            e.launches += 1;
        }
        ctx->event_pool.push_back(p.a);
        ctx->event_pool.push_back(p.b);
    }
    ctx->pending.clear();
    return FZP_OK;
}
extern "C" int fzp_prof_enable(fzp_ctx *ctx, int on) {
    FZP_TRY(fzp_prof_flush(ctx));
    ctx->prof = on != 0;
    ctx->prof_level = on;
    return FZP_OK;
}
extern "C" int fzp_prof_reset(fzp_ctx *ctx) {
    FZP_TRY(fzp_prof_flush(ctx));
    ctx->prof_tab.clear();
    return FZP_OK;
}
extern "C" int fzp_prof_get(fzp_ctx *ctx, const char *name, double *total_ms, int64_t *launches) {
    FZP_TRY(fzp_prof_flush(ctx));
    auto it = ctx->prof_tab.find(name);
    if (total_ms) *total_ms = it == ctx->prof_tab.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == ctx->prof_tab.end() ? 0 : it->second.launches;
    return FZP_OK;
}
extern "C" int fzp_prof_names(fzp_ctx *ctx, char **out) {
    FZP_TRY(fzp_prof_flush(ctx));
    std::string s;
    for (auto &kv : ctx->prof_tab) s += kv.first + "\n";
    *out = (char *)malloc(s.size() + 1);
    memcpy(*out, s.c_str(), s.size() + 1);
    return FZP_OK;
}

// ---------------------------------------------------------------- growing text buffer
struct TextBuf {
    char *p = nullptr;
    size_t n = 0, cap = 0;
    bool reserve(size_t extra) {
        if (n + extra + 1 <= cap) return true;
        size_t nc = cap ? cap * 2 : 1 << 16;
        while (nc < n + extra + 1) nc *= 2;
        char *q = (char *)realloc(p, nc);
        if (!q) return false;
        p = q;
        cap = nc;
        return true;
    }
    inline void putc_(char c) { p[n++] = c; }
    inline void puti(long long v) {   // caller reserved >= 21 bytes.  (r6: two digits per division, 32-bit arithmetic below 2^32 -- the small texts' numbers were 4 ms of a step's host CPU)
        static const char P2[] = "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
                                 "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
        char t[24];
        int k = 24;
        const bool neg = v < 0;
        unsigned long long u = neg ? 0ULL - (unsigned long long)v : (unsigned long long)v;
        while (u >> 32) { const unsigned d = (unsigned)(u % 100); u /= 100; t[--k] = P2[2 * d + 1]; t[--k] = P2[2 * d]; }
        uint32_t w = (uint32_t)u;
        while (w >= 100) { const uint32_t d = w % 100; w /= 100; t[--k] = P2[2 * d + 1]; t[--k] = P2[2 * d]; }
        if (w >= 10) { t[--k] = P2[2 * w + 1]; t[--k] = P2[2 * w]; } else t[--k] = (char)('0' + w);
        if (neg) p[n++] = '-';
        memcpy(p + n, t + k, (size_t)(24 - k));
        n += (size_t)(24 - k);
    }
    inline void put(const char *s, size_t len) { memcpy(p + n, s, len); n += len; }
    int finish(char **text, size_t *len) {
        if (!reserve(1)) return FZP_ENOMEM;
        p[n] = 0;
        *text = p;
        *len = n;
        return FZP_OK;
    }
};

// ---------------------------------------------------------------- SAM parser (phasing.py:42-75)
namespace {
inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }
struct Tok { const char *s; size_t n; };
int split_ws(const char *l, size_t n, Tok *t, int maxt) {
    int k = 0;
    size_t i = 0;
    while (i < n) {
        while (i < n && is_ws(l[i])) i++;
        if (i >= n) break;
        size_t j = i;
        while (j < n && !is_ws(l[j])) j++;
        if (k < maxt) t[k] = {l + i, j - i};
        k++;
        i = j;
    }
    return k;
}
bool tok_int(Tok t, long long *v) {
    if (t.n == 0 || t.n > 19) return false;
    size_t i = 0;
    bool neg = false;
    if (t.s[0] == '-' || t.s[0] == '+') { neg = t.s[0] == '-'; i = 1; }
    if (i == t.n) return false;
    long long x = 0;
    for (; i < t.n; i++) {
        if (t.s[i] < '0' || t.s[i] > '9') return false;
        x = x * 10 + (t.s[i] - '0');
    }
    *v = neg ? -x : x;
    return true;
}
inline int op_code(char c) {
    switch (c) {
        case 'M': return FZP_OP_M;
        case 'I': return FZP_OP_I;
        case 'D': return FZP_OP_D;
        case 'N': return FZP_OP_N;
        case 'S': return FZP_OP_S;
        case 'H': return FZP_OP_H;
        case 'P': return FZP_OP_P;
        case '=': return FZP_OP_EQ;
        case 'X': return FZP_OP_X;
    }
    return -1;
}
// finditer over r"(\d+)([MIDNSHP=X])" (phasing.py:12): digits not followed by an op letter are skipped
bool cigar_next(const char *c, size_t n, size_t *i, long long *adv, int *op) {
    size_t k = *i;
    while (k < n) {
        if (c[k] < '0' || c[k] > '9') { k++; continue; }
        long long v = 0;
        while (k < n && c[k] >= '0' && c[k] <= '9') { if (v < (1LL << 40)) v = v * 10 + (c[k] - '0'); k++; }   // saturates: such lengths are refused below
        if (k < n) {
            int o = op_code(c[k]);
            if (o >= 0) { *adv = v; *op = o; *i = k + 1; return true; }
        }
    }
    *i = n;
    return false;
}
template <class T>
T *dup_vec(const std::vector<T> &v) {
    T *p = (T *)malloc((v.size() ? v.size() : 1) * sizeof(T));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}
}  // namespace

extern "C" void fzp_alnset_free(fzp_alnset *a) {
    if (!a) return;
    free(a->rec_qid); free(a->rec_pos); free(a->cig_off); free(a->cigar); free(a->seq_off); free(a->seq);
    free(a->qname_off); free(a->qnames);
    free(a);
}

extern "C" int fzp_parse_sam(const char *sam, size_t len, fzp_alnset **out) {
    if (!out || (!sam && len)) return FZP_EINVAL;
    *out = nullptr;
    std::vector<int32_t> rec_qid, rec_pos;
    std::vector<int64_t> cig_off{0}, seq_off{0}, qname_off{0};
    std::vector<uint32_t> cigar;
    std::vector<uint8_t> seq;
    std::string qnames;
    std::unordered_map<std::string, int32_t> name_to_id;
    int32_t last_pos = -1, max_span = 0;
    int64_t n_columns = 0;
    size_t off = 0;
    long long line_no = 0;
    while (off < len) {
        const char *l = sam + off;
        const char *e = (const char *)memchr(l, '\n', len - off);
        size_t ln = e ? (size_t)(e - l) : len - off;
        off += ln + (e ? 1 : 0);
        line_no++;
        Tok t[11];
        int nt = split_ws(l, ln, t, 11);
        if (nt == 0) { fzp_set_error("SAM line %lld is empty (reference: IndexError)", line_no); return FZP_EINVAL; }
        if (t[0].s[0] == '@') continue;                                  // phasing.py:44-45
        if (nt < 10) { fzp_set_error("SAM line %lld has %d fields", line_no, nt); return FZP_EINVAL; }
        // phasing.py:47-54: q_id by first appearance, assigned BEFORE any filter
        std::string qn(t[0].s, t[0].n);
        auto it = name_to_id.find(qn);
        int32_t q_id;
        if (it == name_to_id.end()) {
            q_id = (int32_t)name_to_id.size();
            name_to_id.emplace(qn, q_id);
            qnames += qn;
            qname_off.push_back((int64_t)qnames.size());
        } else q_id = it->second;
        long long flag, pos1;
        if (!tok_int(t[1], &flag) || !tok_int(t[3], &pos1)) { fzp_set_error("SAM line %lld: bad FLAG/POS", line_no); return FZP_EINVAL; }
        long long POS = pos1 - 1;                                        // phasing.py:57
        const char *cg = t[5].s; size_t cn = t[5].n;
        // phasing.py:63-75
        long long skip_base = 0, total_aln_pos = 0, adv; int op; size_t ci = 0;
        while (cigar_next(cg, cn, &ci, &adv, &op)) { total_aln_pos += adv; if (op == FZP_OP_S) skip_base += adv; }
        if (total_aln_pos == 0) { fzp_set_error("SAM line %lld: CIGAR has no ops (reference: ZeroDivisionError)", line_no); return FZP_EZERODIV; }
        if (1.0 - 1.0 * (double)skip_base / (double)total_aln_pos < 0.1) continue;   // IEEE double, as written
        if (total_aln_pos < 2000) continue;
        if (POS < 0 || POS > 0x7ffffff0LL) { fzp_set_error("SAM line %lld: POS out of range", line_no); return FZP_EINVAL; }
        if (POS < last_pos) {
            fzp_set_error("SAM line %lld: POS %lld after %d -- input is not coordinate-sorted", line_no, POS + 1, last_pos + 1);
            return FZP_EUNSORTED;
        }
        // ops are stored verbatim; the walk semantics (phasing.py:77-96) live in the kernels
        long long rp = 0, qp = 0, cols = 0; ci = 0;
        size_t sn = t[9].n;
        while (cigar_next(cg, cn, &ci, &adv, &op)) {
            if (adv >= (1LL << 28)) { fzp_set_error("SAM line %lld: CIGAR op length too large", line_no); return FZP_EINVAL; }
            if (op == FZP_OP_S || op == FZP_OP_I) qp += adv;
            else if (op == FZP_OP_M || op == FZP_OP_EQ || op == FZP_OP_X) { qp += adv; rp += adv; cols += adv; if ((size_t)qp > sn) { fzp_set_error("SAM line %lld: CIGAR consumes more bases than SEQ holds (reference: IndexError)", line_no); return FZP_EINVAL; } }
            else if (op == FZP_OP_D) rp += adv;
            if (adv > 0) cigar.push_back((uint32_t)(adv << 4) | (uint32_t)op);
        }
        if (POS + rp > 0x7ffffff0LL) { fzp_set_error("SAM line %lld: alignment runs past 2^31", line_no); return FZP_EINVAL; }
        rec_qid.push_back(q_id);
        rec_pos.push_back((int32_t)POS);
        cig_off.push_back((int64_t)cigar.size());
        seq.insert(seq.end(), (const uint8_t *)t[9].s, (const uint8_t *)t[9].s + sn);
        seq_off.push_back((int64_t)seq.size());
        last_pos = (int32_t)POS;
        if (rp > max_span) max_span = (int32_t)rp;
        n_columns += cols;
    }
    fzp_alnset *a = (fzp_alnset *)calloc(1, sizeof(fzp_alnset));
    if (!a) return FZP_ENOMEM;
    a->n_rec = (int64_t)rec_qid.size();
    a->rec_qid = dup_vec(rec_qid); a->rec_pos = dup_vec(rec_pos);
    a->cig_off = dup_vec(cig_off); a->cigar = dup_vec(cigar);
    a->seq_off = dup_vec(seq_off); a->seq = dup_vec(seq);
    a->n_qid = (int32_t)name_to_id.size();
    a->qname_off = dup_vec(qname_off);
    a->qnames = (char *)malloc(qnames.size() + 1);
    if (a->qnames) memcpy(a->qnames, qnames.c_str(), qnames.size() + 1);
    a->last_pos = last_pos;
    a->max_ref_span = max_span;
    a->n_columns = n_columns;
    if (!a->rec_qid || !a->rec_pos || !a->cig_off || !a->cigar || !a->seq_off || !a->seq || !a->qname_off || !a->qnames) {
        fzp_alnset_free(a);
        return FZP_ENOMEM;
    }
    *out = a;
    return FZP_OK;
}

extern "C" void fzp_result_free(fzp_result *r) {
    if (!r) return;
    free(r->sites); free(r->vmap_qid); free(r->arows); free(r->pvars); free(r->preads);
    memset(r, 0, sizeof *r);
}

// ---------------------------------------------------------------- serializers
extern "C" int fzp_format_variant_pos(const fzp_site *s, int64_t n, char **text, size_t *len) {
    TextBuf b;
    for (int64_t i = 0; i < n; i++) {                                    // phasing.py:124
        if (!b.reserve(160)) return FZP_ENOMEM;
        b.puti((long long)s[i].pos + 1); b.putc_(' '); b.putc_((char)s[i].ref_base); b.putc_(' '); b.puti(s[i].total);
        for (int k = 0; k < 4; k++) { b.putc_(' '); b.putc_((char)s[i].base[k]); b.putc_(' '); b.puti(s[i].count[k]); }
        b.putc_('\n');
    }
    return b.finish(text, len);
}

extern "C" int fzp_format_variant_map(const fzp_site *s, int64_t n, const int32_t *q, char **text, size_t *len) {
    TextBuf b;
    for (int64_t i = 0; i < n; i++) {                                    // phasing.py:125-128
        int64_t r = s[i].row_off;
        for (int a = 0; a < 2; a++)
            for (int32_t k = 0; k < s[i].count[a]; k++, r++) {
                if (!b.reserve(64)) return FZP_ENOMEM;
                b.puti((long long)s[i].pos + 1); b.putc_(' '); b.putc_((char)s[i].ref_base); b.putc_(' ');
                b.putc_((char)s[i].base[a]); b.putc_(' '); b.puti(q[r]); b.putc_('\n');
            }
    }
    return b.finish(text, len);
}

extern "C" int fzp_format_q_id_map(const fzp_alnset *a, char **text, size_t *len) {
    TextBuf b;
    for (int32_t q = 0; q < a->n_qid; q++) {                             // phasing.py:132-134
        size_t nl = (size_t)(a->qname_off[q + 1] - a->qname_off[q]);
        if (!b.reserve(nl + 32)) return FZP_ENOMEM;
        b.puti(q); b.putc_(' '); b.put(a->qnames + a->qname_off[q], nl); b.putc_('\n');
    }
    return b.finish(text, len);
}

static inline void actg_pair(const fzp_site &s, char *x, char *y) {
    // the two alleles in CPython-2.7 dict order A < C < T < G (phasing.py:175,181)
    char a = (char)s.base[0], c = (char)s.base[1];
    if (py2_rank((uint8_t)a) < py2_rank((uint8_t)c)) { *x = a; *y = c; } else { *x = c; *y = a; }
}

extern "C" int fzp_format_atable(const fzp_site *s, const fzp_arow *r, int64_t n, char **text, size_t *len) {
    TextBuf b;
    for (int64_t i = 0; i < n; i++) {                                    // phasing.py:199
        if (!b.reserve(160)) return FZP_ENOMEM;
        char b11, b12, b21, b22;
        actg_pair(s[r[i].site1], &b11, &b12);
        actg_pair(s[r[i].site2], &b21, &b22);
        b.puti((long long)s[r[i].site1].pos + 1); b.putc_(' '); b.putc_(b11); b.putc_(' '); b.putc_(b12); b.putc_(' ');
        b.puti((long long)s[r[i].site2].pos + 1); b.putc_(' '); b.putc_(b21); b.putc_(' '); b.putc_(b22);
        for (int k = 0; k < 4; k++) { b.putc_(' '); b.puti(r[i].n[k]); }
        b.putc_('\n');
    }
    return b.finish(text, len);
}

extern "C" int fzp_format_phased_variants(const fzp_site *s, const fzp_pvar *v, int64_t n, char **text, size_t *len) {
    TextBuf b;
    int64_t i = 0;
    while (i < n) {                                                      // phasing.py:411-421
        int64_t j = i;
        long long mn = 0, mx = 0;
        while (j < n && v[j].block == v[i].block) {
            long long p = (long long)s[v[j].site].pos + 1;
            if (j == i || p < mn) mn = p;
            if (j == i || p > mx) mx = p;
            j++;
        }
        long long cnt = j - i;
        if (!b.reserve(200)) return FZP_ENOMEM;
        b.put("P ", 2); b.puti(v[i].block); b.putc_(' '); b.puti(mn); b.putc_(' '); b.puti(mx); b.putc_(' ');
        b.puti(mx - mn); b.putc_(' '); b.puti(cnt); b.putc_(' ');
        {   // Python-2 `print` of a float: '%.12g', plus '.0' when that looks like an int
            char t[64];
            snprintf(t, sizeof t, "%.12g", 1.0 * (double)(mx - mn) / (double)cnt);
            if (!strpbrk(t, ".enN")) strcat(t, ".0");
            b.put(t, strlen(t));
        }
        b.putc_('\n');
        for (int64_t k = i; k < j; k++) {
            if (!b.reserve(200)) return FZP_ENOMEM;
            long long p = (long long)s[v[k].site].pos + 1;
            char rb = (char)s[v[k].site].ref_base;
            b.put("V ", 2); b.puti(v[k].block); b.putc_(' ');
            const size_t p_at = b.n;                       // (the position stands three times in the row: spelled once, copied twice)
            b.puti(p);
            const size_t p_n = b.n - p_at;
            b.putc_(' ');
            b.put(b.p + p_at, p_n); b.putc_('_'); b.putc_(rb); b.putc_('_'); b.putc_((char)v[k].b1); b.putc_(' ');
            b.put(b.p + p_at, p_n); b.putc_('_'); b.putc_(rb); b.putc_('_'); b.putc_((char)v[k].b2); b.putc_(' ');
            b.puti(v[k].lext); b.putc_(' '); b.puti(v[k].rext); b.putc_(' '); b.puti(v[k].lscore); b.putc_(' '); b.puti(v[k].rscore);
            b.putc_('\n');
        }
        i = j;
    }
    return b.finish(text, len);
}

extern "C" int fzp_format_phased_reads(const fzp_pread *r, int64_t n, const char *ctg_id, const int64_t *qname_off,
                                       const char *qnames, int32_t n_qid, char **text, size_t *len) {
    TextBuf b;
    size_t cn = strlen(ctg_id);
    for (int64_t i = 0; i < n; i++) {                                    // phasing.py:478-480
        if (r[i].q_id < 0 || r[i].q_id >= n_qid) { free(b.p); fzp_set_error("phased read q_id %d not in q_id_map (reference: KeyError)", r[i].q_id); return FZP_EINVAL; }
        size_t nl = (size_t)(qname_off[r[i].q_id + 1] - qname_off[r[i].q_id]);
        if (!b.reserve(cn + nl + 128)) return FZP_ENOMEM;
        b.puti(r[i].q_id); b.putc_(' '); b.put(ctg_id, cn); b.putc_(' '); b.puti(r[i].block); b.putc_(' '); b.puti(r[i].phase); b.putc_(' ');
        b.puti(r[i].n0); b.putc_(' '); b.puti(r[i].n1); b.putc_(' '); b.put(qnames + qname_off[r[i].q_id], nl); b.putc_('\n');
    }
    return b.finish(text, len);
}

extern "C" int fzp_format_sam(const fzp_alnset *a, const char *ctg_id, const int32_t *flags, char **text, size_t *len) {
    TextBuf b;
    size_t cn = strlen(ctg_id);
    static const char OPS[] = "MIDNSHP=X";
    for (int64_t r = 0; r < a->n_rec; r++) {
        int32_t q = a->rec_qid[r];
        size_t nl = (size_t)(a->qname_off[q + 1] - a->qname_off[q]);
        size_t sl = (size_t)(a->seq_off[r + 1] - a->seq_off[r]);
        size_t nc = (size_t)(a->cig_off[r + 1] - a->cig_off[r]);
        if (!b.reserve(nl + cn + sl + nc * 12 + 128)) return FZP_ENOMEM;
        b.put(a->qnames + a->qname_off[q], nl); b.putc_('\t'); b.puti(flags ? flags[r] : 0); b.putc_('\t'); b.put(ctg_id, cn); b.putc_('\t');
        b.puti((long long)a->rec_pos[r] + 1); b.put("\t254\t", 5);
        for (int64_t k = a->cig_off[r]; k < a->cig_off[r + 1]; k++) { b.puti(a->cigar[k] >> 4); b.putc_(OPS[a->cigar[k] & 15]); }
        b.put("\t*\t0\t0\t", 7); b.put((const char *)a->seq + a->seq_off[r], sl); b.put("\t*\n", 3);
    }
    return b.finish(text, len);
}

// ---------------------------------------------------------------- get_phasing_readmap (phasing_readmap.py:8-51)
namespace {
void split_nl(const char *s, size_t n, std::vector<Tok> &out) {   // text.split('\n')
    size_t i = 0;
    for (;;) {
        const char *e = (const char *)memchr(s + i, '\n', n - i);
        size_t ln = e ? (size_t)(e - (s + i)) : n - i;
        out.push_back({s + i, ln});
        if (!e) break;
        i += ln + 1;
    }
}
}  // namespace

// rid_to_phase.all from the gathered records (get_rid_to_phase_all, unzip.py:303-314: the concatenation of every rid_to_phase.<ctg>, whose rows
// phasing_readmap.py:47-51 prints as '%09d ctg block phase'); records in the order given -- fzp_allgather_rid_to_phase returns the file's order
extern "C" int fzp_format_rid_to_phase_all(const fzp_r2p *recs, int64_t n, const char *const *ctg_ids, int32_t n_ctg, char **text, size_t *len) {
    if ((n && !recs) || n < 0 || !ctg_ids || !text || !len) { fzp_set_error("fzp_format_rid_to_phase_all: bad arguments"); return FZP_EINVAL; }
    std::vector<size_t> cl((size_t)(n_ctg > 0 ? n_ctg : 0));
    for (int32_t c = 0; c < n_ctg; c++) cl[(size_t)c] = strlen(ctg_ids[c]);
    TextBuf b;
    for (int64_t i = 0; i < n; i++) {
        const fzp_r2p &r = recs[i];
        if (r.ctg < 0 || r.ctg >= n_ctg) { free(b.p); fzp_set_error("rid_to_phase record %lld: contig index %d of %d", (long long)i, r.ctg, n_ctg); return FZP_EINVAL; }
        if (!b.reserve(cl[(size_t)r.ctg] + 64)) return FZP_ENOMEM;
        char t[16];
        const int k = snprintf(t, sizeof t, "%09d", r.arid);
        b.put(t, (size_t)k); b.putc_(' '); b.put(ctg_ids[r.ctg], cl[(size_t)r.ctg]); b.putc_(' '); b.puti(r.block); b.putc_(' '); b.puti(r.phase); b.putc_('\n');
    }
    return b.finish(text, len);
}

extern "C" int fzp_readmap(const char *phased_reads, size_t pr_len, const char *rawread_ids, size_t rr_len,
                           const char *pread_ids, size_t pi_len, const char *p2c, size_t pc_len,
                           const char *ctg_id, int32_t ctg_index, fzp_r2p **recs, int64_t *n_recs, char **text, size_t *len) {
    std::vector<Tok> rid_to_oid, pid_to_fid;
    split_nl(rawread_ids ? rawread_ids : "", rr_len, rid_to_oid);        // lines 17-18
    split_nl(pread_ids ? pread_ids : "", pi_len, pid_to_fid);
    std::unordered_map<std::string, std::pair<int, int>> rid_to_phase;   // lines 29-33, last line wins
    size_t off = 0;
    while (off < pr_len) {
        const char *l = phased_reads + off;
        const char *e = (const char *)memchr(l, '\n', pr_len - off);
        size_t ln = e ? (size_t)(e - l) : pr_len - off;
        off += ln + (e ? 1 : 0);
        Tok t[7];
        if (split_ws(l, ln, t, 7) < 7) { fzp_set_error("phased_reads: short row"); return FZP_EINVAL; }
        long long bk, ph;
        if (!tok_int(t[2], &bk) || !tok_int(t[3], &ph)) { fzp_set_error("phased_reads: bad block/phase"); return FZP_EINVAL; }
        rid_to_phase[std::string(t[6].s, t[6].n)] = {(int)bk, (int)ph};
    }
    size_t cn = strlen(ctg_id);
    std::unordered_map<long long, std::pair<int, int>> arid_to_phase;    // keyed by pread id ('%09d' is injective)
    off = 0;
    while (off < pc_len) {
        const char *l = p2c + off;
        const char *e = (const char *)memchr(l, '\n', pc_len - off);
        size_t ln = e ? (size_t)(e - l) : pc_len - off;
        off += ln + (e ? 1 : 0);
        Tok t[4];
        int nt = split_ws(l, ln, t, 4);
        if (nt < 2) { fzp_set_error("pread_to_contigs: short row"); return FZP_EINVAL; }
        if (!(t[1].n >= cn && memcmp(t[1].s, ctg_id, cn) == 0)) continue;            // startswith, line 41
        long long rank, pid;
        if (nt < 4 || !tok_int(t[3], &rank)) { fzp_set_error("pread_to_contigs: bad rank"); return FZP_EINVAL; }
        if (rank != 0) continue;                                                      // line 43
        if (!tok_int(t[0], &pid) || pid < 0 || (size_t)pid >= pid_to_fid.size() || pid > 0x7fffffffLL) { fzp_set_error("pread_to_contigs: pread id out of range"); return FZP_EINVAL; }
        Tok fid = pid_to_fid[(size_t)pid];                                            // lines 20-23
        const char *s1 = (const char *)memchr(fid.s, '/', fid.n);
        if (!s1) { fzp_set_error("pread_ids: '%.*s' has no '/'", (int)fid.n, fid.s); return FZP_EINVAL; }
        s1++;
        size_t rem = fid.n - (size_t)(s1 - fid.s);
        const char *s2 = (const char *)memchr(s1, '/', rem);
        Tok mid = {s1, s2 ? (size_t)(s2 - s1) : rem};
        long long raw;
        if (!tok_int(mid, &raw) || raw < 0) { fzp_set_error("pread_ids: bad raw-read field"); return FZP_EINVAL; }
        raw /= 10;                                                                    // py2 int division
        if ((size_t)raw >= rid_to_oid.size()) { fzp_set_error("rawread_ids: id %lld out of range", raw); return FZP_EINVAL; }
        Tok oid = rid_to_oid[(size_t)raw];
        auto it = rid_to_phase.find(std::string(oid.s, oid.n));
        arid_to_phase[pid] = it == rid_to_phase.end() ? std::pair<int, int>{-1, 0} : it->second;   // line 46
    }
    // canonical order: ascending '%09d' string == ascending pread id below 10^9 (py2 dict order is unspecified)
    std::vector<std::pair<std::string, std::pair<long long, std::pair<int, int>>>> rows;
    rows.reserve(arid_to_phase.size());
    for (auto &kv : arid_to_phase) {
        char key[32];
        snprintf(key, sizeof key, "%09lld", kv.first);
        rows.push_back({key, {kv.first, kv.second}});
    }
    std::sort(rows.begin(), rows.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    TextBuf b;
    fzp_r2p *rr = (fzp_r2p *)malloc((rows.size() ? rows.size() : 1) * sizeof(fzp_r2p));
    if (!rr) return FZP_ENOMEM;
    for (size_t i = 0; i < rows.size(); i++) {                                        // lines 49-51
        if (!b.reserve(cn + 80)) { free(rr); return FZP_ENOMEM; }
        b.put(rows[i].first.c_str(), rows[i].first.size()); b.putc_(' '); b.put(ctg_id, cn); b.putc_(' ');
        b.puti(rows[i].second.second.first); b.putc_(' '); b.puti(rows[i].second.second.second); b.putc_('\n');
        rr[i] = {(int32_t)rows[i].second.first, ctg_index, rows[i].second.second.first, rows[i].second.second.second};
    }
    if (recs) *recs = rr; else free(rr);
    if (n_recs) *n_recs = (int64_t)rows.size();
    if (text) return b.finish(text, len);
    free(b.p);
    return FZP_OK;
}
