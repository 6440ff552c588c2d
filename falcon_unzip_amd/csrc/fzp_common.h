// fzp_common.h -- internals shared by the HIP translation units of libfzphase.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "fzphase.h"

void fzp_set_error(const char *fmt, ...);

#define FZP_HIP(x)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            fzp_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));         \
            return FZP_EDEVICE;                                                                      \
        }                                                                                            \
    } while (0)
#define FZP_TRY(x)                 \
    do {                           \
        int rc_ = (x);             \
        if (rc_ != FZP_OK) return rc_; \
    } while (0)

constexpr int WAVE = 64;

// ---------------------------------------------------------------- caching device allocator (fzp_host.hip)
// Steady-state passes of the hot path must not call hipMalloc/hipFree (both are slow and hipFree
// synchronises): freed blocks are kept in size buckets and handed out again.  Every ctx owns its pool:
// fzp_bind(ctx) -- the first thing every entry point does -- selects the device and makes the ctx's pool the
// calling thread's current one; a block always returns to the pool it came from, whoever frees it.  Inside a
// ctx all work is ordered by its streams' events, which is what makes host-side reuse of a freed block safe;
// two ctxs (same or different devices, same or different threads) never see each other's blocks.
struct DevPool;
struct fzp_ctx;
int fzp_bind(fzp_ctx *ctx);          // hipSetDevice(ctx->device) + current pool = ctx's
int usable_cores();                  // threads this process may really use: affinity mask and cgroup CPU quota, not the machine's thread count (fzp_host.hip)
int cores_per_rank();                // ... divided by the ranks of the node (LOCAL_WORLD_SIZE): what the host thread pools are sized by
// A few words from the device, NOW: a one-wave kernel on `st` posts them (and a sequence number behind them) into mapped pinned memory and the calling thread spins on the
// sequence number -- half the round trip of hipMemcpyAsync + hipStreamSynchronize (tools/ubench/fetch_latency.hip: 12 against 25 us between two dependent kernels), which
// is what the count read-backs between the stages of a step cost.  Up to four pieces of up to 256 bytes, each a multiple of 4 bytes from a 4-byte aligned device address.
struct fzp_fetch_piece { void *host; const void *dev; size_t bytes; };
int fzp_fetch(fzp_ctx *ctx, hipStream_t st, const fzp_fetch_piece *pieces, int n_pieces);
inline int fzp_fetch(fzp_ctx *ctx, hipStream_t st, void *host, const void *dev, size_t bytes) { const fzp_fetch_piece p{host, dev, bytes}; return fzp_fetch(ctx, st, &p, 1); }
struct fzp_fill_piece { void *dev; size_t bytes; uint32_t word; };
int fzp_fill(fzp_ctx *ctx, hipStream_t st, const fzp_fill_piece *pieces, int n_pieces);      // every region set to its 32-bit word, one launch per twelve regions (fzp_host.hip)
int fzp_read_back(fzp_ctx *ctx, hipStream_t st, void *host, const void *dev, size_t bytes);
// a stage's total(s) and its per-contig begins in ONE fetch when the begins fit a piece (<= 31 contigs), the total first and the begins by copy otherwise (r5: K2..K5 each
// read a count back, launched the kernel it sizes, and then read the per-contig begins back -- which the scan behind the count had already fixed)
inline int fzp_fetch_with_begins(fzp_ctx *ctx, hipStream_t st, uint64_t *tot, const uint64_t *tot_dev, int n_tot, int64_t *h_begin, const int64_t *begin_dev, int n_ctg) {
    const size_t bb = ((size_t)n_ctg + 1) * sizeof(int64_t);
    if (bb <= 256) { const fzp_fetch_piece fp[2] = {{tot, tot_dev, (size_t)n_tot * 8}, {h_begin, begin_dev, bb}}; return fzp_fetch(ctx, st, fp, 2); }
    FZP_TRY(fzp_fetch(ctx, st, tot, tot_dev, (size_t)n_tot * 8));
    return fzp_read_back(ctx, st, h_begin, begin_dev, bb);
}
      // fzp_fetch where it fits (<= 256 bytes), copy + stream wait otherwise
void *fzp_dev_alloc(size_t bytes);   // from the calling thread's current pool; nullptr on failure
void fzp_dev_free(void *p);
void fzp_dev_trim();                  // give everything cached in the current pool back to the driver
// pinned host staging blocks (bulk D2H of results, H2D of inputs): cached per ctx like device blocks
void *fzp_pinned_acquire(fzp_ctx *ctx, size_t bytes, size_t *cap);   // nullptr on failure
void fzp_pinned_release(fzp_ctx *ctx, void *p);
// host segments -> one device buffer at the given byte offsets: chunks staged through pinned blocks by a few threads,
// H2D copies queued behind each other; returns when everything has arrived (`st` is only used for small inputs)
int fzp_upload_segments(fzp_ctx *ctx, void *dst_dev, const std::vector<const void *> &src, const std::vector<size_t> &dst_off, const std::vector<size_t> &len, hipStream_t st);

// ---------------------------------------------------------------- device buffer
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;   // capacity in elements
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) fzp_dev_free(p);
        p = nullptr; n = 0;
    }
    int alloc(size_t count) {   // grow-only
        if (count <= n && p) return FZP_OK;
        release();
        size_t bytes = (count ? count : 1) * sizeof(T);
        p = (T *)fzp_dev_alloc(bytes);
        if (!p) {
            fzp_set_error("device allocation of %zu bytes failed", bytes);
            return FZP_ENOMEM;
        }
        n = count ? count : 1;
        return FZP_OK;
    }
    int upload(const T *h, size_t count, hipStream_t s) {
        FZP_TRY(alloc(count));
        if (count) FZP_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
        return FZP_OK;
    }
    int download(T *h, size_t count, hipStream_t s, size_t off = 0) const {
        if (count) FZP_HIP(hipMemcpyAsync(h, p + off, count * sizeof(T), hipMemcpyDeviceToHost, s));
        return FZP_OK;
    }
    int zero(size_t count, hipStream_t s) {
        if (count) FZP_HIP(hipMemsetAsync(p, 0, count * sizeof(T), s));
        return FZP_OK;
    }
};

template <typename T> inline fzp_fill_piece fzp_zeroes(DevBuf<T> &b, size_t count) { return fzp_fill_piece{b.p, count * sizeof(T), 0u}; }
template <typename T> inline fzp_fill_piece fzp_ones(DevBuf<T> &b, size_t count) { return fzp_fill_piece{b.p, count * sizeof(T), 0xffffffffu}; }

// ---------------------------------------------------------------- context
struct ProfEntry {
    double ms = 0;
    int64_t launches = 0;
};
struct PendingEvent {
    std::string name;
    hipEvent_t a, b;
};

struct fzp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // trace-back of chunk k runs here while the DP of chunk k+1 runs on `stream`
    hipStream_t stream3 = nullptr;   // K1: the wave-per-read DP of the long reads runs here beside the bit-sliced DP of the others on `stream`
    bool prof = false;
    int prof_level = 0;               // fzp_prof_enable: 1 = every bracket, 2 = the DP stage (k1_sw) only
    std::map<std::string, ProfEntry> prof_tab;
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
    DevBuf<uint64_t> scan_tmp[3];
    std::shared_ptr<DevPool> pool;   // this ctx's cached device blocks
    uint32_t *fetch_slot = nullptr;   // fzp_fetch: 8 x 64 payload words + a sequence number, in mapped pinned memory (fzp_host.hip)
    uint64_t fetch_seq = 0;
    std::mutex pin_mu;
    std::vector<std::pair<void *, size_t>> pin_free, pin_live;   // pinned host blocks: cached / handed out
    struct WorkPool *workers = nullptr;    // fzp_pipe.hip: host threads of the per-contig text / read-map work, kept between calls
    struct GroupPool *gpool = nullptr;     // fzp_pipe.hip: host buffers of fzp_phase_contigs_files' contig groups, kept between calls (no fresh pages per call)
    struct FileWriter *writer = nullptr;   // fzp_pipe.hip: background file writes of FZP_PIPE_ASYNC_WRITES calls (joined by fzp_pipe_flush / ctx destroy)
    std::vector<fzp_ctx *> lanes;    // fzp_phase_contigs: the extra lanes' contexts, kept (with their warm block caches) for the next call
    hipEvent_t ev_pf = nullptr;      // "K2/K3 results are final": their download starts on stream2 while K4/K5 run
    hipEvent_t ev_late = nullptr;    // fzp_batch_result_begin: behind the block / read records' copies
    hipEvent_t ev_pf_done = nullptr; // ... and "that download is over" (what fzp_batch_result_all waits for: stream2 may hold later copies)
    int n_cu = 256;
};

// RAII bracket: records HIP events on the ctx stream around a kernel launch when profiling is on.
struct ProfScope {
    fzp_ctx *c;
    PendingEvent ev;
    bool on;
    hipStream_t st;
    ProfScope(fzp_ctx *ctx, const char *name, hipStream_t stream = nullptr);
    ~ProfScope();
};
int fzp_prof_flush(fzp_ctx *ctx);
void fzp_writer_destroy(fzp_ctx *ctx);   // fzp_pipe.hip
// fzp_align.hip, for fzp_pipe.hip's BAM by-product: the records of every aligned read with 'M' CIGARs + strand flags + a copy of the contig (device
// part, calling thread); the '=' / 'X' split is host-only and runs on a writer thread
int fzp_align_alnset_unsplit(fzp_ctx *ctx, fzp_alnjob *j, int32_t ctg, const int64_t *name_off, const char *names, fzp_alnset **out, std::vector<int32_t> *flags,
                             std::shared_ptr<std::vector<uint8_t>> *ref);
void fzp_alnset_split_eqx(fzp_alnset *a, const uint8_t *ref);

// ---------------------------------------------------------------- scans (fzp_scan.hip)
// out[i] = sum_{j<i} in[j] over n uint32 items (in may alias out); *total_dev (device u64) gets the sum.
int fzp_exclusive_scan_u32(fzp_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint64_t *total_dev);
int fzp_exclusive_scan_u32_end(fzp_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint64_t *total_dev);      // ... and out[n] = the total: a CSR's closing offset without a copy of its own
int fzp_exclusive_scan_u64_inplace(fzp_ctx *ctx, uint64_t *v, size_t n, uint64_t *total_dev);

// ---------------------------------------------------------------- wave helpers (device)
#ifdef __HIPCC__
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += t;
    }
    return v;
}
// the same scan (and a running maximum) on the DPP network, no LDS traffic: four shifts inside each row of 16 lanes, then the
// row totals passed on with row_bcast:15 (rows 1 and 3 take the last lane of the row before) and row_bcast:31 (rows 2 and 3 take lane 31)
__device__ __forceinline__ uint32_t wave_incl_scan_u32_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_maxscan_u32_dpp(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}
__device__ __forceinline__ int32_t wave_sum_i32(int32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// wave-uniform sum without LDS traffic: 4 DPP butterfly steps inside each row of 16 lanes, then the 4 rows via SGPRs
__device__ __forceinline__ int32_t wave_sum_i32_dpp(int32_t v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);   // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int32_t wave_min_i32(int32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ int32_t wave_max_i32(int32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, int src) { return __shfl(v, src, 64); }

// base letter <-> 2-bit code in the order A C G T (the order phasing.py:108 counts in)
__device__ __forceinline__ int sym_code(uint8_t s) {
    // branch-free: bits 1-2 of the letter give A 0, C 1, T 2, G 3; one xor puts G before T; the letter is then checked
    const uint32_t x = ((uint32_t)s >> 1) & 3u;
    const uint32_t code = x ^ (x >> 1);
    const uint32_t expect = (0x54474341u >> (code * 8u)) & 0xffu;      // 'A' 'C' 'G' 'T'
    return expect == (uint32_t)s ? (int)code : 4;
}
__device__ __host__ __forceinline__ uint8_t code_sym(int c) { return (uint8_t)("ACGT"[c & 3]); }
// CPython-2.7 iteration order of a {allele: ...} dict: A < C < T < G (phasing.py:175,181)
__device__ __host__ __forceinline__ int py2_rank(uint8_t b) { return b == 'A' ? 0 : b == 'C' ? 1 : b == 'T' ? 2 : b == 'G' ? 3 : 4; }
#endif
