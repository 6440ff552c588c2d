// fzp_scan.hip -- exclusive prefix sums on the device (reduce-then-scan, recursive).
// Used for every ordered compaction of the phasing path (site lists, variant_map rows, atable rows,
// phased_reads rows): outputs must come out in the reference's order, so slots are assigned by
// scans, never by atomics.  HBM-bound: reads n*4 B twice, writes n*4 B once.
#include "fzp_common.h"

namespace {
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;                       // per thread
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048 items per workgroup

__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t *total) {
    __shared__ uint64_t wsum[SCAN_THREADS / 64];
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t t = __shfl_up(incl, d, 64);
        if (lane_id() >= d) incl += t;
    }
    int w = threadIdx.x >> 6;
    if (lane_id() == 63) wsum[w] = incl;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_THREADS / 64; i++) {
        if (i < w) base += wsum[i];
        tot += wsum[i];
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// pass 1: per-tile sums
__global__ void __launch_bounds__(SCAN_THREADS) k_tile_sums(const uint32_t *in, size_t n, uint64_t *sums) {
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) s += in[base + i];
    uint64_t tot;
    (void)block_excl_scan_u64(s, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// scan of up to SCAN_TILE u64 values in one workgroup (in place), total to *total
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_small_u64(uint64_t *v, size_t n, uint64_t *total) {
    size_t base = (size_t)threadIdx.x * SCAN_ITEMS;
    uint64_t x[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        x[i] = (base + i < n) ? v[base + i] : 0;
        s += x[i];
    }
    uint64_t tot;
    uint64_t off = block_excl_scan_u64(s, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) v[base + i] = off;
        off += x[i];
    }
    if (threadIdx.x == 0 && total) *total = tot;
}

// u64 multi-tile variant used for the recursion levels
__global__ void __launch_bounds__(SCAN_THREADS) k_tile_sums_u64(const uint64_t *in, size_t n, uint64_t *sums) {
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) s += in[base + i];
    uint64_t tot;
    (void)block_excl_scan_u64(s, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(SCAN_THREADS) k_apply_u64(uint64_t *v, size_t n, const uint64_t *tile_off) {
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint64_t x[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        x[i] = (base + i < n) ? v[base + i] : 0;
        s += x[i];
    }
    uint64_t tot;
    uint64_t off = block_excl_scan_u64(s, &tot) + tile_off[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) v[base + i] = off;
        off += x[i];
    }
}

// pass 3: local scan + tile offset
__global__ void __launch_bounds__(SCAN_THREADS) k_apply_u32(const uint32_t *in, uint32_t *out, size_t n, const uint64_t *tile_off) {
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t x[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        x[i] = (base + i < n) ? in[base + i] : 0;
        s += x[i];
    }
    uint64_t tot;
    uint64_t off = block_excl_scan_u64(s, &tot) + tile_off[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = (uint32_t)off;
        off += x[i];
    }
}

// r6: the middle launch folded away.  A tile's offset is the sum of the tile sums before it -- at most 2 048 of them, which a workgroup adds up itself in the time a launch
// takes to start (a step's eighteen scans were fifty-four launches of 4.5 us each; now thirty-six, and a scan that fits one tile is one).  The last tile leaves the total.
__global__ void __launch_bounds__(SCAN_THREADS) k_apply_sums_u32(const uint32_t *in, uint32_t *out, size_t n, const uint64_t *tile_sums, uint64_t *total, int with_end) {
    uint64_t part = 0;
    for (unsigned i = threadIdx.x; i < blockIdx.x; i += SCAN_THREADS) part += tile_sums[i];
    uint64_t before;
    (void)block_excl_scan_u64(part, &before);
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t x[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        x[i] = (base + i < n) ? in[base + i] : 0;
        s += x[i];
    }
    uint64_t tot;
    uint64_t off = block_excl_scan_u64(s, &tot) + before;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = (uint32_t)off;
        off += x[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        if (total) *total = before + tot;
        if (with_end) out[n] = (uint32_t)(before + tot);      // (a CSR's closing offset: the caller gave the array n + 1 places)
    }
}

int scan_u64_inplace(fzp_ctx *ctx, uint64_t *v, size_t n, uint64_t *total_dev, int level) {
    if (n <= (size_t)SCAN_TILE) {
        hipLaunchKernelGGL(k_scan_small_u64, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, v, n, total_dev);
        return FZP_OK;
    }
    if (level >= 3) {
        fzp_set_error("scan: input too large");
        return FZP_EINVAL;
    }
    size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    FZP_TRY(ctx->scan_tmp[level].alloc(tiles));
    uint64_t *sums = ctx->scan_tmp[level].p;
    hipLaunchKernelGGL(k_tile_sums_u64, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream, v, n, sums);
    FZP_TRY(scan_u64_inplace(ctx, sums, tiles, total_dev, level + 1));
    hipLaunchKernelGGL(k_apply_u64, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream, v, n, sums);
    return FZP_OK;
}
}  // namespace

static int scan_u32(fzp_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint64_t *total_dev, bool with_end);
int fzp_exclusive_scan_u32(fzp_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint64_t *total_dev) { return scan_u32(ctx, in, out, n, total_dev, false); }
// ... and out[n] = the total (as 32 bits): the array has n + 1 places, total_dev is asked for
int fzp_exclusive_scan_u32_end(fzp_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint64_t *total_dev) {
    if (!total_dev) { fzp_set_error("scan: the closing offset comes from the total"); return FZP_EINVAL; }
    return scan_u32(ctx, in, out, n, total_dev, true);
}
static int scan_u32(fzp_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint64_t *total_dev, bool with_end) {
    if (n == 0) {
        if (total_dev) FZP_HIP(hipMemsetAsync(total_dev, 0, sizeof(uint64_t), ctx->stream));
        if (with_end) FZP_HIP(hipMemsetAsync(out, 0, sizeof(uint32_t), ctx->stream));
        return FZP_OK;
    }
    ProfScope ps(ctx, "scan");
    size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    static const bool three_pass = getenv("FZP_SCAN_3PASS") != nullptr;      // (A/B runs: the r1-r5 form)
    if (tiles <= (size_t)SCAN_TILE && !three_pass) {
        uint64_t *sums = nullptr;
        if (tiles > 1) {
            FZP_TRY(ctx->scan_tmp[0].alloc(tiles));
            sums = ctx->scan_tmp[0].p;
            hipLaunchKernelGGL(k_tile_sums, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream, in, n, sums);
        }
        hipLaunchKernelGGL(k_apply_sums_u32, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream, in, out, n, sums, total_dev, with_end ? 1 : 0);
        FZP_HIP(hipGetLastError());
        return FZP_OK;
    }
    FZP_TRY(ctx->scan_tmp[0].alloc(tiles));
    uint64_t *sums = ctx->scan_tmp[0].p;
    hipLaunchKernelGGL(k_tile_sums, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream, in, n, sums);
    FZP_TRY(scan_u64_inplace(ctx, sums, tiles, total_dev, 1));
    hipLaunchKernelGGL(k_apply_u32, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, ctx->stream, in, out, n, sums);
    if (with_end) FZP_HIP(hipMemcpyAsync(out + n, total_dev, sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));      // (the total's low word: little-endian)
    FZP_HIP(hipGetLastError());
    return FZP_OK;
}

int fzp_exclusive_scan_u64_inplace(fzp_ctx *ctx, uint64_t *v, size_t n, uint64_t *total_dev) {
    if (n == 0) {
        if (total_dev) FZP_HIP(hipMemsetAsync(total_dev, 0, sizeof(uint64_t), ctx->stream));
        return FZP_OK;
    }
    ProfScope ps(ctx, "scan");
    FZP_TRY(scan_u64_inplace(ctx, v, n, total_dev, 0));
    FZP_HIP(hipGetLastError());
    return FZP_OK;
}
