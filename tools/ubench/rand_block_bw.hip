// What HBM gives a kernel that reads contiguous blocks of B bytes at scattered places of a 12 GB buffer (r5: the walk of K1's trace-back reads a 256-byte piece of 8-byte
// mask records per walker and 32 steps; its older form 512-byte pieces).  Every wave reads blocks two at a time the way k_tb_walk_h does -- 64 lanes x 8 bytes, lanes 0..31 one
// block, lanes 32..63 another when B = 256 -- sixteen loads in flight per wave, 16 waves per CU.  Prints GB/s per block size.
// Build: hipcc --offload-arch=gfx950 -O3 -o rand_block_bw rand_block_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
// a "load" = one wave instruction of 64 x 8 B = 512 B, made of 512 / B blocks (B <= 512) or part of one block (B > 512)
template <int B>
__global__ void __launch_bounds__(64) k(const uint2 *buf, uint64_t n_blocks, int iters, uint32_t *out, uint64_t seed) {
    const int lane = threadIdx.x;
    uint32_t acc = 0;
    uint64_t s = seed + blockIdx.x * 0x9e3779b97f4a7c15ull;
    for (int it = 0; it < iters; it++) {
        uint2 v[16];
#pragma unroll
        for (int l = 0; l < 16; l++) {
            uint64_t at;
            if constexpr (B <= 512) {
                constexpr int per = 512 / B, lanes_per = 64 / per;
                const uint64_t blk = mix(s + (uint64_t)(it * 16 + l) * per + lane / lanes_per) % n_blocks;
                at = blk * (B / 8) + (lane % lanes_per);
            } else {
                constexpr int loads_per = B / 512;
                const uint64_t blk = mix(s + (uint64_t)((it * 16 + l) / loads_per)) % n_blocks;
                at = blk * (B / 8) + ((it * 16 + l) % loads_per) * 64 + lane;
            }
            v[l] = buf[at];
        }
#pragma unroll
        for (int l = 0; l < 16; l++) acc += v[l].x ^ v[l].y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int B> void run(const uint2 *buf, uint64_t bytes, uint32_t *out) {
    const uint64_t n_blocks = bytes / B;
    const int waves = 256 * 16 * 4, iters = 64;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<B>, dim3(waves), dim3(64), 0, 0, buf, n_blocks, 4, out, 1ull);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<B>, dim3(waves), dim3(64), 0, 0, buf, n_blocks, iters, out, 77ull);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double gb = (double)waves * iters * 16 * 512 / 1e9;
    printf("block %5d B: %.2f GB in %.3f ms = %.0f GB/s\n", B, gb, ms, gb / (ms * 1e-3));
}
int main() {
    const uint64_t bytes = 12ull << 30;
    uint2 *buf; uint32_t *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(buf, 1, bytes));
    run<64>(buf, bytes, out); run<128>(buf, bytes, out); run<256>(buf, bytes, out); run<512>(buf, bytes, out); run<1024>(buf, bytes, out); run<4096>(buf, bytes, out); run<65536>(buf, bytes, out);
    return 0;
}
