// micro-benchmark: can a VALU-bound loop push 16 B per iteration through scalar stores (s_store_dwordx4)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>   // 0: VALU only, 1: + scalar stores, 2: + vector stores of the same bytes (lane 0..1 write 16 B)
__global__ void __launch_bounds__(64) k(uint64_t *out, int iters) {
    const int lane = threadIdx.x;
    uint64_t *dst = out + (size_t)blockIdx.x * (size_t)iters * 2;
    int32_t a = lane * 7 + 1, b = lane ^ 21, c = lane + blockIdx.x, d = 3;
    uint32_t off = 0;
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            uint64_t m0, m1;
            // 12 filler VALU + 2 compares that produce the masks
            asm volatile(
                "v_add_u32 %[a], %[a], %[b]\n\t"
                "v_xor_b32 %[b], %[b], %[c]\n\t"
                "v_add_u32 %[c], %[c], %[d]\n\t"
                "v_max_i32 %[d], %[d], %[a]\n\t"
                "v_add_u32 %[a], %[a], %[b]\n\t"
                "v_xor_b32 %[b], %[b], %[c]\n\t"
                "v_add_u32 %[c], %[c], %[d]\n\t"
                "v_and_b32 %[d], 0xff, %[a]\n\t"
                "v_add_u32 %[a], %[a], %[b]\n\t"
                "v_xor_b32 %[b], %[b], %[c]\n\t"
                "v_add_u32 %[c], %[c], %[d]\n\t"
                "v_and_b32 %[d], 0xff, %[a]\n\t"
                "v_cmp_gt_i32_e64 %[m0], %[a], %[b]\n\t"
                "v_cmp_gt_i32_e64 %[m1], %[c], %[d]\n\t"
                : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [d] "+v"(d), [m0] "=s"(m0), [m1] "=s"(m1));
            if (MODE == 1) {
                __uint128_t both = ((__uint128_t)m1 << 64) | m0;
                asm volatile("s_store_dwordx4 %[v], %[p], %[o]" :: [v] "s"(both), [p] "s"(dst), [o] "s"(off) : "memory");
                off += 16;
            } else if (MODE == 2) {
                if (lane < 2) dst[(off >> 3) + lane] = lane ? m1 : m0;
                off += 16;
            }
        }
        if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (MODE == 1) asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    if (a == 0x12345678 && lane == 63) out[0] = (uint64_t)(a + b + c + d);   // keep the filler alive
}

// verification variant: the masks are functions of (wave, iteration); every stored value is checked on the host.
// The same SGPR quad is rewritten right after each s_store (no wait in between): passes only if the store reads
// its data at issue.
__global__ void __launch_bounds__(64) kv(uint64_t *out, int iters) {
    const int lane = threadIdx.x;
    uint64_t *dst = out + (size_t)blockIdx.x * (size_t)iters * 2;
    uint32_t off = 0;
    for (int it = 0; it < iters; it++) {
        int32_t c0 = (it + blockIdx.x) & 63, c1 = (it * 7 + blockIdx.x * 3) & 63;
        asm volatile(
            "v_cmp_gt_i32_e64 s[60:61], %[c0], %[l]\n\t"
            "v_cmp_gt_i32_e64 s[62:63], %[c1], %[l]\n\t"
            "s_nop 0\n\t"
            "s_store_dwordx4 s[60:63], %[p], %[o]\n\t"
            :: [c0] "v"(c0), [c1] "v"(c1), [l] "v"(lane), [p] "s"(dst), [o] "s"(off) : "memory", "s60", "s61", "s62", "s63");
        off += 16;
    }
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

int main() {
    const int waves = 16384, iters = 8192;
    uint64_t *out;
    CK(hipMalloc(&out, (size_t)waves * iters * 16));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(out, 0, (size_t)waves * iters * 16));
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, out, iters);
            hipEventRecord(e1);
            CK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d rep %d: %.3f ms  (%.1f GB/s stored, %.2f G iter/s)\n", mode, rep, ms, mode ? waves * (double)iters * 16 / ms / 1e6 : 0.0,
                   waves * (double)iters / ms / 1e6);
        }
        if (mode) {   // verify a few values are non-zero / present
            std::vector<uint64_t> h(64);
            CK(hipMemcpy(h.data(), out + (size_t)(waves - 1) * iters * 2 + (size_t)(iters - 32) * 2, 64 * 8, hipMemcpyDeviceToHost));
            int nz = 0; for (auto v : h) nz += v != 0;
            printf("   tail non-zero words: %d / 64\n", nz);
        }
    }
    {
        const int w2 = 4096, it2 = 4096;
        CK(hipMemset(out, 0xff, (size_t)w2 * it2 * 16));
        hipLaunchKernelGGL(kv, dim3(w2), dim3(64), 0, 0, out, it2);
        CK(hipDeviceSynchronize());
        std::vector<uint64_t> h((size_t)w2 * it2 * 2);
        CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (int w = 0; w < w2; w++)
            for (int it = 0; it < it2; it++) {
                int c0 = (it + w) & 63, c1 = (it * 7 + w * 3) & 63;
                uint64_t e0 = c0 ? (~0ull >> (64 - c0)) : 0, e1 = c1 ? (~0ull >> (64 - c1)) : 0;
                if (h[((size_t)w * it2 + it) * 2] != e0 || h[((size_t)w * it2 + it) * 2 + 1] != e1) bad++;
            }
        printf("verify: %zu bad of %zu\n", bad, (size_t)w2 * it2);
    }
    return 0;
}
