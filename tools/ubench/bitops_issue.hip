// Issue cost of the instructions k_swb's step is made of (r5): v_bitop3_b32, the 64-bit shifts by a per-lane amount, v_alignbit_b32, v_and_or / v_lshl_or, plain logic
// ops -- cycles per wave64 instruction with 1 and 2 waves per SIMD, 8 independent chains per wave (no dependency stalls).  Build: hipcc --offload-arch=gfx950 -O3 -o bitops_issue bitops_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int ID>
__global__ void __launch_bounds__(64) k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t v[8], w[8];
    uint64_t q[8];
    for (int i = 0; i < 8; i++) { v[i] = seed * (threadIdx.x + 3u + i); w[i] = (seed >> 3) + i * 77u + threadIdx.x; q[i] = ((uint64_t)v[i] << 32) | w[i]; }
    const uint32_t sh = (threadIdx.x & 1u);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (ID == 0) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                REP8(X)
#undef X
            } else if (ID == 1) {
#define X(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(v[i]) : "v"(w[i]), "v"(w[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (ID == 2) {
#define X(i) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q[i]) : "v"(sh));
                REP8(X)
#undef X
            } else if (ID == 3) {
#define X(i) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q[i]) : "v"(sh));
                REP8(X)
#undef X
            } else if (ID == 4) {
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "v"(sh));
                REP8(X)
#undef X
            } else if (ID == 5) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "v"(w[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (ID == 6) {
#define X(i) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(sh), "v"(w[i]));
                REP8(X)
#undef X
            } else if (ID == 7) {
#define X(i) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(v[i]) : "v"(sh));
                REP8(X)
#undef X
            } else if (ID == 8) {
#define X(i) asm volatile("v_bfe_u32 %0, %0, %1, 3" : "+v"(v[i]) : "v"(sh));
                REP8(X)
#undef X
            } else if (ID == 9) {
#define X(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "v"(w[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (ID == 10) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                REP8(X)
#undef X
            } else if (ID == 11) {
#define X(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[i]));
                REP8(X)
#undef X
            } else if (ID == 12) {      // a dependent chain of bitop3 (one chain: what a lone wave pays for latency)
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n\tv_bitop3_b32 %0, %0, %2, %1 bitop3:0x96\n\tv_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n\tv_bitop3_b32 %0, %0, %2, %1 bitop3:0x96\n\t"
                             "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n\tv_bitop3_b32 %0, %0, %2, %1 bitop3:0x96\n\tv_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n\tv_bitop3_b32 %0, %0, %2, %1 bitop3:0x96" : "+v"(v[0]) : "v"(w[0]), "v"(w[1]));
            } else if (ID == 13) {      // ... and of v_xor
                asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(v[0]) : "v"(w[0]));
            }
        }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) acc ^= v[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
    if (acc == 0x12345u) out[0] = acc;
}
template <int ID>
void run(const char *name, uint32_t *d) {
    for (int W = 1; W <= 2; W++) {
        const int iters = 20000, blocks = 256 * 4 * W;      // W one-wave workgroups per SIMD
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(k<ID>, dim3(blocks), dim3(64), 0, 0, d, 100, 12345u);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k<ID>, dim3(blocks), dim3(64), 0, 0, d, iters, 12345u);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        const double inst = (double)iters * 64.0 * W;      // per SIMD
        printf("%-44s W=%d  %8.3f ms  %6.2f cycles/inst per SIMD @2.4 GHz\n", name, W, ms, ms * 1e-3 * 2.4e9 / inst);
    }
}
int main() {
    uint32_t *d; (void)hipMalloc(&d, 64);
    run<10>("v_add_u32", d); run<0>("v_xor_b32", d); run<1>("v_bitop3_b32 (3 vgpr)", d); run<9>("v_or3_b32", d); run<5>("v_and_or_b32", d); run<6>("v_lshl_or_b32 (vgpr shift)", d);
    run<7>("v_lshlrev_b32 (vgpr shift)", d); run<8>("v_bfe_u32 (vgpr offset)", d); run<4>("v_alignbit_b32 (vgpr shift)", d); run<2>("v_lshlrev_b64 (vgpr shift)", d);
    run<3>("v_lshrrev_b64 (vgpr shift)", d); run<11>("v_lshlrev_b64 (by 1)", d); run<12>("v_bitop3_b32, ONE dependent chain", d); run<13>("v_xor_b32, ONE dependent chain", d);
    return 0;
}
