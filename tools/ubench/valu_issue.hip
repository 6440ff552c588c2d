// micro-benchmark: wave64 VALU issue rate on gfx950 by instruction kind and by waves resident per SIMD.
// Question it answers: is one wave64 integer VALU op per SIMD per 4 cycles the ceiling
// k_sw runs against, or do two or more waves per SIMD co-issue at 2 cycles per op (SIMD-32)?
// Every kernel runs ITERS x 64 instructions of one kind on 8 independent register chains; the grid places exactly
// W waves on every SIMD of every CU (workgroups of 256 x min(W,4) threads, one or two per CU).
// Output: one line per (kind, W): wave-instructions per ns per SIMD and the implied cycles per instruction at the
// measured shader clock (s_memtime is a constant 100 MHz counter here, so the clock comes from a calibrated loop of
// dependent v_add_u32 at 1 wave: those issue back to back at their latency).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum Kind { ADD = 0, MAX, ADD_DPP, MAX_DPP, MOV_DPP, CMP_SGPR, CMP_VCC, CNDMASK, MAX3, READLANE, WRITELANE, PK_ADD_I16, PK_MAX_I16, SW_MIX, SW_MIX_NOLANE, SW_MIX_SEL, CNDMASK_E64, MAX_3REG, N_KINDS };
static const char *kind_name[N_KINDS] = {"v_add_u32", "v_max_i32", "v_add_u32_dpp(wave_shl)", "v_max_i32_dpp(wave_shr)", "v_mov_b32_dpp(wave_shl)",
                                         "v_cmp_eq_i32_e64->sgpr", "v_cmp_eq_u32_e32->vcc", "v_cndmask_b32 (vcc written once, one asm statement)", "v_max3_i32", "v_readlane_b32",
                                         "v_writelane_b32", "v_pk_add_i16", "v_pk_max_i16", "k_sw step mix (13 VALU + 9 SALU + s_store)", "k_sw step mix without readlane/writelane", "k_sw step mix, Hnew = cmp_ge + cndmask_e64 (no max + cmp_eq)", "v_cndmask_b32_e64 (sgpr pair mask)", "v_max_i32 (dst apart from both sources)"};

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(1024) k(uint32_t *out, int iters, uint64_t *sink) {
    int32_t v[8], w[8];
    for (int i = 0; i < 8; i++) { v[i] = threadIdx.x * (i + 3) + 1; w[i] = (threadIdx.x ^ (i * 5)) + 7; }
    uint64_t sm = 0;
    int32_t sx = 0;
    asm volatile("v_cmp_gt_i32_e32 vcc, %0, %1" ::"v"(v[0]), "v"(w[0]) : "vcc");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == ADD) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                R8(X)
#undef X
            } else if (KIND == MAX) {
#define X(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                R8(X)
#undef X
            } else if (KIND == ADD_DPP) {
#define X(i) asm volatile("v_add_u32_dpp %0, %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(w[i]));
                R8(X)
#undef X
            } else if (KIND == MAX_DPP) {
#define X(i) asm volatile("v_max_i32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(w[i]));
                R8(X)
#undef X
            } else if (KIND == MOV_DPP) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(w[i]));
                R8(X)
#undef X
            } else if (KIND == CMP_SGPR) {
#define X(i) { uint64_t m; asm volatile("v_cmp_eq_i32_e64 %0, %1, %2" : "=s"(m) : "v"(v[i]), "v"(w[i])); sm ^= m; }
                R8(X)
#undef X
            } else if (KIND == CMP_VCC) {
#define X(i) asm volatile("v_cmp_eq_u32_e32 vcc, %0, %1" : : "v"(v[i]), "v"(w[i]) : "vcc");
                R8(X)
#undef X
            } else if (KIND == CNDMASK) {
                // one asm statement for the eight of them: between separate statements that each declare vcc clobbered the compiler pads with
                // s_nop (it has to assume a VALU write of vcc just before a read of it as a lane mask), and r2's row measured those pads --
                // 23 cycles per instruction.  vcc is written once, before the loop.
                asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n\tv_cndmask_b32_e32 %1, %1, %9, vcc\n\tv_cndmask_b32_e32 %2, %2, %10, vcc\n\tv_cndmask_b32_e32 %3, %3, %11, vcc\n\t"
                             "v_cndmask_b32_e32 %4, %4, %12, vcc\n\tv_cndmask_b32_e32 %5, %5, %13, vcc\n\tv_cndmask_b32_e32 %6, %6, %14, vcc\n\tv_cndmask_b32_e32 %7, %7, %15, vcc"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                             : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
            } else if (KIND == MAX_3REG) {
                // v_max_i32 with a destination that is neither source (the plain row above reuses its first source)
                asm volatile("v_max_i32 %0, %8, %9\n\tv_max_i32 %1, %9, %10\n\tv_max_i32 %2, %10, %11\n\tv_max_i32 %3, %11, %12\n\t"
                             "v_max_i32 %4, %12, %13\n\tv_max_i32 %5, %13, %14\n\tv_max_i32 %6, %14, %15\n\tv_max_i32 %7, %15, %8"
                             : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                             : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
            } else if (KIND == MAX3) {
#define X(i) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "v"(w[(i + 1) & 7]));
                R8(X)
#undef X
            } else if (KIND == READLANE) {
#define X(i) { int32_t s; asm volatile("v_readlane_b32 %0, %1, 63" : "=s"(s) : "v"(v[i])); sx ^= s; }
                R8(X)
#undef X
            } else if (KIND == WRITELANE) {
#define X(i) asm volatile("v_writelane_b32 %0, %1, 63" : "+v"(v[i]) : "s"(it));
                R8(X)
#undef X
            } else if (KIND == PK_ADD_I16) {
#define X(i) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                R8(X)
#undef X
            } else if (KIND == PK_MAX_I16) {
#define X(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                R8(X)
#undef X
            } else if (KIND == CNDMASK_E64) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "s"(sm));
                R8(X)
#undef X
            } else if (KIND == SW_MIX_SEL) {
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) {
                    int32_t top, bot;
                    uint64_t m0, m1;
                    asm volatile(
                        "s_bfe_u64 s[56:57], %[sm], 0x20000\n\t"
                        "v_mov_b32_dpp %[qc], %[qc] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                        "s_add_u32 %[sx], %[sx], 2\n\t"
                        "v_max_i32_dpp %[mm], %[H], %[H] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                        "v_writelane_b32 %[qc], s56, 63\n\t"
                        "v_cmp_eq_i32_e64 %[m1], %[mm], %[H]\n\t"
                        "v_cmp_eq_u32_e32 vcc, %[qc], %[tc]\n\t"
                        "v_cndmask_b32_e32 %[sc], %[vmis], %[vmat], vcc\n\t"
                        "v_add_u32_dpp %[hd], %[X], %[sc] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                        "v_subrev_u32_e32 %[mm], %[gap], %[mm]\n\t"
                        "s_mov_b32 s58, 1\n\t"
                        "v_cmp_ge_i32_e64 %[m0], %[hd], %[mm]\n\t"
                        "s_lshl1_add_u32 s59, s59, 1\n\t"
                        "v_cndmask_b32_e64 %[X], %[mm], %[hd], %[m0]\n\t"
                        "v_max3_i32 %[kb], %[kb], %[X], %[H]\n\t"
                        "v_readlane_b32 %[top], %[X], 0\n\t"
                        "v_readlane_b32 %[bot], %[X], 63\n\t"
                        "s_add_u32 %[sx], %[sx], 16\n\t"
                        "s_sub_u32 %[sx], %[sx], 1\n\t"
                        "s_cmp_gt_i32 %[top], %[bot]\n\t"
                        "s_cselect_b32 s58, 0, 1\n\t"
                        "v_swap_b32 %[H], %[X]\n\t"
                        : [qc] "+v"(v[0]), [tc] "+v"(v[1]), [H] "+v"(v[2]), [X] "+v"(v[3]), [kb] "+v"(v[4]), [mm] "=&v"(v[5]), [sc] "=&v"(v[6]), [hd] "=&v"(v[7]),
                          [sx] "+s"(sx), [top] "=&s"(top), [bot] "=&s"(bot), [m0] "=&s"(m0), [m1] "=&s"(m1)
                        : [vmis] "v"(w[0]), [vmat] "v"(w[1]), [gap] "s"(193), [sm] "s"(sm)
                        : "vcc", "scc", "s56", "s57", "s58", "s59");
                    sm ^= m0 + m1;
                }
            } else if (KIND == SW_MIX || KIND == SW_MIX_NOLANE) {
                // four DP steps' worth of the interior block's instruction mix (fzp_align.hip SWB_DOWN), no memory:
                // per step 13 VALU (mov_dpp, max_dpp, writelane, cmp->sgpr, cmp->vcc, cndmask, add(_dpp), subrev, max, cmp->sgpr,
                // [max3 every other step], readlane x2) and 9 SALU
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) {
                    int32_t top, bot;
                    uint64_t m0, m1;
                    if (KIND == SW_MIX) {
                        asm volatile(
                            "s_bfe_u64 s[56:57], %[sm], 0x20000\n\t"
                            "v_mov_b32_dpp %[qc], %[qc] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                            "s_add_u32 %[sx], %[sx], 2\n\t"
                            "v_max_i32_dpp %[mm], %[H], %[H] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                            "v_writelane_b32 %[qc], s56, 63\n\t"
                            "v_cmp_eq_i32_e64 %[m1], %[mm], %[H]\n\t"
                            "v_cmp_eq_u32_e32 vcc, %[qc], %[tc]\n\t"
                            "v_cndmask_b32_e32 %[sc], %[vmis], %[vmat], vcc\n\t"
                            "v_add_u32_dpp %[hd], %[X], %[sc] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                            "v_subrev_u32_e32 %[mm], %[gap], %[mm]\n\t"
                            "s_mov_b32 s58, 1\n\t"
                            "v_max_i32_e32 %[X], %[hd], %[mm]\n\t"
                            "v_cmp_eq_i32_e64 %[m0], %[X], %[hd]\n\t"
                            "s_lshl1_add_u32 s59, s59, 1\n\t"
                            "v_max3_i32 %[kb], %[kb], %[X], %[H]\n\t"
                            "v_readlane_b32 %[top], %[X], 0\n\t"
                            "v_readlane_b32 %[bot], %[X], 63\n\t"
                            "s_add_u32 %[sx], %[sx], 16\n\t"
                            "s_sub_u32 %[sx], %[sx], 1\n\t"
                            "s_cmp_gt_i32 %[top], %[bot]\n\t"
                            "s_cselect_b32 s58, 0, 1\n\t"
                            "v_swap_b32 %[H], %[X]\n\t"
                            : [qc] "+v"(v[0]), [tc] "+v"(v[1]), [H] "+v"(v[2]), [X] "+v"(v[3]), [kb] "+v"(v[4]), [mm] "=&v"(v[5]), [sc] "=&v"(v[6]), [hd] "=&v"(v[7]),
                              [sx] "+s"(sx), [top] "=&s"(top), [bot] "=&s"(bot), [m0] "=&s"(m0), [m1] "=&s"(m1)
                            : [vmis] "v"(w[0]), [vmat] "v"(w[1]), [gap] "s"(193), [sm] "s"(sm)
                            : "vcc", "scc", "s56", "s57", "s58", "s59");
                    } else {
                        asm volatile(
                            "s_bfe_u64 s[56:57], %[sm], 0x20000\n\t"
                            "v_mov_b32_dpp %[qc], %[qc] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                            "s_add_u32 %[sx], %[sx], 2\n\t"
                            "v_max_i32_dpp %[mm], %[H], %[H] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                            "v_cmp_eq_i32_e64 %[m1], %[mm], %[H]\n\t"
                            "v_cmp_eq_u32_e32 vcc, %[qc], %[tc]\n\t"
                            "v_cndmask_b32_e32 %[sc], %[vmis], %[vmat], vcc\n\t"
                            "v_add_u32_dpp %[hd], %[X], %[sc] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                            "v_subrev_u32_e32 %[mm], %[gap], %[mm]\n\t"
                            "s_mov_b32 s58, 1\n\t"
                            "v_max_i32_e32 %[X], %[hd], %[mm]\n\t"
                            "v_cmp_eq_i32_e64 %[m0], %[X], %[hd]\n\t"
                            "s_lshl1_add_u32 s59, s59, 1\n\t"
                            "v_max3_i32 %[kb], %[kb], %[X], %[H]\n\t"
                            "s_add_u32 %[sx], %[sx], 16\n\t"
                            "s_sub_u32 %[sx], %[sx], 1\n\t"
                            "s_cmp_gt_i32 %[sx], 5\n\t"
                            "s_cselect_b32 s58, 0, 1\n\t"
                            "v_swap_b32 %[H], %[X]\n\t"
                            : [qc] "+v"(v[0]), [tc] "+v"(v[1]), [H] "+v"(v[2]), [X] "+v"(v[3]), [kb] "+v"(v[4]), [mm] "=&v"(v[5]), [sc] "=&v"(v[6]), [hd] "=&v"(v[7]),
                              [sx] "+s"(sx), [m0] "=&s"(m0), [m1] "=&s"(m1)
                            : [vmis] "v"(w[0]), [vmat] "v"(w[1]), [gap] "s"(193), [sm] "s"(sm)
                            : "vcc", "scc", "s56", "s57", "s58", "s59");
                    }
                    sm ^= m0 + m1;
                }
            }
        }
    }
    int32_t acc = 0;
    for (int i = 0; i < 8; i++) acc += v[i];
    if (acc == 0x12345678) out[threadIdx.x] = (uint32_t)acc + (uint32_t)sm + (uint32_t)sx;
    if (sink && acc == 0x7654321) *sink = sm;
}

// shader clock: a chain of dependent v_add_u32 on one wave per CU issues one instruction every 4..8 cycles, which
// does not give the clock; instead time a fixed count of s_nop-free dependent SALU s_add_u32 (1 per cycle per wave
// is NOT guaranteed either).  So: report instructions/ns and let the reader divide by the clock rocm-smi reports;
// the ratio between W = 1 and W >= 2 is clock-free and is the quantity in question.
template <int KIND>
static void run_kind(uint32_t *d_out, int n_cu, int iters, int insts_per_iter_per_wave, int valu_per_group, double clk_ghz) {
    for (int W : {1, 2, 4, 8}) {
        const int wg_threads = 256 * (W < 4 ? W : 4);
        const int wgs_per_cu = W <= 4 ? 1 : W / 4;
        const int grid = n_cu * wgs_per_cu;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(wg_threads), 0, 0, d_out, iters / 8, (uint64_t *)nullptr);   // warm-up
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(wg_threads), 0, 0, d_out, iters, (uint64_t *)nullptr);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double inst_per_wave = (double)iters * insts_per_iter_per_wave;
        const double per_simd = inst_per_wave * W;                     // instructions one SIMD issued
        const double ns = best * 1e6;
        printf("%-48s W=%d  %8.3f ms  %7.4f inst/ns/SIMD  = %5.2f cycles/inst @%.2f GHz  chip %8.1f Ginst/s (valu-only %8.1f)\n", kind_name[KIND], W, best,
               per_simd / ns, clk_ghz * ns / per_simd, clk_ghz, per_simd * n_cu * 4 / ns, per_simd * n_cu * 4 / ns * valu_per_group / insts_per_iter_per_wave * 64.0 / 64.0);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
}


// ---- second table: plain "op dst, dst, src" forms, to sort instructions into the 2-cycle and the 4-cycle class
#define SIMPLE_KINDS(X) \
    X(0, "v_and_b32", "v_and_b32 %0, %0, %1") \
    X(1, "v_xor_b32", "v_xor_b32 %0, %0, %1") \
    X(2, "v_lshrrev_b32", "v_lshrrev_b32 %0, 2, %0") \
    X(3, "v_sub_u32", "v_sub_u32 %0, %0, %1") \
    X(4, "v_subrev_u32", "v_subrev_u32 %0, %1, %0") \
    X(5, "v_min_u32", "v_min_u32 %0, %0, %1") \
    X(6, "v_max_u32", "v_max_u32 %0, %0, %1") \
    X(7, "v_max_f32", "v_max_f32 %0, %0, %1") \
    X(8, "v_add_f32", "v_add_f32 %0, %0, %1") \
    X(9, "v_bfe_u32", "v_bfe_u32 %0, %0, %1, 2") \
    X(10, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %1") \
    X(11, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 2, %1") \
    X(12, "v_add3_u32", "v_add3_u32 %0, %0, %1, %1") \
    X(13, "v_mov_b32", "v_mov_b32 %0, %1") \
    X(14, "v_cndmask_b32 (vcc written once)", "v_cndmask_b32_e32 %0, %0, %1, vcc") \
    X(15, "v_sub_u32_dpp(wave_ror)", "v_sub_u32_dpp %0, %1, %1 wave_ror:1 row_mask:0xf bank_mask:0xf") \
    X(16, "v_max_i16", "v_max_i16 %0, %0, %1") \
    X(17, "v_add_u16", "v_add_u16 %0, %0, %1") \
    X(18, "v_max_i32 (two chains only, dependent)", "v_max_i32 %0, %0, %1") \
    X(19, "v_add_co_u32->vcc", "v_add_co_u32 %0, vcc, %0, %1") \
    X(20, "v_lshlrev_b64", "v_lshlrev_b64 %0, 1, %0") \
    X(21, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 2") \
    X(22, "v_perm_b32", "v_perm_b32 %0, %0, %1, %1") \
    X(23, "v_med3_i32", "v_med3_i32 %0, %0, %1, %1") \
    X(24, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %1") \
    X(25, "v_pk_sub_i16", "v_pk_sub_i16 %0, %0, %1") \
    X(26, "v_swap_b32", "v_swap_b32 %0, %1") \
    X(27, "v_bitop3_b32", "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96") \
    X(28, "v_lshrrev_b64", "v_lshrrev_b64 %0, 1, %0") \
    X(29, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %1") \
    X(30, "v_xnor_b32", "v_xnor_b32 %0, %0, %1") \
    X(31, "v_not_b32", "v_not_b32 %0, %0") \
    X(32, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %0, %1")

template <int K2>
__global__ void __launch_bounds__(1024) k2(uint32_t *out, int iters) {
    int32_t v[8], w[8];
    for (int i = 0; i < 8; i++) { v[i] = threadIdx.x * (i + 3) + 1; w[i] = (threadIdx.x ^ (i * 5)) + 7; }
    uint64_t v64[8];
    for (int i = 0; i < 8; i++) v64[i] = threadIdx.x * 77ull + i;
    asm volatile("v_cmp_gt_i32_e32 vcc, %0, %1" ::"v"(v[0]), "v"(w[0]) : "vcc");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#define X(id, name, text)                                                                                             \
    if (K2 == id) {                                                                                                   \
        if (id == 20 || id == 28) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(text : "+v"(v64[i]) : "v"(w[i])); } \
        else if (id == 18) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(text : "+v"(v[i & 1]) : "v"(w[i])); } \
        else if (id == 26) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(text : "+v"(v[i]), "+v"(w[i])); } \
        else if (id == 14) { asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n\tv_cndmask_b32_e32 %1, %1, %9, vcc\n\tv_cndmask_b32_e32 %2, %2, %10, vcc\n\tv_cndmask_b32_e32 %3, %3, %11, vcc\n\tv_cndmask_b32_e32 %4, %4, %12, vcc\n\tv_cndmask_b32_e32 %5, %5, %13, vcc\n\tv_cndmask_b32_e32 %6, %6, %14, vcc\n\tv_cndmask_b32_e32 %7, %7, %15, vcc" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7])); } \
        else if (id == 15 || id == 19) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(text : "+v"(v[i]) : "v"(w[i]) : "vcc"); } \
        else { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(text : "+v"(v[i]) : "v"(w[i])); }          \
    }
            SIMPLE_KINDS(X)
#undef X
        }
    }
    int32_t acc = 0;
    for (int i = 0; i < 8; i++) acc += v[i] + w[i] + (int32_t)v64[i];
    if (acc == 0x12345678) out[threadIdx.x] = (uint32_t)acc;
}

template <int K2>
static void run_simple(uint32_t *d_out, int n_cu, int iters, const char *name, double clk_ghz) {
    for (int W : {1, 2, 8}) {
        const int wg_threads = 256 * (W < 4 ? W : 4);
        const int grid = n_cu * (W <= 4 ? 1 : W / 4);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k2<K2>, dim3(grid), dim3(wg_threads), 0, 0, d_out, iters / 8);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k2<K2>, dim3(grid), dim3(wg_threads), 0, 0, d_out, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double per_simd = (double)iters * 64 * W, ns = best * 1e6;
        printf("%-48s W=%d  %8.3f ms  %7.4f inst/ns/SIMD  = %5.2f cycles/inst @%.2f GHz\n", name, W, best, per_simd / ns, clk_ghz * ns / per_simd, clk_ghz);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
}

int main(int argc, char **argv) {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int n_cu = prop.multiProcessorCount;
    const double clk_ghz = argc > 1 ? atof(argv[1]) : prop.clockRate * 1e-6;
    printf("# %s  %d CUs  clockRate %.3f GHz (used for cycles/inst)\n", prop.gcnArchName, n_cu, clk_ghz);
    uint32_t *d_out;
    CK(hipMalloc(&d_out, 1 << 20));
    const int iters = 20000;
    run_kind<ADD>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<MAX>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<ADD_DPP>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<MAX_DPP>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<MOV_DPP>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<CMP_SGPR>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<CMP_VCC>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<CNDMASK>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<MAX3>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<READLANE>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<WRITELANE>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<PK_ADD_I16>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<PK_MAX_I16>(d_out, n_cu, iters, 64, 64, clk_ghz);
    // the mixes: per outer iteration 8 x 4 steps; VALU per step 14 (13 + v_swap standing in for the role swap) / 11
    run_kind<SW_MIX>(d_out, n_cu, iters / 4, 8 * 4 * 14, 8 * 4 * 14, clk_ghz);
    run_kind<SW_MIX_NOLANE>(d_out, n_cu, iters / 4, 8 * 4 * 11, 8 * 4 * 11, clk_ghz);
    run_kind<SW_MIX_SEL>(d_out, n_cu, iters / 4, 8 * 4 * 14, 8 * 4 * 14, clk_ghz);
    run_kind<CNDMASK_E64>(d_out, n_cu, iters, 64, 64, clk_ghz);
    run_kind<MAX_3REG>(d_out, n_cu, iters, 64, 64, clk_ghz);
    if (argc > 2) { CK(hipFree(d_out)); return 0; }
#define X(id, name, text) run_simple<id>(d_out, n_cu, iters, name, clk_ghz);
    SIMPLE_KINDS(X)
#undef X
    CK(hipFree(d_out));
    return 0;
}
