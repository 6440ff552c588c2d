// Why did k_revcomp's first word-at-a-time form (its 16-base window through kmer_at: two adjacent 32-bit words as one 64-bit window) return wrong UPPER halves in ~0.2 % of the
// words, different ones every run (HISTORY.md section 14, r5)?  This probe runs (a) that exact form again and says WHERE the bad words are -- the byte address of the window's
// low word modulo 4096 -- and (b) the bare pattern: every 4-byte-aligned pair of a buffer whose word i holds i, read as the compiler reads kmer_at's two words, checked in place.
// Build: hipcc --offload-arch=gfx950 -O3 --save-temps -o unaligned_pair unaligned_pair.hip   (the .s beside it shows what the two loads became)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
__device__ __forceinline__ uint32_t base_at(const uint32_t *__restrict__ pk, int64_t i) { return (pk[i >> 4] >> ((i & 15) * 2)) & 3u; }
__device__ __forceinline__ uint32_t kmer_at(const uint32_t *__restrict__ pk, int64_t p, int k) {
    uint64_t w = (uint64_t)pk[p >> 4] | ((uint64_t)pk[(p >> 4) + 1] << 32);
    uint32_t key = (uint32_t)(w >> ((p & 15) * 2));
    return k < 16 ? (key & ((1u << (2 * k)) - 1u)) : key;
}
__device__ __forceinline__ uint32_t rc_key(uint32_t key, int k) {
    uint32_t x = ~key;
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = __builtin_bswap32(x);
    return k < 16 ? (x >> (32 - 2 * k)) : x;
}
// FORM 0: a base at a time (the reference); 1: the window through kmer_at (the failing form); 2: kmer_at's window, but the destination written through a pointer that is
// NOT restrict (the r5 observation: "only with __restrict__")
// 3: kmer_at's two loads, the 64-bit window pinned in registers before the shift (empty asm); 4: the shift amount pinned; 5: the window's two halves shifted as 32-bit words;
// 6: form 1 + what the kernel SAW (low word, high word, shift) written beside the result
template <int FORM>
__global__ void k_revcomp(const uint32_t *__restrict__ pk, const int64_t *__restrict__ woff, const int64_t *__restrict__ len, uint32_t *__restrict__ out, uint4 *__restrict__ dbg = nullptr) {
    const int64_t sq = blockIdx.x;
    const int64_t n = len[sq];
    const int64_t nw = ((n + 15) / 16 + 8 + 1) & ~1LL;
    const uint32_t *src = pk + woff[sq];
    uint32_t *dst = out + woff[sq];
    for (int64_t w = (int64_t)blockIdx.y * 256 + threadIdx.x; w < nw; w += (int64_t)gridDim.y * 256) {
        uint32_t v = 0;
        if (FORM) {
            const int64_t left = n - w * 16;
            if (left > 0) {
                const int64_t s0 = left - 16;
                uint32_t key;
                if (FORM == 2 && s0 >= 0) {      // the same window by ONE spelled-out 8-byte load into registers of its own
                    uint64_t w2;
                    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(w2) : "v"(src + (s0 >> 4)) : "memory");
                    key = (uint32_t)(w2 >> ((s0 & 15) * 2));
                } else if (FORM == 7) {      // form 1 in a kernel that holds 24 registers instead of 16 (does it matter that the shift amount sits in the wave's LAST register?)
                    asm volatile("" ::: "v23");
                    key = s0 >= 0 ? kmer_at(src, s0, 16) : src[0] << (2 * (uint32_t)(-s0));
                } else if (FORM >= 3 && s0 >= 0) {
                    uint64_t w2 = (uint64_t)src[s0 >> 4] | ((uint64_t)src[(s0 >> 4) + 1] << 32);
                    uint32_t sh = (uint32_t)(s0 & 15) * 2u;
                    if (FORM == 3) asm volatile("" : "+v"(w2));
                    if (FORM == 4) asm volatile("" : "+v"(sh));
                    if (FORM == 5) { const uint32_t lo = (uint32_t)w2, hi = (uint32_t)(w2 >> 32); key = sh ? (lo >> sh) | (hi << (32u - sh)) : lo; }
                    else key = (uint32_t)(w2 >> sh);
                    if (FORM == 6) dbg[woff[sq] + w] = make_uint4((uint32_t)w2, (uint32_t)(w2 >> 32), sh, key);
                } else key = s0 >= 0 ? kmer_at(src, s0, 16) : src[0] << (2 * (uint32_t)(-s0));
                v = rc_key(key, 16);
                if (left < 16) v &= (1u << (2 * (uint32_t)left)) - 1u;
            }
        } else {
            for (int m = 0; m < 16; m++) { const int64_t x = w * 16 + m; if (x < n) v |= (3u - base_at(src, n - 1 - x)) << (2 * m); }
        }
        dst[w] = v;
    }
}
// (b) the bare pattern: word i of p holds i.  MODE bit 0: the lanes walk DOWN the buffer (as k_revcomp's do); bit 1: the destination registers are not the address registers
template <int MODE>
__global__ void k_pairs(const uint32_t *__restrict__ p, int64_t n, unsigned long long *__restrict__ bad, unsigned long long *__restrict__ where) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j + 1 < n; j += (int64_t)gridDim.x * 256) {
        const int64_t i = (MODE & 1) ? n - 2 - j : j;
        uint64_t w;      // (ONE 8-byte load at a 4-byte aligned address, spelled out: left to itself the compiler keeps two loads here, and merges them in k_revcomp)
        if (MODE & 2) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(p + i) : "memory");
        else asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p + i) : "memory");
        if ((uint32_t)w != (uint32_t)i || (uint32_t)(w >> 32) != (uint32_t)(i + 1)) { const unsigned long long k = atomicAdd(bad, 1ull); if (k < 64) where[k] = (unsigned long long)i; }
    }
}
int main() {
    srand(5);
    const int NS = 40;
    std::vector<uint32_t> pk; std::vector<int64_t> woff, len;
    for (int s = 0; s < NS; s++) {
        int64_t n = s < 20 ? s * 3 + 1 : 100000 + rand() % 400000;
        int64_t nw = ((n + 15) / 16 + 8 + 1) & ~1LL;
        woff.push_back((int64_t)pk.size()); len.push_back(n);
        std::vector<uint32_t> w(nw, 0);
        for (int64_t i = 0; i < n; i++) w[i >> 4] |= (uint32_t)(rand() & 3) << (2 * (i & 15));
        pk.insert(pk.end(), w.begin(), w.end());
    }
    for (int i = 0; i < 16; i++) pk.push_back(0);
    uint32_t *dpk, *o0, *o1; int64_t *dwo, *dl;
    hipMalloc(&dpk, pk.size() * 4); hipMalloc(&o0, pk.size() * 4); hipMalloc(&o1, pk.size() * 4); hipMalloc(&dwo, NS * 8); hipMalloc(&dl, NS * 8);
    hipMemcpy(dpk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dwo, woff.data(), NS * 8, hipMemcpyHostToDevice); hipMemcpy(dl, len.data(), NS * 8, hipMemcpyHostToDevice);
    printf("buffer at %p (offset in its 4 KB page: %llu)\n", (void *)dpk, (unsigned long long)((uintptr_t)dpk & 4095));
    std::vector<uint32_t> a(pk.size()), b(pk.size());
    uint4 *ddbg; hipMalloc(&ddbg, pk.size() * 16);
    std::vector<uint4> hdbg(pk.size());
    const char *what[8] = {"", "window through kmer_at", "window by one spelled-out 8-byte load", "kmer_at, window pinned before the shift", "kmer_at, shift amount pinned", "kmer_at's loads, 32-bit shifts", "kmer_at + what it saw", "window through kmer_at, 24 registers"};
    for (int round = 0; round < 14; round++) {
        const int form = round >= 12 ? 7 : 1 + round % 6;      // 1..6, 1..6, then 7 twice
        hipMemset(o0, 0xee, pk.size() * 4); hipMemset(o1, 0xee, pk.size() * 4);
        hipLaunchKernelGGL(k_revcomp<0>, dim3(NS, 64), dim3(256), 0, 0, dpk, dwo, dl, o0, (uint4 *)nullptr);
        if (form == 1) hipLaunchKernelGGL(k_revcomp<1>, dim3(NS, 64), dim3(256), 0, 0, dpk, dwo, dl, o1, ddbg);
        if (form == 2) hipLaunchKernelGGL(k_revcomp<2>, dim3(NS, 64), dim3(256), 0, 0, dpk, dwo, dl, o1, ddbg);
        if (form == 3) hipLaunchKernelGGL(k_revcomp<3>, dim3(NS, 64), dim3(256), 0, 0, dpk, dwo, dl, o1, ddbg);
        if (form == 4) hipLaunchKernelGGL(k_revcomp<4>, dim3(NS, 64), dim3(256), 0, 0, dpk, dwo, dl, o1, ddbg);
        if (form == 5) hipLaunchKernelGGL(k_revcomp<5>, dim3(NS, 64), dim3(256), 0, 0, dpk, dwo, dl, o1, ddbg);
        if (form == 7) hipLaunchKernelGGL(k_revcomp<7>, dim3(NS, 64), dim3(256), 0, 0, dpk, dwo, dl, o1, ddbg);
        if (form == 6) { hipLaunchKernelGGL(k_revcomp<6>, dim3(NS, 64), dim3(256), 0, 0, dpk, dwo, dl, o1, ddbg); hipMemcpy(hdbg.data(), ddbg, pk.size() * 16, hipMemcpyDeviceToHost); }
        hipMemcpy(a.data(), o0, pk.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o1, pk.size() * 4, hipMemcpyDeviceToHost);
        long bad = 0, bad_hi_only = 0, n_explained = 0;
        std::map<int, long> shift_minus_4lane;
        std::map<long, long> by_page_off;      // the low word's byte offset in its 4 KB page / 64
        for (int s = 0; s < NS; s++) {
            const int64_t n = len[s], nw = ((n + 15) / 16 + 8 + 1) & ~1LL;
            for (int64_t w = 0; w < nw; w++)
                if (a[woff[s] + w] != b[woff[s] + w]) {
                    bad++;
                    const int64_t s0 = n - w * 16 - 16;
                    const uintptr_t addr = (uintptr_t)(dpk + woff[s] + (s0 >= 0 ? (s0 >> 4) : 0));
                    by_page_off[(long)((addr & 4095) / 64)]++;
                    // a wrong UPPER word of the window shows in the LOW bases of the reverse complement: is the rest right?
                    const uint32_t sh = s0 >= 0 ? (uint32_t)(s0 & 15) * 2u : 0u;
                    const uint32_t diff = a[woff[s] + w] ^ b[woff[s] + w];
                    if (sh && (diff >> (sh)) == 0) bad_hi_only++;
                    if ((form == 1 || form == 7) && s0 >= 0 && n_explained >= 0) {      // which shift of the RIGHT two words gives what came out?
                        const uint64_t w64 = (uint64_t)pk[woff[s] + (s0 >> 4)] | ((uint64_t)pk[woff[s] + (s0 >> 4) + 1] << 32);
                        int found = -1;
                        for (int q = 0; q < 64 && found < 0; q++) {
                            uint32_t x = ~(uint32_t)(w64 >> q);
                            x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
                            x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
                            x = __builtin_bswap32(x);
                            if (x == b[woff[s] + w]) found = q;
                        }
                        if (found >= 0) { n_explained++; shift_minus_4lane[(found - 4 * (int)(w & 63)) & 63]++; }
                        if (bad <= 8) printf("    word %lld (lane %lld of its wave): the right two words shifted by %d give it (the right shift is %u)\n", (long long)w, (long long)(w & 63), found, sh);
                    }
                    if (form == 6 && bad <= 12) {
                        const uint4 d = hdbg[woff[s] + w];
                        const int64_t wi = s0 >> 4;
                        printf("    saw lo %08x hi %08x shift %u key %08x | memory holds lo %08x hi %08x, shift should be %u\n", d.x, d.y, d.z, d.w, pk[woff[s] + wi], pk[woff[s] + wi + 1], sh);
                    }
                    if (bad <= 6) printf("  round %d seq %d n %lld word %lld: got %08x want %08x; window's low word at page offset %llu, shift %u\n", round, s, (long long)n, (long long)w, b[woff[s] + w], a[woff[s] + w], (unsigned long long)(addr & 4095), sh);
                }
        }
        printf("round %d (%s): %ld bad words of %zu (%ld of them wrong only where the window's upper word lands); low word's offset in its 4 KB page, in 64-byte lines:", round, what[form], bad, pk.size(), bad_hi_only);
        for (auto &kv : by_page_off) printf(" %ld:%ld", kv.first, kv.second);
        printf("\n");
        if (form == 1 || form == 7) { printf("  explained as the right data under a wrong shift: %ld of %ld; (shift - 4 * lane) mod 64:", n_explained, bad); for (auto &kv : shift_minus_4lane) printf(" %d:%ld", kv.first, kv.second); printf("\n"); }
    }
    // (b)
    const int64_t N = 64 << 20;
    uint32_t *p; unsigned long long *dbad, *dwhere;
    hipMalloc(&p, N * 4); hipMalloc(&dbad, 8); hipMalloc(&dwhere, 64 * 8);
    { std::vector<uint32_t> h((size_t)N); for (int64_t i = 0; i < N; i++) h[(size_t)i] = (uint32_t)i; hipMemcpy(p, h.data(), (size_t)N * 4, hipMemcpyHostToDevice); }
    for (int round = 0; round < 8; round++) {
        hipMemset(dbad, 0, 8);
        const int mode = round & 3;
        if (mode == 0) hipLaunchKernelGGL(k_pairs<0>, dim3(4096), dim3(256), 0, 0, p, N, dbad, dwhere);
        if (mode == 1) hipLaunchKernelGGL(k_pairs<1>, dim3(4096), dim3(256), 0, 0, p, N, dbad, dwhere);
        if (mode == 2) hipLaunchKernelGGL(k_pairs<2>, dim3(4096), dim3(256), 0, 0, p, N, dbad, dwhere);
        if (mode == 3) hipLaunchKernelGGL(k_pairs<3>, dim3(4096), dim3(256), 0, 0, p, N, dbad, dwhere);
        unsigned long long hb = 0, hw[64];
        hipMemcpy(&hb, dbad, 8, hipMemcpyDeviceToHost); hipMemcpy(hw, dwhere, 64 * 8, hipMemcpyDeviceToHost);
        printf("pairs mode %d (%s, %s): %llu bad pairs of %lld", mode, (mode & 1) ? "lanes walk down" : "lanes walk up", (mode & 2) ? "vdst != vaddr" : "vdst may be vaddr", hb, (long long)N - 1);
        for (unsigned long long k = 0; k < hb && k < 8; k++) printf(" [i=%llu page offset %llu]", hw[k], (unsigned long long)(((uintptr_t)p + hw[k] * 4) & 4095));
        printf("\n");
    }
    return 0;
}
