// k_revcomp word-at-a-time (r5) against the base-at-a-time form it replaced, on random sequences of 1..500 000 bases, four rounds.  The first word-at-a-time form took its
// 16-base window through kmer_at (one 64-bit window made of two adjacent words) and came back with wrong upper halves in ~0.2 % of the words, different ones every run; with two
// plain word loads and a funnel shift the two forms agree.  Build: hipcc --offload-arch=gfx950 -O3 -o revcomp_check revcomp_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
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
template <int NEW>
__global__ void k_revcomp(const uint32_t *__restrict__ pk, const int64_t *__restrict__ woff, const int64_t *__restrict__ len, uint32_t *__restrict__ out) {
    const int64_t sq = blockIdx.x;
    const int64_t n = len[sq];
    const int64_t nw = ((n + 15) / 16 + 8 + 1) & ~1LL;
    const uint32_t *src = pk + woff[sq];
    uint32_t *dst = out + woff[sq];
    for (int64_t w = (int64_t)blockIdx.y * 256 + threadIdx.x; w < nw; w += (int64_t)gridDim.y * 256) {
        uint32_t v = 0;
        if (NEW) {
            const int64_t left = n - w * 16;
            if (left > 0) {
                const int64_t s0 = left - 16;
                uint32_t key;
                if (s0 >= 0) {
                    const uint32_t sh = (uint32_t)(s0 & 15) * 2u;
                    const uint32_t lo = src[s0 >> 4], hi = src[(s0 >> 4) + 1];
                    key = sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
                } else key = src[0] << (2 * (uint32_t)(-s0));
                v = rc_key(key, 16);
                if (left < 16) v &= (1u << (2 * (uint32_t)left)) - 1u;
            }
        } else {
            for (int m = 0; m < 16; m++) { const int64_t x = w * 16 + m; if (x < n) v |= (3u - base_at(src, n - 1 - x)) << (2 * m); }
        }
        dst[w] = v;
    }
}
int main2() {
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
    hipLaunchKernelGGL(k_revcomp<0>, dim3(NS, 64), dim3(256), 0, 0, dpk, dwo, dl, o0);
    hipLaunchKernelGGL(k_revcomp<1>, dim3(NS, 64), dim3(256), 0, 0, dpk, dwo, dl, o1);
    std::vector<uint32_t> a(pk.size()), b(pk.size());
    hipMemcpy(a.data(), o0, pk.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o1, pk.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int s = 0; s < NS; s++) { int64_t nw = ((len[s] + 15) / 16 + 8 + 1) & ~1LL; for (int64_t w = 0; w < nw; w++) if (a[woff[s] + w] != b[woff[s] + w]) { if (bad < 5) printf("seq %d n %lld word %lld: new %08x old %08x\n", s, (long long)len[s], (long long)w, b[woff[s] + w], a[woff[s] + w]); bad++; } }
    printf("bad words: %d\n", bad);
    {   // host reference for the first bad sequence
        int s = 22; int64_t n = len[s];
        auto base = [&](int64_t i) { return (pk[woff[s] + (i >> 4)] >> ((i & 15) * 2)) & 3u; };
        for (int64_t w = 5054; w < 5062; w++) { uint32_t v = 0; for (int m = 0; m < 16; m++) { int64_t x = w * 16 + m; if (x < n) v |= (3u - base(n - 1 - x)) << (2 * m); } printf("w %lld host %08x old %08x new %08x\n", (long long)w, v, a[woff[s] + w], b[woff[s] + w]); }
        int firstbad = -1; int64_t nw = ((n + 15) / 16 + 8 + 1) & ~1LL; int cnt = 0; int64_t lastbad = -1;
        for (int64_t w = 0; w < nw; w++) if (a[woff[s] + w] != b[woff[s] + w]) { if (firstbad < 0) firstbad = (int)w; lastbad = w; cnt++; }
        printf("seq 22: first bad %d last bad %lld count %d of %lld\n", firstbad, (long long)lastbad, cnt, (long long)nw);
    }
    return 0;
}
int main() { int r = 0; for (int k = 0; k < 4; k++) r |= main2(); return r; }
