#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
extern "C" int64_t rag_lz4_compress_bound(int64_t n);
extern "C" int64_t rag_lz4_block_compress(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap);
extern "C" uint32_t rag_xxh32(const uint8_t* p, int64_t n, uint32_t seed);
static bool decode(const std::vector<uint8_t>& src, size_t n, std::vector<uint8_t>& out) {
    size_t i = 0; out.clear();
    while (i < n) {
        uint8_t tok = src[i++]; size_t lit = tok >> 4;
        if (lit == 15) { uint8_t b; do { if (i >= n) return false; b = src[i++]; lit += b; } while (b == 255); }
        if (i + lit > n) return false;
        out.insert(out.end(), src.begin() + i, src.begin() + i + lit); i += lit;
        if (i >= n) break;
        if (i + 2 > n) return false;
        size_t off = src[i] | (src[i + 1] << 8); i += 2;
        size_t ml = tok & 15;
        if (ml == 15) { uint8_t b; do { if (i >= n) return false; b = src[i++]; ml += b; } while (b == 255); }
        ml += 4;
        if (off == 0 || off > out.size()) return false;
        size_t st = out.size() - off;
        for (size_t j = 0; j < ml; ++j) out.push_back(out[st + j]);
    }
    return true;
}
int main() {
    std::mt19937_64 rng(1);
    for (int t = 0; t < 20000; ++t) {
        size_t n = rng() % (t % 50 == 0 ? 200000 : 600);
        std::vector<uint8_t> a(n);
        int mode = rng() % 4;
        for (size_t i = 0; i < n; ++i) a[i] = mode == 0 ? rng() : mode == 1 ? (uint8_t)(i / 7) : mode == 2 ? (uint8_t)(rng() % 3) : (i > 40 ? a[i - 1 - rng() % 40] : rng());
        int64_t cap = rag_lz4_compress_bound(n);
        std::vector<uint8_t> d(cap);          // exact-size heap buffers: ASan sees any overrun
        int64_t c = rag_lz4_block_compress(a.data(), n, d.data(), cap);
        if (c < 0) { printf("compress failed n=%zu\n", n); return 1; }
        std::vector<uint8_t> back;
        if (!decode(d, c, back) || back != a) { printf("mismatch at trial %d n=%zu\n", t, n); return 1; }
        // a too-small destination must be refused, not overrun
        if (c > 4) { std::vector<uint8_t> small(c - 1); if (rag_lz4_block_compress(a.data(), n, small.data(), c - 1) != -1) { printf("no refusal\n"); return 1; } }
        (void)rag_xxh32(a.data(), n, (uint32_t)t);
    }
    printf("fuzz ok\n");
    return 0;
}
