// rag_lz4.cpp — LZ4 block compressor + xxHash32 for the `compressed` document payload mode
// (reference services/retrieval/api.py:516-523: lz4.frame.compress(msgspec.json.encode(docs)); the
// generation node reads it back with lz4.frame.decompress, services/generation/service.py:429).
// Written from the public LZ4 block / frame format descriptions; host code, no GPU involved.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rag_amd.h"

namespace {

inline uint32_t read32(const uint8_t* p) {
    uint32_t v;
    std::memcpy(&v, p, 4);
    return v;
}
inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// one sequence: literals [lit, lit + nlit), then a match of `mlen` bytes `offset` back (mlen == 0: last literals)
uint8_t* put_sequence(uint8_t* op, const uint8_t* op_end, const uint8_t* lit, int64_t nlit, int64_t offset, int64_t mlen) {
    const int64_t need = 1 + nlit / 255 + 1 + nlit + 2 + (mlen > 0 ? (mlen - 4) / 255 + 1 : 0);
    if (op_end - op < need) return nullptr;
    uint8_t* token = op++;
    int64_t l = nlit;
    if (l >= 15) {
        *token = 15 << 4;
        for (l -= 15; l >= 255; l -= 255) *op++ = 255;
        *op++ = (uint8_t)l;
    } else {
        *token = (uint8_t)(l << 4);
    }
    if (nlit > 0) std::memcpy(op, lit, (size_t)nlit);
    op += nlit;
    if (mlen > 0) {
        *op++ = (uint8_t)(offset & 0xFF);
        *op++ = (uint8_t)(offset >> 8);
        int64_t m = mlen - 4;
        if (m >= 15) {
            *token |= 15;
            for (m -= 15; m >= 255; m -= 255) *op++ = 255;
            *op++ = (uint8_t)m;
        } else {
            *token |= (uint8_t)m;
        }
    }
    return op;
}

}  // namespace

extern "C" int64_t rag_lz4_compress_bound(int64_t n) { return n < 0 ? 0 : n + n / 255 + 16; }

extern "C" int64_t rag_lz4_block_compress(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap) {
    if ((!src && n > 0) || !dst || n < 0 || cap < 0) return -1;
    uint8_t* op = dst;
    const uint8_t* const op_end = dst + cap;
    int64_t anchor = 0;
    if (n >= 13) {
        constexpr int kHashBits = 14;
        std::vector<int32_t> table((size_t)1 << kHashBits, -1);
        const int64_t mflimit = n - 12;    // a match may not start in the last 12 bytes
        const int64_t matchlimit = n - 5;  // nor run into the last 5
        int64_t ip = 0;
        while (ip < mflimit) {
            const uint32_t seq = read32(src + ip);
            const uint32_t h = (seq * 2654435761u) >> (32 - kHashBits);
            const int64_t ref = table[h];
            table[h] = (int32_t)ip;
            if (ref >= 0 && ip - ref <= 65535 && read32(src + ref) == seq) {
                int64_t mlen = 4;
                while (ip + mlen < matchlimit && src[ref + mlen] == src[ip + mlen]) ++mlen;
                op = put_sequence(op, op_end, src + anchor, ip - anchor, ip - ref, mlen);
                if (!op) return -1;
                ip += mlen;
                anchor = ip;
            } else {
                ++ip;
            }
        }
    }
    op = put_sequence(op, op_end, src + anchor, n - anchor, 0, 0);
    if (!op) return -1;
    return op - dst;
}

extern "C" uint32_t rag_xxh32(const uint8_t* p, int64_t n, uint32_t seed) {
    constexpr uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const uint8_t* const end = p + (n > 0 ? n : 0);
    uint32_t h;
    if (n >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t* const limit = end - 16;
        do {
            v1 = rotl32(v1 + read32(p) * P2, 13) * P1;
            v2 = rotl32(v2 + read32(p + 4) * P2, 13) * P1;
            v3 = rotl32(v3 + read32(p + 8) * P2, 13) * P1;
            v4 = rotl32(v4 + read32(p + 12) * P2, 13) * P1;
            p += 16;
        } while (p <= limit);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = seed + P5;
    }
    h += (uint32_t)(n > 0 ? n : 0);
    while (p + 4 <= end) {
        h = rotl32(h + read32(p) * P3, 17) * P4;
        p += 4;
    }
    while (p < end) {
        h = rotl32(h + (*p) * P5, 11) * P1;
        ++p;
    }
    h ^= h >> 15;
    h *= P2;
    h ^= h >> 13;
    h *= P3;
    h ^= h >> 16;
    return h;
}
