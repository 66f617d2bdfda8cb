// rag_text.cpp — host-side pair encoder of the STAND-IN tokenizer (rag_inference_pipeline_amd/model_source.py
// HashTokenizer), for ASCII text.  No vocabulary exists offline, so synthetic models map lower-cased word / punctuation
// pieces to ids by crc32; a deployment with a real checkpoint tokenises with the `tokenizers` library instead.  With the
// cross-encoder pass at 3.5 ms per 640 pairs the Python loop over pairs (regex, list building: ~8 us per pair) had
// become the slowest stage of a rerank batch (reference reranker.py:237-246 builds the same pairs with
// tokenizer(pairs, padding=True, truncation=True)); this does the same work in one call and writes the PACKED arrays
// the transformer takes (ids, token types, cu_seqlens).  Host code, no GPU involved.
#include <cstdint>
#include <cstring>

#include "../../include/rag_amd.h"

namespace {

struct Crc32Table {
    uint32_t t[256];
    Crc32Table() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[i] = c;
        }
    }
};
const Crc32Table kCrc;

inline bool is_word(uint8_t c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
// what Python's re `\s` matches among ASCII characters (str patterns): \t \n \v \f \r, \x1c-\x1f, space
inline bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); }

// Pieces of one text -> ids appended at out[0 .. cap); returns the count, -1 on a non-ASCII byte, -2 if cap is too small.
// A piece is a maximal run of [A-Za-z0-9_] (lower-cased) or one other non-space character:
// re.compile(r"\w+|[^\w\s]").findall(text.lower()) restricted to ASCII.
int64_t pieces(const uint8_t* s, int64_t n, int32_t first_id, int32_t span, int32_t* out, int64_t cap) {
    int64_t count = 0, i = 0;
    while (i < n) {
        const uint8_t c = s[i];
        if (c >= 0x80) return -1;
        if (is_space(c)) {
            ++i;
            continue;
        }
        uint32_t crc = 0xFFFFFFFFu;
        if (is_word(c)) {
            while (i < n && s[i] < 0x80 && is_word(s[i])) {
                uint8_t ch = s[i++];
                if (ch >= 'A' && ch <= 'Z') ch = (uint8_t)(ch + 32);
                crc = kCrc.t[(crc ^ ch) & 0xFF] ^ (crc >> 8);
            }
        } else {
            crc = kCrc.t[(crc ^ c) & 0xFF] ^ (crc >> 8);
            ++i;
        }
        if (count >= cap) return -2;
        out[count++] = first_id + (int32_t)((crc ^ 0xFFFFFFFFu) % (uint32_t)span);
    }
    return count;
}

}  // namespace

extern "C" int64_t rag_hash_encode_pairs(const uint8_t* const* first, const int64_t* first_len, const uint8_t* const* second,
                                         const int64_t* second_len, int64_t n_pairs, int32_t max_length, int32_t roberta,
                                         int32_t first_id, int32_t span, int32_t cls_id, int32_t sep_id, int32_t* ids_out,
                                         int32_t* types_out, int32_t* cu_out, int64_t cap) {
    if (!first || !first_len || !second || !second_len || !ids_out || !cu_out || n_pairs < 0 || span <= 0 || max_length < 0)
        return -3;
    const int64_t n_special = roberta ? 4 : 3;
    int64_t at = 0;
    cu_out[0] = 0;
    for (int64_t p = 0; p < n_pairs; ++p) {
        // layout in place: [cls] a... [sep] ([sep]) b... [sep]; a and b are written at their final positions once their
        // truncated lengths are known, so tokenise both into the free space first
        if (cap - at < n_special) return -2;
        int32_t* pa = ids_out + at + 1;
        const int64_t na = pieces(first[p], first_len[p], first_id, span, pa, cap - at - n_special);
        if (na < 0) return na;
        int32_t* pb = pa + na + (roberta ? 2 : 1);
        const int64_t room_b = cap - (pb - ids_out) - 1;
        if (room_b < 0) return -2;
        const int64_t nb = pieces(second[p], second_len[p], first_id, span, pb, room_b);
        if (nb < 0) return nb;
        int64_t ka = na, kb = nb;   // longest_first truncation: drop from the longer side, the second on ties
        while (ka + kb + n_special > max_length && (ka || kb)) {
            if (kb >= ka) --kb; else --ka;
        }
        int32_t* dst = ids_out + at;
        *dst++ = cls_id;
        dst += ka;                                   // a's first ka ids are already in place
        *dst++ = sep_id;
        if (roberta) *dst++ = sep_id;
        if (dst != pb) std::memmove(dst, pb, (size_t)kb * sizeof(int32_t));
        dst += kb;
        *dst++ = sep_id;
        const int64_t len = dst - (ids_out + at);
        if (types_out) {
            const int64_t n0 = roberta ? len : ka + 2;   // BERT: [cls] a [sep] are type 0, b [sep] type 1
            for (int64_t j = 0; j < len; ++j) types_out[at + j] = j < n0 ? 0 : 1;
        }
        at += len;
        cu_out[p + 1] = (int32_t)at;
    }
    return at;
}
