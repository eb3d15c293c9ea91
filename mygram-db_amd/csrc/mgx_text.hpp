// mgx_text.hpp — the ONE host implementation of the reference's text rules that define gram identity and doc length,
// shared by the column builder (mgx_columns.cpp, inside libmygram_gpu.so) and the C++17 shim (csrc/shim/, inside
// libmygram_shim.so): UTF-8 decoding, the CJK-ideograph predicate, code-point windows. Header-only so that both
// libraries compile the same code. Paths are relative to the reference tree.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

namespace mgx::text {

// src/utils/string_utils.cpp:94-164 (TryParseUtf8Char): bytes consumed, or -1 for an invalid sequence
inline int ParseUtf8(const uint8_t* d, size_t avail, uint32_t* cp) {
  const uint8_t b0 = d[0];
  if (b0 < 0x80) {
    *cp = b0;
    return 1;
  }
  if ((b0 & 0xE0) == 0xC0) {
    if (b0 < 0xC2 || avail < 2 || (d[1] & 0xC0) != 0x80) return -1;
    *cp = ((b0 & 0x1Fu) << 6) | (d[1] & 0x3Fu);
    return 2;
  }
  if ((b0 & 0xF0) == 0xE0) {
    if (avail < 3 || (d[1] & 0xC0) != 0x80 || (d[2] & 0xC0) != 0x80) return -1;
    const uint32_t c = ((b0 & 0x0Fu) << 12) | ((d[1] & 0x3Fu) << 6) | (d[2] & 0x3Fu);
    if (c < 0x800 || (c >= 0xD800 && c <= 0xDFFF)) return -1;
    *cp = c;
    return 3;
  }
  if ((b0 & 0xF8) == 0xF0) {
    if (b0 > 0xF4 || avail < 4 || (d[1] & 0xC0) != 0x80 || (d[2] & 0xC0) != 0x80 || (d[3] & 0xC0) != 0x80) return -1;
    const uint32_t c = ((b0 & 0x07u) << 18) | ((d[1] & 0x3Fu) << 12) | ((d[2] & 0x3Fu) << 6) | (d[3] & 0x3Fu);
    if (c < 0x10000 || c > 0x10FFFF) return -1;
    *cp = c;
    return 4;
  }
  return -1;
}

// src/utils/string_utils.cpp:241-272 (CodepointsToUtf8, one code point)
inline size_t EncodeUtf8(uint32_t c, uint8_t* o) {
  if (c <= 0x7F) {
    o[0] = static_cast<uint8_t>(c);
    return 1;
  }
  if (c <= 0x7FF) {
    o[0] = static_cast<uint8_t>(0xC0 | (c >> 6));
    o[1] = static_cast<uint8_t>(0x80 | (c & 0x3F));
    return 2;
  }
  if (c <= 0xFFFF) {
    o[0] = static_cast<uint8_t>(0xE0 | (c >> 12));
    o[1] = static_cast<uint8_t>(0x80 | ((c >> 6) & 0x3F));
    o[2] = static_cast<uint8_t>(0x80 | (c & 0x3F));
    return 3;
  }
  o[0] = static_cast<uint8_t>(0xF0 | (c >> 18));
  o[1] = static_cast<uint8_t>(0x80 | ((c >> 12) & 0x3F));
  o[2] = static_cast<uint8_t>(0x80 | ((c >> 6) & 0x3F));
  o[3] = static_cast<uint8_t>(0x80 | (c & 0x3F));
  return 4;
}

// src/utils/string_utils.cpp:441-448 — kana is NOT an ideograph here
inline bool IsCjkIdeograph(uint32_t c) {
  return (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0x3400 && c <= 0x4DBF) || (c >= 0x20000 && c <= 0x2A6DF) ||
         (c >= 0x2A700 && c <= 0x2B73F) || (c >= 0x2B740 && c <= 0x2B81F) || (c >= 0xF900 && c <= 0xFAFF);
}
// src/server/search_pipeline.cpp:70-78 (knows one more extension block than the string_utils predicate)
inline bool IsCjkIdeographPipeline(uint32_t c) { return IsCjkIdeograph(c) || (c >= 0x2B820 && c <= 0x2CEAF); }

// Utf8ToCodepoints (:199-218): invalid bytes are skipped. `span` (optional): byte [begin, end) of every code point.
inline void Decode(const uint8_t* text, size_t len, std::vector<uint32_t>* cps,
                   std::vector<std::pair<uint32_t, uint32_t>>* span = nullptr) {
  cps->clear();
  if (span) span->clear();
  size_t i = 0;
  while (i < len) {
    uint32_t c = 0;
    const int k = ParseUtf8(text + i, len - i, &c);
    if (k > 0) {
      cps->push_back(c);
      if (span) span->emplace_back(static_cast<uint32_t>(i), static_cast<uint32_t>(i + k));
      i += static_cast<size_t>(k);
    } else {
      ++i;
    }
  }
}

// The window that STARTS at code point p under hybrid n-grams (GenerateHybridNgrams :452-509; plain n-grams are the
// case ascii_n == kanji_n): its size in code points, or 0 when no window starts there (too close to the end, or it
// would mix ideographs with other code points while cross_boundary is off).
inline int WindowAt(const std::vector<uint32_t>& cps, size_t p, int ascii_n, int kanji_n, bool cross) {
  const bool cjk = IsCjkIdeograph(cps[p]);
  const int w = cjk ? kanji_n : ascii_n;
  if (w <= 0 || p + static_cast<size_t>(w) > cps.size()) return 0;
  if (!cross)
    for (int j = 1; j < w; ++j)
      if (IsCjkIdeograph(cps[p + j]) != cjk) return 0;
  return w;
}

}  // namespace mgx::text
