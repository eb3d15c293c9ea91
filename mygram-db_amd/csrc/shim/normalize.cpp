// normalize.cpp — mygram::utils::NormalizeText for the host layer (reference: src/utils/string_utils.cpp:295-380).
// Two branches, chosen at build time exactly as the reference's USE_ICU does:
//   MGX_USE_ICU   NFKC (unorm2) -> width transliteration (utrans "Fullwidth-Halfwidth" / "Halfwidth-Fullwidth") ->
//                 full lower-casing (u_strToLower, default locale), on UTF-16 through ICU's C API;
//   otherwise     ASCII lower-casing only (the reference's own fallback, :371-377).
// Either way invalid UTF-8 fails closed: "" and one more on the failure counter (:363-366).
#include "mygram_shim.hpp"

#include "../mgx_text.hpp"

#include <atomic>
#include <cstring>
#include <vector>

#ifdef MGX_USE_ICU
#include <unicode/uloc.h>
#include <unicode/unorm2.h>
#include <unicode/ustring.h>
#include <unicode/utrans.h>
#endif

namespace mygram::utils {

namespace {

std::atomic<uint64_t> g_failures{0};

bool ValidUtf8(std::string_view text) {  // IsValidUtf8, string_utils.cpp:543-555
  const auto* d = reinterpret_cast<const uint8_t*>(text.data());
  size_t i = 0;
  while (i < text.size()) {
    uint32_t cp = 0;
    const int n = mgx::text::ParseUtf8(d + i, text.size() - i, &cp);
    if (n < 0) return false;
    i += static_cast<size_t>(n);
  }
  return true;
}

#ifdef MGX_USE_ICU
// One transliterator per thread and direction (they are not thread-safe to share); closed at thread exit.
struct Translit {
  UTransliterator* t = nullptr;
  UErrorCode status = U_ZERO_ERROR;
  explicit Translit(const char16_t* id) { t = utrans_openU(id, -1, UTRANS_FORWARD, nullptr, 0, nullptr, &status); }
  ~Translit() {
    if (t != nullptr) utrans_close(t);
  }
  Translit(const Translit&) = delete;
  Translit& operator=(const Translit&) = delete;
};

bool IcuNormalize(std::string_view text, bool nfkc, std::string_view width, bool lower, std::string* out) {
  UErrorCode st = U_ZERO_ERROR;
  // UTF-8 -> UTF-16 (valid input: at most one unit per byte)
  std::vector<UChar> a(text.size() + 1), b;
  int32_t n = 0;
  u_strFromUTF8(a.data(), static_cast<int32_t>(a.size()), &n, text.data(), static_cast<int32_t>(text.size()), &st);
  if (U_FAILURE(st)) return false;

  if (nfkc) {
    st = U_ZERO_ERROR;
    const UNormalizer2* nz = unorm2_getNFKCInstance(&st);
    if (U_FAILURE(st) || nz == nullptr) return false;
    b.resize(static_cast<size_t>(n) * 3 + 16);
    st = U_ZERO_ERROR;
    int32_t m = unorm2_normalize(nz, a.data(), n, b.data(), static_cast<int32_t>(b.size()), &st);
    if (st == U_BUFFER_OVERFLOW_ERROR) {  // U+FDFA expands 18x; rare, so sized on demand
      b.resize(static_cast<size_t>(m) + 1);
      st = U_ZERO_ERROR;
      m = unorm2_normalize(nz, a.data(), n, b.data(), static_cast<int32_t>(b.size()), &st);
    }
    if (U_FAILURE(st)) return false;
    a.swap(b);
    n = m;
  }

  if (width == "narrow" || width == "wide") {
    thread_local Translit narrow(u"Fullwidth-Halfwidth");
    thread_local Translit wide(u"Halfwidth-Fullwidth");
    Translit& tr = width == "narrow" ? narrow : wide;
    if (U_FAILURE(tr.status) || tr.t == nullptr) return false;
    // in place; either direction can lengthen the text (a voiced half-width kana is two units for one)
    a.resize(static_cast<size_t>(n) * 2 + 16);
    int32_t len = n, limit = n;
    st = U_ZERO_ERROR;
    utrans_transUChars(tr.t, a.data(), &len, static_cast<int32_t>(a.size()), 0, &limit, &st);
    if (U_FAILURE(st)) return false;
    n = len;
  }

  if (lower) {
    b.resize(static_cast<size_t>(n) * 3 + 16);
    st = U_ZERO_ERROR;
    int32_t m = u_strToLower(b.data(), static_cast<int32_t>(b.size()), a.data(), n, nullptr, &st);
    if (st == U_BUFFER_OVERFLOW_ERROR) {
      b.resize(static_cast<size_t>(m) + 1);
      st = U_ZERO_ERROR;
      m = u_strToLower(b.data(), static_cast<int32_t>(b.size()), a.data(), n, nullptr, &st);
    }
    if (U_FAILURE(st)) return false;
    a.swap(b);
    n = m;
  }

  out->resize(static_cast<size_t>(n) * 3 + 1);
  int32_t bytes = 0;
  st = U_ZERO_ERROR;
  u_strToUTF8(out->data(), static_cast<int32_t>(out->size()), &bytes, a.data(), n, &st);
  if (U_FAILURE(st)) return false;
  out->resize(static_cast<size_t>(bytes));
  return true;
}
#endif

}  // namespace

bool NormalizeTextUsesIcu() {
#ifdef MGX_USE_ICU
  return true;
#else
  return false;
#endif
}

uint64_t GetTextNormalizationFailureCount() { return g_failures.load(std::memory_order_relaxed); }
void ResetTextNormalizationFailureCountForTesting() { g_failures.store(0, std::memory_order_relaxed); }

std::string NormalizeText(std::string_view text, bool nfkc, std::string_view width, bool lower) {
  if (!ValidUtf8(text)) {
    g_failures.fetch_add(1, std::memory_order_relaxed);
    return {};
  }
#ifdef MGX_USE_ICU
  // ASCII needs no tables: NFKC and the narrowing transliterator leave it alone and lower-casing it is a byte
  // operation — in every locale but the three whose 'I' does not lower to 'i' (the default locale decides, as it does
  // for UnicodeString::toLower()). Query terms are mostly ASCII; this keeps ICU off the planner's hot path.
  static const bool ascii_lower_is_plain = [] {
    const char* lang = uloc_getDefault();
    return !(lang && (std::strncmp(lang, "tr", 2) == 0 || std::strncmp(lang, "az", 2) == 0 || std::strncmp(lang, "lt", 2) == 0));
  }();
  if (width != "wide" && ascii_lower_is_plain) {
    bool ascii = true;
    for (const char c : text) ascii = ascii && static_cast<unsigned char>(c) < 0x80;
    if (ascii) {
      std::string out(text);
      if (lower)
        for (char& c : out)
          if (c >= 'A' && c <= 'Z') c = static_cast<char>(c + 32);
      return out;
    }
  }
  std::string out;
  if (!IcuNormalize(text, nfkc, width, lower, &out)) {
    g_failures.fetch_add(1, std::memory_order_relaxed);
    return {};
  }
  return out;
#else
  (void)nfkc;
  (void)width;
  std::string out(text);
  if (lower)
    for (char& c : out)
      if (c >= 'A' && c <= 'Z') c = static_cast<char>(c + 32);
  return out;
#endif
}

}  // namespace mygram::utils
