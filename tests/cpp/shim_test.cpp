// shim_test.cpp — the reference's own known-answer tests, written against the shim classes exactly the way the
// reference's gtest files call the originals (expected values transcribed as data from /root/reference/tests; the
// file:line of every case is given). Needs a GPU; built and run by tests/test_gpu_shim.py.
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/mygram_tools.h"
#include "../../mygram-db_amd/csrc/shim/mygram_shim.hpp"

using mygramdb::index::BM25Params;
using mygramdb::index::BM25Scorer;
using mygramdb::index::DocId;
using mygramdb::index::Index;
using mygramdb::query::ResultSorter;
using mygramdb::query::SortOrder;

static int g_failed = 0, g_checked = 0;
#define EXPECT(cond)                                                      \
  do {                                                                    \
    ++g_checked;                                                          \
    if (!(cond)) {                                                        \
      ++g_failed;                                                         \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);         \
    }                                                                     \
  } while (0)
using V = std::vector<DocId>;

static V Range(DocId a, DocId b, int step) {
  V v;
  for (long x = a; step > 0 ? x <= b : x >= static_cast<long>(b); x += step) v.push_back(static_cast<DocId>(x));
  return v;
}

int main() {
  {  // tests/utils/string_utils_test.cpp:136-283 through Index::NormalizeText (index.h:321-323)
    using mygram::utils::NormalizeText;
    EXPECT(NormalizeText("ABC", false, "keep", true) == "abc");
    EXPECT(NormalizeText("ABC", false, "keep", false) == "ABC");
    mygram::utils::ResetTextNormalizationFailureCountForTesting();
    EXPECT(NormalizeText(std::string("abc") + static_cast<char>(0xC0) + static_cast<char>(0xAF), true, "keep", true).empty());
    EXPECT(mygram::utils::GetTextNormalizationFailureCount() == 1);
    if (mygram::utils::NormalizeTextUsesIcu()) {
      Index idx(2, 1, 0.0, true, true, "narrow", true);
      EXPECT(idx.NormalizeText("ＡＢＣ") == "abc");            // :217
      Index keep(2, 1);                                         // defaults: nfkc, "keep", lower
      EXPECT(keep.NormalizeText("ﾗｲﾌﾞ") == "ライブ");         // :246
      EXPECT(keep.NormalizeText("ｱｲｳＡＢＣ") == "アイウabc");  // :229
      EXPECT(idx.NormalizeText("　！？") == " !?");            // :282
      Index wide(2, 1, 0.0, true, false, "wide", false);
      EXPECT(wide.NormalizeText("ABC") == "ＡＢＣ");           // :200
      std::printf("NormalizeText: ICU branch\n");
    } else {
      std::printf("NormalizeText: ASCII fallback (built without ICU)\n");
    }
  }
  {  // tests/index/index_search_test.cpp:22-59
    Index index(1);
    index.AddDocument(1, "abc");
    index.AddDocument(2, "bcd");
    index.AddDocument(3, "cde");
    EXPECT(index.SearchAnd({"b"}) == (V{1, 2}));
    EXPECT(index.SearchAnd({"b", "c"}) == (V{1, 2}));
    EXPECT(index.SearchAnd({"c", "d"}) == (V{2, 3}));
    EXPECT(index.SearchAnd({}).empty());   // :423-437
    EXPECT(index.SearchOr({}).empty());
    EXPECT(index.SearchAnd({"a", "x"}, 10, true).empty());  // :568-585 unknown term
    if (!index.LastError().empty()) std::printf("index error: %s\n", index.LastError().c_str());
  }
  {  // :460-489
    Index index(2);
    index.AddDocument(100, "hello");
    index.AddDocument(200, "help");
    index.AddDocument(300, "yellow");
    index.AddDocument(400, "hello world");
    index.AddDocument(500, "shell");
    EXPECT(index.SearchAnd({"he", "el"}, 0, false) == (V{100, 200, 400, 500}));
    EXPECT(index.SearchAnd({"he", "he", "el"}) == (V{100, 200, 400, 500}));  // duplicate grams harmless (:439-446)
  }
  {  // :587-620 Roaring chain + top-10 reverse; :622-660 limit/reverse matrix
    Index index(1);
    for (DocId i = 1; i <= 15000; ++i) index.AddDocument(i, "ab");
    for (DocId i = 15001; i <= 15100; ++i) index.AddDocument(i, "a");
    EXPECT(index.SearchAnd({"a", "b"}, 10, true) == Range(15000, 14991, -1));
    EXPECT(index.SearchAnd({"a", "b"}, 10, false) == Range(1, 10, 1));
    EXPECT(index.SearchAnd({"a"}, 5, true) == Range(15100, 15096, -1));
    EXPECT(index.SearchAnd({"a", "b"}).size() == 15000);
  }
  {  // OR / NOT: :101-120, :202-222, :227-258
    Index index(1);
    index.AddDocument(1, "abc");
    index.AddDocument(2, "def");
    index.AddDocument(3, "ghi");
    index.AddDocument(4, "jkl");
    EXPECT(index.SearchOr({"a", "d"}) == (V{1, 2}));
    EXPECT(index.SearchOr({"a", "d", "g"}) == (V{1, 2, 3}));
    EXPECT(index.SearchOr({"z"}).empty());
    EXPECT(index.SearchOr({"a", "z"}) == (V{1}));
    EXPECT(index.SearchNot({1, 2, 3, 4}, {"a", "d"}) == (V{3, 4}));
    EXPECT(index.SearchNot({1, 2, 3, 4}, {"a", "d", "g"}) == (V{4}));
    EXPECT(index.SearchNot({1, 2, 3, 4}, {"z"}) == (V{1, 2, 3, 4}));
    EXPECT(index.SearchNot({1, 2, 3, 4}, {}) == (V{1, 2, 3, 4}));
  }
  {  // tests/index/search_by_threshold_test.cpp:41-167
    Index index(2);
    index.AddDocument(1, "hello");
    index.AddDocument(2, "help");
    index.AddDocument(3, "world");
    EXPECT(index.SearchByThreshold({"he", "el", "ll", "lo"}, 4) == (V{1}));
    EXPECT(index.SearchByThreshold({"he", "el", "ll", "lo"}, 2) == (V{1, 2}));
    EXPECT(index.SearchByThreshold({"he", "wo"}, 1) == (V{1, 2, 3}));
    EXPECT(index.SearchByThreshold({"he", "el", "zz"}, 2) == (V{1, 2}));
    EXPECT(index.SearchByThreshold({}, 1).empty());
    EXPECT(index.SearchByThreshold({"he"}, 0).empty());
    EXPECT(index.SearchByThreshold({"zz", "yy", "xx"}, 1).empty());
    EXPECT(index.SearchByThreshold({"he", "el", "lp"}, 3) == (V{2}));
    EXPECT(index.SearchByThreshold({"he", "he", "wo"}, 2).empty());
    EXPECT(index.SearchByThreshold({"he", "he"}, 2).empty());
  }
  {  // tests/index/index_search_test.cpp:662-701 FilterByNgrams
    Index index(1, 0);
    for (DocId d = 1; d <= 6000; ++d) index.AddDocument(d, d % 3 == 0 ? "ab" : "a");
    V cand, want;
    for (DocId d = 1; d <= 6000; d += 7) {
      cand.push_back(d);
      if (d % 3 == 0) want.push_back(d);
    }
    EXPECT(index.FilterByNgrams(cand, {"a", "b"}) == want);
    EXPECT(index.FilterByNgrams(cand, {"a", "z"}).empty());
    EXPECT(index.FilterByNgrams(cand, {}) == cand);
    EXPECT(index.FilterByNgrams({}, {"a"}).empty());
    EXPECT(index.FilterByNgrams({3, 3, 6000, 6001}, {"b"}) == (V{3, 3, 6000}));
  }
  {  // tests/index/index_basic_test.cpp:103-121 RemoveDocument, :126-144 UpdateDocument (after the index was built: the
     // live row + delta index of a mutable table)
    Index index(1);
    index.AddDocument(1, "abc");
    index.AddDocument(2, "bcd");
    EXPECT(index.Count("a") == 1 && index.Count("b") == 2 && index.Count("c") == 2);
    index.RemoveDocument(1, "abc");
    EXPECT(index.Count("a") == 0 && index.Count("b") == 1 && index.Count("c") == 1 && index.Count("d") == 1);
    EXPECT(index.SearchAnd({"b"}) == (V{2}));
    EXPECT(index.SearchOr({"a", "d"}) == (V{2}));
    Index upd(1);
    upd.AddDocument(1, "abc");
    EXPECT(upd.Count("a") == 1 && upd.Count("d") == 0);
    upd.UpdateDocument(1, "abc", "bcd");
    EXPECT(upd.Count("a") == 0 && upd.Count("b") == 1 && upd.Count("c") == 1 && upd.Count("d") == 1);
    EXPECT(upd.SearchAnd({"d"}) == (V{1}) && upd.SearchAnd({"a"}).empty());
    // :256-339 updates that empty posting lists; repeated updates of one document
    upd.UpdateDocument(1, "bcd", "xyz");
    EXPECT(upd.Count("x") == 1 && upd.Count("y") == 1 && upd.Count("z") == 1);
    EXPECT(upd.Count("b") == 0 && upd.Count("c") == 0 && upd.Count("d") == 0);
    Index rep(1);
    rep.AddDocument(1, "a");
    EXPECT(rep.Count("a") == 1);
    rep.UpdateDocument(1, "a", "b");
    rep.UpdateDocument(1, "b", "c");
    rep.UpdateDocument(1, "c", "d");
    rep.UpdateDocument(1, "d", "e");
    EXPECT(rep.Count("e") == 1 && rep.Count("a") == 0 && rep.Count("b") == 0 && rep.Count("c") == 0 && rep.Count("d") == 0);
    EXPECT(rep.SearchAnd({"e"}) == (V{1}));
  }
  {  // a table with gaps in its id range that changes: the NOT universe is the ids that exist NOW
     // (DocumentStore::GetAllDocIds), whichever index holds them
    using namespace mygramdb::search_pipeline;
    using mygramdb::query::NodeType;
    using mygramdb::query::QueryNode;
    Index index(2, 0, 0.0, false);
    index.AddDocument(1, "machine learning");
    index.AddDocument(2, "deep learning");
    index.AddDocument(5, "old cats");  // ids 3, 4 do not exist
    auto not_of = [](const char* t) {
      auto n = std::make_unique<QueryNode>(NodeType::NOT);
      n->children.push_back(std::make_unique<QueryNode>(std::string(t)));
      return n;
    };
    BatchQuery q;
    q.ast = not_of("machine");
    q.order = SortOrder::ASC;
    auto r0 = ExecuteBatch(index, {q});
    EXPECT(r0.has_value() && (*r0)[0].results == (V{2, 5}));
    index.RemoveDocument(2, "deep learning");
    EXPECT(index.AddDocument(4, "new cats"));          // an id inside the old range that never existed
    EXPECT(index.AddDocument(9, "machine cats"));      // and one beyond it
    EXPECT(!index.AddDocument(5, "again"));            // a live id is refused
    auto r1 = ExecuteBatch(index, {q});
    EXPECT(r1.has_value() && (*r1)[0].results == (V{4, 5}) && (*r1)[0].total == 2);
    BatchQuery c;
    c.terms = {"cats"};
    c.order = SortOrder::ASC;
    auto r2 = ExecuteBatch(index, {c});
    EXPECT(r2.has_value() && (*r2)[0].results == (V{4, 5, 9}));
    // a staleness bound: changes recorded inside the bound wait for it (or for a forced application)
    index.SetMutationStaleness(std::chrono::seconds(3600));
    index.RemoveDocument(9, "machine cats");
    auto rs = ExecuteBatch(index, {c});
    EXPECT(rs.has_value() && (*rs)[0].results == (V{4, 5, 9}));  // not yet
    EXPECT(index.ApplyMutations(/*force=*/true).empty());
    rs = ExecuteBatch(index, {c});
    EXPECT(rs.has_value() && (*rs)[0].results == (V{4, 5}));
    EXPECT(index.AddDocument(9, "machine cats"));
    index.SetMutationStaleness(std::chrono::microseconds(0));
    EXPECT(index.Compact().empty());
    auto r3 = ExecuteBatch(index, {q, c});
    EXPECT(r3.has_value() && (*r3)[0].results == (V{4, 5}) && (*r3)[1].results == (V{4, 5, 9}));
    EXPECT(index.SearchOr({"ca", "ma"}) == (V{1, 4, 5, 9}));
    EXPECT(index.GetMutationStats().delta_documents == 0);
  }
  {  // tests/index/index_basic_test.cpp:149-184 UpdateDocumentMaintainsTopNOrdering
    Index index(1);
    const DocId kBase = 512;
    for (DocId d = 1; d <= kBase; ++d) index.AddDocument(d, "aaaa");
    index.AddDocument(kBase + 1, "zzzz");
    EXPECT(index.SearchAnd({"a"}, 3, true) == (V{kBase, kBase - 1, kBase - 2}));
    EXPECT(index.Count("a") == kBase);
    index.UpdateDocument(kBase, "aaaa", "zzzz");
    EXPECT(index.SearchAnd({"a"}, 3, true) == (V{kBase - 1, kBase - 2, kBase - 3}));
    EXPECT(index.Count("a") == kBase - 1);
    index.UpdateDocument(kBase + 1, "zzzz", "aaaa");
    EXPECT(index.SearchAnd({"a"}, 3, true) == (V{kBase + 1, kBase - 1, kBase - 2}));
    EXPECT(index.Count("a") == kBase);
    // the other single operators over both indexes
    EXPECT(index.SearchOr({"z"}) == (V{kBase}));
    EXPECT(index.SearchNot({1, 2, kBase, kBase + 1}, {"z"}) == (V{1, 2, kBase + 1}));
    EXPECT(index.SearchByThreshold({"a", "z"}, 1).size() == kBase + 1);
    EXPECT(index.FilterByNgrams({kBase + 1, 5, kBase, 5, 9999}, {"a"}) == (V{kBase + 1, 5, 5}));
    auto sc = BM25Scorer::ScoreDocuments({kBase, kBase + 1, 7, 9999}, {"a"}, {index.Count("a")}, index, kBase + 1, 4.0, BM25Params{});
    EXPECT(sc.has_value() && sc->size() == 4);
    if (sc.has_value() && sc->size() == 4) {
      EXPECT((*sc)[0].score == 0.0);                       // "zzzz" now: no occurrence of the term
      EXPECT((*sc)[1].score > 0.0 && (*sc)[1].score == (*sc)[2].score);  // "aaaa" in the delta = "aaaa" in the main index
      EXPECT((*sc)[3].score == 0.0);                       // not a document
    }
    // and the batched pipeline: the page holds the delta's document under its table id
    mygramdb::search_pipeline::BatchQuery q;
    q.terms = {"a"};
    q.limit = 3;
    auto r = mygramdb::search_pipeline::ExecuteBatch(index, {q});
    EXPECT(r.has_value() && (*r)[0].total == kBase && (*r)[0].results == (V{kBase + 1, kBase - 1, kBase - 2}));
    if (r.has_value()) EXPECT((*r)[0].total_candidates == kBase && (*r)[0].after_filters == kBase);
    // a mixed batch (a docid page and a SORT _score query) on a table with a delta: each group is merged on its own
    mygramdb::search_pipeline::BatchQuery qs2 = q;
    qs2.sort_by_score = true;
    qs2.limit = 2;
    auto r2 = mygramdb::search_pipeline::ExecuteBatch(index, {q, qs2, q});
    EXPECT(r2.has_value());
    if (r2.has_value()) {
      EXPECT((*r2)[0].results == (V{kBase + 1, kBase - 1, kBase - 2}) && (*r2)[2].results == (*r2)[0].results);
      EXPECT((*r2)[1].total == kBase && (*r2)[1].results.size() == 2 && (*r2)[1].scores.size() == 2 &&
             (*r2)[1].scores[0] == (*r2)[1].scores[1] && (*r2)[1].scores[0] > 0.0);
    } else {
      std::printf("mixed batch on a mutable table: %s\n", r2.error().message().c_str());
    }
  }
  {  // tests/index/bm25_scorer_test.cpp:21-48
    EXPECT(std::fabs(BM25Scorer::ComputeIDF(100, 10) - std::log(90.5 / 10.5 + 1.0)) < 1e-10);
    EXPECT(BM25Scorer::ComputeIDF(0, 10) == 0.0);
    EXPECT(BM25Scorer::ComputeIDF(10, 20) == BM25Scorer::ComputeIDF(10, 10));
  }
  {  // tests/server/http_server_search_test.cpp:498-529 worked example; bm25_scorer_test.cpp:175-184 error case
    Index index(5);  // "alpha" is one 5-gram
    index.AddDocument(1, "alpha");
    index.AddDocument(2, "alpha alpha alpha");
    auto scored = BM25Scorer::ScoreDocuments({1, 2}, {"alpha"}, {2}, index, 2, 11.0, BM25Params{1.2, 0.75});
    EXPECT(scored.has_value());
    if (scored) {
      EXPECT(std::fabs((*scored)[0].score - 0.2346905145964735) < 1e-12);
      EXPECT(std::fabs((*scored)[1].score - 0.25652219037288959) < 1e-12);
      EXPECT((*scored)[0].doc_id == 1 && (*scored)[1].doc_id == 2);
    }
    auto bad = BM25Scorer::ScoreDocuments({1}, {"alpha", "world"}, {1}, index, 1, 1.0, {});
    EXPECT(!bad.has_value() && bad.error().code() == mygram::utils::ErrorCode::kInvalidArgument);
    EXPECT(bad.error().message().find("identical lengths") != std::string::npos);
    // tests/query/bm25_sort_test.cpp:15-75
    EXPECT(ResultSorter::SortByScore(index, {1, 2, 3, 4}, {1.0, 3.0, 2.0, 4.0}, SortOrder::DESC, 0, 0) == (V{4, 2, 3, 1}));
    EXPECT(ResultSorter::SortByScore(index, {1, 2, 3}, {3.0, 1.0, 2.0}, SortOrder::ASC, 0, 0) == (V{2, 3, 1}));
    EXPECT(ResultSorter::SortByScore(index, {1, 2, 3, 4, 5}, {1, 5, 3, 2, 4}, SortOrder::DESC, 3, 0) == (V{2, 5, 3}));
    EXPECT(ResultSorter::SortByScore(index, {1, 2, 3, 4}, {1, 4, 3, 2}, SortOrder::DESC, 2, 1) == (V{3, 4}));
    EXPECT(ResultSorter::SortByScore(index, {3, 1, 2}, {1, 1, 1}, SortOrder::ASC, 0, 0) == (V{1, 2, 3}));
    EXPECT(ResultSorter::SortByScore(index, {3, 1, 2}, {1, 1, 1}, SortOrder::DESC, 0, 0) == (V{3, 2, 1}));
    EXPECT(ResultSorter::SortByScore(index, {}, {}, SortOrder::DESC, 0, 0).empty());
  }
  {  // tests/server/search_pipeline_test.cpp:916-995 fixture through the batched entry
    using namespace mygramdb::search_pipeline;
    Index index(2, 0, 0.0, false);
    index.AddDocument(1, "machine learning basics");
    index.AddDocument(2, "deep learning techniques");
    index.AddDocument(3, "old article about cats");
    auto status1 = index.AddFilterBitmap({1, 2});
    EXPECT(status1.has_value());
    std::vector<BatchQuery> qs(3);
    qs[0].terms = {"learning"};
    qs[1].terms = {"learning", "machine"};
    qs[1].not_terms = {"basics"};
    qs[1].filters = {{status1 ? *status1 : 0u, false}};
    qs[2].terms = {"LEARNING", "zzzz"};  // unknown gram: empty before the device
    for (auto& q : qs) q.order = SortOrder::ASC;
    auto r = ExecuteBatch(index, qs);
    EXPECT(r.has_value());
    if (r) {
      EXPECT((*r)[0].results == (V{1, 2}) && (*r)[0].total == 2);
      EXPECT((*r)[1].results.empty());
      EXPECT((*r)[1].total_candidates == 1 && (*r)[1].after_intersection == 1 && (*r)[1].after_not == 0 &&
             (*r)[1].after_filters == 0);
      EXPECT((*r)[2].empty_term_detected && (*r)[2].results.empty());
    } else {
      std::printf("ExecuteBatch error: %s\n", r.error().message().c_str());
    }
  }
  {  // SearchNot over an all_docs with repeats: std::set_difference semantics (index.cpp:480-484)
    Index index(1);
    index.AddDocument(1, "ab");
    index.AddDocument(2, "b");
    index.AddDocument(3, "a");
    EXPECT(index.SearchNot({1, 1, 2, 2, 3}, {"a"}) == (V{1, 2, 2}));  // one copy of doc 1 goes, doc 3 goes, doc 2 stays twice
    EXPECT(mygramdb::index::LastDeviceError().empty());
  }
  {  // terms shorter than one n-gram: SearchTermDocuments' substring branch (search_pipeline.cpp:438-446,
     // src/query/substring_search.h:24-42) — positive, later, and NOT terms
    using namespace mygramdb::search_pipeline;
    Index index(2, 0, 0.0, false);
    index.AddDocument(1, "a cat");
    index.AddDocument(2, "x ray");
    index.AddDocument(3, "taxi x");
    index.AddDocument(4, "dog");
    std::vector<BatchQuery> qs(4);
    qs[0].terms = {"x"};                 // the only term: every doc whose text contains "x"
    qs[1].terms = {"ta", "x"};           // a later term filters the candidates
    qs[2].terms = {"a"};
    qs[2].not_terms = {"x"};             // a NOT term excludes by substring
    qs[3].terms = {"Q"};                 // normalised to "q": contained in nothing
    for (auto& q : qs) q.order = SortOrder::ASC;
    auto r = ExecuteBatch(index, qs);
    EXPECT(r.has_value());
    if (r) {
      EXPECT((*r)[0].results == (V{2, 3}) && (*r)[0].total_candidates == 2);
      EXPECT((*r)[1].results == (V{3}) && (*r)[1].total_candidates == 1 && (*r)[1].after_intersection == 1);
      EXPECT((*r)[2].results == (V{1}) && (*r)[2].after_intersection == 3 && (*r)[2].after_not == 1);
      EXPECT((*r)[3].results.empty() && (*r)[3].total == 0);
    } else {
      std::printf("ExecuteBatch error: %s\n", r.error().message().c_str());
    }
  }
  {  // the same worked example on the default bigram index: "alpha" spans four n-grams, so tf comes from the text
     // (CountTermOccurrences, bm25_scorer.cpp:27-45) — identical scores
    Index index(2);
    index.AddDocument(1, "alpha");
    index.AddDocument(2, "alpha alpha alpha");
    index.AddDocument(3, "alp pha lph ha");  // every bigram of "alpha", not the term: tf 0 -> score 0
    auto scored = BM25Scorer::ScoreDocuments({1, 2, 3, 77}, {"alpha"}, {2}, index, 2, 11.0, BM25Params{1.2, 0.75});
    EXPECT(scored.has_value());
    if (scored) {
      EXPECT(std::fabs((*scored)[0].score - 0.2346905145964735) < 1e-12);
      EXPECT(std::fabs((*scored)[1].score - 0.25652219037288959) < 1e-12);
      EXPECT((*scored)[2].score == 0.0 && (*scored)[3].score == 0.0);
    } else {
      std::printf("ScoreDocuments error: %s\n", scored.error().message().c_str());
    }
  }
  {  // SORT _score through the batched entry with a multi-gram term: df by text scan of the term's candidates
     // (search_pipeline.cpp:542-565), tf by CountTermOccurrences, ranking by SortByScore
    using namespace mygramdb::search_pipeline;
    Index index(2, 0, 0.0, false);
    index.AddDocument(1, "machine learning basics");
    index.AddDocument(2, "deep learning learning techniques");
    index.AddDocument(3, "old article about cats");
    index.AddDocument(4, "lea ear arn rni nin ing");  // n-gram candidate without the term
    std::vector<BatchQuery> qs(1);
    qs[0].terms = {"learning"};
    qs[0].sort_by_score = true;
    qs[0].limit = 10;
    qs[0].order = SortOrder::DESC;
    auto r = ExecuteBatch(index, qs);
    EXPECT(r.has_value());
    if (r) {
      const double n = 4.0, df = 2.0, avgdl = (23.0 + 33.0 + 22.0 + 23.0) / 4.0;
      const double idf = std::log((n - df + 0.5) / (df + 0.5) + 1.0);
      auto bm = [&](double tf, double dl) { return idf * (tf * (1.2 + 1.0)) / (tf + 1.2 * (1.0 - 0.75 + 0.75 * dl / avgdl)); };
      EXPECT((*r)[0].total == 3);  // verify_text off: doc 4 stays, with score 0
      EXPECT((*r)[0].results == (V{2, 1, 4}));
      EXPECT((*r)[0].scores.size() == 3 && (*r)[0].scores[0] == bm(2.0, 33.0) && (*r)[0].scores[1] == bm(1.0, 23.0) &&
             (*r)[0].scores[2] == 0.0);
    } else {
      std::printf("ExecuteBatch error: %s\n", r.error().message().c_str());
    }
  }
  {  // boolean expressions through the batched entry (EvaluateBooleanAstExpanded, search_pipeline.cpp:327-378):
     // NOT = DocumentStore::GetAllDocIds minus the child, so ids that were never added must not appear
    using namespace mygramdb::search_pipeline;
    using mygramdb::query::NodeType;
    using mygramdb::query::QueryNode;
    Index index(2, 0, 0.0, false);
    index.AddDocument(1, "machine learning basics");
    index.AddDocument(2, "deep learning techniques");
    index.AddDocument(3, "old article about cats");
    index.AddDocument(7, "learning cats");  // ids 4..6 do not exist
    auto term = [](const char* t) { return std::make_unique<QueryNode>(std::string(t)); };
    auto node = [](NodeType t, std::vector<std::unique_ptr<QueryNode>> kids) {
      auto n = std::make_unique<QueryNode>(t);
      n->children = std::move(kids);
      return n;
    };
    auto kids = [](std::unique_ptr<QueryNode> a, std::unique_ptr<QueryNode> b = nullptr) {
      std::vector<std::unique_ptr<QueryNode>> v;
      v.push_back(std::move(a));
      if (b) v.push_back(std::move(b));
      return v;
    };
    std::vector<BatchQuery> qs(4);
    // (machine OR deep) AND NOT techniques
    qs[0].ast = node(NodeType::AND, kids(node(NodeType::OR, kids(term("machine"), term("deep"))),
                                         node(NodeType::NOT, kids(term("techniques")))));
    qs[1].ast = node(NodeType::NOT, kids(term("machine")));                       // every existing doc but 1
    qs[2].ast = node(NodeType::OR, kids(term("cats"), term("zzzzzz")));           // unknown term = empty set in the tree
    qs[3].ast = node(NodeType::AND, kids(term("learning"), node(NodeType::NOT, kids(term("cats")))));
    qs[3].not_terms = {"deep"};
    for (auto& q : qs) q.order = SortOrder::ASC;
    auto r = ExecuteBatch(index, qs);
    EXPECT(r.has_value());
    if (r) {
      EXPECT((*r)[0].results == (V{1}));
      EXPECT((*r)[1].results == (V{2, 3, 7}) && (*r)[1].total == 3);
      EXPECT((*r)[2].results == (V{3, 7}));
      EXPECT((*r)[3].results == (V{1}));
    } else {
      std::printf("ExecuteBatch(ast) error: %s\n", r.error().message().c_str());
    }
    // a leaf shorter than one n-gram is the documents whose text contains it (regular_term_search ->
    // SearchTermDocuments -> SearchNormalizedSubstring, search_pipeline.cpp:1446-1455, :438-446), inside the tree too
    Index sub(2, 0, 0.0, false);
    sub.AddDocument(1, "hello world");
    sub.AddDocument(2, "help a");
    sub.AddDocument(3, "yellow");
    sub.AddDocument(4, "a hello");
    std::vector<BatchQuery> ss(4);
    ss[0].ast = node(NodeType::AND, kids(term("hello"), term("a")));   // "a": docs 2, 4
    ss[1].ast = node(NodeType::OR, kids(term("a"), term("yellow")));
    ss[2].ast = node(NodeType::NOT, kids(term("a")));
    ss[3].ast = node(NodeType::AND, kids(term("hello"), term("a")));
    ss[3].sort_by_score = true;
    for (auto& q : ss) q.order = SortOrder::ASC;
    ss[3].order = SortOrder::DESC;
    auto rs = ExecuteBatch(sub, ss);
    EXPECT(rs.has_value());
    if (rs) {
      EXPECT((*rs)[0].results == (V{4}));
      EXPECT((*rs)[1].results == (V{2, 3, 4}));
      EXPECT((*rs)[2].results == (V{1, 3}));
      EXPECT((*rs)[3].results == (V{4}) && (*rs)[3].scores.size() == 1 && (*rs)[3].scores[0] > 0.0);
    } else {
      std::printf("ExecuteBatch(ast, substring leaf) error: %s\n", rs.error().message().c_str());
    }
  }
  {  // SORT _score over the boolean and FUZZY branches: the batch path must return what the reference's composition
     // returns — the branch's result set, BM25Scorer::ScoreDocuments over the POSITIVE terms (search_handler.cpp:428-456;
     // NOT-excluded terms are not scored: tests/server/search_pipeline_test.cpp:1350-1392), ResultSorter::SortByScore
    using namespace mygramdb::search_pipeline;
    using mygramdb::query::NodeType;
    using mygramdb::query::QueryNode;
    Index index(2, 0, 0.0, false);
    index.AddDocument(1, "machine learning basics");
    index.AddDocument(2, "deep learning techniques");
    index.AddDocument(3, "old article about cats");
    index.AddDocument(4, "learning learning learning");
    index.AddDocument(5, "cats and machine cats");
    index.AddDocument(6, "basics of basics");
    auto term = [](const char* t) { return std::make_unique<QueryNode>(std::string(t)); };
    auto node = [](NodeType t, std::vector<std::unique_ptr<QueryNode>> kids) {
      auto n = std::make_unique<QueryNode>(t);
      n->children = std::move(kids);
      return n;
    };
    auto kids = [](std::unique_ptr<QueryNode> a, std::unique_ptr<QueryNode> b = nullptr) {
      std::vector<std::unique_ptr<QueryNode>> v;
      v.push_back(std::move(a));
      if (b) v.push_back(std::move(b));
      return v;
    };
    // single-gram leaves: their df is the posting count, so the composition needs nothing but the shim's own operators
    struct Case {
      std::shared_ptr<const QueryNode> ast;
      std::vector<std::string> scored;  // CollectAstScoringTerms
    };
    std::vector<Case> cases;
    cases.push_back({node(NodeType::OR, kids(term("ba"), term("ca"))), {"ba", "ca"}});
    cases.push_back({node(NodeType::AND, kids(node(NodeType::OR, kids(term("ba"), term("ca"))), term("ma"))), {"ba", "ca", "ma"}});
    cases.push_back({node(NodeType::AND, kids(term("le"), node(NodeType::NOT, kids(term("de"))))), {"le"}});
    const uint64_t n_docs = index.Bm25DocCount();
    const double avgdl = index.Bm25AvgDocLength();
    for (const Case& c : cases) {
      BatchQuery plain, scored;
      plain.ast = scored.ast = c.ast;
      plain.order = SortOrder::ASC;
      plain.limit = 0;
      scored.sort_by_score = true;
      scored.limit = 4;
      auto a = ExecuteBatch(index, {plain});
      auto b = ExecuteBatch(index, {scored});
      EXPECT(a.has_value() && b.has_value());
      if (!a || !b) {
        std::printf("scored expression: %s\n", (!a ? a.error() : b.error()).message().c_str());
        continue;
      }
      std::vector<uint64_t> dfs;
      for (const auto& t : c.scored) dfs.push_back(index.PostingSize(t));
      auto sc = BM25Scorer::ScoreDocuments((*a)[0].results, c.scored, dfs, index, n_docs, avgdl, BM25Params{});
      EXPECT(sc.has_value());
      if (!sc) continue;
      std::vector<double> scores;
      for (const auto& sd : *sc) scores.push_back(sd.score);
      const V want = ResultSorter::SortByScore(index, (*a)[0].results, scores, SortOrder::DESC, 4, 0);
      EXPECT((*b)[0].results == want);
      EXPECT((*b)[0].total == (*a)[0].results.size());
      for (size_t i = 0; i < want.size() && i < (*b)[0].scores.size(); ++i) {
        double ws = 0;
        for (size_t k = 0; k < (*a)[0].results.size(); ++k)
          if ((*a)[0].results[k] == want[i]) ws = scores[k];
        EXPECT((*b)[0].scores[i] == ws);
      }
    }
    // FUZZY 1 SORT _score: "learnig" matches docs 1, 2 (and 4: le ea ar rn ni of its six bigrams); the EXACT term occurs
    // in no text, so every score is 0 and the docid decides (DESC: larger first)
    BatchQuery fz;
    fz.terms = {"learnig"};
    fz.fuzzy_max_distance = 1;
    fz.sort_by_score = true;
    fz.limit = 10;
    auto f = ExecuteBatch(index, {fz});
    EXPECT(f.has_value());
    if (f) {
      EXPECT((*f)[0].results == (V{4, 2, 1}));
      EXPECT((*f)[0].scores.size() == 3 && (*f)[0].scores[0] == 0.0);
    } else {
      std::printf("scored FUZZY: %s\n", f.error().message().c_str());
    }
    // ... and with the exact spelling the term is scored: the doc that repeats it ranks first
    fz.terms = {"learning"};
    auto f2 = ExecuteBatch(index, {fz});
    EXPECT(f2.has_value());
    if (f2) {
      EXPECT(!(*f2)[0].results.empty() && (*f2)[0].results[0] == 4);
      EXPECT((*f2)[0].scores.size() == (*f2)[0].results.size() && (*f2)[0].scores[0] > (*f2)[0].scores.back());
    }
  }
  {  // the reference's EXACT signatures at the BM25 call sites (search_handler.cpp:454,469 / http_server.cpp:572,584):
     // ScoreDocuments(..., const DocumentStore&, ...) and SortByScore(results, scores, order, limit, offset)
    Index index(2, 0, 0.0, false);
    index.AddDocument(1, "alpha");
    index.AddDocument(2, "alpha alpha alpha");
    index.AddDocument(3, "beta");
    mygramdb::storage::DocumentStore doc_store(index);
    const V cands = index.SearchAnd({"al"});
    EXPECT(cands == (V{1, 2}));
    auto a = BM25Scorer::ScoreDocuments(cands, {"al"}, {index.PostingSize("al")}, doc_store, index.Bm25DocCount(),
                                        index.Bm25AvgDocLength(), BM25Params{});
    auto b = BM25Scorer::ScoreDocuments(cands, {"al"}, {index.PostingSize("al")}, index, index.Bm25DocCount(),
                                        index.Bm25AvgDocLength(), BM25Params{});
    EXPECT(a.has_value() && b.has_value());
    if (a && b) {
      EXPECT(a->size() == 2 && (*a)[0].score == (*b)[0].score && (*a)[1].score == (*b)[1].score && (*a)[1].score > (*a)[0].score);
      std::vector<double> scores{(*a)[0].score, (*a)[1].score};
      EXPECT(ResultSorter::SortByScore(cands, scores, SortOrder::DESC, 10, 0) == (V{2, 1}));
      EXPECT(ResultSorter::SortByScore(cands, scores, SortOrder::ASC, 1, 0) == (V{1}));
      EXPECT(mygramdb::index::LastDeviceError().empty());
    }
    EXPECT(doc_store.IsStoreTextsEnabled() && doc_store.Size() == 3);
  }
  {  // FILTER conditions + FACET on typed columns fed through AddDocument(doc, text, filters):
     // tests/server/search_pipeline_test.cpp:725-910 (SearchPipelineFilterParityTest), facet_handler_test.cpp:203-258
    using namespace mygramdb::search_pipeline;
    using mygramdb::query::FilterCondition;
    using mygramdb::query::FilterOp;
    using mygramdb::storage::FilterValue;
    Index index(2, 0, 0.0, false);
    index.AddDocument(1, "text zero", {{"status", FilterValue{int64_t{1}}}, {"category", FilterValue{std::string("tech")}}, {"score", FilterValue{85.5}}});
    index.AddDocument(2, "text one", {{"status", FilterValue{int64_t{2}}}, {"category", FilterValue{std::string("sports")}}, {"score", FilterValue{92.0}}});
    index.AddDocument(3, "text two", {{"status", FilterValue{int64_t{1}}}, {"category", FilterValue{std::string("tech")}}, {"score", FilterValue{78.0}}});
    index.AddDocument(4, "text three", {{"status", FilterValue{int64_t{3}}}, {"category", FilterValue{std::string("music")}}, {"score", FilterValue{60.0}}});
    index.AddDocument(5, "text four", {});  // NULL in every column
    auto run = [&](std::vector<FilterCondition> conds) {
      BatchQuery q;
      q.terms = {"text"};
      q.order = SortOrder::ASC;
      q.filter_conditions = std::move(conds);
      auto r = ExecuteBatch(index, {q});
      EXPECT(r.has_value());
      return r ? (*r)[0].results : V{};
    };
    EXPECT(run({{"status", FilterOp::EQ, "1"}}) == (V{1, 3}));                                   // :778
    EXPECT(run({{"status", FilterOp::NE, "1"}}) == (V{2, 4, 5}));                                // :792 (NULL passes NE)
    EXPECT(run({{"category", FilterOp::EQ, "tech"}}) == (V{1, 3}));                              // :804
    EXPECT(run({{"CATEGORY", FilterOp::EQ, "tech"}}) == (V{1, 3}));                              // :818
    EXPECT(run({{"status", FilterOp::EQ, "1"}, {"category", FilterOp::EQ, "tech"}}) == (V{1, 3}));  // :832
    EXPECT(run({{"status", FilterOp::EQ, "999"}}).empty());                                      // :847
    EXPECT(run({{"status", FilterOp::EQ, "1"}, {"score", FilterOp::GT, "80.0"}}) == (V{1}));     // :893
    EXPECT(run({{"score", FilterOp::LTE, "78"}}) == (V{3, 4}));
    EXPECT(run({{"nosuch", FilterOp::EQ, "1"}}).empty());
    EXPECT(run({{"nosuch", FilterOp::NE, "1"}}) == (V{1, 2, 3, 4, 5}));
    BatchQuery all;
    auto f = ExecuteFacet(index, all, "category");
    EXPECT(f.has_value());
    if (f) {
      EXPECT(f->matched_documents == 5 && f->total_values == 3);
      EXPECT(!f->value_counts.empty() && f->display[0] == "tech" && f->value_counts[0].second == 2);
      EXPECT(f->value_counts[0].first == std::string("\x0B") + "tech");  // SerializeFilterValue: string tag + bytes
    }
    BatchQuery paged;
    paged.limit = 1;
    paged.offset = 1;
    auto g = ExecuteFacet(index, paged, "status");
    EXPECT(g.has_value() && g->total_values == 3 && g->value_counts.size() == 1 && g->value_counts[0].second == 1);
    EXPECT(!ExecuteFacet(index, all, "nosuch").has_value());
  }
  {  // tests/server/search_pipeline_test.cpp:1492-1512 MixedScriptBoundaryFragmentRequiresExactTextMatch, and the
     // caller-driven verify_text filter
    using namespace mygramdb::search_pipeline;
    Index index(2, 1, 0.0, false);
    index.AddDocument(1, "machine learning basics");
    index.AddDocument(2, "deep learning techniques");
    index.AddDocument(3, "old article about cats");
    index.AddDocument(4, "\xe4\xba\xac\xe9\x83\xbd");                              // 京都
    index.AddDocument(5, "\xe4\xba\xac\xe3\x82\xbf\xe3\x83\xaf\xe3\x83\xbc");  // 京タワー
    index.AddDocument(6, "lea ear arn rni nin ing");
    std::vector<BatchQuery> qs(3);
    qs[0].terms = {"\xe4\xba\xac\xe3\x82\xbf"};  // 京タ
    qs[1].terms = {"learning"};
    qs[2].terms = {"learning"};
    qs[2].verify_text = true;
    for (auto& q : qs) q.order = SortOrder::ASC;
    auto r = ExecuteBatch(index, qs);
    EXPECT(r.has_value());
    if (r) {
      EXPECT((*r)[0].results == (V{5}));
      EXPECT((*r)[1].results == (V{1, 2, 6}));
      EXPECT((*r)[2].results == (V{1, 2}) && (*r)[2].total == 2 && (*r)[2].after_filters == 3);
    } else {
      std::printf("ExecuteBatch(exact) error: %s\n", r.error().message().c_str());
    }
  }
  {  // FUZZY (ExecuteWithFuzzy, search_pipeline.cpp:1659-1744) on the reference's pipeline fixture
     // (tests/server/search_pipeline_test.cpp:916-951, :2034-2052): "learnig" has the bigrams le ea ar rn ni ig,
     // theta = 6 - 1*2 = 4; docs 1 and 2 hold five of them
    using namespace mygramdb::search_pipeline;
    Index index(2, 0, 0.0, false);
    index.AddDocument(1, "machine learning basics");
    index.AddDocument(2, "deep learning techniques");
    index.AddDocument(3, "old article about cats");
    std::vector<BatchQuery> qs(4);
    qs[0].terms = {"learnig"};
    qs[0].fuzzy_max_distance = 1;
    qs[1].terms = {"learnig", "machine"};   // AND across terms, in the order given
    qs[1].fuzzy_max_distance = 1;
    qs[2].terms = {"zzzzzz"};               // no known gram: empty, not an error
    qs[2].fuzzy_max_distance = 2;
    qs[3].terms = {"learnig"};
    qs[3].fuzzy_max_distance = 1;
    qs[3].not_terms = {"deep"};
    for (auto& q : qs) q.order = SortOrder::ASC;
    auto r = ExecuteBatch(index, qs);
    EXPECT(r.has_value());
    if (r) {
      EXPECT((*r)[0].results == (V{1, 2}) && (*r)[0].total_candidates == 2);
      EXPECT((*r)[1].results == (V{1}));
      EXPECT((*r)[2].results.empty() && (*r)[2].total == 0);
      EXPECT((*r)[3].results == (V{1}) && (*r)[3].after_not == 1);
    } else {
      std::printf("ExecuteBatch(fuzzy) error: %s\n", r.error().message().c_str());
    }
    // BatchExecutor: fresh batches through re-used batch objects, two in flight; deep OFFSET through the same entry
    BatchExecutor::Options opt;
    opt.depth = 2;
    opt.planner_threads = 2;
    BatchExecutor ex(index, opt);
    std::vector<BatchQuery> a(2), b(1);
    a[0].terms = {"learning"};
    a[0].sort_by_score = true;
    a[0].limit = 10;
    a[1].terms = {"cats"};
    b[0].terms = {"learning"};
    b[0].sort_by_score = true;
    b[0].limit = 5;
    b[0].offset = 1;  // second-ranked doc only
    for (int round = 0; round < 3; ++round) {
      auto ta = ex.Submit(a);
      auto tb = ex.Submit(b);
      EXPECT(ta.has_value() && tb.has_value());
      auto full = ex.Submit(a);
      EXPECT(!full.has_value());  // both slots hold unfetched batches
      if (!ta || !tb) break;
      auto ra = ex.Wait(*ta);
      auto rb = ex.Wait(*tb);
      EXPECT(ra.has_value() && rb.has_value());
      if (ra && rb) {
        EXPECT((*ra)[0].total == 2 && (*ra)[0].results.size() == 2 && (*ra)[1].results == (V{3}));
        EXPECT((*rb)[0].results.size() == 1 && (*rb)[0].results[0] == (*ra)[0].results[1]);
        EXPECT((*rb)[0].scores.size() == 1 && (*rb)[0].scores[0] == (*ra)[0].scores[1]);
      }
    }
    // the allocation-free serving loop: Submit(vector&&) hands an earlier batch's objects back, WaitInto fills a vector
    // the caller keeps (entries of a previous, LARGER batch must not leak into a smaller one)
    std::vector<BatchQuery> build;
    std::vector<BatchResult> kept;
    for (int round = 0; round < 4; ++round) {
      const bool big = round % 2 == 0;
      build.resize(big ? 2 : 1);
      for (auto& q : build) q = BatchQuery{};
      build[0].terms = {"learning"};
      build[0].sort_by_score = true;
      build[0].limit = 10;
      if (big) build[1].terms = {"cats"};
      auto t = ex.Submit(std::move(build));
      EXPECT(t.has_value());
      if (!t) break;
      const auto err = ex.WaitInto(*t, &kept);
      EXPECT(err.code() == mygram::utils::ErrorCode::kSuccess);
      EXPECT(kept.size() == (big ? 2u : 1u));
      EXPECT(kept[0].total == 2 && kept[0].results.size() == 2 && kept[0].scores.size() == 2);
      if (big) EXPECT(kept[1].results == (V{3}));
    }
    EXPECT(ex.WaitInto(12345, &kept).code() != mygram::utils::ErrorCode::kSuccess);  // unknown ticket
  }
  {  // No silent empty result (the reference tests its own allocation failure path,
     // tests/index/posting_list_test.cpp:188-228): with device allocations failing, the reference-signature methods
     // return {} AND report through LastDeviceError(); with the hook off the same call succeeds and the error is clear
    Index index(1, 0, 0.0);
    for (DocId d = 1; d <= 6000; ++d) index.AddDocument(d, d % 3 == 0 ? "ab" : "a");
    EXPECT(index.SearchAnd({"b"}).size() == 2000 && mygramdb::index::LastDeviceError().empty());
    V cand;
    for (DocId d = 1; d <= 6000; d += 7) cand.push_back(d);
    const V ok = index.FilterByNgrams(cand, {"a", "b"});
    EXPECT(!ok.empty() && mygramdb::index::LastDeviceError().empty());
    // (an operator leases arenas the index keeps from call to call — the steady state allocates nothing — so the failure
    // is injected where an allocation must happen: the first calls on an index that has only been built)
    Index cold(1, 0, 0.0);
    for (DocId d = 1; d <= 6000; ++d) cold.AddDocument(d, d % 3 == 0 ? "ab" : "a");
    EXPECT(cold.Finalize().empty());
    mgxt_fail_device_allocs(0, 1000);
    const V failed = cold.FilterByNgrams(cand, {"a", "b"});
    EXPECT(failed.empty() && !mygramdb::index::LastDeviceError().empty());
    const V sorted = ResultSorter::SortByScore(cold, ok, std::vector<double>(ok.size(), 1.0), SortOrder::DESC, 10, 0);
    EXPECT(sorted.empty() && !mygramdb::index::LastDeviceError().empty());
    EXPECT(cold.SearchAnd({"a", "b"}).empty() && !mygramdb::index::LastDeviceError().empty());
    {
      Index fresh(2, 0, 0.0);  // its device index cannot even be built
      fresh.AddDocument(1, "hello");
      EXPECT(fresh.SearchAnd({"he"}).empty() && !mygramdb::index::LastDeviceError().empty());
    }
    mgxt_fail_device_allocs(0, 0);
    EXPECT(index.FilterByNgrams(cand, {"a", "b"}) == ok && mygramdb::index::LastDeviceError().empty());
    EXPECT(ResultSorter::SortByScore(index, ok, std::vector<double>(ok.size(), 1.0), SortOrder::DESC, 10, 0).size() == 10);
  }
  std::printf("shim_test: %d checks, %d failed\n", g_checked, g_failed);
  return g_failed == 0 ? 0 : 1;
}
