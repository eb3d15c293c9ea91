// mgx_host.hpp — host-side internals shared by the translation units of libmygram_gpu.so.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/mygram_gpu.h"

namespace mgx {

struct Columns;
int BuildColumns(const mgx_build_params& bp, const uint8_t* text_bytes, const uint64_t* text_off,
                 uint32_t first_doc_id, uint64_t n_docs, Columns** out, std::string* err);
int ColumnsFromMgix(const uint8_t* data, uint64_t len, uint32_t first_doc_id, uint64_t n_docs, Columns** out,
                    mgx_mgix_info* info, std::string* err);
void ColumnsView(const Columns* c, mgx_columns_view* v);
bool ColumnsLookup(const Columns* c, const uint8_t* gram, size_t len, uint32_t* id);
void DestroyColumns(Columns* c);
struct DumpData;
int DumpOpen(const uint8_t* data, uint64_t len, const char* table, DumpData** out, std::string* err);
void DumpDestroy(DumpData* d);
void DumpView(const DumpData* d, mgx_dump_view* v);
bool DumpFilterColumn(const DumpData* d, uint32_t i, mgx_dump_filter_column* o);
Columns* DumpTakeColumns(DumpData* d);

// thread-local last-error message
void SetError(const std::string& msg);
int Fail(int code, const std::string& msg);

}  // namespace mgx
