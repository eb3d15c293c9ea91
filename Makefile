# Builds libmygram_gpu.so (HIP kernels for gfx950 + host C ABI) and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := mygram-db_amd/csrc
LIB := mygram-db_amd/libmygram_gpu.so
# -ffp-contract=off: BM25 arithmetic must round exactly like the reference's fp64 expression (no fused multiply-add)
CXXFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wextra -Wno-unused-parameter -Iinclude
HIPFLAGS := --offload-arch=$(ARCH) $(CXXFLAGS)

OBJS := $(CSRC)/mgx_kernels.o $(CSRC)/mgx_api.o $(CSRC)/mgx_columns.o $(CSRC)/mgx_tools.o

SHIM := mygram-db_amd/libmygram_shim.so

all: $(LIB) $(SHIM) oracle

$(CSRC)/mgx_kernels.o: $(CSRC)/mgx_kernels.hip $(CSRC)/mgx_internal.hpp $(CSRC)/mgx_launch.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/mgx_api.o: $(CSRC)/mgx_api.cpp $(CSRC)/mgx_internal.hpp $(CSRC)/mgx_launch.hpp $(CSRC)/mgx_host.hpp include/mygram_gpu.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@
$(CSRC)/mgx_columns.o: $(CSRC)/mgx_columns.cpp $(CSRC)/mgx_host.hpp $(CSRC)/mgx_text.hpp include/mygram_gpu.h
	g++ $(CXXFLAGS) -pthread -c $< -o $@
$(CSRC)/mgx_tools.o: $(CSRC)/mgx_tools.cpp include/mygram_tools.h
	g++ $(CXXFLAGS) -pthread -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -pthread

# host C++17 layer (reference-signature classes, ExecuteBatch, BatchExecutor) + its C face for ctypes / cgo / JNI callers
# ICU (NormalizeText: NFKC / width / lower) when the system has it, like the reference's USE_ICU; MGX_NO_ICU=1 forces
# the ASCII fallback. mgxs_normalize_uses_icu() / NormalizeTextUsesIcu() report which branch was built.
ICU_OK := $(if $(MGX_NO_ICU),,$(shell printf '\043include <unicode/unorm2.h>\nint main(){return 0;}' | g++ -x c++ - -licuuc -licui18n -o /dev/null 2>/dev/null && echo 1))
ICU_FLAGS := $(if $(ICU_OK),-DMGX_USE_ICU,)
ICU_LIBS := $(if $(ICU_OK),-licui18n -licuuc,)
SHIM_SRC := $(CSRC)/shim/mygram_shim.cpp $(CSRC)/shim/shim_capi.cpp $(CSRC)/shim/normalize.cpp
$(SHIM): $(CSRC)/mgx_text.hpp $(SHIM_SRC) $(CSRC)/shim/mygram_shim.hpp include/mygram_shim_c.h include/mygram_gpu.h $(LIB)
	g++ $(CXXFLAGS) $(ICU_FLAGS) -shared -o $@ $(SHIM_SRC) -Lmygram-db_amd -lmygram_gpu $(ICU_LIBS) -Wl,-rpath,'$$ORIGIN' -pthread

oracle:
	$(MAKE) -s -C oracle

# timing-ablation build of the library (tools only: MGX_DEBUG_SKIP then skips kernel phases; never shipped as the product)
ablation:
	$(HIPCC) $(HIPFLAGS) -DMGX_ABLATION -c $(CSRC)/mgx_kernels.hip -o /tmp/mgx_kernels_abl.o
	$(HIPCC) $(HIPFLAGS) -DMGX_ABLATION -x hip -c $(CSRC)/mgx_api.cpp -o /tmp/mgx_api_abl.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o mygram-db_amd/libmygram_gpu_ablation.so /tmp/mgx_kernels_abl.o /tmp/mgx_api_abl.o $(CSRC)/mgx_columns.o $(CSRC)/mgx_tools.o -pthread

clean:
	rm -f $(OBJS) $(LIB) $(SHIM)
	$(MAKE) -C oracle clean
.PHONY: all oracle clean ablation
