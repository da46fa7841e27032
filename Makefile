# Top-level build: the gfx950 engine (HIP, C ABI) and the CPU checker under oracle/.
# hipcc cross-compiles for gfx950 without a GPU.  No FP contraction, no fast-math:
# the numerics contract depends on it.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
# EXTRA=-DSCL_DIAGNOSTICS builds the SC-distance kernels' phase ablation / stamps / occupancy overrides in
# (scripts/ablate_k1.sh); the product build has none of them
EXTRA   ?=
CSRC    := scl_slam_amd/csrc
LIBDIR  := scl_slam_amd/lib
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
            -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-result -Iinclude -I$(CSRC) $(EXTRA)
SRCS    := $(CSRC)/engine.hip $(CSRC)/sc_distance.hip $(CSRC)/ringkey_topk.hip $(CSRC)/make_sc.hip $(CSRC)/icp.hip $(CSRC)/voxel.hip $(CSRC)/sharded_front.hip $(CSRC)/sc_screen.hip $(CSRC)/sc_masked.hip $(CSRC)/sc_matrix.hip $(CSRC)/messages.hip $(CSRC)/iris.hip $(CSRC)/device_sort.hip
OBJS    := $(SRCS:.hip=.o)

all: $(LIBDIR)/libscl_engine.so oracle tests/cpp/adapter_check tests/cpp/libmock_rccl.so

$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.hpp) include/scl_engine.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libscl_engine.so: $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

oracle:
	$(MAKE) -C oracle

# C++ host-side adapter (include/scl/scan_context_hip_descriptor.hpp) type-checked and linked
# against the C ABI; runs on the GPU box (tests/test_gpu_adapter.py)
tests/cpp/adapter_check: tests/cpp/adapter_check.cpp tests/cpp/pcl_types_for_adapter_check.h include/scl/scan_context_hip_descriptor.hpp include/scl/lidar_iris_hip_descriptor.hpp $(LIBDIR)/libscl_engine.so
	g++ -std=c++14 -O2 -Wall -Iinclude -Itests/cpp -o $@ tests/cpp/adapter_check.cpp -L$(LIBDIR) -lscl_engine -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)'

# TEST INFRASTRUCTURE: the stand-in collective the sharded front's G > 1 test loads through SCL_RCCL_LIB (never linked into the product)
tests/cpp/libmock_rccl.so: tests/cpp/mock_rccl.cpp
	$(HIPCC) -x hip --offload-arch=$(ARCH) -O2 -std=c++17 -fPIC -shared -o $@ $<

# Sanitizers + fuzzing over everything that builds without a GPU (log: profiles/rNN/sanitize.txt):
#   1. the CPU checker under ASan + UBSan: the whole `-m "not gpu"` suite against oracle/liboracle_asan.so
#   2. the checker's worker pool under TSan (oracle/tools/tsan_pool_driver)
#   3. libFuzzer + ASan + UBSan over the host-only parsers of untrusted bytes: the ROS wire decoders (messages.hip, compiled as host
#      C++) and the database dump parser (db_file.hpp) -- FUZZ_SECONDS (600) of mutation from tests/cpp/fuzz_seeds.py's corpus
CLANGXX ?= /opt/rocm/lib/llvm/bin/clang++
FUZZ_SECONDS ?= 600
SAN_LOG ?= profiles/r05/sanitize.txt
# (-asan-globals=0: ROCm's clang registers the binary's instrumented globals twice and ASan reports every string literal as an ODR
#  violation before main() runs; heap, stack and UB checks -- what a parser of untrusted bytes can get wrong -- are unaffected)
tests/cpp/fuzz_host: tests/cpp/fuzz_host.cpp $(CSRC)/messages.hip $(CSRC)/db_file.hpp include/scl_messages.h include/scl_engine.h
	$(CLANGXX) -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=fuzzer,address,undefined -fno-sanitize-recover=undefined -mllvm -asan-globals=0 \
	    -Iinclude -I$(CSRC) -o $@ tests/cpp/fuzz_host.cpp

sanitize: tests/cpp/fuzz_host
	$(MAKE) -C oracle liboracle_asan.so tools/tsan_pool_driver
	@mkdir -p $(dir $(SAN_LOG)) /tmp/scl_fuzz_corpus
	@echo "== make sanitize: $$(date -u +%Y-%m-%dT%H:%MZ), $$(gcc --version | head -1), $$($(CLANGXX) --version | head -1)" > $(SAN_LOG)
	@echo "== 1. CPU tests against oracle/liboracle_asan.so (-fsanitize=address,undefined, libasan preloaded into python)" >> $(SAN_LOG)
	SCL_ORACLE_LIB=oracle/liboracle_asan.so LD_PRELOAD=$$(gcc -print-file-name=libasan.so):$$(gcc -print-file-name=libubsan.so) \
	    ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
	    python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider > /tmp/scl_san_tests.log 2>&1; \
	    grep -E "runtime error|ERROR: AddressSanitizer|SUMMARY" /tmp/scl_san_tests.log | head -5 >> $(SAN_LOG); tail -2 /tmp/scl_san_tests.log >> $(SAN_LOG)
	@echo "== 2. worker pool under ThreadSanitizer (oracle/tools/tsan_pool_driver)" >> $(SAN_LOG)
	TSAN_OPTIONS=halt_on_error=1 oracle/tools/tsan_pool_driver >> $(SAN_LOG) 2>&1
	@echo "== 3. libFuzzer + ASan + UBSan: wire decoders (messages.hip) and dump parser (db_file.hpp), $(FUZZ_SECONDS) s" >> $(SAN_LOG)
	python tests/cpp/fuzz_seeds.py /tmp/scl_fuzz_corpus
	tests/cpp/fuzz_host -max_total_time=$(FUZZ_SECONDS) -max_len=4096 -print_final_stats=1 /tmp/scl_fuzz_corpus 2>&1 | grep -E "stat::|ERROR|SUMMARY|Done|cov:" | tail -12 >> $(SAN_LOG)
	@echo "== done: no finding above means none was reported (every tool stops at its first)" >> $(SAN_LOG)
	@cat $(SAN_LOG)

clean:
	rm -f $(OBJS) $(LIBDIR)/libscl_engine.so tests/cpp/adapter_check tests/cpp/libmock_rccl.so
	$(MAKE) -C oracle clean

.PHONY: all oracle clean sanitize
