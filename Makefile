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
SRCS    := $(CSRC)/engine.hip $(CSRC)/sc_distance.hip $(CSRC)/ringkey_topk.hip $(CSRC)/make_sc.hip $(CSRC)/icp.hip $(CSRC)/voxel.hip $(CSRC)/sharded_front.hip $(CSRC)/sc_screen.hip $(CSRC)/sc_masked.hip $(CSRC)/sc_matrix.hip $(CSRC)/messages.hip $(CSRC)/iris.hip
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

clean:
	rm -f $(OBJS) $(LIBDIR)/libscl_engine.so tests/cpp/adapter_check tests/cpp/libmock_rccl.so
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
