# Builds the product library (HIP, gfx950 only) and the CPU oracle (test infra).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -fno-slp-vectorize -Wall -Wno-unused-function
SRC := beamform_amd/csrc
OBJ := build/obj
LIB := beamform_amd/lib/libbfcore.so

HIP_SRCS := $(SRC)/das_fused.hip $(SRC)/das_fused_gen.hip $(SRC)/das_f64_w64.hip $(SRC)/pipeline.hip $(SRC)/convert.hip $(SRC)/resample.hip
# bin-pipeline kernels: one object per supported FFT size (JACK periods 64 ... 4096 frames -> -DBF_NFFT=128 ... 8192)
BIN_SRCS := pipeline_kernels stft_istft mask_kernels cov_kernels gsc_gss_kernels
NFFTS := 128 256 512 1024 2048 4096 8192
CPP_SRCS := $(SRC)/capi.cpp $(SRC)/config.cpp $(SRC)/wavio.cpp
OBJS := $(patsubst $(SRC)/%.hip,$(OBJ)/%.o,$(HIP_SRCS)) $(patsubst $(SRC)/%.cpp,$(OBJ)/%.o,$(CPP_SRCS)) \
        $(foreach n,$(NFFTS),$(foreach b,$(BIN_SRCS),$(OBJ)/$(b)_n$(n).o))
HDRS := $(wildcard $(SRC)/*.hpp) include/bfcore.h

all: $(LIB) oracle examples/file_node examples/theta_scan examples/shard_node

examples/shard_node: examples/shard_node.cpp include/bfcore.h $(LIB)
	$(HIPCC) -O2 -std=c++17 -x hip --offload-arch=$(ARCH) examples/shard_node.cpp -o $@ -Lbeamform_amd/lib -lbfcore -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN/../beamform_amd/lib'

examples/theta_scan: examples/theta_scan.cpp include/bfcore.h $(LIB)
	$(HIPCC) -O2 -std=c++17 -x hip --offload-arch=$(ARCH) examples/theta_scan.cpp -o $@ -Lbeamform_amd/lib -lbfcore -Wl,-rpath,'$$ORIGIN/../beamform_amd/lib'

examples/file_node: examples/file_node.cpp include/bf_node_shim.hpp include/bfcore.h $(LIB)
	$(HIPCC) -O2 -std=c++17 -x hip --offload-arch=$(ARCH) examples/file_node.cpp -o $@ -Lbeamform_amd/lib -lbfcore -Wl,-rpath,'$$ORIGIN/../beamform_amd/lib'

$(OBJ)/%.o: $(SRC)/%.hip $(HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

define BIN_RULE
$(OBJ)/%_n$(1).o: $(SRC)/%.hip $(HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DBF_NFFT=$(1) -c $$< -o $$@
endef
$(foreach n,$(NFFTS),$(eval $(call BIN_RULE,$(n))))

$(OBJ)/%.o: $(SRC)/%.cpp $(HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p beamform_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle -s

ubench:
	for f in valu_rate fetch_calib rowbc vmem_issue rsq_test f64_rate launch_lds realtime_cal power_clock; do $(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-result tools/ubench/$$f.hip -o tools/ubench/$$f.bin; done

# which kernels run for which (node, period, layout, microphones, directions, dump): traced from the library itself (needs a GPU)
dispatch-table: $(LIB)
	python tools/dispatch_table.py docs/DISPATCH.md

emul:
	g++ -O2 -std=c++17 -fPIC -shared -o tests/host_emul/libemul.so tests/host_emul/emul.cpp

clean:
	rm -rf build $(LIB) tests/host_emul/libemul.so examples/file_node examples/theta_scan examples/shard_node
	$(MAKE) -C oracle clean

.PHONY: all oracle emul ubench clean dispatch-table
