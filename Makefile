# Builds the product (HIP codec library + HDF5 filter plugin) and the test-only oracle.
#   make            -> deltarice_amd/libdeltarice_hip.so, deltarice_amd/plugin/libh5deltarice.so
#   make oracle     -> oracle/libdeltarice_oracle.so (+ oracle/_ref when /root/reference exists)
# hipcc cross-compiles gfx950 code objects without a GPU.

HIPCC     ?= /opt/rocm/bin/hipcc
CC        ?= gcc
ARCH      ?= gfx950
HDF5_DIR  ?= /opt/conda
HIPFLAGS  ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function

CSRC      := deltarice_amd/csrc
HIP_LIB   := deltarice_amd/libdeltarice_hip.so
PLUGIN    := deltarice_amd/plugin/libh5deltarice.so
HIP_SRCS  := $(CSRC)/drx_kernels.hip $(CSRC)/drx_api.hip
HIP_HDRS  := $(CSRC)/drx_internal.h include/deltarice_hip.h

.PHONY: all hip plugin oracle clean
all: hip plugin

hip: $(HIP_LIB)
$(HIP_LIB): $(HIP_SRCS) $(HIP_HDRS)
	$(HIPCC) $(HIPFLAGS) -shared $(HIP_SRCS) -o $@

plugin: $(PLUGIN)
$(PLUGIN): $(CSRC)/h5z_deltarice.c include/deltarice_h5filter.h include/deltarice_hip.h $(HIP_LIB)
	@mkdir -p deltarice_amd/plugin
	$(CC) -O2 -std=gnu11 -Wall -fPIC -shared -Iinclude -I$(HDF5_DIR)/include $< -o $@ \
	    -Ldeltarice_amd -ldeltarice_hip -Wl,-rpath,'$$ORIGIN/..' -ldl

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f $(HIP_LIB) $(PLUGIN)
	$(MAKE) -C oracle clean
