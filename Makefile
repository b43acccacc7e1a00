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

H5IO      := deltarice_amd/libdeltarice_h5io.so

.PHONY: all hip plugin h5io oracle clean
all: hip plugin h5io

hip: $(HIP_LIB)
$(HIP_LIB): $(HIP_SRCS) $(HIP_HDRS)
	$(HIPCC) $(HIPFLAGS) -shared $(HIP_SRCS) -o $@

plugin: $(PLUGIN)
$(PLUGIN): $(CSRC)/h5z_deltarice.c include/deltarice_h5filter.h include/deltarice_hip.h $(HIP_LIB)
	@mkdir -p deltarice_amd/plugin
	$(CC) -O2 -std=gnu11 -Wall -fPIC -shared -Iinclude -I$(HDF5_DIR)/include $< -o $@ \
	    -Ldeltarice_amd -ldeltarice_hip -Wl,-rpath,'$$ORIGIN/..' -ldl

# direct-chunk file <-> VRAM path; links the application's libhdf5 and the HIP runtime C API
h5io: $(H5IO)
$(H5IO): $(CSRC)/h5_direct.c include/deltarice_h5io.h include/deltarice_hip.h $(HIP_LIB)
	$(CC) -O2 -std=gnu11 -Wall -fPIC -shared -Iinclude -I$(HDF5_DIR)/include -I/opt/rocm/include $< -o $@ \
	    -Ldeltarice_amd -ldeltarice_hip -L$(HDF5_DIR)/lib -lhdf5 -L/opt/rocm/lib -lamdhip64 -ldl \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(HDF5_DIR)/lib -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f $(HIP_LIB) $(PLUGIN) $(H5IO)
	$(MAKE) -C oracle clean
