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
HIP_SRCS  := $(CSRC)/drx_encode_kernels.hip $(CSRC)/drx_encode_stream.hip $(CSRC)/drx_decode_kernels.hip $(CSRC)/drx_blocks.hip $(CSRC)/drx_pieces.hip $(CSRC)/drx_iir.hip $(CSRC)/drx_api.hip
HIP_OBJS  := $(HIP_SRCS:.hip=.o)
HIP_HDRS  := $(CSRC)/drx_internal.h $(CSRC)/drx_device.h $(CSRC)/drx_encode.h $(CSRC)/drx_walk.h $(CSRC)/drx_iir_math.h include/deltarice_hip.h

H5IO      := deltarice_amd/libdeltarice_h5io.so

# Python binding (the reference's `deltaRice.h5`, src/h5.pyx) and the install step of its setup.py
# (`install --h5plugin --h5plugin-dir=DIR`, setup.py:186-227; default dir = setup.py:44)
PYTHON    ?= python3
CYTHON    ?= cython
PY_SUFFIX := $(shell $(PYTHON) -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
PY_INC    := $(shell $(PYTHON) -c "import sysconfig; print(sysconfig.get_paths()['include'])")
PYEXT     := deltaRice/h5$(PY_SUFFIX)
PLUGIN_DIR ?= /usr/local/hdf5/lib/plugin

.PHONY: all hip plugin h5io pyext install-plugin oracle asan check-asm clean
all: hip plugin h5io pyext

hip: $(HIP_LIB)
# one object per translation unit (no device code crosses them), so that a change to one kernel file rebuilds that file
$(CSRC)/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(HIP_LIB): $(HIP_OBJS)
	$(PYTHON) tools/check_asm_hazards.py $(HIP_OBJS)
	$(HIPCC) $(HIPFLAGS) -shared $(HIP_OBJS) -o $@
# the hazards a hand-written asm statement can hide from the compiler (VALU-written SGPR base within five wait states of a
# vector-memory instruction; a returning atomic's destination touched before its s_waitcnt), on the gfx950 disassembly
check-asm: $(HIP_OBJS)
	$(PYTHON) tools/check_asm_hazards.py $(HIP_OBJS)

plugin: $(PLUGIN)
$(PLUGIN): $(CSRC)/h5z_deltarice.c include/deltarice_h5filter.h include/deltarice_hip.h $(HIP_LIB)
	@mkdir -p deltarice_amd/plugin
	$(CC) -O2 -std=gnu11 -Wall -fPIC -shared -Iinclude -I$(HDF5_DIR)/include $< -o $@ \
	    -Ldeltarice_amd -ldeltarice_hip -Wl,-rpath,'$$ORIGIN/..' -ldl

# direct-chunk file <-> VRAM path; links the application's libhdf5 and the HIP runtime C API
h5io: $(H5IO)
$(H5IO): $(CSRC)/h5_direct.c include/deltarice_h5io.h include/deltarice_hip.h $(HIP_LIB)
	$(CC) -O2 -std=gnu11 -Wall -fPIC -shared -Iinclude -I$(HDF5_DIR)/include -I/opt/rocm/include $< -o $@ \
	    -Ldeltarice_amd -ldeltarice_hip -L$(HDF5_DIR)/lib -lhdf5 -L/opt/rocm/lib -lamdhip64 -ldl -lpthread \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(HDF5_DIR)/lib -Wl,-rpath,/opt/rocm/lib

# deltaRice/h5.pyx -> deltaRice/h5.cpython-*.so, linked against the plugin library (whose callback runs the HIP codec).
# h5py is needed to IMPORT the module, not to build it.
# A box without cython or the Python headers fails `make all` loudly (the drop-in module would be missing from a build that
# exits 0); `make SKIP_PYEXT=1` builds the codec library and the plugin without it.
HAVE_PYEXT_TOOLS := $(shell command -v $(CYTHON) >/dev/null 2>&1 && [ -f "$(PY_INC)/Python.h" ] && echo yes)
ifeq ($(HAVE_PYEXT_TOOLS),yes)
pyext: $(PYEXT)
else
pyext:
ifeq ($(SKIP_PYEXT),1)
	@echo "pyext: skipped (SKIP_PYEXT=1; $(CYTHON) or $(PY_INC)/Python.h not found); deltaRice.h5 needs it, the codec and the HDF5 plugin do not"
else
	@echo "pyext: $(CYTHON) or $(PY_INC)/Python.h not found -- the reference's Python surface (deltaRice.h5) cannot be built."; \
	 echo "       \`make SKIP_PYEXT=1\` builds the codec library and the HDF5 plugin without it."; exit 1
endif
endif
.PHONY: pyext-strict
pyext-strict: $(PYEXT)
$(PYEXT): deltaRice/h5.pyx include/deltarice_h5filter.h $(PLUGIN)
	@mkdir -p build/pyext
	$(CYTHON) -3 -o build/pyext/h5.c deltaRice/h5.pyx
	$(CC) -O2 -fPIC -shared -Wall -Wno-unused-function -I$(PY_INC) -Iinclude -I$(HDF5_DIR)/include build/pyext/h5.c -o $@ \
	    -Ldeltarice_amd/plugin -l:libh5deltarice.so -Wl,-rpath,'$$ORIGIN/../deltarice_amd/plugin'

# Copies the filter plugin to where HDF5 looks for plugins (HDF5_PLUGIN_PATH, default /usr/local/hdf5/lib/plugin)
# and the codec library one directory above it, where the plugin's $ORIGIN/.. run path finds it.
install-plugin: $(PLUGIN) $(HIP_LIB)
	install -d $(PLUGIN_DIR)
	install -m 755 $(PLUGIN) $(PLUGIN_DIR)/libh5deltarice.so
	install -m 755 $(HIP_LIB) $(PLUGIN_DIR)/../libdeltarice_hip.so
	@echo "Installed HDF5 filter plugins to $(PLUGIN_DIR)"

oracle:
	$(MAKE) -C oracle all
# CPU-side sanitizer builds of the restatement, the plugin's C source and the HDF5 test driver (tests/test_sanitizers.py)
asan: $(HIP_LIB)
	$(MAKE) -C oracle asan

clean:
	rm -f $(HIP_LIB) $(HIP_OBJS) $(PLUGIN) $(H5IO) $(PYEXT)
	rm -rf build/pyext
	$(MAKE) -C oracle clean
