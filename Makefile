# Builds the MI355X-native library (hand-written HIP for gfx950 behind the C ABI in include/),
# the CPU oracle (test infrastructure) and the host-compiled arithmetic check used by CPU tests.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off + correctly rounded div/sqrt: the arithmetic contract shared with oracle/
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DOCML_BASIC_ROUNDED_OPERATIONS \
            -fPIC -Wall -Wno-unused-function -Wno-pass-failed
PKG      := cuda-sfm_amd
CSRC     := $(PKG)/csrc
BUILD    := build
LIB      := $(PKG)/lib/libsfm_amd.so
COMMLIB  := $(PKG)/lib/libsfm_amd_rccl.so
SRCS     := $(wildcard $(CSRC)/*.hip)
OBJS     := $(patsubst $(CSRC)/%.hip,$(BUILD)/%.o,$(SRCS))
HDRS     := $(wildcard $(CSRC)/*.hpp) include/sfm_amd.h

DEMO     := $(PKG)/host/two_view_demo
HDEMO    := $(PKG)/host/homography_demo
SDEMO    := $(PKG)/host/sift_demo
MAINAPP  := $(PKG)/host/sfm_main

IOTEST   := tests/cpp/io_test
GEOMTEST := tests/cpp/geom_test

all: $(LIB) $(COMMLIB) oracle hostcheck $(DEMO) $(HDEMO) $(SDEMO) $(MAINAPP) $(IOTEST) $(GEOMTEST)

$(BUILD)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) $(FLAGS_$*) -c $< -o $@

# the lane-solve kernels are long chains of scalar FP32 operations: the SLP vectoriser pairs some of them into v_pk_* and pays
# for it with register moves (366 v_mov in 2473 instructions); without it the same arithmetic needs fewer issue slots
FLAGS_ransac := -fno-slp-vectorize
# match_fused's two interleaved exact chains: paired into v_pk_fma_f32 they need a register move per operand
FLAGS_match_fused := -fno-slp-vectorize

$(LIB): $(OBJS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -lpthread

# the RCCL exchange step lives in its own library: libsfm_amd.so itself has no RCCL dependency
$(COMMLIB): $(CSRC)/comm.cpp include/sfm_amd_comm.h include/sfm_amd.h $(LIB)
	$(HIPCC) -x hip --cuda-host-only -O2 -fPIC -shared -Iinclude -o $@ $< -L$(PKG)/lib -lsfm_amd -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle

# plain C++ host program on the facade headers: no HIP headers, links only the C-ABI library
$(DEMO): $(PKG)/host/two_view_demo.cpp $(PKG)/host/sfm.h $(PKG)/host/cudaSift.h include/sfm_amd.h $(LIB)
	g++ -O2 -std=c++14 -Wall -o $@ $< -L$(PKG)/lib -lsfm_amd -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib

$(HDEMO): $(PKG)/host/homography_demo.cpp $(PKG)/host/geomFuncs.h $(PKG)/host/cudaSift.h include/sfm_amd.h $(LIB)
	g++ -O2 -std=c++14 -ffp-contract=off -Wall -o $@ $< -L$(PKG)/lib -lsfm_amd -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib

$(SDEMO): $(PKG)/host/sift_demo.cpp $(PKG)/host/cudaImage.h $(PKG)/host/sfm_io.h $(PKG)/host/cudaSift.h include/sfm_amd.h $(LIB)
	g++ -O2 -std=c++14 -Wall -o $@ $< -L$(PKG)/lib -lsfm_amd -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib

$(MAINAPP): $(PKG)/host/sfm_main.cpp $(PKG)/host/cudaImage.h $(PKG)/host/sfm.h $(PKG)/host/sfm_io.h $(PKG)/host/cudaSift.h include/sfm_amd.h $(LIB)
	g++ -O2 -std=c++14 -Wall -o $@ $< -L$(PKG)/lib -lsfm_amd -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib

$(IOTEST): tests/cpp/io_test.cpp $(PKG)/host/sfm_io.h $(PKG)/host/cudaSift.h include/sfm_amd.h
	g++ -O2 -std=c++14 -Wall -o $@ $<

$(GEOMTEST): tests/cpp/geom_test.cpp $(PKG)/host/geomFuncs.h $(PKG)/host/sfm_io.h $(PKG)/host/cudaSift.h include/sfm_amd.h
	g++ -O2 -std=c++14 -ffp-contract=off -Wall -o $@ $<

hostcheck: tests/hostcheck/libhostcheck.so

tests/hostcheck/libhostcheck.so: tests/hostcheck/hostcheck.hip $(CSRC)/device_math.hpp $(CSRC)/sift_math.hpp $(CSRC)/prefilter_math.hpp $(CSRC)/match_prefilter_math.hpp
	$(HIPCC) -x hip --cuda-host-only -O2 -ffp-contract=off -mfma -fPIC -shared -Wno-pass-failed -o $@ $<

clean:
	rm -rf $(BUILD) $(LIB) $(DEMO) $(HDEMO) $(SDEMO) $(MAINAPP) $(IOTEST) $(GEOMTEST) tests/hostcheck/libhostcheck.so
	$(MAKE) -C oracle clean

.PHONY: all oracle hostcheck clean
