# Builds the MI355X-native library (hand-written HIP for gfx950 behind the C ABI in include/),
# the CPU oracle (test infrastructure) and the host-compiled arithmetic check used by CPU tests.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off + correctly rounded div/sqrt: the arithmetic contract shared with oracle/
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DOCML_BASIC_ROUNDED_OPERATIONS \
            -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-pass-failed
PKG      := cuda-sfm_amd
CSRC     := $(PKG)/csrc
BUILD    := build
BUILD_AB := build/ab
LIB      := $(PKG)/lib/libsfm_amd.so
COMMLIB  := $(PKG)/lib/libsfm_amd_rccl.so
# Two flavours of the same sources:
#   libsfm_amd.so     the product: what include/sfm_amd.h declares, nothing else (sfm_ransac_params.reserved[] must be 0)
#   libsfm_amd_ab.so  the lab bench (-DSFM_AB=1): + the A/B switches behind reserved[], the recorded slower kernel variants
#                     ($(CSRC)/ab/*.hip) and the probe / trace hooks of include/sfm_amd_ab.h.  Loaded only by tests/ and
#                     profiles/ (`import cuda_sfm_amd_ab`), never by the product path.
LIB_AB   := $(PKG)/lib/libsfm_amd_ab.so
SRCS     := $(wildcard $(CSRC)/*.hip)
SRCS_AB  := $(wildcard $(CSRC)/ab/*.hip)
OBJS     := $(patsubst $(CSRC)/%.hip,$(BUILD)/%.o,$(SRCS))
OBJS_AB  := $(patsubst $(CSRC)/%.hip,$(BUILD_AB)/%.o,$(SRCS)) $(patsubst $(CSRC)/ab/%.hip,$(BUILD_AB)/ab_%.o,$(SRCS_AB))
HDRS     := $(wildcard $(CSRC)/*.hpp) include/sfm_amd.h include/sfm_amd_ab.h

DEMO     := $(PKG)/host/two_view_demo
HDEMO    := $(PKG)/host/homography_demo
SDEMO    := $(PKG)/host/sift_demo
MAINAPP  := $(PKG)/host/sfm_main

IOTEST   := tests/cpp/io_test
GEOMTEST := tests/cpp/geom_test

all: $(LIB) $(LIB_AB) $(COMMLIB) oracle hostcheck fakeccl $(DEMO) $(HDEMO) $(SDEMO) $(MAINAPP) $(IOTEST) $(GEOMTEST)

$(BUILD)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) $(FLAGS_$*) -c $< -o $@

$(BUILD_AB)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(BUILD_AB)
	$(HIPCC) $(HIPFLAGS) -DSFM_AB=1 $(FLAGS_$*) -c $< -o $@

$(BUILD_AB)/ab_%.o: $(CSRC)/ab/%.hip $(HDRS)
	@mkdir -p $(BUILD_AB)
	$(HIPCC) $(HIPFLAGS) -DSFM_AB=1 -I$(CSRC) -c $< -o $@

# the lane-solve kernels are long chains of scalar FP32 operations: the SLP vectoriser pairs some of them into v_pk_* and pays
# for it with register moves (366 v_mov in 2473 instructions); without it the same arithmetic needs fewer issue slots
FLAGS_ransac := -fno-slp-vectorize
# match_fused's two interleaved exact chains: paired into v_pk_fma_f32 they need a register move per operand
FLAGS_match_fused := -fno-slp-vectorize

$(LIB): $(OBJS) $(CSRC)/exports.map
	@mkdir -p $(PKG)/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--version-script=$(CSRC)/exports.map -o $@ $(OBJS) -lpthread

$(LIB_AB): $(OBJS_AB) $(CSRC)/exports.map
	@mkdir -p $(PKG)/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--version-script=$(CSRC)/exports.map -o $@ $(OBJS_AB) -lpthread

ab: $(LIB_AB)

# the RCCL exchange step lives in its own library: libsfm_amd.so itself has no RCCL dependency
$(COMMLIB): $(CSRC)/comm.cpp include/sfm_amd_comm.h include/sfm_amd.h $(LIB)
	$(HIPCC) -x hip --cuda-host-only -O2 -fPIC -fvisibility=hidden -shared -Iinclude -Wl,--version-script=$(CSRC)/exports.map -o $@ $< -L$(PKG)/lib -lsfm_amd -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

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

# CPU-only: the host-compiled arithmetic check of the non-GPU tests (the fakeccl target, which needs the gfx950 build and the RCCL
# header, is a target of its own and part of `all`)
hostcheck: tests/hostcheck/libhostcheck.so

# TEST HARNESS: comm.cpp linked against a shared-memory stand-in for the nine RCCL calls it makes, so that a 1-GPU box can run the
# exchange code with two real ranks (tests/test_gpu_fakeccl.py); the product's libsfm_amd_rccl.so is linked against librccl
FAKECCL := tests/fake_ccl/libsfm_amd_fakeccl.so
fakeccl: $(FAKECCL)
$(FAKECCL): $(CSRC)/comm.cpp tests/fake_ccl/fake_ccl.cpp include/sfm_amd_comm.h include/sfm_amd.h $(LIB)
	$(HIPCC) -x hip --cuda-host-only -O2 -fPIC -fvisibility=hidden -shared -Iinclude -Wl,--version-script=$(CSRC)/exports.map -o $@ $(CSRC)/comm.cpp tests/fake_ccl/fake_ccl.cpp -L$(PKG)/lib -lsfm_amd -lrt -lpthread -Wl,-rpath,'$$ORIGIN/../../$(PKG)/lib'

tests/hostcheck/libhostcheck.so: tests/hostcheck/hostcheck.hip $(CSRC)/device_math.hpp $(CSRC)/sift_math.hpp $(CSRC)/prefilter_math.hpp $(CSRC)/match_prefilter_math.hpp
	$(HIPCC) -x hip --cuda-host-only -O2 -ffp-contract=off -mfma -fPIC -shared -Wno-pass-failed -o $@ $<

clean:
	rm -rf $(BUILD) $(LIB) $(LIB_AB) $(DEMO) $(HDEMO) $(SDEMO) $(MAINAPP) $(IOTEST) $(GEOMTEST) tests/hostcheck/libhostcheck.so tests/fake_ccl/libsfm_amd_fakeccl.so
	$(MAKE) -C oracle clean

.PHONY: all ab oracle hostcheck fakeccl clean
