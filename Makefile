# Builds the shipped library and the command-line renderer for gfx950.
#   make            -> raytracer_amd/libraytracer_amd.so, raytracer_amd/raytracer,
#                      raytracer_amd/libraytracer_amd_probe.so (measurement and test hooks, not needed to render)
#   make oracle     -> the CPU checker used by tests (oracle/)
# No FMA contraction, no fast-math: results must match the reference's x86
# arithmetic bit for bit.

HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
SRC := raytracer_amd/csrc
OUT := raytracer_amd
CXXFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$(SRC) -Wall -Wno-unused-result
HIPFLAGS := --offload-arch=$(ARCH) $(CXXFLAGS)

OBJS := build/kernels.o build/cells.o build/bvh_build.o build/sort_rays.o build/static_trace.o build/upload_device.o build/context.o build/api.o build/host_scene.o build/wide_tree.o
PROBE_OBJS := build/variants.o build/probe_api.o
HDRS := $(wildcard $(SRC)/*.hpp) include/raytracer_amd.h include/raytracer_amd_probe.h include/rt_pmath.h

all: $(OUT)/libraytracer_amd.so $(OUT)/libraytracer_amd_probe.so $(OUT)/raytracer

build/%.o: $(SRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

build/%.o: $(SRC)/%.cpp $(HDRS)
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(OUT)/libraytracer_amd.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(OBJS)

$(OUT)/libraytracer_amd_probe.so: $(PROBE_OBJS) $(OUT)/libraytracer_amd.so
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(PROBE_OBJS) -L$(OUT) -lraytracer_amd -Wl,-rpath,'$$ORIGIN'

$(OUT)/raytracer: $(SRC)/cli_main.cpp $(OUT)/libraytracer_amd.so include/raytracer_amd.h
	$(HIPCC) $(CXXFLAGS) -o $@ $< -L$(OUT) -lraytracer_amd -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(OUT)/libraytracer_amd.so $(OUT)/libraytracer_amd_probe.so $(OUT)/raytracer

.PHONY: all oracle clean
