# Build the MI355X (gfx950) C-ABI library in-tree; `make emu` builds the CPU SIMT emulation of the same sources
# (test infrastructure only, see tests/emu/README.md).
HIPCC ?= /opt/rocm/bin/hipcc
HOSTCXX ?= /opt/rocm/lib/llvm/bin/clang++
CSRC := diffusestylegesture_amd/csrc
# one tag for the library and the bare code object: dsg_aql.h refuses a dsg_kernels.hsaco built from other sources
TAG := $(shell cat $(CSRC)/dsg_hip.cpp $(CSRC)/dsg_kernels.h $(CSRC)/dsg_fused.h $(CSRC)/dsg_batched.h $(CSRC)/dsg_stream.h $(CSRC)/dsg_aql.h include/dsg.h | cksum | cut -d' ' -f1)
LIB := $(CSRC)/libdsg_hip.so
EMU := tests/emu/_build/libdsg_emu.so

all: $(LIB) $(CSRC)/dsg_kernels.hsaco

# dsg_bvh.cpp is plain host C++ (the BVH post-processing behind the same C ABI), compiled along
$(LIB): $(CSRC)/dsg_hip.cpp $(CSRC)/dsg_kernels.h $(CSRC)/dsg_fused.h $(CSRC)/dsg_batched.h $(CSRC)/dsg_stream.h $(CSRC)/dsg_aql.h include/dsg.h $(CSRC)/dsg_bvh.cpp
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -DDSG_BUILD_TAG=$(TAG)u $(CSRC)/dsg_hip.cpp $(CSRC)/dsg_bvh.cpp -L/opt/rocm/lib -lhsa-runtime64 -lpthread -o $@

# the device side of the same translation unit as a bare code object: loaded through the HSA loader by the AQL
# submission path (dsg_aql.h), which needs kernel descriptors the HIP runtime does not hand out
$(CSRC)/dsg_kernels.hsaco: $(CSRC)/dsg_hip.cpp $(CSRC)/dsg_kernels.h $(CSRC)/dsg_fused.h $(CSRC)/dsg_batched.h $(CSRC)/dsg_stream.h $(CSRC)/dsg_aql.h include/dsg.h
	$(HIPCC) --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 -std=c++17 -Wno-pass-failed -DDSG_BUILD_TAG=$(TAG)u \
	    -Rpass-analysis=kernel-resource-usage $(CSRC)/dsg_hip.cpp -o $@ 2> $(CSRC)/dsg_kernels.resources.txt || (cat $(CSRC)/dsg_kernels.resources.txt >&2; exit 1)

# timeline build: the same sources with first-wave-start / last-wave-end stamps in every step kernel (dsg_kernels.h: TlScope), as a
# library + its OWN code object with the same tag, so that tools/aql_timeline.py traces the fence-free AQL path itself
STAMPS_SRC := $(CSRC)/dsg_hip.cpp $(CSRC)/dsg_kernels.h $(CSRC)/dsg_fused.h $(CSRC)/dsg_batched.h $(CSRC)/dsg_stream.h $(CSRC)/dsg_aql.h include/dsg.h
stamps: $(CSRC)/libdsg_hip_stamps.so $(CSRC)/dsg_kernels_stamps.hsaco
$(CSRC)/libdsg_hip_stamps.so: $(STAMPS_SRC) $(CSRC)/dsg_bvh.cpp
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -DDSG_STAMPS=1 -DDSG_BUILD_TAG=$(TAG)u $(CSRC)/dsg_hip.cpp $(CSRC)/dsg_bvh.cpp -L/opt/rocm/lib -lhsa-runtime64 -lpthread -o $@
$(CSRC)/dsg_kernels_stamps.hsaco: $(STAMPS_SRC)
	$(HIPCC) --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 -std=c++17 -Wno-pass-failed -DDSG_STAMPS=1 -DDSG_BUILD_TAG=$(TAG)u $(CSRC)/dsg_hip.cpp -o $@

# marks build: the stamps build + DSG_TL_MARK phase marks inside the kernels (tools/aql_timeline.py --lib marks)
marks: $(CSRC)/libdsg_hip_marks.so $(CSRC)/dsg_kernels_marks.hsaco
$(CSRC)/libdsg_hip_marks.so: $(STAMPS_SRC) $(CSRC)/dsg_bvh.cpp
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -DDSG_STAMPS=2 -DDSG_BUILD_TAG=$(TAG)u $(CSRC)/dsg_hip.cpp $(CSRC)/dsg_bvh.cpp -L/opt/rocm/lib -lhsa-runtime64 -lpthread -o $@
$(CSRC)/dsg_kernels_marks.hsaco: $(STAMPS_SRC)
	$(HIPCC) --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 -std=c++17 -Wno-pass-failed -DDSG_STAMPS=2 -DDSG_BUILD_TAG=$(TAG)u $(CSRC)/dsg_hip.cpp -o $@

# development build: bf16 only (a third of the instantiations, a third of the compile time), under its OWN file names so that it can never be
# mistaken for the product library:   make dev [DEVNAME=devA] [DEVFLAGS="-DDSG_STAMPS=2 -DDSG_X_..."]   ->   libdsg_hip_$(DEVNAME).so + dsg_kernels_$(DEVNAME).hsaco
# (DSG_LIB=.../libdsg_hip_devA.so python tools/...; two variants built under two names are A/B-ed on ONE box in one gpurun call: tools/ab_dev.sh)
DEVFLAGS ?=
DEVNAME ?= dev
DEVDEF := -DDSG_DEV_BF16_ONLY=1 $(DEVFLAGS) '-DDSG_HSACO_NAME="dsg_kernels_$(DEVNAME).hsaco"' -DDSG_BUILD_TAG=$(TAG)u
dev: $(CSRC)/libdsg_hip_$(DEVNAME).so $(CSRC)/dsg_kernels_$(DEVNAME).hsaco
$(CSRC)/libdsg_hip_$(DEVNAME).so: $(STAMPS_SRC) $(CSRC)/dsg_bvh.cpp
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed $(DEVDEF) $(CSRC)/dsg_hip.cpp $(CSRC)/dsg_bvh.cpp -L/opt/rocm/lib -lhsa-runtime64 -lpthread -o $@
$(CSRC)/dsg_kernels_$(DEVNAME).hsaco: $(STAMPS_SRC)
	$(HIPCC) --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 -std=c++17 -Wno-pass-failed $(DEVDEF) \
	    -Rpass-analysis=kernel-resource-usage $(CSRC)/dsg_hip.cpp -o $@ 2> $(CSRC)/dsg_kernels_$(DEVNAME).resources.txt || (cat $(CSRC)/dsg_kernels_$(DEVNAME).resources.txt >&2; exit 1)

# the kernels that were measured slower and removed from the library: compile check only (experiments/, outside the package)
experiments: experiments/experiments.hip experiments/dsg_rejected_kernels.h experiments/dsg_stream_ln.h $(CSRC)/dsg_kernels.h $(CSRC)/dsg_fused.h $(CSRC)/dsg_batched.h $(CSRC)/dsg_stream.h
	$(HIPCC) --offload-arch=gfx950 --cuda-device-only -O3 -std=c++17 -Wno-pass-failed -I$(CSRC) -c experiments/experiments.hip -o /dev/null

emu: $(EMU)
$(EMU): $(CSRC)/dsg_hip.cpp $(CSRC)/dsg_kernels.h $(CSRC)/dsg_fused.h $(CSRC)/dsg_batched.h $(CSRC)/dsg_stream.h include/dsg.h tests/emu/shim/hip/hip_runtime.h tests/emu/emu_rt.cpp $(CSRC)/dsg_bvh.cpp
	mkdir -p tests/emu/_build
	$(HOSTCXX) -O2 -g -std=c++17 -fPIC -shared -pthread -Itests/emu/shim -DDSG_EMU=1 \
	    $(CSRC)/dsg_hip.cpp tests/emu/emu_rt.cpp $(CSRC)/dsg_bvh.cpp -o $@

# micro-probes behind the numbers in DESIGN.md s5 (tools/*.cpp; run on the GPU box, logs under profiles/)
PROBES := xcd_probe persist_probe dep_probe icache_probe loadpath_probe layer_probe noise_probe
tools: $(addprefix tools/_build/,$(PROBES)) tools/_build/aql_probe tools/_build/aql_kernels.hsaco
tools/_build/%: tools/%.cpp
	mkdir -p tools/_build
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 $< -o $@
tools/_build/aql_kernels.hsaco: tools/aql_kernels.hip
	mkdir -p tools/_build
	$(HIPCC) --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 $< -o $@
tools/_build/aql_probe: tools/aql_probe.cpp
	mkdir -p tools/_build
	g++ -O2 -std=c++17 -I/opt/rocm/include $< -L/opt/rocm/lib -lhsa-runtime64 -Wl,-rpath,/opt/rocm/lib -o $@

clean:
	rm -f $(LIB) $(EMU) $(CSRC)/dsg_kernels.hsaco $(CSRC)/dsg_kernels.resources.txt
.PHONY: all emu stamps marks dev experiments tools clean
