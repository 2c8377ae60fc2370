# Build the C-ABI libraries.
#   make hip   -> star_amd/libstar_hip.so       (gfx950, the product)
#   make emu   -> tools/hostemu/libstar_emu.so   (SIMT emulator build, tests only)
#   make oracle-> oracle C helpers (none yet)
CSRC := star_amd/csrc
SRCS := $(wildcard $(CSRC)/*.cpp)
HIPCC ?= hipcc
CLANGXX ?= /opt/rocm/lib/llvm/bin/clang++
HIP_OBJS := $(patsubst $(CSRC)/%.cpp,build/hip/%.o,$(SRCS))
EMU_OBJS := $(patsubst $(CSRC)/%.cpp,build/emu/%.o,$(SRCS)) build/emu/hostemu.o
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-value -Wno-inline-asm -ffp-contract=fast
EMUFLAGS := -O2 -g -std=c++17 -fPIC -DSTAR_HOSTEMU=1 -DSTAR_BENCH_VARIANTS=1 -mavx2 -mf16c -mfma -Wno-unused-value -Wno-psabi -Wno-psabi

all: hip emu
hip: star_amd/libstar_hip.so
emu: tools/hostemu/libstar_emu.so

# attention kernels: no NaN-canonicalising v_max in front of every fmaxf on MFMA outputs (39 extra VALU per key tile);
# masked scores are finite (-1e30 / -30000), so NaNs can only come from NaN inputs and propagate either way
build/hip/attn.o: HIPFLAGS += -fno-honor-nans -fno-slp-vectorize
build/hip/attn7.o: HIPFLAGS += -fno-honor-nans -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form
build/hip/gemm_as.o: HIPFLAGS += -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form
build/hip/gemm_tq.o: HIPFLAGS += -fno-honor-nans -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form
# precise header dependencies (-MMD): touching norm.h no longer recompiles the GEMM instantiation units (minutes each).
# An object WITHOUT its .d file (a build tree from before -MMD, or a .d deleted by hand) has unknown header dependencies: it
# depends on every header until its .d exists again (secondary expansion: the prerequisite list is computed per target), so a
# struct layout change in ctx.h / gemm.h can never be linked against a stale object.
HDRS := $(wildcard $(CSRC)/*.h $(CSRC)/*.inc) include/star_hip.h
.SECONDEXPANSION:
build/hip/%.o: $(CSRC)/%.cpp $$(if $$(wildcard build/hip/$$*.d),,$$(HDRS))
	@mkdir -p build/hip
	$(HIPCC) $(HIPFLAGS) -MMD -MP -c $< -o $@
star_amd/libstar_hip.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $^ -o $@

build/emu/%.o: $(CSRC)/%.cpp $$(if $$(wildcard build/emu/$$*.d),,$$(HDRS))
	@mkdir -p build/emu
	$(CLANGXX) $(EMUFLAGS) -MMD -MP -c $< -o $@
build/emu/hostemu.o: tools/hostemu/hostemu.cpp $(CSRC)/hostemu.h
	@mkdir -p build/emu
	$(CLANGXX) $(EMUFLAGS) -c $< -o $@
tools/hostemu/libstar_emu.so: $(EMU_OBJS)
	$(CLANGXX) -shared -fPIC $^ -o $@ -lm -lpthread

# bench-only build: the product sources + timing ablations / losing A/B variants (-DSTAR_BENCH_VARIANTS); loaded explicitly by tools/
BENCH_OBJS := $(patsubst $(CSRC)/%.cpp,build/bench/%.o,$(SRCS))
bench: tools/bench/libstar_hip_bench.so
build/bench/attn.o: HIPFLAGS += -fno-honor-nans -fno-slp-vectorize
build/bench/attn7.o: HIPFLAGS += -fno-honor-nans -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form
build/bench/gemm_as.o: HIPFLAGS += -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form
build/bench/gemm_tq.o: HIPFLAGS += -fno-honor-nans -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form
build/bench/%.o: $(CSRC)/%.cpp $$(if $$(wildcard build/bench/$$*.d),,$$(HDRS))
	@mkdir -p build/bench
	$(HIPCC) $(HIPFLAGS) -DSTAR_BENCH_VARIANTS=1 -MMD -MP -c $< -o $@
tools/bench/libstar_hip_bench.so: $(BENCH_OBJS)
	@mkdir -p tools/bench
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $^ -o $@

# torch-free timing harness over the C ABI (dlopen()s a build of the library): seconds instead of minutes per A/B on a fresh GPU box
cbench: tools/cbench/cbench tools/cbench/cbench_emu tools/probe/mfma_valu_overlap
tools/probe/mfma_valu_overlap: tools/probe/mfma_valu_overlap.hip
	$(HIPCC) -O2 --offload-arch=gfx950 $< -o $@
tools/cbench/cbench: tools/cbench/cbench.cpp include/star_hip.h
	$(HIPCC) -O2 --offload-arch=gfx950 $< -o $@ -ldl
tools/cbench/cbench_emu: tools/cbench/cbench.cpp include/star_hip.h
	$(CXX) -O2 -DCBENCH_EMU $< -o $@ -ldl

-include $(wildcard build/hip/*.d build/emu/*.d build/bench/*.d)

clean:
	rm -rf build star_amd/libstar_hip.so tools/hostemu/libstar_emu.so tools/bench tools/cbench/cbench tools/cbench/cbench_emu
.PHONY: all hip emu bench cbench clean
