# Build recipes for the CMDA gfx950 kernel library, the CPU emulator build of the same
# sources (test infrastructure) -- see __graft_entry__.build().
ROCM      ?= /opt/rocm
HIPCC     ?= $(ROCM)/bin/hipcc
CLANGXX   ?= $(ROCM)/lib/llvm/bin/clang++
CSRC      := cmda_amd/csrc
SRCS      := $(wildcard $(CSRC)/*.hip)
HIP_OBJS  := $(patsubst $(CSRC)/%.hip,build/hip/%.o,$(SRCS))
EMU_OBJS  := $(patsubst $(CSRC)/%.hip,build/emu/%.o,$(SRCS)) build/emu/hip_emu.o
HIPFLAGS  := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$(CSRC) -Iinclude -Wno-unused-result
EMUFLAGS  := -x c++ -std=c++17 -O2 -g -fPIC -DCMDA_EMU -Itests/emu -I$(CSRC) -Iinclude -pthread -ffp-contract=off -Wno-unknown-pragmas -Wno-pass-failed

all: hip emu
hip: cmda_amd/libcmda_hip.so
emu: tests/emu/libcmda_emu.so

HDRS      := $(wildcard $(CSRC)/*.h) include/cmda_hip.h
build/hip/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/hip
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

cmda_amd/libcmda_hip.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(HIP_OBJS)

build/emu/%.o: $(CSRC)/%.hip $(HDRS) tests/emu/hip_emu.h
	@mkdir -p build/emu
	$(CLANGXX) $(EMUFLAGS) -c $< -o $@

build/emu/hip_emu.o: tests/emu/hip_emu.cpp tests/emu/hip_emu.h
	@mkdir -p build/emu
	$(CLANGXX) $(EMUFLAGS) -c $< -o $@

tests/emu/libcmda_emu.so: $(EMU_OBJS)
	$(CLANGXX) -shared -fPIC -pthread -o $@ $(EMU_OBJS)

# tuning build: the same library with the LDS-DMA GEMM's phase stamps compiled in (tools/gemm_phase.py)
build/hip_timing/gemm.o: $(CSRC)/gemm.hip $(wildcard $(CSRC)/gemm_*.hip) $(HDRS)
	@mkdir -p build/hip_timing
	$(HIPCC) $(HIPFLAGS) -DCMDA_GEMM_TIMING -c $< -o $@
timing: build/hip_timing/gemm.o $(HIP_OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o build/libcmda_hip_timing.so build/hip_timing/gemm.o $(filter-out build/hip/gemm.o build/hip/gemm_t%.o build/hip/gemm_reg%.o,$(HIP_OBJS))

# tuning build of the ping-pong GEMM with per-segment s_memtime stamps (tools/dbg/pp_phase.py)
build/hip_pptiming/gemm_pp.o: $(CSRC)/gemm_pp.hip $(HDRS)
	@mkdir -p build/hip_pptiming
	$(HIPCC) $(HIPFLAGS) -DCMDA_PP_TIMING -c $< -o $@
pptiming: build/hip_pptiming/gemm_pp.o $(HIP_OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o build/libcmda_hip_pptiming.so build/hip_pptiming/gemm_pp.o $(filter-out build/hip/gemm_pp.o,$(HIP_OBJS))

build/hip_x3timing/gemm_x3_lean.o: $(CSRC)/gemm_x3_lean.hip $(HDRS)
	@mkdir -p build/hip_x3timing
	$(HIPCC) $(HIPFLAGS) -DCMDA_X3_TIMING -c $< -o $@
x3timing: build/hip_x3timing/gemm_x3_lean.o $(HIP_OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o build/libcmda_hip_x3timing.so build/hip_x3timing/gemm_x3_lean.o $(filter-out build/hip/gemm_x3_lean.o,$(HIP_OBJS))

build/hip_leantiming/gemm_lean.o: $(CSRC)/gemm_lean.hip $(HDRS)
	@mkdir -p build/hip_leantiming
	$(HIPCC) $(HIPFLAGS) -DCMDA_LEAN_TIMING -c $< -o $@
leantiming: build/hip_leantiming/gemm_lean.o $(HIP_OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o build/libcmda_hip_leantiming.so build/hip_leantiming/gemm_lean.o $(filter-out build/hip/gemm_lean.o,$(HIP_OBJS))

clean:
	rm -rf build cmda_amd/libcmda_hip.so tests/emu/libcmda_emu.so
.PHONY: all hip emu timing pptiming x3timing leantiming clean
