# Build of the MI355X engine. `make` = the shipped gfx950 library; `make emu` = test-only emulator build.
# One object per translation unit (the kernels are split over kernels/launch_*.cpp), so `make -j` builds them in parallel.
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
CXX_EMU ?= $(ROCM)/lib/llvm/bin/clang++
CSRC := piper_amd/csrc
SRCS := $(CSRC)/engine.cpp $(CSRC)/engine_pack.cpp $(CSRC)/engine_launch.cpp $(CSRC)/engine_issue.cpp $(CSRC)/kernels/launch_conv.cpp $(CSRC)/kernels/launch_bf3.cpp $(CSRC)/kernels/launch_front.cpp $(CSRC)/kernels/launch_tail.cpp \
        $(CSRC)/pe_api.cpp $(CSRC)/policy.cpp $(CSRC)/weights.cpp $(CSRC)/onnx_reader.cpp $(CSRC)/piper_shim.cpp
HDRS := include/piper.hpp $(CSRC)/engine.h $(CSRC)/engine_internal.h $(wildcard $(CSRC)/kernels/*.h) $(CSRC)/pe_rt.h $(CSRC)/policy.h $(CSRC)/weights.h \
        $(CSRC)/unicode_tables.h include/piper_hip.h
LIB := piper_amd/libpiper_hip.so
EMULIB := tests/emu/libpiper_hip_emu.so
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result -Wno-unused-value
EMUFLAGS := -DPE_EMU -O2 -mfma -std=c++17 -Wno-psabi -fPIC -Itests/emu
OBJ := $(patsubst $(CSRC)/%.cpp,build/gfx950/%.o,$(SRCS))
OBJ_STAMPS := $(patsubst $(CSRC)/%.cpp,build/stamps/%.o,$(SRCS))
OBJ_EMU := $(patsubst $(CSRC)/%.cpp,build/emu/%.o,$(SRCS)) build/emu/hip_emu.o

all: $(LIB)

build/gfx950/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIB): $(OBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(OBJ) -o $@

emu: $(EMULIB)
build/emu/%.o: $(CSRC)/%.cpp $(HDRS) tests/emu/hip_emu.h
	@mkdir -p $(dir $@)
	$(CXX_EMU) $(EMUFLAGS) -c $< -o $@
build/emu/hip_emu.o: tests/emu/hip_emu.cpp tests/emu/hip_emu.h
	@mkdir -p $(dir $@)
	$(CXX_EMU) $(EMUFLAGS) -c $< -o $@
$(EMULIB): $(OBJ_EMU)
	$(CXX_EMU) -shared -fPIC $(OBJ_EMU) -o $@

# tuning build: the same library with phase timestamps in the small kernels (scripts/stamps.py); never shipped
stamps: piper_amd/libpiper_hip_stamps.so
build/stamps/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -DPE_STAMPS -c $< -o $@
piper_amd/libpiper_hip_stamps.so: $(OBJ_STAMPS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(OBJ_STAMPS) -o $@

# C++ callers of the piper:: API (mirror of the reference's test.cpp)
tests/cpp/test_piper: tests/cpp/test_piper.cpp $(LIB) include/piper.hpp
	g++ -O1 -std=c++17 -Iinclude tests/cpp/test_piper.cpp -o $@ -Lpiper_amd -lpiper_hip -Wl,-rpath,'$$ORIGIN/../../piper_amd'
tests/cpp/test_piper_emu: tests/cpp/test_piper.cpp $(EMULIB) include/piper.hpp
	g++ -O1 -std=c++17 -Iinclude tests/cpp/test_piper.cpp -o $@ -Ltests/emu -lpiper_hip_emu -Wl,-rpath,'$$ORIGIN/../emu'

clean:
	rm -rf build $(LIB) $(EMULIB) tests/cpp/test_piper tests/cpp/test_piper_emu
.PHONY: all emu stamps clean
