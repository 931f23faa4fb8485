# Build of the MI355X engine. `make` = the shipped gfx950 library; `make emu` = test-only emulator build.
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
CXX_EMU ?= $(ROCM)/lib/llvm/bin/clang++
CSRC := piper_amd/csrc
SRCS := $(CSRC)/engine.cpp $(CSRC)/pe_api.cpp $(CSRC)/weights.cpp $(CSRC)/onnx_reader.cpp $(CSRC)/piper_shim.cpp
HDRS := include/piper.hpp $(CSRC)/engine.h $(wildcard $(CSRC)/kernels/*.h) $(CSRC)/pe_rt.h $(CSRC)/weights.h include/piper_hip.h
LIB := piper_amd/libpiper_hip.so
EMULIB := tests/emu/libpiper_hip_emu.so

all: $(LIB)

$(LIB): $(SRCS) $(HDRS)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -x hip $(SRCS) -o $@ -Wno-unused-result -Wno-unused-value

emu: $(EMULIB)

# tuning build: the same library with phase timestamps in the small kernels (scripts/stamps.py); never shipped
stamps: piper_amd/libpiper_hip_stamps.so
piper_amd/libpiper_hip_stamps.so: $(SRCS) $(HDRS)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPE_STAMPS -x hip $(SRCS) -o $@ -Wno-unused-result -Wno-unused-value

$(EMULIB): $(SRCS) $(HDRS) tests/emu/hip_emu.cpp tests/emu/hip_emu.h
	$(CXX_EMU) -DPE_EMU -O2 -g -std=c++17 -Wno-psabi -fPIC -shared -Itests/emu $(SRCS) tests/emu/hip_emu.cpp -o $@

# C++ callers of the piper:: API (mirror of the reference's test.cpp)
tests/cpp/test_piper: tests/cpp/test_piper.cpp $(LIB) include/piper.hpp
	g++ -O1 -std=c++17 -Iinclude tests/cpp/test_piper.cpp -o $@ -Lpiper_amd -lpiper_hip -Wl,-rpath,'$$ORIGIN/../../piper_amd'
tests/cpp/test_piper_emu: tests/cpp/test_piper.cpp $(EMULIB) include/piper.hpp
	g++ -O1 -std=c++17 -Iinclude tests/cpp/test_piper.cpp -o $@ -Ltests/emu -lpiper_hip_emu -Wl,-rpath,'$$ORIGIN/../emu'

clean:
	rm -f $(LIB) $(EMULIB) tests/cpp/test_piper tests/cpp/test_piper_emu
.PHONY: all emu stamps clean
