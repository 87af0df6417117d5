# Builds libcat_amd.so (hand-written HIP for gfx950 + the C ABI) and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
# gfx950 (MI355X) is the only supported target: the kernels' tiles assume its 160 KB of LDS per CU (ntt.hip refuses others)
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Iinclude
SRC := cat_amd/csrc/capi.cpp cat_amd/csrc/comm.cpp cat_amd/csrc/ntt.hip cat_amd/csrc/polyops.hip cat_amd/csrc/keyswitch.hip cat_amd/csrc/bfv.hip
HDR := include/cat_amd.h cat_amd/csrc/context.hpp cat_amd/csrc/host_math.hpp cat_amd/csrc/modarith.cuh cat_amd/csrc/rows8.cuh
OBJ := $(patsubst cat_amd/csrc/%,build/%.o,$(SRC))
LIB := cat_amd/lib/libcat_amd.so

all: $(LIB) oracle facade_test random_test modulus_test latency_facade

$(LIB): $(OBJ)
	@mkdir -p cat_amd/lib
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJ) -ldl

build/%.hip.o: cat_amd/csrc/%.hip $(HDR)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

build/%.cpp.o: cat_amd/csrc/%.cpp $(HDR)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

oracle:
	$(MAKE) -s -C oracle

# C++ facade parity test (needs a GPU to run; plain g++: the facade only sees the C ABI)
facade_test: tests/cpp/facade_test
tests/cpp/facade_test: tests/cpp/test_facade.cpp cat_amd/host/seal_hip.hpp cat_amd/host/seal_hip_random.hpp cat_amd/host/seal_hip_crypto.hpp include/cat_amd.h $(LIB) oracle
	g++ -std=c++17 -O2 -Wall $< -o $@ -Lcat_amd/lib -lcat_amd -Loracle -lseal_oracle \
	    -Wl,-rpath,'$$ORIGIN/../../cat_amd/lib' -Wl,-rpath,'$$ORIGIN/../../oracle' -Wl,-rpath-link,/opt/rocm/lib

# host-only check of the facade's PRNG and samplers against the oracle (and oracle/_ref at run time)
random_test: tests/cpp/random_test
tests/cpp/random_test: tests/cpp/test_random.cpp cat_amd/host/seal_hip_random.hpp oracle
	g++ -std=c++17 -O2 -Wall $< -o $@ -Loracle -lseal_oracle -ldl -Wl,-rpath,'$$ORIGIN/../../oracle'

modulus_test: tests/cpp/modulus_test
tests/cpp/modulus_test: tests/cpp/test_modulus.cpp cat_amd/host/seal_hip_modulus.hpp
	g++ -std=c++17 -O2 -Wall $< -o $@

# single-ciphertext latencies through seal::Evaluator at the reference's published configuration (needs a GPU to run)
latency_facade: tools/latency_facade
tools/latency_facade: tools/latency_facade.cpp cat_amd/host/seal_hip.hpp cat_amd/host/seal_hip_crypto.hpp include/cat_amd.h $(LIB)
	g++ -std=c++17 -O2 -Wall $< -o $@ -Lcat_amd/lib -lcat_amd -Wl,-rpath,'$$ORIGIN/../cat_amd/lib' -Wl,-rpath-link,/opt/rocm/lib

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean facade_test random_test modulus_test latency_facade
