# Builds libmcmc_hip.so (gfx950 only) in-tree.  `make` here, or __graft_entry__.build().
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := mcmc_amd/csrc
OUT     := mcmc_amd/lib
OBJ     := build/obj
HIPFLAGS := -O3 --offload-arch=$(ARCH) -fPIC -std=c++17 -Wall -Wno-unused-function -Iinclude -I$(CSRC)

HIP_SRCS := $(wildcard $(CSRC)/*.hip)
CPP_SRCS := $(wildcard $(CSRC)/*.cpp)
OBJS := $(patsubst $(CSRC)/%.hip,$(OBJ)/%.o,$(HIP_SRCS)) $(patsubst $(CSRC)/%.cpp,$(OBJ)/%.o,$(CPP_SRCS))
HDRS := $(wildcard $(CSRC)/*.hpp) include/mcmc_hip.h

all: $(OUT)/libmcmc_hip.so

$(OBJ)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(OUT)/libmcmc_hip.so: $(OBJS)
	@mkdir -p $(OUT)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# Host-sanitizer build: the same sources with AddressSanitizer + UndefinedBehaviorSanitizer on the HOST
# code only (GPU sanitizers are not available on this pool: every -fsanitize= sits behind -Xarch_host) and -DMCMC_DRY_CTX, which
# adds a device-less test context so that the argument validation of every entry point runs on a box without a GPU.
# tests/abi_check.cpp drives it; tests/test_host_sanitizer.py runs `make asan` and the binary.
ASAN_OBJ := build/asan
ASANFLAGS := -O1 -g --offload-arch=$(ARCH) -fPIC -std=c++17 -Wall -Wno-unused-function -Iinclude -I$(CSRC) -DMCMC_DRY_CTX \
             -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined \
             -Xarch_host -fno-omit-frame-pointer
ASAN_OBJS := $(patsubst $(CSRC)/%.hip,$(ASAN_OBJ)/%.o,$(HIP_SRCS)) $(patsubst $(CSRC)/%.cpp,$(ASAN_OBJ)/%.o,$(CPP_SRCS))

$(ASAN_OBJ)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(ASAN_OBJ)
	$(HIPCC) $(ASANFLAGS) -c $< -o $@

$(ASAN_OBJ)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(ASAN_OBJ)
	$(HIPCC) $(ASANFLAGS) -x hip -c $< -o $@

$(ASAN_OBJ)/abi_check.o: tests/abi_check.cpp include/mcmc_hip.h
	@mkdir -p $(ASAN_OBJ)
	$(HIPCC) $(ASANFLAGS) -x hip -c $< -o $@

$(ASAN_OBJ)/abi_check: $(ASAN_OBJS) $(ASAN_OBJ)/abi_check.o
	$(HIPCC) --offload-arch=$(ARCH) -fsanitize=address -fsanitize=undefined -fno-gpu-sanitize -o $@ $^

asan: $(ASAN_OBJ)/abi_check

clean:
	rm -rf build $(OUT)/libmcmc_hip.so

.PHONY: all asan clean
