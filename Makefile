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

clean:
	rm -rf build $(OUT)/libmcmc_hip.so

.PHONY: all clean
