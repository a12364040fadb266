# Top-level build: everything lands in-tree next to its sources.
#
#   raytracer_amd/csrc/librt_hip.so   the gfx950 device layer + C ABI (include/rt_hip.h)
#   raytracer_amd/host/librt_host.so  the C++ host side (scene build) with C entry points
#   raytracer_amd/host/rt_main        the headless frame-loop executable
#   oracle/                           the checker (see oracle/Makefile)

HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950

CSRC := raytracer_amd/csrc
HOST := raytracer_amd/host

HIP_SOURCES := $(wildcard $(CSRC)/*.hip)
HIP_HEADERS := $(wildcard $(CSRC)/*.h) include/rt_hip.h
HOST_LIB_SOURCES := $(HOST)/model.cpp $(HOST)/scene.cpp $(HOST)/host_capi.cpp
HOST_HEADERS := $(wildcard $(HOST)/*.h) include/rt_hip.h

HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -ffp-contract=off -fno-fast-math -munsafe-fp-atomics \
            -Wall -Wno-unused-function -Iinclude -I$(CSRC)
CXXFLAGS := -O2 -std=c++17 -fPIC -pthread -ffp-contract=off -fno-fast-math -Wall -Wextra -Iinclude -I$(HOST)

all: device host oracle

device: $(CSRC)/librt_hip.so
host: $(HOST)/librt_host.so $(HOST)/rt_main
oracle:
	$(MAKE) -C oracle all

$(CSRC)/librt_hip.so: $(HIP_SOURCES) $(HIP_HEADERS)
	$(HIPCC) $(HIPFLAGS) $(HIP_SOURCES) -o $@

$(HOST)/librt_host.so: $(HOST_LIB_SOURCES) $(HOST_HEADERS)
	$(CXX) $(CXXFLAGS) -shared $(HOST_LIB_SOURCES) -o $@

ROCM ?= /opt/rocm
# tiled.cpp holds the multi-GPU gather: HIP runtime API and RCCL from plain C++ (no device code)
$(HOST)/rt_main: $(HOST)/main.cpp $(HOST)/raytracer.cpp $(HOST)/tiled.cpp $(HOST)/librt_host.so $(CSRC)/librt_hip.so $(HOST_HEADERS)
	$(CXX) $(CXXFLAGS) -D__HIP_PLATFORM_AMD__ -I$(ROCM)/include $(HOST)/main.cpp $(HOST)/raytracer.cpp $(HOST)/tiled.cpp -o $@ \
	    -L$(HOST) -lrt_host -L$(CSRC) -lrt_hip -L$(ROCM)/lib -lrccl -lamdhip64 \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/../csrc' -Wl,-rpath,$(ROCM)/lib

clean:
	rm -f $(CSRC)/librt_hip.so $(HOST)/librt_host.so $(HOST)/rt_main
	$(MAKE) -C oracle clean

.PHONY: all device host oracle clean stamps variant resources

# Diagnostic build with per-phase cycle stamps in the kernels (never loaded by default; RT_HIP_LIB selects it).
stamps: $(CSRC)/librt_hip_stamps1.so $(CSRC)/librt_hip_stamps2.so $(CSRC)/librt_hip_stamps3.so $(CSRC)/librt_hip_stamps4.so
$(CSRC)/librt_hip_stamps%.so: $(HIP_SOURCES) $(HIP_HEADERS)
	$(HIPCC) $(HIPFLAGS) -DRT_STAMPS=$* $(HIP_SOURCES) -o $@

# Experimental build with extra defines: make variant NAME=x DEFS="-DRT_AABB_THREADS=64" -> librt_hip_var_x.so
variant:
	$(HIPCC) $(HIPFLAGS) $(DEFS) $(HIP_SOURCES) -o $(CSRC)/librt_hip_var_$(NAME).so

# Register, scratch and LDS use of every kernel (compiler remarks), e.g. `make resources | less`.
resources:
	@for f in $(HIP_SOURCES); do $(HIPCC) $(filter-out -shared,$(HIPFLAGS)) $(DEFS) --cuda-device-only -c -Rpass-analysis=kernel-resource-usage $$f -o /dev/null 2>&1 | \
	  grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: //' | paste - - - - - - ; done
