# Build of the MI355X path tracer.  `make` builds everything in-tree:
#   hobbyraytracer_amd/lib/libhrt_hip.so    HIP kernels + C ABI (include/hrt.h), gfx950 only
#   hobbyraytracer_amd/lib/libhrt_host.so   host plumbing + C ABI (include/hrt_host.h)
#   hobbyraytracer_amd/bin/hobbyraytracer   the CLI (drop-in for the reference's executable)
#   oracle/liboracle.so                     CPU restatement (test infrastructure only)
# -ffp-contract=off everywhere: CPU oracle and GPU kernels must round identically.

HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
CXXFLAGS := -O2 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function
HIPFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=$(ARCH) -Wno-unused-value

PKG      := hobbyraytracer_amd
HOST_SRC := $(PKG)/host/classes.cpp $(PKG)/host/bvh_build.cpp $(PKG)/host/scene.cpp $(PKG)/host/yaml_lite.cpp \
            $(PKG)/host/image_io.cpp $(PKG)/host/jpeg_lite.cpp $(PKG)/host/assets.cpp $(PKG)/host/host_api.cpp
HOST_HDR := $(wildcard $(PKG)/host/*.h) $(wildcard $(PKG)/csrc/*.h) $(wildcard include/*.h)

all: $(PKG)/lib/libhrt_hip.so $(PKG)/lib/libhrt_host.so $(PKG)/bin/hobbyraytracer oracle/liboracle.so

$(PKG)/lib/libhrt_hip.so: $(PKG)/csrc/hrt_hip.hip $(PKG)/csrc/hrt_lbvh.hip $(PKG)/csrc/hrt_sahbvh.hip $(HOST_HDR)
	@mkdir -p $(PKG)/lib build
	$(HIPCC) $(HIPFLAGS) -c -o build/hrt_hip.o $(PKG)/csrc/hrt_hip.hip
	$(HIPCC) $(HIPFLAGS) -c -o build/hrt_lbvh.o $(PKG)/csrc/hrt_lbvh.hip
	$(HIPCC) $(HIPFLAGS) -c -o build/hrt_sahbvh.o $(PKG)/csrc/hrt_sahbvh.hip
	$(HIPCC) $(HIPFLAGS) -shared -o $@ build/hrt_hip.o build/hrt_lbvh.o build/hrt_sahbvh.o -ldl

$(PKG)/lib/libhrt_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(PKG)/lib
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -lz

$(PKG)/bin/hobbyraytracer: $(PKG)/host/main.cpp $(PKG)/host/render.cpp $(PKG)/lib/libhrt_host.so $(PKG)/lib/libhrt_hip.so $(HOST_HDR)
	@mkdir -p $(PKG)/bin
	$(CXX) $(CXXFLAGS) -o $@ $(PKG)/host/main.cpp $(PKG)/host/render.cpp -L$(PKG)/lib -lhrt_host -lhrt_hip \
	    -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib -lpthread

oracle/liboracle.so: oracle/oracle.cpp $(wildcard $(PKG)/csrc/*.h) include/hrt.h
	$(MAKE) -C oracle

clean:
	rm -rf $(PKG)/lib $(PKG)/bin oracle/liboracle.so oracle/_ref build

.PHONY: all clean
