# Builds the three shared libraries in-tree:
#   raytracer_amd/librtgpu.so   HIP render path behind include/rtgpu.h (gfx950)
#   raytracer_amd/librthost.so  C++ mirror of the reference's Scene API (host only)
#   oracle/liboracle.so         CPU checker (test infrastructure, never linked by the two above)
HIPCC   ?= hipcc
CXX     ?= g++
ARCH    ?= gfx950

GPU_SRC  := raytracer_amd/csrc/rtgpu.hip
GPU_HDR  := raytracer_amd/csrc/kernels.h raytracer_amd/csrc/dev_packet.h raytracer_amd/csrc/dev_rt.h raytracer_amd/csrc/dev_types.h include/rtgpu.h
HOST_SRC := raytracer_amd/host/mesh.cpp raytracer_amd/host/scene.cpp raytracer_amd/host/png.cpp raytracer_amd/host/jpeg.cpp raytracer_amd/host/sdl.cpp raytracer_amd/host/capi.cpp
HOST_HDR := raytracer_amd/host/scene.hpp raytracer_amd/host/sdl.hpp raytracer_amd/host/rmath.hpp include/rtgpu.h

all: raytracer_amd/librtgpu.so raytracer_amd/librthost.so oracle/liboracle.so

# -ffp-contract=off: the reference never fuses a multiply-add; neither may the device code
raytracer_amd/librtgpu.so: $(GPU_SRC) $(GPU_HDR)
	$(HIPCC) -O3 -std=c++17 -fPIC -shared --offload-arch=$(ARCH) -ffp-contract=off -o $@ $(GPU_SRC)

raytracer_amd/librthost.so: $(HOST_SRC) $(HOST_HDR) raytracer_amd/librtgpu.so
	$(CXX) -O2 -std=c++17 -fPIC -shared -pthread -ffp-contract=off -o $@ $(HOST_SRC) -Lraytracer_amd -lrtgpu -lz -Wl,-rpath,'$$ORIGIN'

oracle/liboracle.so: oracle/oracle.cpp oracle/oracle.h include/rtgpu.h
	$(CXX) -O2 -std=c++17 -fPIC -shared -pthread -ffp-contract=off -fno-fast-math -o $@ oracle/oracle.cpp

clean:
	rm -f raytracer_amd/librtgpu.so raytracer_amd/librthost.so oracle/liboracle.so

.PHONY: all clean
