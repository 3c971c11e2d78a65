# Builds the three shared libraries (and two small programs, raytracer_amd/div_check.bin and fetch_calib.bin) in-tree:
#   raytracer_amd/librtgpu.so   HIP render path behind include/rtgpu.h (gfx950)
#   raytracer_amd/librthost.so  C++ mirror of the reference's Scene API (host only)
#   oracle/liboracle.so         CPU checker (test infrastructure, never linked by the two above)
HIPCC   ?= hipcc
CXX     ?= g++
ARCH    ?= gfx950

GPU_SRC  := raytracer_amd/csrc/rtgpu.hip
GPU_HDR  := raytracer_amd/csrc/kernels.h raytracer_amd/csrc/dev_packet.h raytracer_amd/csrc/dev_rt.h raytracer_amd/csrc/dev_types.h raytracer_amd/csrc/bvh_build.h include/rtgpu.h
BVH_SRC  := raytracer_amd/csrc/bvh_build.hip
BVH_HDR  := raytracer_amd/csrc/bvh_build.h raytracer_amd/csrc/dev_types.h
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off
HOST_SRC := raytracer_amd/host/mesh.cpp raytracer_amd/host/scene.cpp raytracer_amd/host/png.cpp raytracer_amd/host/jpeg.cpp raytracer_amd/host/sdl.cpp raytracer_amd/host/capi.cpp
HOST_HDR := raytracer_amd/host/scene.hpp raytracer_amd/host/sdl.hpp raytracer_amd/host/rmath.hpp include/rtgpu.h

all: raytracer_amd/librtgpu.so raytracer_amd/librthost.so oracle/liboracle.so raytracer_amd/div_check.bin raytracer_amd/fetch_calib.bin

# -ffp-contract=off: the reference never fuses a multiply-add; neither may the device code
# two translation units (the tree builder pulls in the rocPRIM sort and compiles slowly; no device code calls across them)
build/obj/rtgpu.o: $(GPU_SRC) $(GPU_HDR)
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c -o $@ $(GPU_SRC)

build/obj/bvh_build.o: $(BVH_SRC) $(BVH_HDR)
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c -o $@ $(BVH_SRC)

raytracer_amd/librtgpu.so: build/obj/rtgpu.o build/obj/bvh_build.o
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ build/obj/rtgpu.o build/obj/bvh_build.o

raytracer_amd/librthost.so: $(HOST_SRC) $(HOST_HDR) raytracer_amd/librtgpu.so
	$(CXX) -O2 -std=c++17 -fPIC -shared -pthread -ffp-contract=off -o $@ $(HOST_SRC) -Lraytracer_amd -lrtgpu -lz -Wl,-rpath,'$$ORIGIN'

# a self-check of one device function (dev_rt.h div_by_norm against the plain divisions, bit for bit); the GPU tests run it
raytracer_amd/div_check.bin: tools/div_check.hip raytracer_amd/csrc/dev_rt.h raytracer_amd/csrc/dev_types.h
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -ffp-contract=off -Iinclude -o $@ tools/div_check.hip

# a measurement program, not part of the product: what the FETCH_SIZE counter reports for known byte counts (tools/fetch_calib_report.py)
raytracer_amd/fetch_calib.bin: tools/fetch_calib.hip
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -o $@ tools/fetch_calib.hip

oracle/liboracle.so: oracle/oracle.cpp oracle/oracle.h include/rtgpu.h
	$(CXX) -O2 -std=c++17 -fPIC -shared -pthread -ffp-contract=off -fno-fast-math -o $@ oracle/oracle.cpp

clean:
	rm -f raytracer_amd/librtgpu.so raytracer_amd/librthost.so oracle/liboracle.so raytracer_amd/div_check.bin raytracer_amd/fetch_calib.bin

.PHONY: all clean
