# Builds the HIP library (gfx950) and the CPU oracle.  hipcc cross-compiles without a GPU.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Iinclude -Ivisual_slam_amd/csrc -Wall -Wno-unused-function
CSRC := visual_slam_amd/csrc
LIB := visual_slam_amd/lib/libvslam_hip.so
OBJS := $(CSRC)/ba.o $(CSRC)/match.o $(CSRC)/track.o $(CSRC)/comm.o $(CSRC)/capi.o
HDRS := include/vslam_hip.h $(wildcard $(CSRC)/*.h)

HOST := visual_slam_amd/host
HOSTLIB := visual_slam_amd/lib/libvslam_host.so
HARNESS := tests/cpp/adapter_harness

all: $(LIB) $(HOSTLIB) $(HARNESS) oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p visual_slam_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

# C++ host adapters keeping the reference's class interfaces, above the C ABI
$(HOSTLIB): $(HOST)/optimizer.cc $(HOST)/superpoint_matcher.cc $(HOST)/*.h include/vslam_hip.h $(LIB)
	g++ -O2 -std=c++14 -fPIC -shared -Wall -Iinclude -I$(HOST) -o $@ $(HOST)/optimizer.cc $(HOST)/superpoint_matcher.cc -Lvisual_slam_amd/lib -lvslam_hip -Wl,-rpath,'$$ORIGIN'

$(HARNESS): tests/cpp/adapter_harness.cc $(HOSTLIB)
	g++ -O2 -std=c++14 -Wall -Iinclude -I$(HOST) -o $@ tests/cpp/adapter_harness.cc -Lvisual_slam_amd/lib -lvslam_host -lvslam_hip -Wl,-rpath,'$$ORIGIN/../../visual_slam_amd/lib'

oracle:
	$(MAKE) -s -C oracle

clean:
	rm -f $(OBJS) $(LIB) $(HOSTLIB) $(HARNESS)
	$(MAKE) -s -C oracle clean

.PHONY: all oracle clean
