# Builds the HIP library (gfx950) and the CPU oracle.  hipcc cross-compiles without a GPU.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Iinclude -Ivisual_slam_amd/csrc -Wall -Wno-unused-function
CSRC := visual_slam_amd/csrc
LIB := visual_slam_amd/lib/libvslam_hip.so
OBJS := $(CSRC)/ba.o $(CSRC)/match.o $(CSRC)/capi.o
HDRS := include/vslam_hip.h $(wildcard $(CSRC)/*.h)

all: $(LIB) oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p visual_slam_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -s -C oracle

clean:
	rm -f $(OBJS) $(LIB)
	$(MAKE) -s -C oracle clean

.PHONY: all oracle clean
