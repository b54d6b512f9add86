# Makefile -- the same builds __graft_entry__.build() drives from Python, for C++ integrators.
#   make lib         beamforming-lk_amd/libawpu_hip.so   (hipcc, gfx950 only; there is no CPU fallback)
#   make oracle      oracle/liboracle_das.so (+ oracle/_ref from the reference tree when it is present)
#   make example     examples/heatmap_min: the C ABI from plain C (needs an MI355X to do more than report its absence)
#   make host-test   tests/host/test_mimo_worker: the C++ mirror (MIMOWorkerHip, AWProcessingUnitHip,
#                    PipelineHip) against the oracle; needs an MI355X to run (--nogpu checks the failure path);
#                    tests/host/test_exact_signatures: class AWProcessingUnit with the reference's own signatures
#                    (-DAWPU_WITH_OPENCV; cv::Mat from OPENCV_CFLAGS)
HIPCC ?= /opt/rocm/bin/hipcc
PKG := beamforming-lk_amd
CSRC := $(PKG)/csrc
LIB := $(PKG)/libawpu_hip.so
KERNEL_SRC := $(CSRC)/das_kernels.hip $(CSRC)/das_fast.hip $(CSRC)/awpu_hip.cpp $(CSRC)/geometry_host.cpp
HOST_SRC := $(PKG)/host/mimo_worker_hip.cpp $(PKG)/host/aw_processing_unit_hip.cpp $(PKG)/host/pipeline_hip.cpp \
            $(PKG)/host/aw_processing_unit.cpp
# OPENCV_CFLAGS: where <opencv2/core.hpp> lives; defaults to the tests' few-line stand-in for cv::Mat (no OpenCV here)
OPENCV_CFLAGS ?= -Itests/host/mock_opencv

.PHONY: lib oracle host-test example trips clean
lib: $(LIB)

$(LIB): $(KERNEL_SRC) $(CSRC)/das_kernels.h $(CSRC)/das_fast_trip.inc include/awpu_hip.h
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-result -Werror=inline-asm -x hip \
	    -Iinclude -I$(CSRC) $(KERNEL_SRC) -o $@

# the hand-scheduled inner loops are GENERATED at build time (tools/gen_trip_asm.py documents the schedule and its knobs;
# the include is not tracked)
$(CSRC)/das_fast_trip.inc: tools/gen_trip_asm.py
	python3 tools/gen_trip_asm.py

trips: $(CSRC)/das_fast_trip.inc

oracle:
	$(MAKE) -C oracle

host-test: $(LIB) oracle
	g++ -O2 -std=c++17 -pthread -Iinclude -I$(PKG)/host -Ioracle tests/host/test_mimo_worker.cpp $(HOST_SRC) \
	    -L$(PKG) -lawpu_hip -Loracle -loracle_das -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -Wl,-rpath,'$$ORIGIN/../../oracle' -Wl,-rpath,/opt/rocm/lib \
	    -o tests/host/test_mimo_worker
	g++ -O2 -std=c++17 -pthread -DAWPU_WITH_OPENCV $(OPENCV_CFLAGS) -Iinclude -I$(PKG)/host -Ioracle tests/host/test_exact_signatures.cpp $(HOST_SRC) \
	    -L$(PKG) -lawpu_hip -Loracle -loracle_das -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -Wl,-rpath,'$$ORIGIN/../../oracle' -Wl,-rpath,/opt/rocm/lib \
	    -o tests/host/test_exact_signatures

example: $(LIB)
	gcc -O2 -Wall -Iinclude examples/heatmap_min.c -L$(PKG) -lawpu_hip -lm -Wl,-rpath,'$$ORIGIN/../$(PKG)' \
	    -Wl,-rpath,/opt/rocm/lib -o examples/heatmap_min
	gcc -O2 -Wall -Iinclude examples/live_call_rate.c -L$(PKG) -lawpu_hip -lm -Wl,-rpath,'$$ORIGIN/../$(PKG)' \
	    -Wl,-rpath,/opt/rocm/lib -o examples/live_call_rate

clean:
	rm -f $(LIB) tests/host/test_mimo_worker tests/host/test_exact_signatures examples/heatmap_min examples/live_call_rate
	$(MAKE) -C oracle clean
