# Builds everything in-tree:
#   yadcc_amd/libydc.so        HIP kernels + C-ABI (+ host C++ dispatcher), gfx950
#   oracle/liboracle.so, oracle/_ref/libyadcc_ref.so   (test infrastructure)
#   tests/model/libmodel.so    (test tool)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := yadcc_amd/csrc
HIP_SRCS := $(CSRC)/ydc_api.hip $(wildcard $(CSRC)/*.cc)
HDRS := $(wildcard $(CSRC)/*.h) include/yadcc_dispatch.h

all: lib oracle model native

lib: yadcc_amd/libydc.so
yadcc_amd/libydc.so: $(HIP_SRCS) $(HDRS)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
	    -Iinclude -I$(CSRC) -o $@ $(HIP_SRCS)

# Measurement build: the matching kernel leaves phase stamps behind (tools/phase_probe.py).
probe: yadcc_amd/libydc_probe.so
yadcc_amd/libydc_probe.so: $(HIP_SRCS) $(HDRS)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
	    -DYDC_PHASE_PROBE -Iinclude -I$(CSRC) -o $@ $(HIP_SRCS)

oracle:
	$(MAKE) -s -C oracle
model:
	$(MAKE) -s -C tests/model
# Native (C++) drive of the dispatcher through the scheduler harness; plain g++, links libydc.so.
native: tests/native/harness_test tests/native/td_linearize_gpu tools/td_native_bench tools/hbm_calib tests/tools/launch_probe tests/tools/overlap_probe tests/tools/atomic_probe
	$(MAKE) -s -C tests/native all
# Counter calibration microbenchmark (tools/calibrate.sh runs it under rocprofv3 on the GPU box).
tools/hbm_calib: tools/hbm_calib.hip
	$(HIPCC) --offload-arch=$(ARCH) -O2 -o $@ tools/hbm_calib.hip
# Launch latency / resident mailbox and two-queue overlap microbenchmarks.
tests/tools/launch_probe: tests/tools/launch_probe.hip
	$(HIPCC) --offload-arch=$(ARCH) -O2 -o $@ tests/tools/launch_probe.hip
tests/tools/overlap_probe: tests/tools/overlap_probe.hip
	$(HIPCC) --offload-arch=$(ARCH) -O2 -o $@ tests/tools/overlap_probe.hip
tests/tools/atomic_probe: tests/tools/atomic_probe.hip
	$(HIPCC) --offload-arch=$(ARCH) -O2 -o $@ tests/tools/atomic_probe.hip
tools/td_native_bench: tools/td_native_bench.cc tools/parked_workload.h yadcc_amd/libydc.so $(HDRS)
	g++ -O2 -std=c++17 -Wall -I$(CSRC) -Iinclude -Itools -o $@ tools/td_native_bench.cc \
	    -Lyadcc_amd -lydc -Wl,-rpath,'$$ORIGIN/../yadcc_amd' -lpthread
# The same tool over the CPU stand-in of the device API (host-side profiling without a GPU).
tools/td_native_bench_stub: tools/td_native_bench.cc tools/parked_workload.h $(HDRS)
	$(MAKE) -s -C tests/native all
	g++ -O2 -g -std=c++17 -Wall -I$(CSRC) -Iinclude -Itools -o $@ tools/td_native_bench.cc \
	    -Ltests/native -ltd_stub -Wl,-rpath,'$$ORIGIN/../tests/native' -lpthread
tests/native/harness_test: tests/native/harness_test.cc tests/native/scheduler_harness.cc tests/native/scheduler_harness.h yadcc_amd/libydc.so $(HDRS)
	g++ -O2 -std=c++17 -Wall -I$(CSRC) -Iinclude -Itests/native -o $@ tests/native/harness_test.cc tests/native/scheduler_harness.cc \
	    -Lyadcc_amd -lydc -Wl,-rpath,'$$ORIGIN/../../yadcc_amd' -lpthread
# Concurrent callers through the C-ABI against the real library (verified against the reference by
# tests/test_task_dispatcher_gpu.py).
tests/native/td_linearize_gpu: tests/native/td_linearize.cc yadcc_amd/libydc.so include/yadcc_dispatch.h
	g++ -O2 -std=c++17 -Wall -Iinclude -o $@ tests/native/td_linearize.cc \
	    -Lyadcc_amd -lydc -Wl,-rpath,'$$ORIGIN/../../yadcc_amd' -lpthread
# Sanitizer builds of the host class against the CPU stand-in of the device API (no GPU).
tsan asan:
	$(MAKE) -s -C tests/native $@
clean:
	rm -f yadcc_amd/libydc.so tests/model/libmodel.so
	$(MAKE) -C tests/native clean
	$(MAKE) -C oracle clean
.PHONY: all lib probe oracle model native tsan asan clean
