# Top-level convenience targets (plain make; no cmake needed).
#   make            libreevr_amd.so (hipcc, gfx950) + the oracle's C restatement
#   make ref        the untouched reference built where it lies (needs /root/reference)
#   make example    examples/host_block_loop + host_many_channels (g++, link the C ABI)
#   make test       CPU test-suite;   make test-gpu   on an MI355X
HIPCC ?= /opt/rocm/bin/hipcc
CSRC  := reevr_amd/csrc
LIB   := $(CSRC)/libreevr_amd.so

all: $(LIB) oracle

$(LIB): $(CSRC)/rvc_kernels.hip $(CSRC)/rvc_sweep.hip $(CSRC)/rvc_impulse.hip $(CSRC)/rvc_plan.cpp $(CSRC)/rvc_state.cpp $(CSRC)/rvc_schedule.cpp $(CSRC)/rvc_abi.cpp $(CSRC)/rvc_set.h $(CSRC)/rvc_internal.h $(CSRC)/rvc_fft_lds.hpp include/reevr_amd/rvc.h
	python -m reevr_amd.build --force     # hipcc -c per source (rvc_sweep.hip with -fno-slp-vectorize), then link

oracle:
	$(MAKE) -C oracle

ref:
	$(MAKE) -C oracle ref

example: $(LIB)
	g++ -O2 -std=c++17 -I include examples/host_block_loop.cpp -L $(CSRC) -lreevr_amd \
	  -Wl,-rpath,'$$ORIGIN/../$(CSRC)' -o examples/host_block_loop
	g++ -O2 -std=c++17 -I include examples/host_many_channels.cpp -L $(CSRC) -lreevr_amd \
	  -Wl,-rpath,'$$ORIGIN/../$(CSRC)' -o examples/host_many_channels

test: all
	python -m pytest tests -q -m "not gpu"

test-gpu: all
	python -m pytest tests -q -m gpu

clean:
	rm -f $(LIB) examples/host_block_loop examples/host_many_channels
	$(MAKE) -C oracle clean

.PHONY: all oracle ref example test test-gpu clean
