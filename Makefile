# Builds libsim3opt.so (hipcc, gfx950), the CPU oracle and the C++ example without Python.
#   make            -> sim3opt_amd/libsim3opt.so, oracle/liboracle_sim3.so
#   make example    -> examples/direct_pgo (testDirectSim3Optimization on the g2o-named shim)
HIPCC ?= /opt/rocm/bin/hipcc
CSRC  := sim3opt_amd/csrc
SRCS  := $(CSRC)/engine.hip $(CSRC)/capi.cpp $(CSRC)/graph.cpp $(CSRC)/kitti_io.cpp \
         $(CSRC)/comm.cpp $(CSRC)/eval.cpp $(CSRC)/stepwise.cpp $(CSRC)/map_io.hip
HDRS  := $(wildcard $(CSRC)/*.hpp) include/sim3opt.h

all: sim3opt_amd/libsim3opt.so oracle/liboracle_sim3.so

sim3opt_amd/libsim3opt.so: $(SRCS) $(HDRS)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-result -o $@ $(SRCS) -ldl

oracle/liboracle_sim3.so: oracle/sim3_oracle.c oracle/sim3_oracle.h
	$(MAKE) -C oracle liboracle_sim3.so

example: sim3opt_amd/libsim3opt.so examples/direct_pgo.cpp include/sim3opt_g2o.hpp
	g++ -std=c++17 -Wall -DSIM3OPT_G2O_NAMES -Iinclude examples/direct_pgo.cpp -Lsim3opt_amd -lsim3opt \
	    -Wl,-rpath,$(CURDIR)/sim3opt_amd -o examples/direct_pgo

clean:
	rm -f sim3opt_amd/libsim3opt.so oracle/liboracle_sim3.so examples/direct_pgo

.PHONY: all example clean
