# Builds libsim3opt.so (hipcc, gfx950), the CPU oracle and the C++ examples without Python.
#   make                 -> sim3opt_amd/libsim3opt.so, oracle/liboracle_sim3.so
#   make example         -> examples/direct_pgo (testDirectSim3Optimization on the g2o-named shim)
#   make conformance     -> tests/cxx/shim_conformance, tests/cxx/ba_shim_conformance (one block per
#                           method of the g2o-named shims; EIGEN_INC=-I/usr/include/eigen3 for a real Eigen)
#   make LIB=/some/where/libsim3opt.so   builds the library elsewhere (used by the tests)
HIPCC ?= /opt/rocm/bin/hipcc
CSRC  := sim3opt_amd/csrc
# one list of translation units, shared with sim3opt_amd/build.py
SRCS  := $(addprefix $(CSRC)/,$(shell cat $(CSRC)/SOURCES))
HDRS  := $(wildcard $(CSRC)/*.hpp) include/sim3opt.h include/sim3opt_bench.h $(CSRC)/SOURCES
LIB   ?= sim3opt_amd/libsim3opt.so
LIBDIR = $(abspath $(dir $(LIB)))
EIGEN_INC ?= -Itests/mock_eigen

all: $(LIB) oracle/liboracle_sim3.so

$(LIB): $(SRCS) $(HDRS)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-result -o $@ $(SRCS) -ldl

oracle/liboracle_sim3.so: oracle/sim3_oracle.c oracle/sim3_oracle.h
	$(MAKE) -C oracle liboracle_sim3.so

example: $(LIB) examples/direct_pgo.cpp include/sim3opt_g2o.hpp
	g++ -std=c++17 -Wall -DSIM3OPT_G2O_NAMES -Iinclude examples/direct_pgo.cpp -L$(LIBDIR) -lsim3opt \
	    -Wl,-rpath,$(LIBDIR) -o examples/direct_pgo

conformance: $(LIB) tests/cxx/shim_conformance.cpp tests/cxx/ba_shim_conformance.cpp include/sim3opt_g2o.hpp include/sim3opt_g2o_ba.hpp
	g++ -std=c++17 -Wall -DSIM3OPT_G2O_NAMES -Iinclude $(EIGEN_INC) tests/cxx/shim_conformance.cpp \
	    -L$(LIBDIR) -lsim3opt -Wl,-rpath,$(LIBDIR) -o tests/cxx/shim_conformance
	g++ -std=c++17 -Wall -DSIM3OPT_G2O_BA_NAMES -Iinclude $(EIGEN_INC) tests/cxx/ba_shim_conformance.cpp \
	    -L$(LIBDIR) -lsim3opt -Wl,-rpath,$(LIBDIR) -o tests/cxx/ba_shim_conformance

clean:
	rm -f sim3opt_amd/libsim3opt.so oracle/liboracle_sim3.so examples/direct_pgo tests/cxx/shim_conformance \
	    tests/cxx/ba_shim_conformance

.PHONY: all example conformance clean
