# libtrajopt_hip.so for gfx950 (hipcc cross-compiles without a GPU) and the CPU oracle used by the tests.
#   make lib      -> trajectory_optimization_amd/libtrajopt_hip.so   (same command as trajectory_optimization_amd/_lib.py)
#   make oracle   -> oracle/_build/liboracle.so
#   make test     -> the CPU test suite (the GPU suite: python -m pytest tests -q -m gpu on an MI355X)
HIPCC ?= /opt/rocm/bin/hipcc
SRC := trajectory_optimization_amd/csrc
LIB := trajectory_optimization_amd/libtrajopt_hip.so

lib: $(LIB)

$(LIB): $(wildcard $(SRC)/*.hip) $(wildcard $(SRC)/*.hpp) include/trajopt_hip.h
	$(HIPCC) -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -shared $(SRC)/trajopt_hip.hip -o $@

oracle:
	$(MAKE) -C oracle

test: lib oracle
	python -m pytest tests -q -m "not gpu"

.PHONY: lib oracle test
