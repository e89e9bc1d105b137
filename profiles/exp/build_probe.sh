#!/bin/bash
# Diagnostic library: libjirf_hip with phase stamps compiled into gemm_xs_kernel (-DJIRF_XS_PROBE).
# Run from the repository root after `make -C image-restoration_amd`; writes profiles/exp/libjirf_probe.so.
set -e
mkdir -p profiles/exp/build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DJIRF_XS_PROBE ${PROBE_DEFS} -c image-restoration_amd/csrc/gemm.hip -o profiles/exp/build/gemm.o
objs=$(ls image-restoration_amd/build/*.o | grep -v /gemm.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o profiles/exp/libjirf_probe.so profiles/exp/build/gemm.o $objs
