#!/bin/bash
# libsigma_hip_probe.so = the library with k_cg_coop's phase timers compiled in (-DSGM_COOP_PROBE; tools/probes/coop_probe.py)
set -e
cd "$(dirname "$0")/../../sigma_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I/opt/rocm/include"
/opt/rocm/bin/hipcc $FLAGS -DSGM_COOP_PROBE -c sgm_cg.hip -o /tmp/sgm_cg_probe.o
OBJS=$(ls *.o | grep -v sgm_cg.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/libsigma_hip_probe.so $OBJS /tmp/sgm_cg_probe.o -ldl
