#!/bin/bash
# a tuning build of the index kernels with cycle marks between their phases (work-item 0 of every workgroup), on the per-table C3 leg
cd ${GRAFT_REPO_ROOT:-.}
cp mash_amd/libmashgpu.so /tmp/libmashgpu.keep
cd mash_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DIX_PHASE_CLOCKS -c index_build.hip -o build/index_build.o 2>&1 | grep -E "error" ; \
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DIX_PHASE_CLOCKS -x hip -c host_index.cpp -o build/host_index.o 2>&1 | grep -E " error"; \
  make 2>&1 | tail -1; cd ../..
MASHGPU_IX_CLOCKS=1 python tools/prof_leg.py --leg ${1:-c3} --steps 3 --cold 2>&1 | grep "ix clocks" | tail -15
cp /tmp/libmashgpu.keep mash_amd/libmashgpu.so
