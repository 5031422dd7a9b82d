#!/bin/bash
# development: a second library with the phase timers of sage_bcm.hip compiled in (dance_amd/libdancehip_prof.so; scripts/sage_prof.py loads it)
cd /root/repo/dance_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DDH_BUILDING -ffp-contract=off -DDH_SB_PROF -c sage_bcm.hip -o /tmp/sage_bcm_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdancehip_prof.so $(ls ../../build/csrc/*.o | grep -v sage_bcm.o) /tmp/sage_bcm_prof.o
