#!/bin/bash
# development: ablation builds of the kNN filter kernel (dance_amd/libdancehip_knnabl{1,2,3}.so; load with DANCE_HIP_LIB=...)
cd /root/repo/dance_amd/csrc
for v in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DDH_BUILDING -ffp-contract=off -DDH_KNN_ABL=$v -c knn_filter.hip -o /tmp/knn_filter_abl$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdancehip_knnabl$v.so $(ls ../../build/csrc/*.o | grep -v knn_filter.o) /tmp/knn_filter_abl$v.o
done
