#!/bin/bash
# A/B builds of libdancehip.so that differ only in gemm_f32.hip's compile-time switches, timed at the headline shapes.
#   build (here, no GPU):   bash scripts/gemm_variants.sh build
#   run (on the GPU box):   bash scripts/gemm_variants.sh run  > gpurun_out/gemm_variants.jsonl
# Variants: name:flags
VARIANTS=(
  "lg1:-DDH_GEMM_LIVEGROUPS=1"
  "lg0:-DDH_GEMM_LIVEGROUPS=0"
)
R=$(cd "$(dirname "$0")/.." && pwd)
V=$R/build/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DDH_BUILDING -ffp-contract=off"
if [ "$1" = build ]; then
  make -C $R/dance_amd/csrc -j16 > /dev/null || exit 1
  mkdir -p $V
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}; fl=${v#*:}
    ( cd $R/dance_amd/csrc && /opt/rocm/bin/hipcc $FLAGS $fl -c gemm_f32.hip -o $V/gemm_f32_$name.o &&
      objs=$(ls $R/build/csrc/*.o | grep -v "/gemm_f32.o") &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libdancehip_$name.so $objs $V/gemm_f32_$name.o ) || exit 1
    echo built $name
  done
else
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    DANCE_HIP_LIB=$V/libdancehip_$name.so timeout 300 python $R/scripts/gemm_variants.py $name
  done
fi
