#!/bin/bash
# A/B builds for scripts/overlap_diag.py: the GEMM's wave priority next to the resident aggregation.
#   build (here): bash scripts/overlap_diag.sh build      run (GPU box): bash scripts/overlap_diag.sh run
VARIANTS=("prio3:-DDH_GEMM_SETPRIO=3")
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
  timeout 600 python $R/scripts/overlap_diag.py default > $R/gpurun_out/r05b_overlap_diag_default.json
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    DANCE_HIP_LIB=$V/libdancehip_$name.so timeout 300 python $R/scripts/overlap_diag.py $name quick > $R/gpurun_out/r05b_overlap_diag_$name.json
  done
fi
