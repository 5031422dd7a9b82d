#!/bin/bash
# round 3, call q: the P = 2 code path of bench.py and of the sharded layer with the real kernels — two ranks on ONE GPU over gloo (RCCL refuses
# duplicate devices); functional check only
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03q; mkdir -p $O; cd $R
export DANCE_AMD_BENCH_ONE_GPU=1 DANCE_AMD_BENCH_BACKEND=gloo
for P in 2 4; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 --master-port 2951$P bench.py --gpus $P --steps 3 --warmup 1 --cells 200000 > $O/bench_p$P.json 2> $O/bench_p$P.err
echo "P=$P rc=$?"; tail -c 1800 $O/bench_p$P.json; grep -v '^W\|^\[W\|amdgpu.ids\|^\*\|OMP_NUM' $O/bench_p$P.err | tail -12
done
