#!/bin/bash
# GPU: the gene <- cell split-K kernel: parity tests, then the 1M-cell timing
TAG=${TAG:-r04u}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_sage_dense.py -x -q -k "splitk or mfma" 2>&1 | tail -15
DANCE_AMD_SAGE_MFMA=bcm timeout 600 python -m pytest tests/test_gpu_sage_dense.py -x -q -k "mfma" 2>&1 | tail -5
TAG=$TAG timeout 900 python scripts/sage_splitk_time.py 2> gpurun_out/$TAG/splitk.err | tail -5
tail -5 gpurun_out/$TAG/splitk.err
