#!/bin/bash
# GPU: after the split-K wiring — sage + scdeepsort tests, then the full GPU suite, then the timing
TAG=${TAG:-r04z}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_sage_dense.py tests/test_gpu_scdeepsort.py -x -q 2>&1 | tail -8
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
TAG=$TAG timeout 600 python scripts/sage_splitk_time.py 2> gpurun_out/$TAG/splitk.err | grep "^f32\|^bf16" | cut -c1-200
