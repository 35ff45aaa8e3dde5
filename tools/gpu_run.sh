#!/bin/bash
set -u
O=gpurun_out/c44; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
bash tools/profile.sh r02_k 2>&1 | tail -1
bash tools/profile.sh r02_k_m2 --config 3 2>&1 | tail -1
timeout 120 python tools/fuzz_gpu.py 45 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
