#!/bin/bash
# one test file on the GPU box: gpu_one.sh <pytest args>
mkdir -p gpurun_out/r02o
timeout -k 10 600 python -m pytest "$@" -m gpu -q -x > gpurun_out/r02o/pytest.log 2>&1; rc=$?
tail -n 15 gpurun_out/r02o/pytest.log; echo "pytest rc=$rc"
