#!/bin/bash
# the GPU suite, log under gpurun_out/<tag>/ (usage: gpu_tests.sh <tag> [pytest args])
tag=${1:-r02t}; shift || true
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=12 "$@" > $out/pytest.log 2>&1; rc=$?
grep -E "^(FAILED|ERROR)|passed|failed|R4 " $out/pytest.log | tail -40
echo "pytest rc=$rc"
exit $rc
