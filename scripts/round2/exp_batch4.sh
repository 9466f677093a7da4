#!/bin/bash
# role-skip timing builds (results wrong): how much each role constrains the tick, barriers on
for m in 2 1 128 64 32; do
  bash scripts/round2/exp_pipe.sh skip$m -DLMC_EXP_SKIP=$m 2>&1 | grep launch_ms
done
