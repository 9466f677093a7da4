#!/bin/bash
# phase-shift experiments: delay some waves' tick start (results stay correct: only timing moves)
bash scripts/round2/exp_pipe.sh sleepT1_6 -DLMC_EXP_SLEEP_MASK=2 -DLMC_EXP_SLEEP_N=6 2>&1 | grep -E "launch_ms|pytest"
bash scripts/round2/exp_pipe.sh sleepN_8 -DLMC_EXP_SLEEP_MASK=128 -DLMC_EXP_SLEEP_N=8 2>&1 | grep -E "launch_ms|pytest"
bash scripts/round2/exp_pipe.sh sleepT135_6 -DLMC_EXP_SLEEP_MASK=42 -DLMC_EXP_SLEEP_N=6 2>&1 | grep -E "launch_ms|pytest"
bash scripts/round2/exp_pipe.sh sleepT24_6 -DLMC_EXP_SLEEP_MASK=20 -DLMC_EXP_SLEEP_N=6 2>&1 | grep -E "launch_ms|pytest"
