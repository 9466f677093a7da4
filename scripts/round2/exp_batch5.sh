#!/bin/bash
bash scripts/round2/exp_pipe.sh sched1 -DLMC_PIPE_SCHED=1
