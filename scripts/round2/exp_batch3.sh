#!/bin/bash
bash scripts/round2/exp_obj.sh rows_base lmc_step_rows "--prior l2" -fno-slp-vectorize
bash scripts/round2/exp_obj.sh rows_slp lmc_step_rows "--prior l2"
