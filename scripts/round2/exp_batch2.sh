#!/bin/bash
bash scripts/round2/exp_pipe.sh nohstore -DLMC_EXP_NO_HSTORE
bash scripts/round2/exp_pipe.sh halfhstore -DLMC_EXP_HALF_HSTORE
