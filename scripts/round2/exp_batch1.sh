#!/bin/bash
bash scripts/round2/exp_pipe.sh base
bash scripts/round2/exp_pipe.sh early -DLMC_EXP_EARLY_HANDOFF
