#!/bin/bash
# role pairing of the all-live RT links of a chained prox (ME-TV as configured): C0 = L + C | T1 + T5 | T2 + T4 | T3 + N, C1 = L + T5 | T1 + T2 | T3 + N | T4 + C, C2 = L + T4 | T1 + T3 | T2 + C | T5 + N
B="python bench.py --no-hbm-probe --no-cpu-baseline --ncvx me --ncvx-iters 50 --ncvx-rtol 1e-4 --tv-rtol 1e-4 --steps 20 --warmup 40"
export LMC_BENCH_AS_CONFIGURED=0
for rep in 1 2; do for v in C0 C1 C2; do
  LMC_ATOMI_LIB=$PWD/build/var/liblmc_atomi_$v.so $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), d['config']['ncvx_exit']['reruns_after_round_1_2_3_4'])"
done; done
