#!/bin/bash
# round 2: re-profile every kernel DESIGN quotes (60-step traces + PMC passes) and collect the summaries under gpurun_out/prof_r02_*
# usage: gpu_profiles.sh a|b   (two calls: each stays well inside one gpurun limit)
set -u
part=${1:-a}
if [ $part = a ]; then
  bash scripts/profile.sh r02_pipe > /dev/null 2>&1 && echo pipe done
  LMC_ROWS_PAIR=0 bash scripts/profile.sh r02_rows --prior l2 > /dev/null 2>&1 && echo rows done
  bash scripts/profile.sh r02_rowspair --prior l2 > /dev/null 2>&1 && echo rowspair done
  bash scripts/profile.sh r02_rowspairnm --prior l2 --no-moments > /dev/null 2>&1 && echo rowspairnm done
  bash scripts/profile.sh r02_c2 --prior l2 --size 256 --chains 128 > /dev/null 2>&1 && echo c2 done
  LMC_BLOCK_PAIR=0 bash scripts/profile.sh r02_block --prior haar --data mask > /dev/null 2>&1 && echo block done
  bash scripts/profile.sh r02_blockpair --prior haar --data mask > /dev/null 2>&1 && echo blockpair done
  bash scripts/profile.sh r02_blockpairnm --prior haar --data mask --no-moments > /dev/null 2>&1 && echo blockpairnm done
  bash scripts/profile.sh r02_ulpda --alg ulpda > /dev/null 2>&1 && echo ulpda done
  tags="pipe rows rowspair rowspairnm c2 block blockpair blockpairnm ulpda"
else
  for k in 1 2 3; do bash scripts/profile.sh r02_warm$k --tv-warm --tv-iters $k > /dev/null 2>&1 && echo warm$k done; done
  bash scripts/profile.sh r02_wide877tv --size 667 --width 877 --chains 512 > /dev/null 2>&1 && echo wide877tv done
  bash scripts/profile.sh r02_wide877l2 --size 667 --width 877 --chains 512 --prior l2 > /dev/null 2>&1 && echo wide877l2 done
  bash scripts/profile.sh r02_metv --ncvx me --ncvx-iters 50 > /dev/null 2>&1 && echo metv done
  tags="warm1 warm2 warm3 wide877tv wide877l2 metv"
fi
for t in $tags; do echo "=== $t"; head -6 gpurun_out/prof_r02_$t/summary.txt | cut -c1-200; grep -E "VALU busy|FETCH_SIZE|WRITE_SIZE" gpurun_out/prof_r02_$t/summary.txt | head -8; done
if [ $part = a ]; then
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "import json,sys; d=json.load(sys.stdin); print('peak_measured', d['roofline']['peak_measured'], 'launch_ms', d['roofline']['launch_ms'])"
fi
