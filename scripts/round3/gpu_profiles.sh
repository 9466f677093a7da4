#!/bin/bash
# round 3: kernel traces (60 timed steps) + PMC passes of the kernels DESIGN quotes, under gpurun_out/prof_r03_*  (usage: gpu_profiles.sh a|b|c)
set -u
part=${1:-a}
P="bash scripts/profile.sh"
if [ $part = a ]; then
  $P r03_pipe > /dev/null 2>&1 && echo pipe done
  $P r03_pipert --tv-rtol 1e-4 --warmup 60 > /dev/null 2>&1 && echo pipert done
  $P r03_pipe7 --blur-k 7 > /dev/null 2>&1 && echo pipe7 done
  $P r03_pipemc --ncvx mc > /dev/null 2>&1 && echo pipemc done
  tags="pipe pipert pipe7 pipemc"
elif [ $part = b ]; then
  $P r03_rowspair --prior l2 > /dev/null 2>&1 && echo rowspair done
  LMC_ROWS_PAIR=0 $P r03_rows7 --prior l2 --blur-k 7 > /dev/null 2>&1 && echo rows7 done
  $P r03_blockpair --prior haar --data mask > /dev/null 2>&1 && echo blockpair done
  $P r03_c5 --config 5 > /dev/null 2>&1 && echo c5 done
  $P r03_c2 --config 2 > /dev/null 2>&1 && echo c2 done
  tags="rowspair rows7 blockpair c5 c2"
else
  $P r03_ulpda --alg ulpda > /dev/null 2>&1 && echo ulpda done
  $P r03_ulpda7 --alg ulpda --blur-k 7 > /dev/null 2>&1 && echo ulpda7 done
  $P r03_metv --ncvx me --ncvx-iters 50 --ncvx-rtol 1e-4 --tv-rtol 1e-4 --warmup 30 > /dev/null 2>&1 && echo metv done
  $P r03_mymala --alg mymala > /dev/null 2>&1 && echo mymala done
  tags="ulpda ulpda7 metv mymala"
fi
for t in $tags; do echo "=== $t"; head -6 gpurun_out/prof_r03_$t/summary.txt | cut -c1-200; grep -E "VALU busy|FETCH_SIZE|WRITE_SIZE" gpurun_out/prof_r03_$t/summary.txt | head -8; done
