#!/bin/bash
# rehearsals of bench.py's N > 1 code on the one-GPU box: (a) RCCL path at world size 1 (C-ABI collective on torch's communicator, then on one of our own),
# (b) two ranks on the same GPU over gloo under torch.distributed.run
out=gpurun_out/r02d; mkdir -p $out
LMC_BENCH_DIST=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-probe > $out/dist1.json 2> $out/dist1.err; echo "dist1 rc=$?"
python -c "import json;d=json.load(open('$out/dist1.json'));print(d['n_gpus'], round(d['value']), d['config']['collective'][:90])"
LMC_RCCL_COMM=own LMC_BENCH_DIST=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-probe > $out/dist1own.json 2> $out/dist1own.err; echo "dist1own rc=$?"
python -c "import json;d=json.load(open('$out/dist1own.json'));print(d['n_gpus'], round(d['value']), d['config']['collective'][:90])"
LMC_BENCH_BACKEND=gloo LMC_BENCH_DEVICE=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --chains 256 > $out/gloo2.json 2> $out/gloo2.err; echo "gloo2 rc=$?"
tail -1 $out/gloo2.json | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['n_gpus'], d['config']['chains_total'], round(d['value']), d['config']['collective'][:80])"
