#!/bin/bash
out=gpurun_out/r02r; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_abi2.py tests/test_gpu_parity.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -n 6 $out/pytest.log; echo "pytest rc=$rc"
python - <<'PY'
import time, numpy as np, torch, sys
sys.path.insert(0, '.')
import lmc_atomi_amd as la
import bench
H = W = 512; sigma = 0.75
u, h, y = bench.synth_problem(H, W, sigma)
pf = la.L2(Op=la.Convolve2D((H, W), h, offset=(2, 2)), b=y, sigma=1 / sigma ** 2)
for rtol in (0.0, 1e-4):
    smp = la.MYULASampler(pf, la.TV((H, W), sigma=0.3, niter=10, rtol=rtol), (H, W), n_chains=1024, tau=0.2 * sigma ** 2, gamma=sigma ** 2, seed=0)
    smp.step(10); torch.cuda.synchronize(); t0 = time.perf_counter(); smp.step(20); torch.cuda.synchronize()
    print(f"512x512x1024 MYULA TV K=10 rtol={rtol:g}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per iteration")
    smp.close()
PY
