import sys, json, subprocess, os
# same number of pixel-iterations per launch in every row: H*W*C = 512*512*1024
for (H, W, C) in [(512, 512, 1024), (512, 256, 2048), (512, 128, 4096), (512, 64, 8192)]:
    for v in ["split", "stream"]:
        env = dict(os.environ, LMC_VARIANT=v)
        out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-moments",
                              "--size", str(H), "--width", str(W), "--chains", str(C)], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(H, W, C, v, "ms/launch", round(d["roofline"]["launch_ms"], 3), flush=True)
        except Exception as e:
            print(H, W, C, v, "FAILED", out.stderr[-300:])
