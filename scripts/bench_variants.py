import sys, json, subprocess
for v in sys.argv[1:] or ["stream", "split"]:
    import os
    env = dict(os.environ, LMC_VARIANT=v)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "3", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(v, "ms/launch", round(d["roofline"]["launch_ms"], 3), "chain-it/s", int(d["value"]), "frac", round(d["roofline"]["frac"], 4), d["roofline"]["kernel"], flush=True)
    except Exception as e:
        print(v, "FAILED", out.stderr[-500:])
