#!/bin/bash
# keyframes/s as a function of the batch (run through gpurun): one bench.py line per batch size into gpurun_out/sweep_<B>.json
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for b in ${BATCHES:-1 4 16 64 128 256 512}; do
  timeout 400 python bench.py --batch $b --no-cpu-baseline --inputs resident --unique-frames 16 --render-workers 0 > gpurun_out/sweep_$b.json 2> gpurun_out/sweep_$b.err
  python - $b <<'PY'
import json, sys
b = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/sweep_%s.json" % b).read().strip().split("\n")[-1])
    k = d["kernels_ms_per_step"]
    print(b, d["value"], d["ms_per_step"], k.get("lm_window_kernel"), round(sum(v for n, v in k.items() if n.startswith("orb_")), 4), d["roofline"]["frac"])
except Exception as e:
    print(b, "failed", e)
PY
done
