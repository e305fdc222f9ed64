#!/bin/bash
# Kernel trace of the default step with one and with two batches in flight (bench.py --in-flight): how the kernels of the two streams overlap.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
COMMON="--steps 6 --warmup 2 --repeats 1 --no-plain-schedule --no-cpu-baseline --no-config4 --no-reference-pipeline --no-config4-step --no-live-dropin --render-workers 0 --unique-frames 128 --inputs resident"
for P in 1 2; do
  OUT=gpurun_out/prof_r05_inflight$P; mkdir -p $OUT
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py $COMMON --in-flight $P > $OUT/bench.json 2> $OUT/trace.log
  python tools/overlap_from_trace.py $OUT/trace > $OUT/overlap.txt 2>&1
  echo "== --in-flight $P"; cat $OUT/overlap.txt
  python - $OUT/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench line: value %.0f keyframes/s, %.3f ms/step, in_flight %s" % (d["value"], d["ms_per_step"], json.dumps(d["timing"]["in_flight"]["one_batch_in_flight"])))
PY
done
