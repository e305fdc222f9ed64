cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03/gputests_final.log 2>&1
tail -6 gpurun_out/r03/gputests_final.log
( time timeout 600 python bench.py ) > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err
tail -c 400 gpurun_out/r03/bench_final.err
rm -f gpurun_out/r03/batch_sweep.jsonl
for B in 1 4 16 64 128 256 512; do
  timeout 300 python bench.py --batch $B --no-cpu-baseline --inputs resident --unique-frames 16 --render-workers 8 2>/dev/null | tail -1 >> gpurun_out/r03/batch_sweep.jsonl
done
wc -l gpurun_out/r03/batch_sweep.jsonl
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh r03b > gpurun_out/prof_r03b.log 2>&1
tail -40 gpurun_out/prof_r03b.log
