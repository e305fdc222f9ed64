cd $GRAFT_REPO_ROOT
bash tools/profile_sq.sh r03_orb python tools/bench_orb.py --batch 256 --reps 3 2>&1 | tail -12
bash tools/profile_mix.sh r03_orb python tools/bench_orb.py --batch 256 --reps 3 > /dev/null 2>&1
cat gpurun_out/mix_r03_orb/summary.txt | head -8
