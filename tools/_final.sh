cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r02 > /dev/null 2>&1
bash tools/pmc_step.sh pmc_step > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o tr -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py 5 > $GRAFT_REPO_ROOT/gpurun_out/train_step.log 2>&1
f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/r02_train_step_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
