mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-batch32 --no-graph > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-batch32 --no-graph > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch $GRAFT_REPO_ROOT/gpurun_out/pmc_write
