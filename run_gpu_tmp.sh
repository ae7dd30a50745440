mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof6 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 8 --no-cpu-baseline > /dev/null 2>&1
