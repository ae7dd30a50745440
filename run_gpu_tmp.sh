mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2>/dev/null
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_r01 | head -20
