mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short > gpurun_out/t2.log 2>&1; tail -3 gpurun_out/t2.log
for cfg in "--dbg 0" "--dbg 0 --nw 4" "--dbg 0 --nw 8"; do
  python tests/bench_kernels.py --batch 1 --only-qmm $cfg 2>&1 | grep -v amdgpu.ids
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -o bk -- python $GRAFT_REPO_ROOT/tests/bench_kernels.py --batch 1 --nw 4 > /dev/null 2>&1
