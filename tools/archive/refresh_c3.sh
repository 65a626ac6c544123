mkdir -p gpurun_out/r
timeout 900 python -m pytest tests/test_gpu_newton.py tests/test_gpu_estimator.py -x -q 2>&1 | tail -3
python bench.py --workload c3 > gpurun_out/r/c3.json 2> gpurun_out/r/c3.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r/prof_c3 -o c3 -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r/prof_c3.log 2>&1
cd $R; find gpurun_out/r -name "*_kernel_trace.csv" -delete
cut -c1-300 gpurun_out/r/c3.json
