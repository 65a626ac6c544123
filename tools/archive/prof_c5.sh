# rocprofv3 kernel stats of the C5 workload (bench.py --workload c5); summary lands in gpurun_out/prof_c5/
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -o c5 -- python3 $R/bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_c5.log 2>&1
cd $R; find gpurun_out/prof_c5 -name "*_kernel_trace.csv" -delete; find gpurun_out/prof_c5 -size +20M -delete
find gpurun_out/prof_c5 -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200
