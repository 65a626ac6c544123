set -x
mkdir -p gpurun_out/r
python bench.py --steps 10 --warmup 3 > gpurun_out/r/c4.json 2> gpurun_out/r/c4.err
python bench.py --workload c2 > gpurun_out/r/c2.json 2> gpurun_out/r/c2.err
python bench.py --workload c3 > gpurun_out/r/c3.json 2> gpurun_out/r/c3.err
python bench.py --workload c5 > gpurun_out/r/c5.json 2> gpurun_out/r/c5.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r/prof_c4 -o c4 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r/prof_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r/prof_c3 -o c3 -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r/prof_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r/prof_c5 -o c5 -- python3 $R/bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r/prof_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r/prof_c2 -o c2 -- python3 $R/bench.py --workload c2 --no-cpu-baseline > $R/gpurun_out/r/prof_c2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r/pmc_fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r/pmc_write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r/pmc_w.log 2>&1
cd $R; find gpurun_out/r -name "*_kernel_trace.csv" -delete; find gpurun_out/r -size +20M -delete; du -sh gpurun_out/r
