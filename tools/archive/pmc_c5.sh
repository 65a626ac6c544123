# HBM traffic of the C5 (native CSR) bench command: FETCH_SIZE and WRITE_SIZE in separate PMC passes
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r/pmc_c5_fetch -o f -- python3 $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r/pmc_c5f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r/pmc_c5_write -o w -- python3 $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r/pmc_c5w.log 2>&1
cd $R; ls gpurun_out/r/pmc_c5_fetch gpurun_out/r/pmc_c5_write
