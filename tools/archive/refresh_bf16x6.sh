R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r
python bench.py --steps 10 --warmup 3 --option gemm_arith=1 > $R/gpurun_out/r/c4_bf16x6.json 2> $R/gpurun_out/r/c4_bf16x6.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r/prof_c4b -o c4b -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --option gemm_arith=1 > $R/gpurun_out/r/prof_c4b.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r/pmc_mfma_c4b -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --option gemm_arith=1 > $R/gpurun_out/r/pmc_m4b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r/pmc_f4b -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --option gemm_arith=1 > $R/gpurun_out/r/pmc_f4b.log 2>&1
cd $R; find gpurun_out/r -name "*_kernel_trace.csv" -delete; cut -c1-200 gpurun_out/r/c4_bf16x6.json
