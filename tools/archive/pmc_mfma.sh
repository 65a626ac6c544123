# MFMA-pipe busy cycles of the C4 (and C3) bench command, one PMC pass (SQ + GRBM counters only)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r/pmc_mfma_c4 -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r/pmc_m4.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r/pmc_mfma_c3 -o m -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r/pmc_m3.log 2>&1
cd $R; ls -la gpurun_out/r/pmc_mfma_c4 gpurun_out/r/pmc_mfma_c3; tail -2 gpurun_out/r/pmc_m4.log
