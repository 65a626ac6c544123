# Round-2 refresh of the C3 numbers under profiles/ after the shared-partial-sum form: bench line, rocprofv3 kernel stats.
# Run on the GPU box from the repo root:  bash tools/refresh_c3_r02.sh   (results under gpurun_out/r02c3/)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c3
mkdir -p $O
python bench.py --workload c3 --steps 5 --warmup 2 > $O/c3.json 2> $O/c3.err
python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --option row_classes=0 > $O/c3_rowbyrow.json 2> $O/c3_rowbyrow.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_c3.log 2>&1
cd $R
find $O -name "*_kernel_trace.csv" -delete
find $O -size +20M -delete
cut -c1-400 $O/c3.json; echo; cut -c1-200 $O/c3_rowbyrow.json; echo
head -12 $O/prof_c3/*/c3_kernel_stats.csv 2>/dev/null || find $O -name "*kernel_stats.csv" | head
