R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o c3 -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/tr.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/tr/**/c3_kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
print(list(rows[0].keys()), len(rows))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-80:]:
    n=r['Kernel_Name']
    if 'fill' in n: continue
    print('%-60s grid=%s  %.3f ms' % (n[:60], r.get('Grid_Size', r.get('Grid_Size_X')), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
PY
