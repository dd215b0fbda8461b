set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/tc
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tc -o cfg2 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-gemm-roofline > gpurun_out/r05_t_trace.log 2>&1
ls /tmp/tc/* | head
f=$(find /tmp/tc -name "*memory_copy_trace.csv" | head -1)
head -3 $f
python - <<PY
import csv
rows=list(csv.DictReader(open('$f')))
print(len(rows), rows[0].keys())
k=list(csv.DictReader(open('$(find /tmp/tc -name "*kernel_trace.csv" | head -1)')))
adam=[int(r['Start_Timestamp']) for r in k if 'adam_clip' in r['Kernel_Name']]
a,b=adam[-2],adam[-1]
sel=[r for r in rows if a < int(r['Start_Timestamp']) < b]
print('copies in the last step:', len(sel))
for r in sel: print(r.get('Direction'), r.get('Size', r.get('Bytes')), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us at', (int(r['Start_Timestamp'])-a)/1e3)
PY
