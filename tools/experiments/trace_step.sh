# kernel timeline of ONE cfg2 training step (rocprofv3 --kernel-trace): gpurun -- bash tools/experiments/trace_step.sh -> gpurun_out/trace_step.txt
cd /tmp; export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
rm -rf /tmp/tr
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -o t -- python $ROOT/bench.py --no-cpu-baseline --no-alt --no-gemm-roofline --no-other-configs --repeats 1 --steps 2 --warmup 2 > $ROOT/gpurun_out/trace_step.log 2>&1
python - <<PY
import csv, glob
rows=[]
for f in glob.glob('/tmp/tr/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70], r.get('Grid_Size_X','')))
for f in glob.glob('/tmp/tr/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'MEMCPY '+r.get('Direction','')+' '+str(r.get('Bytes','')), ''))
rows.sort()
# last step: find last adam_clip; print from the previous adam_clip end to the last
idx=[i for i,r in enumerate(rows) if 'adam_clip' in r[2]]
a,b=idx[-2]+1, idx[-1]+1
out=open('$ROOT/gpurun_out/trace_step.txt','w')
prev=rows[a-1][1]
for s,e,n,g in rows[a:b]:
    out.write('%9.1f gap %6.1f dur %8.1f  %s %s\n'%((s-rows[a-1][1])/1e3,(s-prev)/1e3,(e-s)/1e3,n,g)); prev=max(prev,e)
out.close()
PY
