set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/tr5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr5 -o cfg5 -- python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-gemm-roofline > gpurun_out/r05_p_cfg5_trace.log 2>&1
cp $(find /tmp/tr5 -name "*kernel_trace.csv" | head -1) /tmp/cfg5_trace.csv
cp $(find /tmp/tr5 -name "*kernel_stats.csv" | head -1) gpurun_out/r05_p_cfg5_kernel_stats.csv
python tools/trace_step.py /tmp/cfg5_trace.csv --summary > gpurun_out/r05_p_cfg5_trace_summary.txt 2>&1
python - <<'PY' > gpurun_out/r05_p_cfg5_segments.txt 2>&1
import csv, re
rows=list(csv.DictReader(open('/tmp/cfg5_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adam_clip' in r['Kernel_Name']]
step=rows[idx[-2]+1:idx[-1]+1]
t0=int(step[0]['Start_Timestamp'])
def short(n): return re.sub(r'\(.*','',n).replace('void ','').replace('nabu::','')[:50]
prev=None; start=None; cnt=0
for r in step:
    k=short(r['Kernel_Name'])
    cls='dec_bwd' if k.startswith(('attn_bwd','gemm_skinny_fused','rows16_kernel')) else k
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    if cls!=prev:
        if prev is not None: print('%10.1f .. %10.1f  %5d x %s'%(start,last_e,cnt,prev))
        prev=cls; start=s; cnt=0
    cnt+=1; last_e=e
print('%10.1f .. %10.1f  %5d x %s'%(start,last_e,cnt,prev))
PY
head -30 gpurun_out/r05_p_cfg5_trace_summary.txt
