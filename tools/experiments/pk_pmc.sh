#!/bin/bash
# Matrix-pipe utilisation, clock and wait breakdown of the packed GEMM kernels at the cfg2 shapes:
#   gpurun -- 'bash tools/experiments/pk_pmc.sh 2'      (argument: planes; NABU_PK_VAR is passed through)
# Two counter passes (kernel trace only), summaries printed and left under gpurun_out/pk_pmc/.
set -u
PL=${1:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pk_pmc
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/tools/experiments/gemm_pk_bench.py --planes $PL --fp32 0 --reps 3"
rm -rf /tmp/q1 /tmp/q2
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/q1 -o a -- $CMD > $OUT/mfma_p$PL.log 2>&1
python $ROOT/tools/pmc_mfma_summary.py $(find /tmp/q1 -name "*counter_collection.csv" | head -1) $OUT/mfma_p$PL.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/q2 -o b -- $CMD > $OUT/sq_p$PL.log 2>&1
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(glob.glob('/tmp/q2/**/*counter_collection.csv', recursive=True)[0])):
    agg[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    wc = v.get('SQ_WAVE_CYCLES', 0)
    if wc < 1e8: continue
    print(k[:60], {c: round(x / wc, 3) for c, x in v.items() if c != 'SQ_WAVE_CYCLES'})
PY
