#!/bin/bash
# Round profile set (run on the GPU box: gpurun -- 'bash tools/profile_round.sh r02').  Writes under
# gpurun_out/<tag>_prof/; the summaries that are judged are then copied into profiles/.
# Counter passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_prof
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-alt --no-gemm-roofline --no-other-configs --repeats 1"
stats() {   # name, args...
  local name=$1; shift
  rm -rf /tmp/p_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -o $name -- $BENCH "$@" > $OUT/${name}.log 2>&1
  cp $(find /tmp/p_$name -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${name}_kernel_stats.csv
}
pmc() {     # name, counters (space separated), args...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/q_$name
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/q_$name -o $name -- $BENCH "$@" > $OUT/${name}.log 2>&1
  cp $(find /tmp/q_$name -name "*counter_collection.csv" | head -1) $OUT/${name}_counters.csv
}
stats cfg2 --steps 3 --warmup 1
stats cfg3 --workload cfg3 --steps 3 --warmup 1
stats cfg5 --workload cfg5 --steps 2 --warmup 1
stats cfg1 --workload cfg1 --steps 5 --warmup 2
pmc cfg2_fetch "FETCH_SIZE" --steps 1 --warmup 1
pmc cfg2_write "WRITE_SIZE" --steps 1 --warmup 1
pmc cfg2_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --steps 1 --warmup 1
pmc cfg2_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" --steps 1 --warmup 1
for w in cfg3 cfg5; do
  pmc ${w}_fetch "FETCH_SIZE" --workload $w --steps 1 --warmup 1
  pmc ${w}_write "WRITE_SIZE" --workload $w --steps 1 --warmup 1
  pmc ${w}_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload $w --steps 1 --warmup 1
done
python $ROOT/tools/pmc_summary.py $OUT/cfg2_fetch_counters.csv $OUT/cfg2_write_counters.csv $OUT/${TAG}_cfg2_pmc_traffic.json > $OUT/cfg2_traffic.txt
python $ROOT/tools/pmc_mfma_summary.py $OUT/cfg2_mfma_counters.csv $OUT/${TAG}_cfg2_pmc_mfma.json > $OUT/cfg2_mfma.txt
for w in cfg3 cfg5; do
  python $ROOT/tools/pmc_summary.py $OUT/${w}_fetch_counters.csv $OUT/${w}_write_counters.csv $OUT/${TAG}_${w}_pmc_traffic.json > $OUT/${w}_traffic.txt
  python $ROOT/tools/pmc_mfma_summary.py $OUT/${w}_mfma_counters.csv $OUT/${TAG}_${w}_pmc_mfma.json > $OUT/${w}_mfma.txt
done
python - <<PY
import csv, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open('$OUT/cfg2_sq_counters.csv')):
    agg[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
out = {}
for k, v in agg.items():
    wc = v.get('SQ_WAVE_CYCLES', 0)
    if wc < 1e7: continue
    out[k] = {c: round(x / wc, 4) for c, x in v.items() if c != 'SQ_WAVE_CYCLES'}
    out[k]['SQ_WAVE_CYCLES'] = wc
json.dump(out, open('$OUT/${TAG}_cfg2_pmc_sq.json', 'w'), indent=1)
for k, v in out.items(): print(k[:70], v)
PY
rm -f $OUT/*_counters.csv    # raw per-dispatch rows: tens of MB
ls -la $OUT
