#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for lib in "" ${NABU_ALT_LIB:-}; do
  NABU_LIB=$lib timeout 120 python tools/experiments/ring_bench.py 2>&1 | tail -1
done
cd /tmp; export TMPDIR=/tmp
for lib in "" ${NABU_ALT_LIB:-}; do
  rm -rf /tmp/q
  (cd $ROOT && NABU_LIB=$lib NSTEPS=1 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/q -o w -- python tools/experiments/ring_bench.py > /tmp/q.log 2>&1)
  python - <<PY
import csv, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open('$(find /tmp/q -name "*counter_collection.csv" | head -1)')):
    if 'persist' in r['Kernel_Name']:
        a = agg[r['Kernel_Name'][:40]]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in agg.items(): print('lib [$lib]', k, n, 'launches', 'WRITE_SIZE per launch %.0f KB' % (v / n))
PY
done
