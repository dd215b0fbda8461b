ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for wl in ${WLS:-cfg3}; do
rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $ROOT/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-gemm-roofline > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
echo "== $wl"; python $ROOT/tools/experiments/trace_overlap.py $f
done
