# kernel statistics of the plain and the regularised cfg3 step side by side: gpurun -- bash tools/experiments/td_prof.sh -> gpurun_out/td_{plain,td}_stats.csv
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in plain td; do
  extra=""; [ $v = td ] && extra="--training-defaults"
  rm -rf /tmp/t_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/t_$v -o $v -- python $R/bench.py --workload cfg3 --no-cpu-baseline --no-alt --no-gemm-roofline --no-other-configs --repeats 1 --steps 4 --warmup 2 $extra > $R/gpurun_out/td_$v.log 2>&1
  cp $(find /tmp/t_$v -name "*kernel_stats.csv" | head -1) $R/gpurun_out/td_${v}_stats.csv
done
