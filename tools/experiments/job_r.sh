set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_speller.py -x -q -m gpu 2>&1 | tail -1
for m in 1 0 1 0; do
NABU_SPELLER_MERGE_FINISH=$m python bench.py --workload cfg5 --steps 8 --warmup 2 --no-cpu-baseline --no-alt --no-gemm-roofline 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 merge=$m', d['ms_per_step'], d['final_loss'])"
done
export TMPDIR=/tmp
rm -rf /tmp/tj
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tj -o cfg5 -- python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-gemm-roofline > gpurun_out/r05_r_trace.log 2>&1
grep -E "finish|rows16|attn_bwd_loc" $(find /tmp/tj -name "*kernel_stats.csv" | head -1) | cut -c1-140
