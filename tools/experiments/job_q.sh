set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_speller.py tests/test_hip_golden.py tests/test_hip_model.py -x -q -m gpu 2>&1 | tail -2
for m in 1 0; do
NABU_ATTN_GRADS_MFMA=$m python bench.py --workload cfg5 --steps 8 --warmup 2 --no-cpu-baseline --no-alt --no-gemm-roofline 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 grads_mfma=$m', d['ms_per_step'], d['final_loss'])"
done
export TMPDIR=/tmp
rm -rf /tmp/tj
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tj -o cfg5 -- python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-gemm-roofline > gpurun_out/r05_q_trace.log 2>&1
cp $(find /tmp/tj -name "*kernel_stats.csv" | head -1) gpurun_out/r05_q_cfg5_kernel_stats.csv
grep -E "param_grads|scatter_rows" gpurun_out/r05_q_cfg5_kernel_stats.csv | cut -c1-160
