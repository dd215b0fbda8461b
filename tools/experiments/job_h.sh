set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "reserve or forward_only or input_bound or recurrent_precision" > gpurun_out/r05_h_newtests.log 2>&1
tail -3 gpurun_out/r05_h_newtests.log
for m in cfg1 warm cfg2; do python tools/experiments/first_launch.py $m 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05_h_first_launch.txt
for ns in 2 8; do
NABU_SPELLER_STREAMS=$ns python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-gemm-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 streams $ns', d['ms_per_step'])"
done | tee gpurun_out/r05_h_streams.txt
python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-gemm-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 default', d['ms_per_step'])" | tee -a gpurun_out/r05_h_streams.txt
