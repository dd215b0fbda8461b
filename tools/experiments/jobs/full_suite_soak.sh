set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r05_s_gputests.log 2>&1
tail -2 gpurun_out/r05_s_gputests.log
python tools/experiments/soak.py cfg5 600 2>&1 | grep -v amdgpu | tail -1
python tools/experiments/soak.py cfg3 1000 2>&1 | grep -v amdgpu | tail -1
python bench.py --workload cfg5 --training-defaults --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-gemm-roofline 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 training-defaults', d['ms_per_step'], d['final_loss'])"
python bench.py --workload cfg3 --training-defaults --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-gemm-roofline 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 training-defaults', d['ms_per_step'], d['final_loss'])"
