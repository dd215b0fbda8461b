set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_speller.py -x -q -m gpu 2>&1 | tail -3
python bench.py --workload cfg5 --training-defaults --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-gemm-roofline 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 training-defaults', d['ms_per_step'], d['final_loss'])"
python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-gemm-roofline 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5', d['ms_per_step'], d['final_loss'])"
