set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_speller.py tests/test_hip_golden.py -x -q -m gpu 2>&1 | tail -2
for wl in cfg5 cfg3; do
python bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-alt --no-gemm-roofline 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d['final_loss'])"
done
