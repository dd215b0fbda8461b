set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_speller.py -x -q -m gpu 2>&1 | tail -2
python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-gemm-roofline 2> gpurun_out/r05_k_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5', d['ms_per_step'], d['final_loss'])"
grep "attn_bwd stamps" gpurun_out/r05_k_err.txt
NABU_ATTN_BWD_MFMA=0 python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-gemm-roofline 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 old kernel', d['ms_per_step'], d['final_loss'])"
true
