set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r05_i_gputests.log 2>&1
tail -3 gpurun_out/r05_i_gputests.log
