# Round 4, call 8: the full -m gpu suite on the current build, then the round's collection (scripts/collect_r04.sh).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04/pytest.log
bash scripts/collect_r04.sh
