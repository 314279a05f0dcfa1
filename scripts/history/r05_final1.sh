R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_final_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r05_final_tests.log
[ $rc -ne 0 ] && exit $rc
bash scripts/history/collect_r05.sh profiles && bash scripts/history/collect_r05.sh fused
