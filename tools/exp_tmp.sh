cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r01e_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r01e_pytest.log
tail -3 gpurun_out/r01e_pytest.log
timeout 1500 bash tools/profile_round.sh r01e
