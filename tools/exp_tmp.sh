cd $GRAFT_REPO_ROOT
FS_WATCHDOG=60 timeout 420 python3 -m pytest tests -m gpu -x -q > gpurun_out/r01h_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r01h_pytest.log
tail -3 gpurun_out/r01h_pytest.log
timeout 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
