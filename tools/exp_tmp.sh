cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/exp26_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/exp26_pytest.log
tail -3 gpurun_out/exp26_pytest.log
SLICES=8 bash tools/ab_variants.sh exp26 default build/variants/libfs_argsregs.so default build/variants/libfs_argsregs.so
