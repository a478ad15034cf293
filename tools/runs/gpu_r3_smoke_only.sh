export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/r03_smoke_last.log 2>&1; echo "exit $?"; tail -2 gpurun_out/r03_smoke_last.log
