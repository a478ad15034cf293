export TMPDIR=/tmp
mkdir -p gpurun_out
T=r3g
# the window wave's own timeline (FS_SCOUT_PROFILE): slots = [fetch] waiting for request / serial stores, [states_chain] fetch, [ranks] solve,
# [rounds] waiting for the request behind a solved window, [writeback] check + prices + reply, [coder] write-back behind the reply, [windows_total] serial wave waiting for replies
FS_WAVES=3 FS_LIB=build/libfastore_amd_prof.so COPIES=1 timeout -k 10 120 python3 tools/ppmd_microbench.py 3000000 > gpurun_out/${T}_scoutprof_3M_w3.txt 2>&1; cat gpurun_out/${T}_scoutprof_3M_w3.txt
