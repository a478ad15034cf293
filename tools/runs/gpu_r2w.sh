export TMPDIR=/tmp
cd /tmp && cd - > /dev/null
bash tools/profile_round.sh r02z > gpurun_out/r02z_profile_round.log 2>&1
tail -3 gpurun_out/r02z_profile_round.log | cut -c1-300
bash tools/pmc_single_stream.sh r02z 3000000 > gpurun_out/r02z_single_stream.log 2>&1
tail -4 gpurun_out/r02z_single_stream.log | cut -c1-300
ls gpurun_out | head -40
