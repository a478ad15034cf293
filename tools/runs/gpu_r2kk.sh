export TMPDIR=/tmp
mkdir -p gpurun_out
T=r2kk
TIMEFORMAT="   process wall %R s"
G=tests/golden/se_lossless.in
run() { ( time FS_TRACE=1 ./fastore_amd/fastore_pack e -i$G -o/tmp/o_$1 -r -f256 -c10 -d8 -w1024 -W1024 ) 2>&1 | grep -v "slice\|batch:\|matcher so\|route\|flush\|close inputs" | cut -c1-200; echo; }
{
echo "== back to back"; run a; run b; run c
echo "== after 6 s idle"; sleep 6; run d
echo "== after 15 s idle"; sleep 15; run e
echo "== FS_MAX_WAVES=512 (9 GB pool) back to back"; export FS_MAX_WAVES=512; run f; run g
} > gpurun_out/${T}_startup.txt 2>&1
cat gpurun_out/${T}_startup.txt
