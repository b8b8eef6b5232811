R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r3prof2; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python3 $R/scripts/scst_decode_profile.py 2 > /dev/null 2>&1; echo c3_stats $?
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 $R/scripts/scst_c5_decode_profile.py 2 > /dev/null 2>&1; echo c5_stats $?
cp $(ls $O/c3/*/*kernel_stats.csv | tail -1) $O/r03_scst_decode_kernel_stats.csv
cp $(ls $O/c5/*/*kernel_stats.csv | tail -1) $O/r03_scst_c5_decode_kernel_stats.csv
rm -rf $O/c3 $O/c5
cd $R
[ -n "$NOBENCH" ] || { timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo bench $?; }
