# round-3 counter passes of the cached decode (eager launches: hipGraph replays under --pmc take tens of minutes); every command bounded
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r3prof; mkdir -p $O
export CXR_PROFILE_EAGER=1
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/dec_fetch -- python3 $R/scripts/scst_decode_profile.py 1 > /dev/null 2>&1; echo dec_fetch $?
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/dec_write -- python3 $R/scripts/scst_decode_profile.py 1 > /dev/null 2>&1; echo dec_write $?
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c5_fetch -- python3 $R/scripts/scst_c5_decode_profile.py 1 > /dev/null 2>&1; echo c5_fetch $?
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c5_write -- python3 $R/scripts/scst_c5_decode_profile.py 1 > /dev/null 2>&1; echo c5_write $?
unset CXR_PROFILE_EAGER
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 $R/scripts/scst_c5_decode_profile.py 2 > /dev/null 2>&1; echo c5_stats $?
cd $R
python scripts/pmc_traffic.py $O/dec_fetch $O/dec_write 255 "CXR_PROFILE_EAGER=1 python3 scripts/scst_decode_profile.py 1 (units = 255 token-steps of one 32-row decode; prefill and encoder launches are in the per-family totals)" > $O/r03_pmc_decode_hbm_traffic.json
python scripts/pmc_traffic.py $O/c5_fetch $O/c5_write 255 "CXR_PROFILE_EAGER=1 python3 scripts/scst_c5_decode_profile.py 1 (units = 255 token-steps of one 32-row decode, 16 studies x 3 images, 128-token prompt)" > $O/r03_pmc_decode_c5_hbm_traffic.json
cp $(ls $O/c5/*/*kernel_stats.csv | tail -1) $O/r03_scst_c5_decode_kernel_stats.csv
rm -rf $O/dec_fetch $O/dec_write $O/c5_fetch $O/c5_write $O/c5/*/*kernel_trace.csv
ls -la $O | tail -8
