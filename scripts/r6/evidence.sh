#!/bin/bash
# Round-6 evidence pass (one gpurun call): in-step GEMM shapes, kernel-trace summaries (TF step, forward only, SCST decode), HBM-side traffic of the TF step and of the
# cached decode from separate counter passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md: x2 on FETCH_SIZE for 16-byte-per-lane loads)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O
cd $R
timeout 300 python scripts/gemm_profile.py > $O/r06_gemm_shapes_instep.txt 2>/dev/null; echo shapes $?
cd /tmp && export TMPDIR=/tmp
TF="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tf_prof -- python3 $R/$TF --steps 20 --warmup 5 > $O/tf_prof.log 2>&1; echo tf_trace $?
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd_prof -- python3 $R/scripts/fwd_profile.py 10 > $O/fwd_prof.log 2>&1; echo fwd_trace $?
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec_prof -- python3 $R/scripts/scst_decode_profile.py > $O/dec_prof.log 2>&1; echo dec_trace $?
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_prof -- python3 $R/scripts/scst_c5_decode_profile.py > $O/c5_prof.log 2>&1; echo c5_trace $?
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tf_fetch -- python3 $R/$TF --steps 3 --warmup 1 > $O/tf_fetch.log 2>&1; echo tf_fetch $?
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tf_write -- python3 $R/$TF --steps 3 --warmup 1 > $O/tf_write.log 2>&1; echo tf_write $?
export CXR_PROFILE_EAGER=1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/dec_fetch -- python3 $R/scripts/scst_decode_profile.py 1 > /dev/null 2>&1; echo dec_fetch $?
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/dec_write -- python3 $R/scripts/scst_decode_profile.py 1 > /dev/null 2>&1; echo dec_write $?
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c5_fetch -- python3 $R/scripts/scst_c5_decode_profile.py 1 > /dev/null 2>&1; echo c5_fetch $?
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c5_write -- python3 $R/scripts/scst_c5_decode_profile.py 1 > /dev/null 2>&1; echo c5_write $?
unset CXR_PROFILE_EAGER
cd $R
for n in tf fwd dec c5; do f=$(ls $O/${n}_prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/r06_${n}_kernel_stats.csv; done
STEPS=$(python - <<PY
import csv, glob
f = sorted(glob.glob("$O/tf_fetch/**/*counter_collection.csv", recursive=True))[-1]
# one softmax-CE launch per optimiser step (AdamW is THREE launches per step on one rank: the round-4 / round-5 scripts divided its count by two and
# so reported 2/3 of the true bytes per step)
print(sum(1 for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("softmax_ce")))
PY
)
echo "optimiser steps in the counter run: $STEPS"
python scripts/pmc_traffic.py $O/tf_fetch $O/tf_write $STEPS "python3 $TF --steps 3 --warmup 1 (units = optimiser steps in the run, untimed pre-steps included)" > $O/r06_pmc_tf_hbm_traffic.json
python scripts/pmc_kernel_table.py $O/tf_fetch $O/tf_write $STEPS > $O/r06_pmc_tf_kernel_table.txt 2>&1
python scripts/pmc_traffic.py $O/dec_fetch $O/dec_write 255 "CXR_PROFILE_EAGER=1 python3 scripts/scst_decode_profile.py 1 (units = 255 token-steps of one 32-row decode; prefill and encoder launches are in the per-family totals)" > $O/r06_pmc_decode_hbm_traffic.json
python scripts/pmc_traffic.py $O/c5_fetch $O/c5_write 255 "CXR_PROFILE_EAGER=1 python3 scripts/scst_c5_decode_profile.py 1 (units = 255 token-steps of one 32-row decode, 16 studies x 3 images, 128-token prompt)" > $O/r06_pmc_decode_c5_hbm_traffic.json
rm -rf $O/tf_fetch $O/tf_write $O/dec_fetch $O/dec_write $O/c5_fetch $O/c5_write $O/tf_prof $O/fwd_prof $O/dec_prof $O/c5_prof
head -14 $O/r06_pmc_tf_kernel_table.txt; head -30 $O/r06_gemm_shapes_instep.txt
