#!/bin/bash
# Round-4 evidence pass: per-shape in-step GEMM table, HBM-side traffic of the TF step from two separate counter passes (FETCH_SIZE, WRITE_SIZE)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; mkdir -p $O
cd $R
timeout 300 python scripts/gemm_profile.py > $O/r04_gemm_shapes_instep.txt 2>/dev/null; echo shapes $?
cd /tmp && export TMPDIR=/tmp
CMD="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 3 --warmup 1"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tf_fetch -- python3 $R/$CMD > $O/tf_fetch.log 2>&1; echo fetch $?
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tf_write -- python3 $R/$CMD > $O/tf_write.log 2>&1; echo write $?
cd $R
STEPS=$(python - <<PY
import csv, glob
f = sorted(glob.glob("$O/tf_fetch/**/*counter_collection.csv", recursive=True))[-1]
print(sum(1 for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("adamw_kernel")) // 2)
PY
)
echo "optimiser steps in the counter run: $STEPS"
python scripts/pmc_traffic.py $O/tf_fetch $O/tf_write $STEPS "python3 $CMD (units = optimiser steps in the run, untimed pre-steps included)" > $O/r04_pmc_tf_hbm_traffic.json
python scripts/pmc_kernel_table.py $O/tf_fetch $O/tf_write $STEPS > $O/r04_pmc_tf_kernel_table.txt 2>&1
rm -rf $O/tf_fetch $O/tf_write
head -12 $O/r04_pmc_tf_kernel_table.txt; python -c "
import json; d=json.load(open('$O/r04_pmc_tf_hbm_traffic.json')); print({k: round(v['hbm_bytes_per_launch']/1e6,1) for k,v in d.items() if isinstance(v,dict) and 'hbm_bytes_per_launch' in v})"
