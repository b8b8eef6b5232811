#!/bin/bash
# Round 6: the two counter passes of the TF step again, with the optimiser steps of the run counted by the softmax-CE launches (one per step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TF="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tf_fetch -- python3 $R/$TF --steps 3 --warmup 1 > $O/tf_fetch.log 2>&1; echo tf_fetch $?
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tf_write -- python3 $R/$TF --steps 3 --warmup 1 > $O/tf_write.log 2>&1; echo tf_write $?
cd $R
STEPS=$(python - <<PY
import csv, glob
f = sorted(glob.glob("$O/tf_fetch/**/*counter_collection.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
print(sum(1 for r in rows if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("softmax_ce")))
PY
)
echo "optimiser steps in the counter run: $STEPS"
python scripts/pmc_traffic.py $O/tf_fetch $O/tf_write $STEPS "python3 $TF --steps 3 --warmup 1 (units = optimiser steps in the run = softmax-CE launches, untimed pre-steps included)" > $O/r06_pmc_tf_hbm_traffic.json
python scripts/pmc_kernel_table.py $O/tf_fetch $O/tf_write $STEPS > $O/r06_pmc_tf_kernel_table.txt 2>&1
rm -rf $O/tf_fetch $O/tf_write
head -16 $O/r06_pmc_tf_kernel_table.txt; tail -1 $O/r06_pmc_tf_kernel_table.txt
