#!/bin/bash
# Round-6 closing pass on the final tree: the default bench line, then the TF-step evidence that the last changes touch (in-step GEMM shapes, kernel-trace
# summaries of the TF step and of the forward, counter passes of the TF step). The decode evidence of scripts/r6/evidence.sh stands (kernels unchanged).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O
cd $R
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r06_bench_full_output.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r6f/r06_bench_full_output.json")
t = open(p).read(); d = json.loads(t[t.index('{"metric"'):])
s = d['scst']
print('tf', round(d['ms_per_step'], 2), round(d['value']), 'frac', round(d['roofline']['frac'], 4), 'achieved', round(d['roofline']['achieved'], 1), 'wgrad', round(d['roofline']['weight_grad_kernel']['achieved'], 1))
print('scst', s['headline_is'], round(s['ms_per_step'], 2), round(s['value'], 3), 'synthetic', round(s['synthetic_ids']['ms_per_step'], 2), 'r512', s['string_round_trip'].get('r512', {}).get('ms_per_step'), 'us/tok', round(s['roofline']['us_per_token_step'], 1), 'frac', round(s['roofline']['frac'], 4))
print('encoder', s.get('encoder_forward_ms'), {k: round(s['encoder_roofline'][k], 4) for k in ('achieved', 'frac')}, 'label', round(s['synthetic_ids']['label_forward']['label_forward_ms'], 2))
for k in ('forward_only', 'tf_single', 'tf_dropin', 'scst_dropin', 'scst_c5', 'beam_generation', 'cpu_baseline'):
    v = d.get(k, {}); print(k, {kk: (round(v[kk], 3) if isinstance(v[kk], float) else v[kk]) for kk in ('value', 'ms_per_step', 'ms', 'frac', 'error', 'ms_per_batch', 'us_per_token_step', 'encoder_forward_ms') if kk in v})
PY
timeout 300 python scripts/gemm_profile.py > $O/r06_gemm_shapes_instep.txt 2>/dev/null; echo shapes $?
cd /tmp && export TMPDIR=/tmp
TF="bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tf_prof -- python3 $R/$TF --steps 20 --warmup 5 > $O/tf_prof.log 2>&1; echo tf_trace $?
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd_prof -- python3 $R/scripts/fwd_profile.py 10 > $O/fwd_prof.log 2>&1; echo fwd_trace $?
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tf_fetch -- python3 $R/$TF --steps 3 --warmup 1 > $O/tf_fetch.log 2>&1; echo tf_fetch $?
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tf_write -- python3 $R/$TF --steps 3 --warmup 1 > $O/tf_write.log 2>&1; echo tf_write $?
cd $R
for n in tf fwd; do f=$(ls $O/${n}_prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/r06_${n}_kernel_stats.csv; done
STEPS=$(python - <<PY
import csv, glob
f = sorted(glob.glob("$O/tf_fetch/**/*counter_collection.csv", recursive=True))[-1]
print(sum(1 for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("softmax_ce")))
PY
)
echo "optimiser steps in the counter run: $STEPS"
python scripts/pmc_traffic.py $O/tf_fetch $O/tf_write $STEPS "python3 $TF --steps 3 --warmup 1 (units = optimiser steps in the run, untimed pre-steps included)" > $O/r06_pmc_tf_hbm_traffic.json
python scripts/pmc_kernel_table.py $O/tf_fetch $O/tf_write $STEPS > $O/r06_pmc_tf_kernel_table.txt 2>&1
rm -rf $O/tf_fetch $O/tf_write $O/tf_prof $O/fwd_prof
head -8 $O/r06_pmc_tf_kernel_table.txt; head -12 $O/r06_gemm_shapes_instep.txt | cut -c1-120; ls -la $O
