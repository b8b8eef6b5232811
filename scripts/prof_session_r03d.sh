R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r3prof4; mkdir -p $O
cd $R
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/tf -- python3 bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 8 --warmup 3 > $O/tf.log 2>&1; echo tf $?
python scripts/timeline.py $O/tf > $O/timeline.txt 2>&1
rm -rf $O/tf
