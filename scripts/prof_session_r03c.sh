R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r3prof3; mkdir -p $O
cd $R
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tf -- python3 bench.py --no-scst --no-extras --no-cpu-baseline --no-dropin --steps 20 --warmup 5 > $O/tf.log 2>&1; echo tf $?
cp $(ls $O/tf/*/*kernel_stats.csv | tail -1) $O/r03_bench_tf_step_2image_kernel_stats.csv
rm -rf $O/tf
tail -1 $O/tf.log | cut -c1-300
