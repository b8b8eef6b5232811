#!/bin/bash
# GPU call 34 of round 4: which of the keys measured before `scst` slows it down (109-111 ms after all of them, 102-104 right after the TF key)
run() { echo -n "$1: "; shift; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scst', round(d['scst']['ms_per_step'],2), 'decode', round(d['scst']['roofline']['decode_ms_per_step'],2), [k for k in d if k in ('forward_only','tf_single','tf_dropin')])"; }
run "no dropin (forward_only + tf_single first)" --no-dropin
run "all extras                                " 
run "no extras                                 " --no-extras
