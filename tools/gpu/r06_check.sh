#!/bin/bash
# round 6: the whole GPU suite (-x) with the margin log + the full bench line (bf16 + the two fp32-storage child runs + cpu baseline)
out=gpurun_out/${1:-r06check}; mkdir -p $out
export CMDA_TEST_MARGINS=$PWD/$out/margins.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $out/gputest.log 2>&1; grep -E "passed|failed" $out/gputest.log | tail -2; grep -E "^FAILED|Error" $out/gputest.log | head -5
unset CMDA_TEST_MARGINS
python bench.py > $out/bench.json 2> $out/err_bench; cut -c1-300 $out/bench.json; python -c "
import json;d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]);print({k:d.get(k) for k in ('ms_per_step','x3_ms_per_step','exact_f32_ms_per_step')});print(d['roofline'].get('mit_blocks'));print(d['roofline'].get('frac'), d['roofline'].get('frac_rocprof'));print(d.get('accuracy',{}).get('bf16'))"
