#!/bin/bash
# round 4, GPU session 5: the quarantine / re-homing placement -- GPU tests touching it, the policy trial on the new
# code, default bench lines (memory held after placement), padded vs contiguous in one process
export TMPDIR=/tmp
O=gpurun_out/r04_s5; mkdir -p $O
( timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_integration_stub.py -q -m gpu --maxfail=10 -k "sliding or outlives or output_planes or budget or rccl or plain or two_ranks or device_batch_and_synth or stub or c_example or headline" ) > $O/pytest_subset.log 2>&1
tail -15 $O/pytest_subset.log
( timeout 1500 python3 tests/vmm_policy_trial.py --cases 80 ) > $O/vmm_policy_trial.json 2> $O/vmm_policy_trial.err
python3 - <<PY
import json
d=json.load(open('$O/vmm_policy_trial.json'))
for r in d['results']: print(r.get('policy'), r.get('what'), r.get('cases'), r.get('cases_with_a_wrong_layer'), r.get('address_space'), r.get('error'))
PY
for i in 1 2 3; do ( timeout 1200 python3 bench.py --no-cpu-baseline --no-single-tile ) > $O/bench_default_$i.json 2> $O/bench_default_$i.err; done
python3 - <<PY
import json
for i in (1,2,3):
    try:
        d=json.loads([l for l in open('$O/bench_default_%d.json' % i) if l.startswith('{"metric"')][-1])
        print(i, d['value'], d['roofline']['frac'], d['roofline'].get('frac_first_come_placement'), d['roofline'].get('frac_kept_placement_probe'), d['roofline'].get('realloc_spread'), d['config']['arena_placement'].get('positions'), d['host_path']['zero_copy_Gpx_s'], d['host_path']['pageable_Gpx_s'])
    except Exception as e: print(i, 'failed', e)
PY
