#!/bin/bash
# the headline only: python bench.py without the extra measurements (one line: value, ms/step, one-in-flight value, BA ms)
python bench.py --no-config4-step --no-live-dropin --no-cpu-baseline --no-plain-schedule --no-config4 --no-reference-pipeline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels_ms_per_step',{})
print('value %.0f  ms/step %.3f  one-in-flight %.0f  BA %.3f  orb %.3f' % (d['value'], d['ms_per_step'], d['config']['extras'].get('one_batch_in_flight_keyframes_per_s',0), k.get('lm_window_kernel',0), sum(v for n,v in k.items() if n.startswith('orb_'))))"
