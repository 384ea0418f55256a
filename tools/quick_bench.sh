#!/bin/bash
# usage: tools/quick_bench.sh <label> [bench args...]   (env vars pass through)
label=$1; shift
python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
t=d.get('twin')
print('%-28s %-26s relu %.2f Gs/s %.3f ms frac %.3f%s' % ('$label', d['config']['workload'].split(':')[0], d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], (' | twin %.2f Gs/s' % (t['value']/1e9)) if t else ''))"
