#!/bin/bash
cd /root/repo
b() { python bench.py --steps 48 --warmup 0 --no-pmc --cpu-seconds 0 --depth $1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['counters']; k=d['kernel_time_ms']
rays=c['numRays']+c['numShadowRays']
print('depth $1: %.1f Msamples/s; rays %d shadow %d; trace %.1f ms -> %.2f Grays/s in k_trace; shade %.1f ms' % (d['value'], c['numRays'], c['numShadowRays'], k['trace'], rays/k['trace']/1e6, k['shade']))"; }
for d in 0 1 2 4 8; do b $d; done
