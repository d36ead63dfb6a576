#!/bin/bash
# the pre kernel's register budget on the workloads it matters for
for i in 1 2; do
for wl in lsystem chain8 cppn_hardcore chain4; do
timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --no-secondary > /tmp/x.json 2>/dev/null; python3 -c "
import json; d=json.load(open('/tmp/x.json')); print('$wl %.2fM' % (d['value']/1e6))"; done; done
