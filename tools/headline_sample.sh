#!/bin/bash
# The default bench line twice on this box (the driver's command): one line per run appended to gpurun_out/<tag>_headline.txt
#   tools/headline_sample.sh <tag>
set -u
TAG=${1:-sample}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do
  python3 $R/bench.py 2>/dev/null | python3 -c "
import sys, json, socket
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); rl = r['roofline']; oc = r.get('other_configs', {})
print('$TAG run $i  value %.1f  ms_per_step %.4f  frac %.4f  ms_per_launch %.4f  sustained %.4f  cpu %.1f  | dsen2_60 %.1f (%.4f)  vdsen2_bf16 %.1f (%.4f)' % (
    r['value'], r['ms_per_step'], rl['frac'], rl['ms_per_launch'], rl['sustained_ms_per_step'], r['cpu_baseline']['value'],
    oc['dsen2_60_fp32']['value'], oc['dsen2_60_fp32']['roofline_frac'], oc['vdsen2_20_bf16']['value'], oc['vdsen2_20_bf16']['roofline_frac']))" >> $R/gpurun_out/${TAG}_headline.txt
done
cat $R/gpurun_out/${TAG}_headline.txt
