#!/bin/bash
# one box: the eager step with / without the one-rank RCCL exchange, wall clock for several channel caps, then rocprofv3 tables of both
out=gpurun_out/rccl_side; mkdir -p $out; export TMPDIR=/tmp
for r in 0 1 0 1; do REDUCER=$r python3 tools/rccl_side_probe.py 30 2>/dev/null | grep reducer; done
for c in 4; do REDUCER=1 NCCL_MAX_NCHANNELS=$c python3 tools/rccl_side_probe.py 30 2>/dev/null | grep reducer; done
for r in 0 1; do
  REDUCER=$r rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$r -o p -- python3 tools/rccl_side_probe.py 20 > /dev/null 2>&1
  f=$(find $out/p$r -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/stats_$r.csv; rm -rf $out/p$r
done
python3 - <<EOF
import csv
def load(p):
    return {r['Name']:(int(r['Calls']), float(r['TotalDurationNs'])/1e6) for r in csv.DictReader(open(p))}
a,b=load('$out/stats_0.csv'),load('$out/stats_1.csv')
rows=sorted(((b.get(k,(0,0))[1]-a.get(k,(0,0))[1],k,a.get(k,(0,0)),b.get(k,(0,0))) for k in set(a)|set(b)), key=lambda r:-abs(r[0]))
print('total ms off/on', sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
for d,k,x,y in rows[:22]: print('%+8.2f ms %-72s %5d %8.2f | %5d %8.2f'%(d,k[:72],x[0],x[1],y[0],y[1]))
EOF
