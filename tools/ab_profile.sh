#!/bin/bash
# same-box A/B of two source trees (this one and a git worktree with its own built library, e.g. _ab_r4): throughput of the
# secondary configs, then a rocprofv3 kernel table of one config in each tree.     tools/ab_profile.sh _ab_r4 c4
other=${1:-_ab_r4}; cfg=${2:-c4}; out=gpurun_out/ab_$cfg; mkdir -p $out
export TMPDIR=/tmp
for d in $other . $other .; do
  (cd $d && STEPS=10 python tools/bench_configs.py c1 $cfg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    if 'value' in j: print('$d', j['config'][:44], j['value'], j['ms_per_step'])")
done
root=$PWD
for d in $other .; do
  tag=$(echo $d | tr -d './_'); tag=${tag:-cur}
  (cd $d && STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$tag -o p -- python3 tools/bench_configs.py $cfg > /dev/null 2>&1)
  f=$(find $out/prof_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $out/stats_$tag.csv
  rm -rf $out/prof_$tag
done
python - <<EOF
import csv
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        d[r['Name']]=(int(r['Calls']), float(r['TotalDurationNs'])/1e6)
    return d
import glob
fs=sorted(glob.glob('$out/stats_*.csv'))
a,b=load(fs[0]),load(fs[1])
print(fs)
rows=[]
for k in set(a)|set(b):
    ca,ta=a.get(k,(0,0.0)); cb,tb=b.get(k,(0,0.0))
    rows.append((tb-ta,k,ca,ta,cb,tb))
rows.sort(key=lambda r:-abs(r[0]))
print('total ms', sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
for d,k,ca,ta,cb,tb in rows[:25]:
    print('%+9.2f ms  %-70s %6d %9.2f | %6d %9.2f'%(d,k[:70],ca,ta,cb,tb))
EOF
