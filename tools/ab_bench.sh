#!/bin/bash
# same-box A/B of the headline step of two trees (this one and a git worktree, e.g. _ab_r4): rocprofv3 kernel tables of
# `bench.py --no-launch-timing --steps 20` in each, then the per-kernel difference in ms per step.   tools/ab_bench.sh _ab_r4
other=${1:-_ab_r4}; out=gpurun_out/ab_bench; mkdir -p $out; export TMPDIR=/tmp; root=$PWD
for d in $other .; do
  tag=$(echo $d | tr -d './_'); tag=${tag:-cur}
  (cd $d && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$tag -o p -- python3 bench.py --no-launch-timing --steps 20 --no-cpu-baseline --no-secondary --no-fp32-side --no-rccl-side > /dev/null 2>&1)
  f=$(find $out/prof_$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/stats_$tag.csv; rm -rf $out/prof_$tag
done
python - <<EOF
import csv, glob
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        d[r['Name']]=(int(r['Calls']), float(r['TotalDurationNs'])/1e6/25)
    return d
fs=sorted(glob.glob('$out/stats_*.csv')); a,b=load(fs[0]),load(fs[1]); print(fs)
rows=[]
for k in set(a)|set(b):
    ca,ta=a.get(k,(0,0.0)); cb,tb=b.get(k,(0,0.0)); rows.append((tb-ta,k,ca,ta,cb,tb))
rows.sort(key=lambda r:-abs(r[0]))
print('ms/step', sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
for d,k,ca,ta,cb,tb in rows[:22]: print('%+7.3f  %-64s %5d %7.3f | %5d %7.3f'%(d,k[:64],ca,ta,cb,tb))
EOF
