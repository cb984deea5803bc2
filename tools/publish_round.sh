#!/bin/bash
# Copies what tools/collect_round.sh left under gpurun_out/round/ into profiles/round${ROUND}_* and rebuilds the derived tables.
set -e
ROUND=${ROUND:-5}
# steps inside one profiled `bench.py --steps 20 --warmup 5` run: graph mode (round 5) = 4 eager warm-up + 1 replay + 20 timed replays
# + 1 eager + 4 event-bracketed eager steps = 30; eager mode (rounds 1-4, --eager) = 25
NSTEPS=${NSTEPS:-30}
O=${O:-gpurun_out/round}
cp $O/bench.json profiles/round${ROUND}_bench.json
cp $O/per_layer.txt profiles/round${ROUND}_per_layer.txt
cp $O/prof_bench/p_kernel_stats.csv profiles/round${ROUND}_bench_kernel_stats.csv
cp $O/traffic.json profiles/round${ROUND}_traffic.json
for c in c4 c5 c5x; do cp $O/traffic_$c.json profiles/round${ROUND}_traffic_$c.json; done
cp $O/other_configs.jsonl profiles/round${ROUND}_other_configs.jsonl
python tools/summarize_profile.py $O/prof_bench/p_kernel_stats.csv $NSTEPS "round ${ROUND}: python3 bench.py --steps 20 --warmup 5 (ResNet-50 tile bag 64 bf16, fwd+bwd+Adam)" > profiles/round${ROUND}_bench_kernel_stats.md
python tools/summarize_profile.py $O/prof_c4/p_kernel_stats.csv 12 "round ${ROUND}: tools/bench_configs.py c4 (EfficientNet-B3 tile bag 64 bf16, BN train, fwd+bwd+Adam), 6 timed + 3 warm-up + 3 event-timed steps" > profiles/round${ROUND}_efficientnet_b3_kernel_stats.md
python tools/summarize_profile.py $O/prof_c5/p_kernel_stats.csv 9 "round ${ROUND}: tools/bench_configs.py c5 (ResNet-50 segment B=8 299x299 bf16, decoder training, Dice)" > profiles/round${ROUND}_c5_kernel_stats.md
python tools/summarize_profile.py $O/prof_c5x/p_kernel_stats.csv 9 "round ${ROUND}: tools/bench_configs.py c5x (ResNet-50 segment B=4 512x512 bf16, decoder training, Dice)" > profiles/round${ROUND}_c5_512_kernel_stats.md
python tools/roofline_c4_c5.py > /dev/null
python tools/check_bench_vs_profile.py profiles/round${ROUND}_bench.json profiles/round${ROUND}_bench_kernel_stats.csv $NSTEPS | tail -1
