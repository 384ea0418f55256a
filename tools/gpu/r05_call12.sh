#!/bin/bash
# r05 GPU call 12: same-box A/B of the evaluate kernels: HEAD (two launches + list), the one-launch build, and its two switches off
O=gpurun_out/r05i; mkdir -p $O
for round in 1 2; do
  for v in nopf pf; do
    L=$PWD/fv-srn_amd/ablate/libfvsrn_$v.so; [ $v = main ] && L=$PWD/fv-srn_amd/libfvsrn.so
    for n in 4194304 16777216; do
      FVSRN_LIBRARY=$L timeout 300 python tools/bench_evaluate.py $n c32l4_fourier_relu c64l6_grid16_relu c64l6_fourier_relu c48l5_fourier_snakealt 2>> $O/err.txt | sed "s/^{/{\"lib\": \"$v\", /" >> $O/ab.jsonl
    done
  done
done
python - <<'PY'
import json, collections
r = collections.defaultdict(list)
for l in open("gpurun_out/r05i/ab.jsonl"):
    d = json.loads(l)
    r[(d["workload"].split(":")[1], d["points"].bit_length() - 1, d["lib"])].append(d["points_per_s"] / 1e9)
for k in sorted(r):
    print("%-28s 2^%d %-8s %s" % (k[0], k[1], k[2], " ".join("%7.2f" % v for v in r[k])))
PY
for v in main; do
  L=$PWD/fv-srn_amd/ablate/libfvsrn_$v.so; [ $v = main ] && L=$PWD/fv-srn_amd/libfvsrn.so
  FVSRN_LIBRARY=$L python bench.py --no-cpu-baseline 2>> $O/err.txt > $O/bench_$v.json
  python -c "import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value']/1e9, d['roofline']['frac'], d['twin']['value']/1e9)"
done
tail -3 $O/err.txt
