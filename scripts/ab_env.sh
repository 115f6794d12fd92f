#!/bin/bash
# A/B of environment switches on the bench: bash scripts/ab_env.sh <out-prefix> <config> NAME=VAR=VALUE ...
# (each arm runs twice, interleaved, so that box drift shows)
out=$1; cfg=$2; shift 2
for rep in 1 2; do
  for arm in "$@"; do
    name=${arm%%=*}; kv=${arm#*=}
    env $kv python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu > ${out}_${cfg}_${name}_${rep}.json 2>> ${out}.err || exit 1
  done
done
