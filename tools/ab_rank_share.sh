#!/bin/bash
# A/B of a rank's share of the C4 genome under environment switches, each configuration several times, interleaved:
#   bash tools/ab_rank_share.sh <out.txt> <world> <rank> "NAME=VALUE ..." "NAME2=VALUE ..." ...   ("-" = no switch)
out=$1; world=$2; rank=$3; shift 3
: > $out
for rep in 1 2 3; do
  for cfg in "$@"; do
    envs=""; [ "$cfg" != "-" ] && envs="$cfg"
    echo -n "[$cfg] " >> $out
    env $envs CS_STEPS=124 python tools/time_rank_share.py $world $rank 2>/dev/null | tail -1 >> $out
  done
done
cat $out
