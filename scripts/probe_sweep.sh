#!/bin/bash
# Dev: hand-off probe variants (steps qstores sleep mode nodrain shards replicas).  Build first:
#   hipcc --offload-arch=gfx950 -O3 -o scripts/persist_probe scripts/persist_probe.hip
cd $(dirname $0)
for cfg in "1 1 0 1 1" "1 1 0 4 1" "1 1 0 16 1" "1 1 0 1 8" "1 1 0 4 8" "1 1 0 16 8" "1 1 1 16 8" "1 1 1 8 4" "1 1 1 16 2" "1 1 1 32 1" "1 1 1 64 1"; do
  set -- $cfg
  echo "=== sleep $1 mode $2 nodrain $3 shards $4 reps $5"
  ./persist_probe 2000 2 $1 $2 $3 $4 $5 | grep -E "rep 1|poll|publish|load"
done
