#!/bin/bash
# A strong-scaled rank's share (shard 0 of N) of dragon 1080p x 1024 spp over slot-pool sizes.
set -u
for n in 8 4 2; do
for slots in 0 8388608 4194304 2097152 1048576; do
  echo "shards $n slots $slots:"
  SHARDS=$n SPS=8 SLOTS=$slots timeout -k 10 120 python tools/probe_shard8.py 2>&1 | grep steps || exit 5
done
done
