#!/bin/bash
# tools/sweep_shard.sh VAR v1 v2 ... : a rank's eighth of the dragon frame (tools/probe_shard8.py) with VAR set to each value
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
VAR=$1; shift
for v in "$@"; do
    export $VAR=$v
    echo "$VAR=$v: $(SHARDS=${SHARDS:-8} SPS=8 timeout -k 10 120 python tools/probe_shard8.py 2>&1 | grep ms | tail -2 | tr '\n' ' ')" || exit 1
done
