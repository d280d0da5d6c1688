#!/bin/bash
# Rehearsal of the multi-process bench path on a ONE-GPU box (run through gpurun): N ranks share GPU 0,
# the library binds tests/cpp/shm_transport.cpp instead of librccl (RCCL refuses several ranks per
# device), torch.distributed runs on gloo.  Checks the code path end to end -- bootstrap, shared tuning,
# K-row exchanges, both scaling modes, the JSON line -- the numbers mean nothing.
#   tools/rehearsal.sh [steps]
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/rehearsal
mkdir -p "$OUT"
cd "$ROOT"
STEPS=${1:-60}
hipcc -O2 -fPIC -shared -std=c++17 -x hip --offload-arch=gfx950 tests/cpp/shm_transport.cpp -o "$OUT/libshm_transport.so" -lrt -lpthread
export GS_RCCL_LIBRARY=$OUT/libshm_transport.so HSA_ENABLE_IPC_MODE_LEGACY=0
port=29611
for mode in "2 weak" "4 weak" "2 strong" "4 strong"; do
  set -- $mode
  port=$((port + 1))
  timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$1" --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus "$1" --steps "$STEPS" --warmup 12 --scaling "$2" --rehearsal > "$OUT/bench_n$1_$2.json" 2> "$OUT/bench_n$1_$2.log" \
      || { tail -20 "$OUT/bench_n$1_$2.log"; exit 1; }
  tail -1 "$OUT/bench_n$1_$2.json"
done
