#!/bin/bash
# N bench.py processes on the N GPUs of one node WITHOUT torch.distributed: plain processes, the collectives are the C ABI's
# (lthip_comm_* = RCCL behind comm.hip), the communicator id travels through a file.
#   tools/run8.sh [N] [bench.py arguments...]        e.g.  tools/run8.sh 8 --scaling strong --steps 3 --warmup 1
# Rank 0 prints the JSON line.  (The driver's own launch -- python -m torch.distributed.run ... bench.py --gpus N -- keeps working;
# this is the torch-free way to start the same measurement.)
N=${1:-8}; shift
idfile=$(mktemp -u /tmp/lthip_comm_id.XXXXXX)
pids=()
for r in $(seq 0 $((N - 1))); do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=$N LONGTAIL_LAUNCH=plain LTHIP_COMM_ID_FILE=$idfile HSA_ENABLE_IPC_MODE_LEGACY=0 \
    python bench.py --gpus $N --collective c "$@" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
rm -f $idfile
exit $rc
