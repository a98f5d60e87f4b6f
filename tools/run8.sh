#!/bin/bash
# N bench.py processes on the N GPUs of one node WITHOUT torch.distributed: plain processes, the collectives are the C ABI's
# (lthip_comm_* = RCCL behind comm.hip), the communicator id travels through a file.
#   tools/run8.sh [N] [bench.py arguments...]        e.g.  tools/run8.sh 8 --scaling strong --steps 3 --warmup 1
# Rank 0 prints the JSON line.  (The driver's own launch -- python -m torch.distributed.run ... bench.py --gpus N -- keeps working, and
# `python bench.py --gpus N [--launch plain]` starts its ranks itself; this is the same thing as a shell loop.)  Every collective of
# the exchange -- the three all-gathers and the sharded first-seen table's all-to-all -- is lthip_comm_allgather / lthip_comm_alltoallv.
#   tools/run8.sh 8 --handshake-only                 10-second check that RCCL sees 8 ranks (id file, lthip_comm_create, reductions)
#   LTHIP_COMM_TRANSPORT=shm tools/run8.sh 2 ...     the same flow on a box with fewer GPUs (shared-memory stand-in for RCCL)
N=${1:-8}; shift
idfile=$(mktemp -u /tmp/lthip_comm_id.XXXXXX)
pids=()
for r in $(seq 0 $((N - 1))); do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=$N LONGTAIL_LAUNCH=plain LTHIP_COMM_ID_FILE=$idfile HSA_ENABLE_IPC_MODE_LEGACY=0 \
    python bench.py --gpus $N "$@" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
rm -f $idfile
exit $rc
