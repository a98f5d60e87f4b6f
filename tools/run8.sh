#!/bin/bash
# N bench.py processes on the N GPUs of one node WITHOUT torch.distributed: plain processes, the collectives are the C ABI's
# (lthip_comm_* = RCCL behind comm.hip), the communicator id travels through a file.
#   tools/run8.sh [N] [bench.py arguments...]        e.g.  tools/run8.sh 8 --scaling strong --steps 3 --warmup 1
# Rank 0 prints the JSON line.  (The driver's own launch -- python -m torch.distributed.run ... bench.py --gpus N -- keeps working, and
# `python bench.py --gpus N [--launch plain]` starts its ranks itself; this is the same thing as a shell loop.)  Every collective of
# the exchange -- the three all-gathers and the sharded first-seen table's all-to-all -- is lthip_comm_allgather / lthip_comm_alltoallv.
#   tools/run8.sh 8 --handshake-only                 10-second check that RCCL sees 8 ranks (id file, lthip_comm_create, reductions)
#   LTHIP_COMM_TRANSPORT=shm tools/run8.sh 2 ...     the same flow on a box with fewer GPUs (shared-memory stand-in for RCCL)
# The two commands of the first 8-GPU lease (BASELINE.json configs[3] and configs[4]), after the handshake:
#   tools/run8.sh 8 --handshake-only
#   tools/run8.sh 8 --scaling strong --steps 3 --warmup 1                                       # configs[3]: the 64 GiB tree of 1 MiB files, sharded by file
#   tools/run8.sh 8 --scaling strong --file-mib 16384 --codec zstd --kind mixed --steps 3 --warmup 1   # configs[4]: 4 x 16 GiB PAK files, ZStd, intra-file segment shard
# (the same through the driver's launcher: python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
#  --master-port P bench.py --gpus 8 ...; `python bench.py --gpus 8 ...` alone runs the handshake first, by itself, and prints ONE JSON
#  line naming the failing stage if first contact fails.)  LTHIP_RCCL_PATH=<librccl.so> picks the RCCL; the line's config.comm says which was bound.
if [ "$1" = "-h" ] || [ "$1" = "--help" ]; then sed -n '2,20p' "$0"; exit 0; fi
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
