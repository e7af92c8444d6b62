#!/bin/bash
# Start N ranks of any program, one per GPU, with the environment libcloudsc2_comm.so (and torchrun-style programs) read:
#   tools/launch_ranks.sh N program [args...]
# RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT are set per rank; rank 0's output comes first, the others follow.
# CLOUDSC2_COMM=shm in the caller's environment rehearses more ranks than the node has GPUs (the ranks then share GPU 0).
n=$1; shift
port=$(python3 -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
export WORLD_SIZE=$n MASTER_ADDR=127.0.0.1 MASTER_PORT=$port CLOUDSC2_COMM_TOKEN=launch_$$_$port HSA_ENABLE_IPC_MODE_LEGACY=0
tmp=$(mktemp -d)
pids=()
for r in $(seq 0 $((n-1))); do
  RANK=$r LOCAL_RANK=$r "$@" > $tmp/out.$r 2> $tmp/err.$r &
  pids+=($!)
done
rc=0
for r in $(seq 0 $((n-1))); do
  wait ${pids[$r]} || rc=$?
done
for r in $(seq 0 $((n-1))); do
  [ $r -gt 0 ] && [ -s $tmp/out.$r ] && echo "--- rank $r stdout"
  cat $tmp/out.$r
done
for r in $(seq 0 $((n-1))); do cat $tmp/err.$r >&2; done
rm -r "$tmp"
exit $rc
