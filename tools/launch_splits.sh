#!/bin/bash
# BASELINE config 4: Breakfast splits 1-4 trained CONCURRENTLY, each by its own data-parallel group of GPUS_PER_GROUP ranks
# (default 2): independent process groups -- separate rendezvous ports, disjoint HIP_VISIBLE_DEVICES -- with a 2-rank gradient
# all-reduce inside each group and nothing across groups.  Aggregate throughput = the sum over groups.
#
#   bash tools/launch_splits.sh --cfg configs/docker/inside.yaml --set trainer.num_epochs 1
#
# Environment:
#   SPLITS="1 2 3 4"        which splits (one group each)
#   GPUS_PER_GROUP=2        ranks (= GPUs) per group
#   DEVICES="0 1 2 3 4 5 6 7"   GPU ids handed out group by group (fewer than SPLITS x GPUS_PER_GROUP: ids are reused and every
#                               group shares its devices with another -- only for plumbing tests; then set MUCON_DIST_BACKEND=gloo,
#                               RCCL refuses two ranks on one GPU)
#   BASE_PORT=29600         group g rendezvous on 127.0.0.1:BASE_PORT+g
#   LOG_DIR=./split_logs    one log per group
#   NCCL_IB_DISABLE         defaults to 1 here: single node, gradients travel over xGMI only (set it to 0 to override)
# Exit code: non-zero if any group failed.
set -u
SPLITS=${SPLITS:-"1 2 3 4"}
GPUS_PER_GROUP=${GPUS_PER_GROUP:-2}
DEVICES=(${DEVICES:-0 1 2 3 4 5 6 7})
BASE_PORT=${BASE_PORT:-29600}
LOG_DIR=${LOG_DIR:-./split_logs}
export NCCL_IB_DISABLE=${NCCL_IB_DISABLE:-1}
mkdir -p "$LOG_DIR"
pids=()
g=0
for split in $SPLITS; do
    devs=()
    for ((r = 0; r < GPUS_PER_GROUP; ++r)); do
        devs+=("${DEVICES[$(((g * GPUS_PER_GROUP + r) % ${#DEVICES[@]}))]}")
    done
    vis=$(printf "%s\n" "${devs[@]}" | awk '!seen[$0]++' | paste -sd, -)   # (ids reused on a small box: each listed once)
    port=$((BASE_PORT + g))
    echo "split $split: devices $vis, rendezvous 127.0.0.1:$port -> $LOG_DIR/split$split.log"
    HIP_VISIBLE_DEVICES=$vis python -m torch.distributed.run --nnodes=1 --nproc-per-node "$GPUS_PER_GROUP" \
        --master-addr 127.0.0.1 --master-port "$port" -m mucon_amd.train_test_mucon "$@" --set dataset.split "$split" \
        --exp-name "split$split" > "$LOG_DIR/split$split.log" 2>&1 &
    pids+=($!)
    g=$((g + 1))
done
rc=0
for pid in "${pids[@]}"; do
    wait "$pid" || rc=1
done
exit $rc
