#!/usr/bin/env bash
# One process per MI355X over RCCL/xGMI: tools/dist_train.sh CONFIG NGPUS [train.py options]
# (same positional interface as the reference's launcher script; single node unless NNODES/NODE_RANK/MASTER_ADDR say otherwise)
set -euo pipefail
if [ "$#" -lt 2 ]; then
    echo "usage: $0 CONFIG NGPUS [--work-dir DIR] [--amp] [--cfg-options k=v ...]" >&2
    exit 2
fi
cfg=$1
ngpus=$2
shift 2
here=$(cd "$(dirname "$0")" && pwd)
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}    # dmabuf IPC: RCCL needs it on this driver
exec python -m torch.distributed.run \
    --nnodes="${NNODES:-1}" --node-rank="${NODE_RANK:-0}" \
    --master-addr="${MASTER_ADDR:-127.0.0.1}" --master-port="${PORT:-29500}" \
    --nproc-per-node="$ngpus" \
    "$here/train.py" "$cfg" --launcher pytorch "$@"
