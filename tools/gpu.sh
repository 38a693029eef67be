#!/bin/bash
# build locally (fail loudly: a stale .so must never travel), then run a script on the GPU box:  tools/gpu.sh <timeout_s> <script>
set -euo pipefail
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" 
/usr/local/graft/bin/gpurun --timeout "$1" -- "bash $2"
