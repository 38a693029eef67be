#!/bin/bash
# build locally (fail loudly: a stale .so must never travel), then run a script on the GPU box:  tools/gpu.sh <timeout_s> <script>
set -euo pipefail
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
# the GPU box has no .git: stamp the snapshot with the commit it is (profiles/traffic.json and friends carry it)
echo "$(git rev-parse --short=12 HEAD)$(git diff --quiet HEAD -- . && echo || echo +dirty)" > .commit_stamp
/usr/local/graft/bin/gpurun --timeout "$1" -- "bash $2"
