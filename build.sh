#!/bin/sh
# rebuild the HIP extension (gfx950) and the CPU oracle in-tree; fails when a compile or link fails
set -eu
cd "$(dirname "$0")"
python -c "import __graft_entry__ as g; g.build_hip(force=True); from oracle import oracle as o; o.build(force=True)"
ls -la gsv-tts-lite_amd/lib/libgsv_hip.so oracle/libgsv_oracle.so
