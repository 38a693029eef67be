#!/bin/sh
# rebuild the HIP extension (gfx950) and the CPU oracle in-tree
cd "$(dirname "$0")" && python -c "import __graft_entry__ as g; g.build_hip(force=True); from oracle import oracle as o; o.build()" 2>&1 | grep -E "error|Error" ; ls -la gsv-tts-lite_amd/lib/libgsv_hip.so
