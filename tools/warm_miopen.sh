#!/bin/bash
# Populate the MIOpen JIT cache on a GPU box (the image has no gfx950 kernel db):
#   gpurun -- 'bash tools/warm_miopen.sh'      then copy gpurun_out/miopen/* into hawkeye_amd/miopen_db/
set -e
mkdir -p gpurun_out/miopen
# start from the seed already in the tree so the result is a superset
cp -rn hawkeye_amd/miopen_db/* gpurun_out/miopen/ 2>/dev/null || true
export HAWKEYE_MIOPEN_DIR=$PWD/gpurun_out/miopen
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernels "$@"
python -c "import __graft_entry__ as g; g.smoke()"
du -sh gpurun_out/miopen
