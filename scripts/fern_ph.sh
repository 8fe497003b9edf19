#!/bin/bash
# gpurun -- "bash scripts/fern_ph.sh": phase cycles of the fused 64-wide backward (the -DNH_PHASE_TIMING build of scripts/build_r64_variant.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp nerf-pytorch_amd/libnerfhip.so /tmp/libnerfhip_product.so
cp nerf-pytorch_amd/libnerfhip_ph.so nerf-pytorch_amd/libnerfhip.so
for m in ${@:-fused}; do echo "== $m"; python scripts/r64_phases.py $m 2>&1 | tail -14; done
cp /tmp/libnerfhip_product.so nerf-pytorch_amd/libnerfhip.so
