#!/bin/bash
# the fused block forward (csrc/seg_fused.hip) against the two launches it replaces, and its knock-out / depth variants, on one box
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/pygho_amd/_lib/variants
echo "== default"; python3 tools/fused_fwd_ab.py 8192 20 2>/dev/null
echo "== default, H not stored"; python3 tools/fused_fwd_ab.py 8192 20 --no-h 2>/dev/null
for v in "$@"; do
  echo "== $v"; PYGHO_AMD_LIB=$V/$v/libpygho_hip.so python3 tools/fused_fwd_ab.py 8192 20 2>/dev/null
done
