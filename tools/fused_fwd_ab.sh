#!/bin/bash
# the fused block forward (csrc/seg_fused.hip) against the two launches it replaces, and its knock-out / depth variants, on one box
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/pygho_amd/_lib/variants
PICK='import sys,json; d=json.loads(sys.stdin.readline()); print({k: (round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ("stores_h","out_bit_identical","h_bit_identical_on_those","separate_ms","linear_bn_act_ms","seg_gmr_ms","fused_ms","fused_frac_of_8TBs")})'
echo "== default"; python3 tools/fused_fwd_ab.py 8192 20 2>/dev/null | python3 -c "$PICK"
echo "== default, H not stored"; python3 tools/fused_fwd_ab.py 8192 20 --no-h 2>/dev/null | python3 -c "$PICK"
for v in "$@"; do
  echo "== $v"; PYGHO_AMD_LIB=$V/$v/libpygho_hip.so python3 tools/fused_fwd_ab.py 8192 20 2>/dev/null | python3 -c "$PICK"
done
