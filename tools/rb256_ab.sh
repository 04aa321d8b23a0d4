#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== MB=2 (default build)"; python3 tools/rb256_bench.py 2>/dev/null
echo "== MB=1"; PYGHO_AMD_LIB=$GRAFT_REPO_ROOT/pygho_amd/_lib/variants/mb1/libpygho_hip.so python3 tools/rb256_bench.py 2>/dev/null
