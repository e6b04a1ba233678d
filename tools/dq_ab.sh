#!/bin/bash
# A/B of the dQ kernel forms inside one gpurun call (variant libraries built by hand under lc2is_amd/csrc/build/var_*)
R=${GRAFT_REPO_ROOT:-/root/repo}
echo "old dq:";            LC2IS_DQ_V2=0 python tools/attn_bench.py --only bwd --check | grep -v "^$"
echo "dq2 waves3 dp_ahead1:"; python tools/attn_bench.py --only bwd --check
for v in 30 21; do echo "dq2 var_$v:"; LC2IS_LIB=$R/lc2is_amd/csrc/build/var_$v/liblc2is_hip.so python tools/attn_bench.py --only bwd --check; done
