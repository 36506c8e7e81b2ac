#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for x in 0 8192 21000 49152 131072; do echo "extra LDS $x"; VISREP_ATTN_EXTRA_LDS=$x ATTN_VARIANTS=1 timeout 300 python tools/attn_time.py 2>&1 | grep "attn variant"; done
