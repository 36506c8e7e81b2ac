#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for d in randn zeros peaked flat vzero; do ATTN_DATA=$d timeout 300 python tools/attn_time.py 2>&1 | grep "attn variant"; done
