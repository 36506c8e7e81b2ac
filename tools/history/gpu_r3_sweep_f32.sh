#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3s
timeout 1200 python -m law_of_vision_representation_in_mllms_amd.sweep --precision fp32 --settings CLIP336 CLIP224 OpenCLIP DINOv2 SigLIP CLIP224+DINOv2 CLIP336+DINOv2 2>/dev/null | tail -1 > gpurun_out/r3s/sweep_vit_fp32.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3s/sweep_vit_fp32.json"))
print("fp32 ViT sweep wall", d["wall_s"], "setup", d["setup_s"], {k:(v.get("a_s"),v.get("c_s")) for k,v in d["per_setting"].items()})
PY
