cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/gpu.sh r5d "tests:exchange_with_device or 768px" "py:tools/diag/thread_dist_diag.py"
for b in 256 255 256 255; do
  VISREP_SWEEP_B224=$b timeout 300 python -m law_of_vision_representation_in_mllms_amd.sweep --settings CLIP336 CLIP224 OpenCLIP DINOv2 CLIP224+DINOv2 --fp32-products 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B224=$b wall', d['wall_s'], {k:(v.get('c_s'),v.get('a_s')) for k,v in d['per_setting'].items()})"
done
