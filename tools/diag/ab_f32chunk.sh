cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in 128 0 128 0; do
  VISREP_F32_CHUNK=$c timeout 300 python -m law_of_vision_representation_in_mllms_amd.sweep --settings CLIP336 CLIP224 OpenCLIP DINOv2 --fp32-products 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunk=$c wall', d['wall_s'], {k:(v.get('c_s'),v.get('a_s')) for k,v in d['per_setting'].items()})"
done
