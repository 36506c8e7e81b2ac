# where a tile round of the persistent GEMM goes, with the epilogue body separated from the waits at its barriers (tools/gemm_tile_timing.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
VISREP_LIB=$PWD/law_of_vision_representation_in_mllms_amd/libvisrep_hip_tiletiming.so timeout 300 python tools/gemm_tile_timing.py
