#!/bin/bash
# PMC passes (counters only, never combined with tracing) over tools/attn_time.py: both head-width-64 attention kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/sq_counters.txt
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_a -- python $R/tools/attn_time.py > $O/pmc_a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_b -- python $R/tools/attn_time.py > $O/pmc_b.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_c -- python $R/tools/attn_time.py > $O/pmc_c.log 2>&1
cd $R
tail -3 $O/pmc_a.log $O/pmc_b.log $O/pmc_c.log
python tools/summarize_pmc.py gpurun_out/r3pmc "attn_fwd" > $O/summary.md 2>&1
cat $O/summary.md
