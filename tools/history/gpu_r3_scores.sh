#!/bin/bash
# Round 3: rocprofv3 kernel trace + PMC passes (counters only) over the CURRENT score kernels (tools/score_bench.py, SURVEY §8(d) sizes).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3scores
mkdir -p $O
cd /tmp && export TMPDIR=/tmp SCORE_BENCH_N=512
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/score_trace -- python $R/tools/score_bench.py > $O/score_bench.json 2> $O/score_trace.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/score_pmc_a -- python $R/tools/score_bench.py > /dev/null 2> $O/score_pmc_a.err
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/score_pmc_c -- python $R/tools/score_bench.py > /dev/null 2> $O/score_pmc_c.err
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/score_pmc_d -- python $R/tools/score_bench.py > /dev/null 2> $O/score_pmc_d.err
cd $R
python tools/summarize_pmc.py gpurun_out/r3scores "ascore|cscore|pck" > $O/summary.md 2>&1
cat $O/score_bench.json | head -60
cat $O/summary.md
