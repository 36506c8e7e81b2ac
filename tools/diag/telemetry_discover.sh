#!/bin/bash
# What board telemetry can an ordinary user read on the GPU box?  (round 6: VERDICT r5 item 2)
O=gpurun_out/telemetry; mkdir -p $O
{
echo "== which"; which amd-smi rocm-smi rocminfo 2>&1
echo "== id"; id
echo "== rocm-smi"; timeout 30 rocm-smi 2>&1 | head -30
echo "== rocm-smi power/clock"; timeout 30 rocm-smi --showpower --showclocks --showperflevel --showmaxpower 2>&1 | head -60
echo "== amd-smi metric"; timeout 60 amd-smi metric -g 0 2>&1 | head -150
echo "== amd-smi static limit"; timeout 60 amd-smi static -g 0 --limit 2>&1 | head -60
echo "== sysfs"; for d in /sys/class/drm/card*/device; do echo $d; ls $d 2>/dev/null | tr '\n' ' '; echo; for h in $d/hwmon/hwmon*; do echo $h; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input power1_cap power1_cap_max freq1_input freq2_input temp1_input temp2_input; do [ -r $h/$f ] && echo "$f = $(cat $h/$f 2>&1)"; done; done; for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent power_dpm_force_performance_level pm_info; do [ -r $d/$f ] && { echo "-- $f"; cat $d/$f 2>&1 | head -20; }; done; done
echo "== gpu_metrics"; ls -la /sys/class/drm/card*/device/gpu_metrics 2>&1
} > $O/discover.log 2>&1
tail -c 6000 $O/discover.log
