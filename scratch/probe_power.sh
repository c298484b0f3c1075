#!/bin/bash
# round 6: what power / clock telemetry can an ordinary user read on the GPU box?
mkdir -p gpurun_out/probe
{
echo "== hwmon"; for d in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $d; ls $d; for f in power1_average power1_input power1_cap power1_cap_max freq1_input freq2_input temp1_input; do [ -r $d/$f ] && echo "$f=$(cat $d/$f)"; done; done
echo "== pp_dpm"; for c in /sys/class/drm/card*/device; do echo $c; for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent current_link_speed; do [ -r $c/$f ] && { echo "-- $f"; cat $c/$f; }; done; ls $c | head -80; done
echo "== gpu_metrics"; for c in /sys/class/drm/card*/device/gpu_metrics; do ls -l $c; python3 -c "import sys;b=open('$c','rb').read();print(len(b),b[:4].hex())"; done
echo "== amdsmi"; python3 -c "import amdsmi; print(amdsmi.__file__)" 2>&1; ls /opt/rocm/share/amd_smi 2>&1 | head; ls /opt/rocm/lib | grep -i smi
echo "== rocm-smi"; timeout 20 rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40
echo "== amd-smi"; timeout 20 amd-smi metric -p -c 2>&1 | head -60
echo "== nproc"; nproc; lscpu | head -20
} > gpurun_out/probe/power_probe.txt 2>&1
