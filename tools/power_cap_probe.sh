#!/bin/bash
# can this box run the bench at a SECOND power cap or a pinned shader clock?  Tries the two rocm-smi knobs as the ordinary
# user gpurun gives us, and if one is accepted runs the DiT-step A/B at that setting and resets it.  Output: gpurun_out/${TAG}_power_cap_probe.txt
cd "$(dirname "$0")/.."
R=$PWD
export PYTHONPATH=$R/friendly-stable-audio-tools_amd:$PYTHONPATH
TAG=${TAG:-r06}
out=gpurun_out/${TAG}_power_cap_probe.txt
{
  echo "== whoami: $(whoami)"
  echo "== rocm-smi --showpower --showmaxpower --showsclkrange --showperflevel"
  rocm-smi --showpower --showmaxpower --showsclkrange --showperflevel 2>&1 | grep -v "^$\|====" | head -20
  echo "== rocm-smi --setpoweroverdrive 1000"
  rocm-smi --setpoweroverdrive 1000 --autorespond y 2>&1 | grep -v "^$\|====" | head -8
  capped=$(rocm-smi --showmaxpower 2>&1 | grep -c "1000")
  echo "== rocm-smi --setperfdeterminism 1900"
  rocm-smi --setperfdeterminism 1900 2>&1 | grep -v "^$\|====" | head -8
  echo "== rocm-smi --setsrange 500 1900"
  rocm-smi --setsrange 500 1900 --autorespond y 2>&1 | grep -v "^$\|====" | head -8
  echo "== after the attempts:"
  rocm-smi --showmaxpower --showsclkrange --showperflevel 2>&1 | grep -v "^$\|====" | head -12
  echo "== amd-smi set (power cap)"
  amd-smi set --gpu 0 --power-cap 1000 2>&1 | head -5
  amd-smi static --gpu 0 --limit 2>&1 | head -30
} > $out 2>&1
# if any knob took, time the DiT step there (ab_r05 with the r05 library + this tree) and reset
if grep -q "Successfully\|successfully" $out; then
  AB_SET="on:prefetch=0" AB_ROUNDS=3 timeout 600 python tools/ab_r05.py 1 8 >> $out 2>&1
  python tools/power_kernels.py >> $out 2>&1
  rocm-smi --resetpoweroverdrive --resetperfdeterminism --resetclocks >> $out 2>&1
  amd-smi reset --gpu 0 --power-cap >> $out 2>&1
fi
tail -30 $out
