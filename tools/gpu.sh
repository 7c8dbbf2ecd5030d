#!/bin/bash
# Developer helper: rebuild the .so, then run a command on a GPU box via gpurun.  usage: tools/gpu.sh <timeout_s> '<command>'
set -e
make -C /root/repo/friendly-stable-audio-tools_amd/csrc -j8 2>&1 | grep -E "error|warning" && exit 1
python -c "import sys; sys.path.insert(0,'/root/repo/friendly-stable-audio-tools_amd'); from stable_audio_tools import _hip; _hip.lib()"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "mkdir -p gpurun_out; $2"
