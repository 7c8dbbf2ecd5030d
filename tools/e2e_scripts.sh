#!/bin/bash
# Developer check on a GPU box: run the two entry-script counterparts end to end on synthetic weights.
set -e
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD/friendly-stable-audio-tools_amd:$PYTHONPATH
T=$(mktemp -d)
cat > $T/cond.yaml <<YML
demo:
  break: {prompt: "Amen break 174 BPM", seconds_start: 0, seconds_total: 12}
  pad: {prompt: "warm analog pad", seconds_start: 0, seconds_total: 30}
YML
python friendly-stable-audio-tools_amd/generate.py --output-dir $T/out --cond-yaml-path $T/cond.yaml --synthetic-weights 0 --sample-steps 8 --batch-size 4 --clip-length --seed 1
ls -la $T/out/demo
mkdir -p $T/in && cp $T/out/demo/break_item-1.wav $T/in/
python friendly-stable-audio-tools_amd/reconstruct_audios.py --audio-dir $T/in --output-dir $T/rec/reconstructed --synthetic-weights 0 --frame-duration 1.0
ls -la $T/rec/reconstructed $T/rec/original
python - <<PY
import sys; sys.path.insert(0, "friendly-stable-audio-tools_amd")
from stable_audio_tools.utils.wav_io import load_wav
a, sr = load_wav("$T/out/demo/break_item-1.wav"); b, _ = load_wav("$T/rec/reconstructed/break_item-1.wav")
print("generated", tuple(a.shape), sr, float(a.abs().max()), "reconstructed", tuple(b.shape), float(b.abs().max()))
assert a.shape == (2, 12 * 44100) and b.shape == a.shape
PY
