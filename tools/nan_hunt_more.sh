set -e
for s in 2 3 4; do RAYZ_HUNT_SEED=$s python tools/nan_hunt.py bouncing10k 1920 2048 bvh f32; done
python tools/nan_hunt.py bouncing 3840 4096 linear f32
python tools/nan_hunt.py mesh 3840 2048 bvh f32
python tools/nan_hunt.py three 1920 16384 linear f32
