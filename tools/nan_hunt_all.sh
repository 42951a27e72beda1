set -e
python tools/nan_hunt.py three 400 32768 linear f32
python tools/nan_hunt.py three 400 8192 linear f64
python tools/nan_hunt.py custom 960 16384 bvh f32
python tools/nan_hunt.py custom 960 2048 bvh f64
python tools/nan_hunt.py bouncing 1920 4096 bvh f32
python tools/nan_hunt.py mesh 1920 2048 bvh f32
python tools/nan_hunt.py bouncing10k 1920 1024 bvh f64
