#!/bin/bash
# Shader clock / power / temperature while the flat-list kernel runs (GPU box, repo root): why its rate differs between
# boxes.  bash tools/clk_probe.sh
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-also --steps 2 --warmup 1 > gpurun_out/clk_bench.json 2> gpurun_out/clk_bench.err &
pid=$!
sleep 9
for i in $(seq 1 12); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|Graphics Package Power\|Average Graphics\|Socket\|junction" | tr -s '\t ' ' ' | tr '\n' '|'; echo
  sleep 2
done
wait $pid
cut -c1-220 gpurun_out/clk_bench.json
