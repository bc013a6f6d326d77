#!/bin/bash
# Polls rocm-smi (socket power, sclk, power cap) while the frame loop runs: is the chip at its power cap under the renderer?   bash tools/power_poll.sh
set -o pipefail
OUT=gpurun_out/power_poll.txt
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v "^$" | head -30 > $OUT
echo "---- under load (bench.py --steps 2000: ~9 s of frames)" >> $OUT
python3 bench.py --steps 2000 --warmup 10 --no-cpu-baseline --no-gpu-eager-baseline --no-sustained --no-chunked --no-variants --no-train --no-shard-rehearsal --steady-seconds 0 > gpurun_out/power_poll_bench.json 2> gpurun_out/power_poll_bench.err &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>&1 | grep -i -E "power|sclk|mclk" | head -8 >> $OUT
  echo "--" >> $OUT
  sleep 0.5
done
wait $BP
echo "---- idle again" >> $OUT
sleep 2
rocm-smi --showpower --showclocks 2>&1 | grep -i -E "power|sclk" | head -6 >> $OUT
cat $OUT
