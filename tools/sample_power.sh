#!/bin/bash
# samples rocm-smi (power, sclk) every 0.5 s while bench.py runs; GPU box only
python bench.py --steps 700 --warmup 4 --cpu-pairs 0 ${@} > /tmp/bench_power.log 2>&1 &
BP=$!
sleep 7
for i in $(seq 1 24); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|Average Graphics Package Power|sclk" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/  */ /g'
  echo
  sleep 0.5
  kill -0 $BP 2>/dev/null || break
done
wait $BP
tail -c 300 /tmp/bench_power.log
