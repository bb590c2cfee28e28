mkdir -p gpurun_out
(python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-step-probe > gpurun_out/bench_clk.json 2>/dev/null) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6 7 8 9 10; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|power\|Temperature (Sensor junction)\|fclk" | tr '\n' ';'; echo
  sleep 1
done > gpurun_out/clocks_under_load.txt
wait $BP
cat gpurun_out/clocks_under_load.txt | cut -c1-400
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr '\n' ';'; echo " (idle)"
