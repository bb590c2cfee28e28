set -x
mkdir -p gpurun_out
for M in audioldm_48k audioldm2-full-large-1150k audioldm2-speech-gigaspeech; do
timeout 1500 python bench.py --model $M --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$M.json 2> gpurun_out/bench_$M.err
python -c "import json;d=json.load(open('gpurun_out/bench_$M.json'));print('$M', d['value'], d['ms_per_step'], d['unet_step_ms'], d['unet_step_frac_of_f32_mfma_peak'], d['roofline']['kernel'], d['roofline']['frac'])"
done
