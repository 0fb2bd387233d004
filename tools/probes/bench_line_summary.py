import json
d=json.loads(open("gpurun_out/r6_bench_final.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["fwd_bwd"]["ms_per_step"], d["config0_on_gpu"]["eager_ms"], d["script_pattern_on_gpu"]["fp16_370x463_eager_ms"], d["script_pattern_on_gpu"]["script_loop_images_per_s"], d["cpu_baseline"]["value"], list(d["strong_shards"]))
