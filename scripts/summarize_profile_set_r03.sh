#!/bin/bash
# After `bash scripts/run_profile_set_r03.sh <tag>` on the GPU box: regenerate profiles/r03_final_* from gpurun_out/<tag>/ and refresh
# profiles/traffic.json (HBM bytes per launch, one record per launch length).
set -e
tag=${1:-r03}
o=gpurun_out/$tag
python3 scripts/summarize_profile.py r03_final $o/kt $o/pmc3 $o/pmc4 --envsteps-per-launch 4096000 --timed-launches 5 > /dev/null
python3 scripts/summarize_profile.py r03_final_chunk100 $o/ckt $o/pmc1 $o/pmc2 $o/pmc5 --envsteps-per-launch 409600 --timed-launches 10 > /dev/null
python3 scripts/summarize_profile.py r03_final_driver_shape $o/dkt $o/dpmc3 $o/dpmc4 --envsteps-per-launch 81920 --timed-launches 5 > /dev/null
python3 scripts/summarize_profile.py r03_final_shard512 $o/s5kt $o/s5pmc1 $o/s5pmc2 --kernel mjb_k_step2_spec --envsteps-per-launch 51200 --timed-launches 10 > /dev/null
python3 scripts/summarize_profile.py r03_final_shard1024 $o/s10kt $o/s10pmc2 --kernel mjb_k_step2_spec --envsteps-per-launch 102400 --timed-launches 10 > /dev/null
python3 scripts/summarize_profile.py r03_other_configs $o/okt --kernel k_ --timed-launches 1 > /dev/null
cp $o/bench.json profiles/r03_final_bench.json
cp $o/bench_driver_shape.json profiles/r03_final_bench_driver_shape.json
python3 - <<'PY'
import json
recs = []
for name, steps, cmd, n in (("r03_final", 1000, "python3 bench.py --no-cpu-baseline --no-host-loop --no-other-configs", 5), ("r03_final_driver_shape", 20, "python3 bench.py --no-cpu-baseline --no-host-loop --no-other-configs --steps 20 --warmup 5", 5)):
    s = json.load(open(f"profiles/{name}_summary.json"))
    recs.append({"model": "humanoid", "global_batch": 4096, "launch_steps": steps, "traffic_bytes_per_launch": s["traffic_bytes_per_launch"],
                 "source": f"profiles/{name}_summary.txt (FETCH_SIZE x2 + WRITE_SIZE, KiB, separate --pmc passes of `{cmd}`, its {n} event-timed launches)"})
json.dump(recs, open("profiles/traffic.json", "w"), indent=1)
print(recs)
PY
