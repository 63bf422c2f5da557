set -x
python bench.py > gpurun_out/bench_r3_final.json 2> gpurun_out/bench_r3_final.err
run() { python bench.py --no-cpu-baseline --no-second-engine "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*', '|', d['value'], d['ms_per_step'], r['kernel'], r['frac'], d['roofline'].get('step_hbm_traffic_GB'), (r.get('hbm_kernels') or {}).get('frac_of_8TBps'), r['all_mfma_kernels']['ms_per_step'])"; }
( run --graph on; run --graph off; run --matrix-dtype f32; run --matrix-dtype bf16; run --height 32 --width 1024 --classes 17 --batch 16; run --height 32 --width 1024 --classes 17 --batch 16 --graph on; run --height 40 --width 1800 --classes 14 --batch 8 --dataset SemanticPOSS ) > gpurun_out/bench_r3_configs.txt
timeout 1200 tools/profile_all.sh 2>&1 | grep -v rocprofv3 | tail -6
