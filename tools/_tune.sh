T=gpurun_out/conv_tiles_gfx950.json; rm -f $T
python - <<'P'
import json
json.dump(dict(meta={}, entries={}), open('gpurun_out/conv_tiles_gfx950.json','w'))
P
python tools/tune_conv_tiles.py --table $T > gpurun_out/tune_r50.log 2>&1 && python tools/tune_conv_tiles.py --table $T --trained-like > gpurun_out/tune_r50tl.log 2>&1 && python tools/tune_conv_tiles.py --table $T --depth 101 > gpurun_out/tune_r101.log 2>&1 && python tools/tune_conv_tiles.py --table $T --depth 101 --infer --batch 16 > gpurun_out/tune_r101inf.log 2>&1; tail -n 2 gpurun_out/tune_*.log; head -30 gpurun_out/tune_r50.log
