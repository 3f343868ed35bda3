timeout -k 10 300 python scripts/tmp/typed_ab.py > gpurun_out/r02_typed_order.jsonl 2>gpurun_out/r02_typed_order.err; cat gpurun_out/r02_typed_order.jsonl
timeout -k 10 300 python scripts/sweep.py --method bicubic "" "STAGE_ORDER=1" "STAGE_ORDER=1,STAGE_ZPB=50" "STAGE_ORDER=1,STAGE_ZPB=13" > gpurun_out/r02_sweep_y.log 2>&1; grep median gpurun_out/r02_sweep_y.log
