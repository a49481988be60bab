set -e
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r3_final_base.json 2> gpurun_out/r3_final_base.err
python bench.py --steps 20 --warmup 5 --cpu-utts 0 --no-recall 2>/dev/null | cut -c1-160
for m in cascaded_plus hybrid_plus_large large; do python bench.py --model $m --steps 10 --warmup 3 --cpu-utts 0 --no-recall 2>/dev/null > gpurun_out/r3_final_$m.json; cut -c1-170 gpurun_out/r3_final_$m.json; done
python bench.py --trainable --steps 5 --warmup 2 --cpu-utts 0 --no-recall 2>/dev/null > gpurun_out/r3_final_trainable.json; cut -c1-170 gpurun_out/r3_final_trainable.json
python bench.py --unfreeze 2 --steps 10 --warmup 3 --cpu-utts 0 --no-recall 2>/dev/null > gpurun_out/r3_final_unfreeze2.json; cut -c1-170 gpurun_out/r3_final_unfreeze2.json
