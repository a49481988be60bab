#!/usr/bin/env python3
"""Host enqueue time of a train step next to its device time (is the host far enough ahead for the encoder / tail overlap?):
    python tools/host_enqueue_time.py base cascaded_plus hybrid_plus_large        (through gpurun; SC_ENC_OVERLAP=0 for one stream)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
for kind in sys.argv[1:]:
    model, trainer, batch, _, _ = bench.make_workload(kind, 64, 160000, False, 0, dev)
    for _ in range(4):
        trainer.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        trainer.step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(kind, "host enqueue %.2f ms/step, with device drain %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100), flush=True)
    del model, trainer, batch
