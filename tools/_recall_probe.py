import sys, json, numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import recall_eval
fx = dict(np.load("tests/golden/recall_eval.npz"))
m = recall_eval.build_model()
for b in (40, 125):
    r = recall_eval.hip_recall(m, fx, batch=b, dump=f"gpurun_out/hip_emb_b{b}.npy")
    print(b, json.dumps(r))
