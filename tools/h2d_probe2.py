import time, torch, json
dev = torch.device("cuda", 0)
B, L = 64, 160000
wav = torch.empty(B, L, pin_memory=True).normal_()
img = torch.randn(B, 512); img_p = img.pin_memory()
ids = torch.arange(B)
cs = torch.cuda.Stream()
big = torch.empty(1 << 28, device=dev)
def busy(n=40):
    for _ in range(n): big.fill_(1.0)
def t(fn, n=5):
    r = []
    for _ in range(n):
        torch.cuda.synchronize(); busy()
        a = time.perf_counter(); x = fn(); r.append((time.perf_counter() - a) * 1e3); torch.cuda.synchronize()
    return [round(v, 3) for v in r]
out = {}
def wav_cs():
    with torch.cuda.stream(cs):
        d = wav.to(dev, non_blocking=True); e = torch.cuda.Event(); e.record(cs)
    return d
out["wav_pinned_on_copy_stream"] = t(wav_cs)
out["img_pageable_main_nonblocking"] = t(lambda: img.to(dev, non_blocking=True))
out["img_pinned_main_nonblocking"] = t(lambda: img_p.to(dev, non_blocking=True))
out["ids_pageable_main_nonblocking"] = t(lambda: ids.to(dev, non_blocking=True))
def img_cs():
    with torch.cuda.stream(cs):
        return img.to(dev, non_blocking=True)
out["img_pageable_copy_stream"] = t(img_cs)
out["is_pinned_call"] = t(lambda: wav.is_pinned())
print(json.dumps(out, indent=1))
