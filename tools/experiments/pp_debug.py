import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speechclip_plus_amd import _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, D = 64, 12, 768
R, T = int(sys.argv[1]), int(sys.argv[2])
qk = torch.randn(B * R + 64, 2 * D, device=dev).to(torch.bfloat16)
vt = torch.randn(D * (B * R + 64), device=dev).to(torch.bfloat16)
valid = torch.full((B,), T, dtype=torch.int32, device=dev)
L = _lib.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
outs = []
for opt in (0, 1, 0):
    L.sc_set_option(7, opt)
    o = torch.zeros(B * R + 64, D, device=dev, dtype=torch.bfloat16)
    assert L.sc_attn_fwd_bf16(qk.data_ptr(), 2 * D, vt.data_ptr(), valid.data_ptr(), o.data_ptr(), D, B, R, H, D, ctypes.c_float(0.125), None, 0, ctypes.c_float(0.0), 99, st) == 0
    torch.cuda.synchronize()
    outs.append(o[: B * R].float().view(B, R, H, 64))
L.sc_set_option(7, 0)
pp, fw, pp2 = outs
print("two ping-pong runs identical:", bool(torch.equal(pp, pp2)))
d = (pp - fw).abs().amax(dim=3)          # [B, R, H]
bad = (d > 0.05).nonzero()
print("bad (b, row, h) count:", len(bad), "first:", bad[:12].tolist())
if len(bad):
    b, r, h = bad[0].tolist()
    q = qk[b * R + r, h * 64: h * 64 + 64].float()
    K = qk[b * R: b * R + T, D + h * 64: D + h * 64 + 64].float()
    V = vt[D * b * R:].view(-1)[: H * 64 * R].view(H, 64, R)[h, :, :T].float()
    s = (K @ q) * 0.125
    p = torch.softmax(s, 0)
    ref = V @ p
    print("scores: max", float(s.max()), "argmax", int(s.argmax()), "second", float(s.topk(2).values[1]), " running max over 64-key tiles:", [round(float(s[i:i+64].max()),1) for i in range(0, T, 64)])
    print("ref   ", [round(float(x), 3) for x in ref[:8]])
    print("pp    ", [round(float(x), 3) for x in pp[b, r, h, :8]])
    print("4wave ", [round(float(x), 3) for x in fw[b, r, h, :8]])
    rows = sorted(set(int(x[1]) for x in bad if x[0] == b and x[2] == h))
    print("bad rows of that (b, h):", rows)
