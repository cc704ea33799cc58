"""dev tool: does the depthwise kernel read outside its tensors? Input, weights and bias sit inside larger buffers whose guard zones are filled with
NaN (fp16 / fp32); the output is compared with the same call on clean guard zones (zeros). Any NaN or difference = an out-of-range read that is USED."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demonet_amd import _lib
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
G = 1 << 16           # guard elements on both sides
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
bad = 0
for (n, h, w, c, k, s) in [(5, 40, 40, 120, 5, 1), (3, 20, 20, 480, 3, 1), (3, 20, 20, 672, 5, 2), (4, 80, 80, 72, 5, 2), (2, 160, 160, 16, 3, 1),
                           (7, 10, 10, 480, 5, 1), (3, 19, 19, 96, 3, 1), (2, 75, 75, 144, 3, 2), (9, 5, 5, 512, 3, 1), (3, 1, 1, 128, 3, 1)]:
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = torch.randn(n * h * w * c, device="cuda", generator=g).half()
    ws_ = (torch.randn(k * k * c, device="cuda", generator=g) * 0.2).half()
    bs = torch.randn(c, device="cuda", generator=g)
    outs = []
    for fill in (0.0, float("nan")):
        X = torch.full((2 * G + xs.numel(),), fill, dtype=torch.float16, device="cuda"); X[G:G + xs.numel()] = xs
        Wt = torch.full((2 * G + ws_.numel(),), fill, dtype=torch.float16, device="cuda"); Wt[G:G + ws_.numel()] = ws_
        B = torch.full((2 * G + c,), fill, dtype=torch.float32, device="cuda"); B[G:G + c] = bs
        O = torch.zeros(n * ho * wo * c, dtype=torch.float16, device="cuda")
        x_, w_, b_ = X[G:], Wt[G:], B[G:]
        _lib.check(L.dn_depthwise_conv(P(x_), P(w_), P(b_), P(O), n, h, w, c, k, s, pad, 1, st))
        torch.cuda.synchronize()
        outs.append(O.clone())
    nan = bool(torch.isnan(outs[1]).any())
    same = bool(torch.equal(outs[0], outs[1]))
    print(f"n={n} {c}x{h}x{w} k{k}s{s}: NaN in output {nan}, equal to clean-guard run {same}")
    bad += nan or not same
print("OUT-OF-RANGE READS USED" if bad else "clean")
