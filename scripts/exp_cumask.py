"""How the 3x3 convolution layers (power-limited) and the fused head (latency-bound) scale with the number of CUs they may use:
the same launches on streams created with hipExtStreamCreateWithCUMask.  If the convolutions lose little on 3/4 of the CUs
(the clock rises as the active silicon shrinks), giving the remaining quarter to the latency-bound stages of the neighbouring
batch would be a net win.  usage: python scripts/exp_cumask.py"""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from himo_amd import _lib
from himo_amd.seflow.model import conv2d_nhwc, ACT_SPLIT_IN, ACT_SPLIT_OUT

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
torch.cuda.init(); torch.zeros(1, device=dev)


def masked_stream(word: int):
    mask = (ctypes.c_uint32 * 8)(*([word] * 8))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


shapes = [("enc1.x", 48, 256, 256, 64, 64, 1), ("enc3.x", 48, 64, 64, 256, 256, 1), ("dec2.u4", 16, 256, 256, 256, 128, 0), ("dec3.u5", 16, 512, 512, 64, 64, 0)]
data = {}
for name, n, h, w, ci, co, epi in shapes:
    x = torch.randn(n, h, w, ci, device=dev)
    xin = conv2d_nhwc(x, torch.randn(3, 3, ci, ci, device=dev) * 0.05, torch.zeros(ci, device=dev), precision="f16x2", act_layout=ACT_SPLIT_OUT)
    data[name] = (xin, torch.randn(3, 3, ci, co, device=dev) * 0.05, torch.zeros(co, device=dev), torch.ones(co, device=dev), torch.zeros(co, device=dev))
    del x
torch.cuda.synchronize()
for label, word in (("256 CUs (all)", 0xffffffff), ("224 CUs (7/8)", 0x7f7f7f7f), ("192 CUs (3/4)", 0x77777777), ("128 CUs (1/2)", 0x55555555), ("64 CUs (1/4)", 0x11111111)):
    st = masked_stream(word)
    line = f"{label:15s}"
    with torch.cuda.stream(st):
        for name, n, h, w, ci, co, epi in shapes:
            xin, wt, b, sc, sh = data[name]
            lay = ACT_SPLIT_IN | ACT_SPLIT_OUT
            for _ in range(2):
                conv2d_nhwc(xin, wt, b, epilogue=epi, scale=sc, shift=sh, precision="f16x2", act_layout=lay)
            st.synchronize()
            _lib.prof_start(only="conv3x3_f16x2")
            for _ in range(6):
                conv2d_nhwc(xin, wt, b, epilogue=epi, scale=sc, shift=sh, precision="f16x2", act_layout=lay)
            st.synchronize()
            ms = sorted(v["avg_ms"] for v in _lib.prof_stop().values())[0]
            line += f"  {name} {ms * 1e3:7.1f} us"
    print(line, flush=True)
