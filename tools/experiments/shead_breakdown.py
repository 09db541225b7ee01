"""Where the s-head's time goes on the GPU (ms per call, Tz = 125, 7 decoding steps): torch.nn.LSTM (MIOpen) and
the torch decoding loop against the persistent HIP kernels."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mucon_amd import ops
from mucon_amd.config import get_cfg_defaults, update_config
from mucon_amd.mucon.models import create_model
dev = "cuda"
m = create_model(update_config(get_cfg_defaults(), [], []), 48, 31, 2048).to(dev).train()
m.set_teacher_forcing(True)
enc = torch.randn(1, 125, 128, device=dev, requires_grad=True)
tfi = torch.tensor([49, 1, 2, 3, 4, 5, 6], device=dev); tft = torch.tensor([1, 2, 3, 4, 5, 6, 48], device=dev)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
W = list(m.fs_encoder_lstm.parameters())
def miopen_f(): return m.fs_encoder_lstm(enc)
def miopen_fb():
    o, (h, c) = m.fs_encoder_lstm(enc); (o.sum() + h.sum() + c.sum()).backward()
def hip_f(): return ops.lstm_forward(enc[0], W)
def hip_fb():
    o, h, c = ops.lstm_forward(enc[0], W); (o.sum() + h.sum() + c.sum()).backward()
def shead_f(): return m.sequence_generation_forward(enc, 7, tfi, tft)
def shead_fb():
    a, b = m.sequence_generation_forward(enc, 7, tfi, tft); (torch.cat(a).sum() + torch.stack(b).sum()).backward()
print("biLSTM MIOpen  fwd %.3f  fwd+bwd %.3f" % (t(miopen_f), t(miopen_fb)))
print("biLSTM HIP     fwd %.3f  fwd+bwd %.3f" % (t(hip_f), t(hip_fb)))
for lstm, dec in ((False, False), (True, False), (True, True)):
    m.native_lstm, m.native_decoder = lstm, dec
    print("s-head  lstm=%s decoder=%s  fwd %.3f  fwd+bwd %.3f" % ("HIP" if lstm else "MIOpen", "HIP" if dec else "torch", t(shead_f), t(shead_fb)))
