import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mucon_amd.config import get_cfg_defaults, update_config
from mucon_amd.mucon.models import create_model
dev="cuda"; cfg=update_config(get_cfg_defaults(),[],[])
m=create_model(cfg,48,31,2048).to(dev).train()
enc=torch.randn(1,125,128,device=dev,requires_grad=True)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
def lstm_f(): return m.fs_encoder_lstm(enc)
def lstm_fb():
    o,(h,c)=m.fs_encoder_lstm(enc); (o.sum()+h.sum()+c.sum()).backward()
tfi=torch.tensor([49,1,2,3,4,5,6],device=dev); tft=torch.tensor([1,2,3,4,5,6,48],device=dev)
def shead_f(): return m.sequence_generation_forward(enc,7,tfi,tft)
def shead_fb():
    a,b=m.sequence_generation_forward(enc,7,tfi,tft); (torch.cat(a).sum()+torch.stack(b).sum()).backward()
print("biLSTM fwd", t(lstm_f)); print("biLSTM fwd+bwd", t(lstm_fb)); print("s-head fwd", t(shead_f)); print("s-head fwd+bwd", t(shead_fb))
with torch.backends.cudnn.flags(enabled=False):
    print("native biLSTM fwd", t(lstm_f)); print("native biLSTM fwd+bwd", t(lstm_fb))
from mucon_amd import ops
W=list(m.fs_encoder_lstm.parameters())
def hip_f(): return ops.lstm_forward(enc[0],W)
def hip_fb():
    o,h,c=ops.lstm_forward(enc[0],W); (o.sum()+h.sum()+c.sum()).backward()
print("HIP biLSTM fwd", t(hip_f)); print("HIP biLSTM fwd+bwd", t(hip_fb))
m.native_lstm=True
print("s-head (HIP lstm) fwd", t(shead_f)); print("s-head (HIP lstm) fwd+bwd", t(shead_fb))
m.native_lstm=False
print("s-head (MIOpen lstm) fwd", t(shead_f)); print("s-head (MIOpen lstm) fwd+bwd", t(shead_fb))
