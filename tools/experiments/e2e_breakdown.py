import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.nn.utils import clip_grad_norm_
from mucon_amd import synth
from mucon_amd.config import get_cfg_defaults, update_config
from mucon_amd.core.datasets import Batch
from mucon_amd.mucon.models import create_model
from mucon_amd.mucon.trainers import SimpleTrainer
dev="cuda"; T,N,C=2000,6,48
cfg=update_config(get_cfg_defaults(),[],[])
torch.manual_seed(0)
model=create_model(cfg,C,31,2048).to(dev); tr=synth.transcript(3,N,C,allow_repeats=False)
trainer=SimpleTrainer(cfg,model,dev); trainer.on_start_epoch(0); model.train()
batch=Batch(feats=torch.randn(1,T,2048),gt_label=torch.from_numpy(synth.segment_labels(4,T,tr)),transcript=torch.from_numpy(tr),
            transcript_tf_input=torch.tensor([C+1]+tr.tolist()),transcript_tf_target=torch.tensor(tr.tolist()+[C]),video_name="s").to(dev)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
acc={}
for it in range(25):
    t0=sync(); trainer.optimizer.zero_grad()
    enc=model.temporal_modeling_forward(batch.feats); t1=sync()
    trs,lens=model.sequence_generation_forward(enc,batch.transcript_tf_target.shape[0],batch.transcript_tf_input,batch.transcript_tf_target); t2=sync()
    seg,logp=model._segmentation_and_logp(enc,T); t3=sync()
    from mucon_amd.mucon.models import MuConForwardOut
    fo=MuConForwardOut(transcript=torch.cat(trs,0),lengths=torch.stack(lens[:-1]),segmentation=seg); fo._logp=logp
    loss=model.loss(batch,fo); t4=sync()
    loss.main.backward(); t5=sync()
    t6=sync(); trainer.fused_step.step(); t7=sync()
    if it>=5:
        for k,v in (("encoder_fwd",t1-t0),("s_head_fwd",t2-t1),("y_head_fwd",t3-t2),("loss",t4-t3),("backward",t5-t4),("clip+sgd",t7-t6)):
            acc[k]=acc.get(k,0)+v
for k,v in acc.items(): print(f"{k:12s} {v/20*1e3:8.3f} ms")
