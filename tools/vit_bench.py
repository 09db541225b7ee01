import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mucon_amd import ops
from mucon_amd.core.viterbi import PoissonModel
C=48; dev="cuda"
for (T,N) in ((16384,64),(4096,8),(2000,6)):
    g=torch.Generator().manual_seed(7)
    tr=torch.randint(0,C,(N,),generator=g).numpy().astype(np.int32)
    mu=np.ones(C); mu[np.unique(tr)]=T/N
    P=PoissonModel(mu).rows_for(tr,30)
    lp=torch.log_softmax(3*torch.randn(T,C,generator=g),dim=1).to(dev)
    for _ in range(3): ops.viterbi_decode_batch([lp],[tr],[P],30,2000)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): ops.viterbi_decode_batch([lp],[tr],[P],30,2000)
    torch.cuda.synchronize(); print(T,N,"ms/video",(time.perf_counter()-t0)/10*1e3)
