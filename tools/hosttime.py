import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from mucon_amd import ops, _lib
lib=_lib.load()
dev=torch.device("cuda",0)
spec=ops.EncoderSpec(); C,B,T=48,8,4096
names, params = bench.make_params(spec, C, dev)
enc_params, wc, bc = params[:-2], params[-2], params[-1]
tape=torch.randn(B,T,2048,device=dev); dlogp=torch.randn(B,T,C,device=dev)/(B*T)
def step(i):
    for p in params: p.grad=None
    enc=ops.encoder_forward(tape, enc_params, spec, training=True, seed=i)
    _,logp=ops.head_forward(enc,wc,bc,T,want_logits=False)
    logp.backward(dlogp)
for i in range(5): step(i)
torch.cuda.synchronize()
import cProfile, pstats
K=50
t0=time.perf_counter()
for i in range(K): step(i)
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print(f"issue {1e3*(t1-t0)/K:.3f} ms/step, total {1e3*(t2-t0)/K:.3f} ms/step")
pr=cProfile.Profile(); pr.enable()
for i in range(K): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
