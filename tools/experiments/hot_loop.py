"""The hot-path step (B=8, T=4096) in a loop, for rocprofv3: TRAINING=0/1 selects dropout replay off/on."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from mucon_amd import ops
dev = torch.device("cuda", 0)
spec = ops.EncoderSpec(); C, B, T = 48, 8, 4096
names, params = bench.make_params(spec, C, dev)
enc_params, wc, bc = params[:-2], params[-2], params[-1]
tape = torch.randn(B, T, 2048, device=dev); dlogp = torch.randn(B, T, C, device=dev) / (B * T)
training = os.environ.get("TRAINING", "1") == "1"
for i in range(12):
    for p in params: p.grad = None
    enc = ops.encoder_forward(tape, enc_params, spec, training=training, seed=i)
    _, logp = ops.head_forward(enc, wc, bc, T, want_logits=False)
    logp.backward(dlogp)
torch.cuda.synchronize()
