"""GPU diagnostic: per-tensor gradient errors of the HIP path vs the f64 oracle, next to the error of
the f32 oracle vs the f64 oracle (the noise floor of an fp32 evaluation incl. ReLU/max-pool kinks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mucon_amd import ops, synth
from oracle import dense as od

def run(B, T, over, seed=91, verbose=False):
    spec, ocfg = ops.EncoderSpec(**over), od.EncoderConfig(**over)
    params_np = od.seeded_params(ocfg, seed)
    tape_np = synth.tape(seed + 1, B, T, 2048)
    Tz = spec.out_length(T)
    w = synth.uniform_pm1(seed + 2, (B, T, 48)); v = synth.uniform_pm1(seed + 3, (B, Tz, 128))
    res = {}
    for dt in (torch.float64, torch.float32):
        g, L = od.hot_path_grads(tape_np, params_np, ocfg, w, v, dt)
        res[dt] = g
    names = ops.param_names(spec)
    P = [torch.tensor(params_np[k], device="cuda", requires_grad=True) for k in names]
    wc = torch.tensor(params_np["conv_classifier.weight"], device="cuda", requires_grad=True)
    bc = torch.tensor(params_np["conv_classifier.bias"], device="cuda", requires_grad=True)
    enc = ops.encoder_forward(torch.tensor(tape_np, device="cuda"), P, spec)
    _, logp = ops.head_forward(enc, wc, bc, T, want_logits=False)
    ((torch.tensor(w, device="cuda") * logp).sum() + (torch.tensor(v, device="cuda") * enc).sum()).backward()
    print(f"--- B={B} T={T} {over} seed={seed}")
    if verbose:
        print(f"{'param':34s} {'hip L2':>10s} {'hip max':>10s} {'f32 L2':>10s} {'f32 max':>10s}")
    worst = (0.0, "")
    for k, t in zip(names + ["conv_classifier.weight", "conv_classifier.bias"], P + [wc, bc]):
        ref = res[torch.float64][k].reshape(-1)
        g = t.grad.cpu().numpy().astype(np.float64).reshape(-1)
        g32 = res[torch.float32][k].astype(np.float64).reshape(-1)
        n, m = np.linalg.norm(ref) + 1e-30, np.abs(ref).max() + 1e-30
        worst = max(worst, (np.linalg.norm(g-ref)/n, k))
        if verbose:
            print(f"{k:34s} {np.linalg.norm(g-ref)/n:10.2e} {np.abs(g-ref).max()/m:10.2e} {np.linalg.norm(g32-ref)/n:10.2e} {np.abs(g32-ref).max()/m:10.2e}")
    print(f"    worst hip rel-L2 {worst[0]:.2e} at {worst[1]}")

if __name__ == "__main__":
    for sd in (91, 191, 291, 391, 491):
        run(1, 2097, {}, sd)
    for sd in (91, 191):
        run(2, 1201, {}, sd)
