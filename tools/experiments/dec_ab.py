"""ON THE GPU BOX: the s-head decoder, one-workgroup kernels against the eight-workgroup kernels (MUCON_DEC_MW=0 / 1): teacher-forced
training shape (7 steps, Tz = 125) forward and forward + backward, greedy decoding (up to 31 steps)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from mucon_amd import _lib, ops
from helpers import shead_params
GOLD = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden", "shead_cases.npz"))
dev = "cuda:0"
P = {k: v.to(dev) for k, v in shead_params(GOLD, "a").items()}
dec = [P[n] for n in ops.DECODER_STATE_NAMES]
def timeit(f, n=100):
    for _ in range(5): r = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6, r
for Tz, steps, teacher in ((125, 7, True), (125, 31, False), (64, 31, True)):
    memory = torch.randn(Tz, 256, device=dev); hn = torch.randn(2, 128, device=dev); cn = torch.randn(2, 128, device=dev)
    tf = torch.randint(0, 48, (steps,), device=dev); tf[0] = 49
    for mw in (0, 1, 0, 1):
        _lib.set_knob("MUCON_DEC_MW", mw)
        with torch.no_grad():
            f = (lambda: ops.decoder_forward_deferred(memory, hn, cn, tf, dec, steps, 48)) if not teacher else (lambda: ops.decoder_forward(memory, hn, cn, tf, dec, steps, True, False, 48))
            dt, r = timeit(f)
        line = f"Tz={Tz} steps={steps} teacher={teacher} MUCON_DEC_MW={mw}: forward {dt:7.1f} us"
        if not teacher: line += f" (ran {int(r[2].item())} steps)"
        if teacher:   # forward + backward through the graph-free route the training step takes
            R1 = torch.randn(steps, 49, device=dev); r2 = torch.randn(steps, device=dev)
            def fb():
                (logp, lens), ctx = ops.run_forward(ops._DecoderFn, memory, hn.reshape(-1), cn.reshape(-1), tf, None, (steps, True, False, 48), *dec)
                return ops.run_backward(ops._DecoderFn, ctx, R1, r2)
            dt2, _ = timeit(fb)
            line += f"   forward + backward {dt2:7.1f} us"
        print(line)
_lib.set_knob("MUCON_DEC_MW", 1)
