"""ON THE GPU BOX: the s-head decoder's forward, one-workgroup kernel against the eight-workgroup kernel (MUCON_DEC_MW=0 / 1):
teacher-forced training shape (7 steps, Tz = 125) and greedy decoding (up to 31 steps)."""
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
for Tz, steps, teacher in ((125, 7, True), (125, 31, False), (64, 31, True)):
    memory = torch.randn(Tz, 256, device=dev); hn = torch.randn(2, 128, device=dev); cn = torch.randn(2, 128, device=dev)
    tf = torch.randint(0, 48, (steps,), device=dev); tf[0] = 49
    for mw in (0, 1, 0, 1):
        _lib.set_knob("MUCON_DEC_MW", mw)
        with torch.no_grad():
            f = lambda: ops.decoder_forward_deferred(memory, hn, cn, tf, dec, steps, 48) if not teacher else ops.decoder_forward(memory, hn, cn, tf, dec, steps, True, False, 48)
            for _ in range(5): r = f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): r = f()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        extra = ""
        if not teacher: extra = f" (ran {int(r[2].item())} steps)"
        print(f"Tz={Tz} steps={steps} teacher={teacher} MUCON_DEC_MW={mw}: {dt*1e6:7.1f} us per forward{extra}")
_lib.set_knob("MUCON_DEC_MW", 1)
