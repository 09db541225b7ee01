"""Per-phase clocks of one decoding step of the persistent decoder kernels (forward and backward), printed by the kernels themselves
when the library is built with the timing hook:
    MUCON_HIPCC_FLAGS=-DDEC_TIMING=3 python -m mucon_amd.build --force      # HERE: stamp step 3
    gpurun -- python tools/dec_phases.py                                     # prints "decoder_fwd step 3: q ... score ... (cycles)"
    python -m mucon_amd.build --force                                        # back to the plain library
(Tz = 125, 7 steps, teacher forcing: the end-to-end leg's shapes.)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mucon_amd.config import get_cfg_defaults, update_config
from mucon_amd.mucon.models import create_model
dev = "cuda"
m = create_model(update_config(get_cfg_defaults(), [], []), 48, 31, 2048).to(dev).train()
m.set_teacher_forcing(True)
enc = torch.randn(1, 125, 128, device=dev, requires_grad=True)
tfi = torch.tensor([49, 1, 2, 3, 4, 5, 6], device=dev); tft = torch.tensor([1, 2, 3, 4, 5, 6, 48], device=dev)
for i in range(3):
    a, b = m.sequence_generation_forward(enc, 7, tfi, tft); (torch.cat(a).sum() + torch.stack(b).sum()).backward()
    torch.cuda.synchronize()
