import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden
from trajectory_optimization_amd import synth
from trajectory_optimization_amd.model import ModelPose
d = load_golden("pose_adam_bundled")
dev = torch.device("cuda:0")
m = ModelPose(points=torch.from_numpy(d["points"]), trans0=torch.from_numpy(d["trans0"]), q0=torch.from_numpy(d["q0"]),
              intrins=torch.from_numpy(synth.K_INTRINS), img_width=synth.IMG_WIDTH, img_height=synth.IMG_HEIGHT, device=dev)
opt = torch.optim.Adam([{"params": [m.trans], "lr": 0.02}, {"params": [m.quat], "lr": 0.02}])
print("ref losses", d["losses"])
for i in range(10):
    opt.zero_grad()
    loss = m()
    loss.backward()
    print(i, "loss", loss.item(), "trans", m.trans.detach().cpu().numpy(), "g", m.trans.grad.cpu().numpy(), "gq", m.quat.grad.cpu().numpy())
    opt.step()
    print("   after step trans", m.trans.detach().cpu().numpy(), "quat", m.quat.detach().cpu().numpy())
