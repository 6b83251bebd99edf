import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd")); sys.path.insert(0, REPO)
import torch, models
from fastdepth_hip.train import TrainEngine
torch.manual_seed(0)
m = models.MobileNetSkipAdd((224, 224), pretrained=False); m.decode_conv6[1].bias.data.fill_(2.8); m = m.cuda().train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = TrainEngine(m, dtype=torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32)
x = torch.rand(B, 3, 224, 224, device="cuda"); t = 0.7 + 9.3 * torch.rand(B, 1, 224, 224, device="cuda")
for _ in range(8): eng.step(x, t)
torch.cuda.synchronize()
