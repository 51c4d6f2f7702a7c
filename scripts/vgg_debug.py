"""Stage-wise check of the VGG stack's backward against torch on CPU (content-only, style-only, per-tap)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd import ops
from wast3d_amd.vgg import VGG19Features, get_features
from tests.test_gpu_vgg import torch_reference_features
model = VGG19Features.random_init(seed=3)
sd = {k: v.cpu() for k, v in model.state_dict().items()}
g = torch.Generator().manual_seed(1)
img = torch.rand(3, 200, 260, generator=g)
for tap in range(5):
    x = img.clone().requires_grad_(True)
    fr = torch_reference_features(x, sd)
    gt = torch.randn(fr[tap].shape, generator=g)
    (fr[tap] * gt).sum().backward()
    xi = img.cuda().requires_grad_(True)
    fh = get_features(xi, model)
    (fh[tap] * gt.cuda()).sum().backward()
    print("tap", tap, "linear functional: grad rel err", float((xi.grad.cpu() - x.grad).abs().max() / x.grad.abs().max()))
# gram term alone
x = img.clone().requires_grad_(True)
fr = torch_reference_features(x, sd)
A = fr[2].reshape(256, -1); G = A @ A.t(); (G ** 2).mean().backward()
xi = img.cuda().requires_grad_(True)
fh = get_features(xi, model)
l = ops.style_loss([torch.zeros(256, 256, device="cuda")], [fh[2]]); l.backward()
print("gram tap2: loss", float(l), float((G ** 2).mean()), "grad rel err", float((xi.grad.cpu() - x.grad).abs().max() / x.grad.abs().max()))
