"""The VGG-19 feature stack (utils/loss_utils.py:66-109) on HIP against torch's CPU convolution with the same seeded weights: the five tapped feature maps,
and d(style + content loss)/d(image) through resize -> 13 convolutions / ReLUs / 4 max-pools.  Pretrained weights are unobtainable here: architecture,
tap points (conv outputs 0/5/10/19/28, pre-ReLU), resize and backward are what is pinned.  Tolerances: features 2e-5 of the layer's largest value
(f32 summation order over up to 4608 terms), image gradient 1e-3 of its largest entry."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def torch_reference_features(img, sd):
    """features[:29] of torchvision's vgg19 written with torch.nn.functional on CPU, taps as in the reference's VGG.forward."""
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512]
    x = F.interpolate(img.unsqueeze(0), size=(112, 112))
    feats, li = [], 0
    for v in cfg:
        if v == "M":
            x = F.max_pool2d(x, 2, 2); li += 1
        else:
            x = F.conv2d(x, sd[f"features.{li}.weight"], sd[f"features.{li}.bias"], padding=1)
            if li in (0, 5, 10, 19, 28):
                feats.append(x)
            x = F.relu(x); li += 2
    return feats


def test_vgg_features_and_image_gradient_match_torch_cpu():
    from wast3d_amd import ops
    from wast3d_amd.vgg import VGG19Features, get_features
    model = VGG19Features.random_init(seed=3)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    img = torch.rand(3, 200, 260, generator=g)
    style = torch.rand(3, 150, 150, generator=g)
    # reference (CPU autograd)
    x = img.clone().requires_grad_(True)
    fr = torch_reference_features(x, sd)
    with torch.no_grad():
        fs = torch_reference_features(style, sd)
    def gram_loss(a, b):
        A = a.reshape(a.shape[1], -1); B = b.reshape(b.shape[1], -1)
        return torch.mean((A @ A.t() - B @ B.t()) ** 2)
    ref_loss = sum(gram_loss(a, b) for a, b in zip(fr, fs)) * 1e-6 + sum(torch.mean((a - b.detach()) ** 2) for a, b in zip(fr, torch_reference_features(img * 0.9, sd)))
    ref_loss.backward()
    # HIP
    xi = img.cuda().requires_grad_(True)
    fh = get_features(xi, model)
    assert [tuple(f.shape) for f in fh] == [(1, 64, 112, 112), (1, 128, 56, 56), (1, 256, 28, 28), (1, 512, 14, 14), (1, 512, 7, 7)]
    for a, b in zip(fh, fr):
        assert float((a.detach().cpu() - b.detach()).abs().max()) <= 2e-5 * float(b.detach().abs().max())
    with torch.no_grad():
        fsh = get_features(style.cuda(), model); fch = get_features((img * 0.9).cuda(), model)
    loss = ops.style_loss(fsh, fh) * 1e-6 + ops.content_loss(fch, fh)
    assert abs(float(loss) - float(ref_loss)) <= 2e-4 * abs(float(ref_loss))
    loss.backward()
    gerr = float((xi.grad.cpu() - x.grad).abs().max()) / float(x.grad.abs().max())
    assert gerr < 1e-3, gerr


def test_vgg_state_dict_roundtrip_and_layer_table():
    from wast3d_amd.vgg import VGG19Features, _layer_table
    tab = _layer_table()
    assert [t[0] for t in tab] == [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28]             # torchvision's conv indices in vgg19.features[:29]
    assert [t[3] for t in tab] == [False, True, False, True, False, False, False, True, False, False, False, True, False]
    m = VGG19Features.random_init(seed=0)
    m2 = VGG19Features.from_state_dict(m.state_dict())
    assert all(torch.equal(a, b) for a, b in zip(m.w, m2.w))
