"""The VGG-19 feature stack of the image-style leg (utils/loss_utils.py:66-109), frozen weights, on the HIP kernels of csrc/vgg.hip.

    model = VGG19Features.from_state_dict(torchvision_vgg19_state_dict)     # keys 'features.{i}.weight' / '.bias', i in 0..28
    feats = get_features(render, model)                                     # list of 5 tensors [1,C,h,w]: taps 0, 5, 10, 19 = relu(conv), tap 28 = conv
    loss  = ops.style_loss(style_feats, feats) + ops.content_loss(content_feats, feats)
    loss.backward()                                                         # -> d loss / d render through the frozen network

The pretrained weights cannot be fetched in this environment; the architecture, tap points, resize and the backward pass are pinned against torch's own
CPU convolution with seeded random weights (tests/test_gpu_vgg.py).

Tap semantics (utils/loss_utils.py:82-93): VGG.forward appends the OUTPUT TENSOR of layers 0, 5, 10, 19, 28; torchvision's ReLUs are inplace=True, so the next
layer rewrites that very tensor and the stored feature of taps 0 / 5 / 10 / 19 is relu(conv) — autograd sees the ReLU too.  Only tap 28 (no ReLU follows
inside features[:29]) is a pre-activation.  Internally the activations are kept pre-ReLU (a convolution applies the ReLU when it loads its input; max-pool
commutes with ReLU, so pooled layers keep pool(z)); the four post-ReLU taps are materialised by w3d_relu and their gradients masked by [z > 0]."""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from ._lib import check, lib, ptr, require_cuda, stream_ptr

# torchvision vgg19.features[:29]: layer index -> ('conv', Cin, Cout) | 'relu' | 'pool'
CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512]
TAPS = (0, 5, 10, 19, 28)                 # utils/loss_utils.py:70-71
POST_RELU_TAPS = (0, 5, 10, 19)           # rewritten in place by the ReLU(inplace=True) that follows them in torchvision's vgg19.features
RESIZE = 112                              # utils/loss_utils.py:107


def _layer_table():
    """[(torchvision layer index, Cin, Cout, pool_after)] of the 13 convolutions of features[:29]."""
    out, idx, cin = [], 0, 3
    for i, v in enumerate(CFG):
        if v == "M":
            out[-1] = out[-1][:3] + (True,)
            idx += 1
        else:
            out.append((idx, cin, v, False))
            cin = v
            idx += 2                      # conv + relu
    return out


class VGG19Features:
    def __init__(self, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], device="cuda"):
        self.layers = _layer_table()
        assert len(weights) == len(self.layers) == 13
        self.w = [w.detach().to(device).float().contiguous() for w in weights]
        self.b = [b.detach().to(device).float().contiguous() for b in biases]
        for (li, cin, cout, _), w in zip(self.layers, self.w):
            assert tuple(w.shape) == (cout, cin, 3, 3), (li, tuple(w.shape))
        # backward-data weights: w'[ci, co, ky, kx] = w[co, ci, 2-ky, 2-kx]
        self.wt = [w.flip(2, 3).transpose(0, 1).contiguous() for w in self.w]

    @classmethod
    def from_state_dict(cls, sd: Dict[str, torch.Tensor], device="cuda", prefix="features."):
        tab = _layer_table()
        return cls([sd[f"{prefix}{li}.weight"] for li, *_ in tab], [sd[f"{prefix}{li}.bias"] for li, *_ in tab], device)

    @classmethod
    def random_init(cls, seed: int = 0, device="cuda"):
        """Kaiming-normal weights (what torchvision initialises VGG with before loading a checkpoint): for tests and synthetic benchmarks."""
        g = torch.Generator().manual_seed(seed)
        ws, bs = [], []
        for _, cin, cout, _ in _layer_table():
            ws.append(torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cout * 9)) ** 0.5)
            bs.append(torch.randn(cout, generator=g) * 0.05)
        return cls(ws, bs, device)

    def state_dict(self, prefix="features."):
        sd = {}
        for (li, *_), w, b in zip(self.layers, self.w, self.b):
            sd[f"{prefix}{li}.weight"], sd[f"{prefix}{li}.bias"] = w, b
        return sd

    def preactivations(self, x: torch.Tensor) -> List[torch.Tensor]:
        """All 13 convolution outputs (pre-ReLU, before their pool) for an already resized image [3,h,w]: diagnostics / tests."""
        require_cuda(x)
        L, s = lib(), stream_ptr()
        cur = x.detach().reshape(3, x.shape[-2], x.shape[-1]).float().contiguous()
        H, W = cur.shape[-2:]
        zs, relu_in = [], False
        for (li, cin, cout, pool), w, b in zip(self.layers, self.w, self.b):
            z = torch.empty(cout, H, W, device=x.device)
            check(L.w3d_conv3x3(ptr(cur), ptr(w), ptr(b), cin, cout, H, W, int(relu_in), ptr(z), s), "w3d_conv3x3")
            zs.append(z)
            if pool:
                p = torch.empty(cout, H // 2, W // 2, device=x.device)
                check(L.w3d_maxpool2(ptr(z), cout, H, W, ptr(p), s), "w3d_maxpool2")
                cur, H, W = p, H // 2, W // 2
            else:
                cur = z
            relu_in = True
        return zs

    def __call__(self, x: torch.Tensor) -> List[torch.Tensor]:
        """x [1,3,h,w] or [3,h,w] (already resized) -> the five tapped feature maps [1,C,h',w']."""
        return list(_VGGFeatures.apply(x, self))


class _VGGFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model: VGG19Features):
        require_cuda(x)
        L, s = lib(), stream_ptr()
        cur = x.detach().reshape(3, x.shape[-2], x.shape[-1]).float().contiguous()
        H, W = cur.shape[-2:]
        zs, taps, relu_in = [], [], False
        for (li, cin, cout, pool), w, b in zip(model.layers, model.w, model.b):
            z = torch.empty(cout, H, W, device=x.device)
            check(L.w3d_conv3x3(ptr(cur), ptr(w), ptr(b), cin, cout, H, W, int(relu_in), ptr(z), s), "w3d_conv3x3")
            zs.append(z)
            if li in POST_RELU_TAPS:
                a = torch.empty_like(z)
                check(L.w3d_relu(ptr(z), z.numel(), ptr(a), s), "w3d_relu")
                taps.append(a.unsqueeze(0))
            elif li in TAPS:
                taps.append(z.unsqueeze(0))
            if pool:
                p = torch.empty(cout, H // 2, W // 2, device=x.device)
                check(L.w3d_maxpool2(ptr(z), cout, H, W, ptr(p), s), "w3d_maxpool2")
                cur, H, W = p, H // 2, W // 2
            else:
                cur = z
            relu_in = True
        ctx.model, ctx.zs, ctx.in_shape = model, zs, x.shape
        return tuple(taps)

    @staticmethod
    def backward(ctx, *g_taps):
        model, zs = ctx.model, ctx.zs
        L, s = lib(), stream_ptr()
        tap_of = {li: g for li, g in zip(TAPS, g_taps)}
        gz = None                                             # gradient wrt the pre-activation of the layer being processed
        g_in = None
        for n in range(len(model.layers) - 1, -1, -1):
            li, cin, cout, pool = model.layers[n]
            z = zs[n]
            H, W = z.shape[-2:]
            gt = tap_of.get(li)
            if g_in is None:                                  # last layer (tap 28, a pre-activation): only its tap gradient
                gz = gt.reshape(cout, H, W).float().contiguous().clone() if gt is not None else torch.zeros_like(z)
            else:
                gz = torch.empty_like(z)
                check(L.w3d_relu_pool_bwd(ptr(z), ptr(g_in), cout, H, W, int(pool), 0, ptr(gz), s), "w3d_relu_pool_bwd")
                if gt is not None:                            # a post-ReLU tap: its gradient passes the same [z > 0] mask
                    gtc = gt.reshape(cout, H, W).float().contiguous()
                    if li in POST_RELU_TAPS:
                        check(L.w3d_relu_pool_bwd(ptr(z), ptr(gtc), cout, H, W, 0, 1, ptr(gz), s), "w3d_relu_pool_bwd(tap)")
                    else:
                        gz.add_(gtc)
            # gradient wrt this convolution's input (post-ReLU [post-pool] activation of the previous layer, or the image)
            g_in = torch.empty(cin, H, W, device=z.device)
            check(L.w3d_conv3x3(ptr(gz), ptr(model.wt[n]), None, cout, cin, H, W, 0, ptr(g_in), s), "w3d_conv3x3(backward-data)")
        return g_in.reshape(ctx.in_shape), None


class _NearestResize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, Ho, Wo):
        require_cuda(img)
        x = img.detach().float().contiguous()
        C, H, W = x.shape[-3:]
        out = torch.empty(C, Ho, Wo, device=x.device)
        check(lib().w3d_nearest_resize(ptr(x), C, H, W, Ho, Wo, ptr(out), stream_ptr()), "w3d_nearest_resize")
        ctx.dims = (C, H, W, Ho, Wo, img.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        C, H, W, Ho, Wo, shape = ctx.dims
        gi = torch.zeros(C, H, W, device=g.device)
        check(lib().w3d_nearest_resize_bwd(ptr(g.contiguous()), C, H, W, Ho, Wo, ptr(gi), stream_ptr()), "w3d_nearest_resize_bwd")
        return gi.reshape(shape), None, None


def get_features(input_tensor: torch.Tensor, model: VGG19Features, preprocessing: bool = True) -> List[torch.Tensor]:
    """Drop-in for utils/loss_utils.py:95-109: image [3,H,W] -> unsqueeze -> nearest resize to 112 x 112 -> the five tapped feature maps."""
    x = _NearestResize.apply(input_tensor, RESIZE, RESIZE)
    return model(x.unsqueeze(0))
