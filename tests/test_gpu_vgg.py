"""The VGG-19 feature stack (utils/loss_utils.py:66-109) on HIP against torch's CPU convolution with the same seeded weights: the five tapped feature maps,
and d(style + content loss)/d(image) through resize -> 13 convolutions / ReLUs / 4 max-pools.  Pretrained weights are unobtainable here: architecture,
tap points (layers 0/5/10/19/28 as the reference's VGG.forward stores them: torchvision's inplace ReLUs rewrite the appended tensors, so taps 0/5/10/19 are
post-ReLU and only 28 is a pre-activation), resize and backward are what is pinned.  Tolerances: features 2e-5 of the layer's largest value
(f32 summation order over up to 4608 terms), image gradient 1e-3 of its largest entry."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def torchvision_like_features29(sd):
    """torchvision.models.vgg19().features[:29] rebuilt layer for layer (torchvision is absent here): Conv2d(3x3, pad 1) / ReLU(inplace=True) / MaxPool2d(2, 2)
    at torchvision's indices, weights from `sd`.  The inplace ReLUs are the point: they alias the tensors the reference's forward loop has appended."""
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512]
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            conv = torch.nn.Conv2d(cin, v, kernel_size=3, padding=1)
            li = len(layers)
            with torch.no_grad():
                conv.weight.copy_(sd[f"features.{li}.weight"]); conv.bias.copy_(sd[f"features.{li}.bias"])
            conv.requires_grad_(False)
            layers += [conv, torch.nn.ReLU(inplace=True)]
            cin = v
    return torch.nn.Sequential(*layers)[:29]


def torch_reference_features(img, sd):
    """The reference's VGG.forward / get_features loop (utils/loss_utils.py:82-109) over the torchvision-shaped module list: the layer outputs 0/5/10/19/28
    are appended and then rewritten in place by the following ReLU."""
    model = torchvision_like_features29(sd)
    assert len(model) == 29 and isinstance(model[28], torch.nn.Conv2d)
    x = F.interpolate(img.unsqueeze(0), size=(112, 112))
    feats = []
    for layer_num, layer in enumerate(model):
        x = layer(x)
        if layer_num in (0, 5, 10, 19, 28):
            feats.append(x)
    return feats


def reference_vjp_with_masks(zs, g_taps, sd, taps=(0, 5, 10, 19, 28)):
    """d(sum_t <tap_t, g_t>)/d(input) (taps 0/5/10/19 post-ReLU, tap 28 the pre-activation) on CPU with torch's own primitives (conv backward-data = conv_transpose2d, max_unpool2d), the ReLU masks and pool
    arg-maxima taken from the GIVEN pre-activations `zs`: a unit whose pre-activation lies within rounding of 0 (or a near-tied pool cell) may fall
    on different sides in two f32 evaluations of a 13-layer network; sharing the masks takes that out of the comparison of the backward kernels."""
    from wast3d_amd.vgg import _layer_table
    tab = _layer_table()
    g_of = dict(zip(taps, g_taps))
    g_in = None
    for n in range(len(tab) - 1, -1, -1):
        li, cin, cout, pool = tab[n]
        z = zs[n].cpu()[None]
        if g_in is None:
            gz = g_of[li].clone() if li in g_of else torch.zeros_like(z)
        else:
            if pool:
                pz, ind = F.max_pool2d(z, 2, 2, return_indices=True)
                gz = F.max_unpool2d(g_in * (pz > 0), ind, 2, output_size=z.shape[-2:])
            else:
                gz = g_in * (z > 0)
            if li in g_of:
                gz = gz + g_of[li] * (z > 0)                   # a post-ReLU tap (every tap but the last layer's)
        g_in = F.conv_transpose2d(gz, sd[f"features.{li}.weight"], padding=1)
    return g_in[0]


def test_vgg_backward_kernels_match_torch_primitives_on_shared_masks():
    """Every pre-activation of the 13 layers against torch's CPU forward, and the vector-Jacobian product of random tap gradients against torch's
    primitives evaluated on the SAME ReLU masks / pool arg-maxima (1e-5 of the largest entry)."""
    from wast3d_amd.vgg import VGG19Features, _VGGFeatures, _NearestResize
    model = VGG19Features.random_init(seed=3)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    img = torch.rand(3, 200, 260, generator=g)
    x112 = F.interpolate(img.unsqueeze(0), size=(112, 112))[0]
    assert torch.equal(_NearestResize.apply(img.cuda(), 112, 112).cpu(), x112)
    zs = model.preactivations(x112.cuda())
    # forward of all 13 layers
    cur, k = x112[None], 0
    for v in [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512]:
        if v == "M":
            cur = F.max_pool2d(cur, 2, 2)
        else:
            li = model.layers[k][0]
            zr = F.conv2d(cur, sd[f"features.{li}.weight"], sd[f"features.{li}.bias"], padding=1)
            assert float((zs[k].cpu() - zr[0]).abs().max()) <= 2e-5 * float(zr.abs().max()), li
            cur, k = F.relu(zr), k + 1
    xi = x112.cuda().requires_grad_(True)
    fh = model(xi)
    g_taps = [torch.randn(f.shape, generator=g) for f in fh]
    torch.autograd.backward(fh, [t.cuda() for t in g_taps])
    ref = reference_vjp_with_masks(zs, g_taps, sd)
    assert float((xi.grad.cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


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
    # end to end against CPU autograd: two independent f32 forwards may put a unit with a pre-activation within rounding of 0 on different sides of its
    # ReLU (the backward kernels themselves are held to 1e-5 on shared masks above) — almost every pixel agrees tightly, none is far off
    rel = (xi.grad.cpu() - x.grad).abs() / float(x.grad.abs().max())
    # (a flipped unit of a deep layer perturbs its whole receptive field slightly: the bound is on the size of the deviation, not on its extent)
    assert float(rel.max()) < 5e-2 and float(rel.mean()) < 1e-3, (float(rel.max()), float(rel.mean()))


def test_vgg_state_dict_roundtrip_and_layer_table():
    from wast3d_amd.vgg import VGG19Features, _layer_table
    tab = _layer_table()
    assert [t[0] for t in tab] == [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28]             # torchvision's conv indices in vgg19.features[:29]
    assert [t[3] for t in tab] == [False, True, False, True, False, False, False, True, False, False, False, True, False]
    m = VGG19Features.random_init(seed=0)
    m2 = VGG19Features.from_state_dict(m.state_dict())
    assert all(torch.equal(a, b) for a, b in zip(m.w, m2.w))


def test_image_style_step_matches_oracle_composition():
    """One iteration of the image-style loop (train_st.py:265-343) composed from this package's drop-in pieces — render -> get_features x3 -> content_loss(layers 2, 3) x 1e1
    + tv_loss x 1e3 + style_loss(layers 0, 1) x 1e-3 -> backward -> optimizer.step() with `_scaling` frozen (train_st.py:229) — against the same step composed from the
    checker's pieces: oracle rasteriser (oracle/raster_ref.c) forward / backward, torch-CPU VGG-19 with the same seeded weights and CPU autograd for the image losses
    (the floating-point reference of the conv stack; pretrained weights are unobtainable here), oracle TV value, numpy activation chain, oracle Adam with the six group
    rates.  Checks: every loss term, dL/d(image), and the parameters after the step."""
    import math
    from oracle import geometry_ref as O
    from oracle import raster_ref as RR
    from wast3d_amd import ops
    from wast3d_amd.camera import Camera
    from wast3d_amd.gaussian_model import GaussianModel, default_opt
    from wast3d_amd.gaussian_renderer import render
    from wast3d_amd.vgg import VGG19Features, get_features
    from types import SimpleNamespace
    N, H, W = 1500, 96, 128
    g = torch.Generator().manual_seed(11)
    pc = GaussianModel.synthetic(N, seed=5, scale=0.05, opacity=0.5, extent=0.8); pc.spatial_lr_scale = 1.0
    with torch.no_grad():
        pc._features_rest.add_(0.05 * torch.randn(pc._features_rest.shape, generator=g).cuda())
        pc._rotation.add_(0.3 * torch.randn(N, 4, generator=g).cuda())
        pc._scaling.add_(0.3 * torch.randn(N, 3, generator=g).cuda())
    opt = default_opt()
    pc.training_setup(opt)
    pc._scaling.requires_grad_(False)
    params = [pc._xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation]
    raw0 = [p.data.detach().cpu().numpy().copy() for p in params]
    cam = Camera.look_at_origin(0.6, 0.45, 4.03, 0.6911112070083618, W, H, device="cuda")
    bg = torch.ones(3, device="cuda")
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    offs = -torch.rand(H, W, 2, generator=g)
    gt_image = torch.rand(3, H, W, generator=g)
    style_image = torch.rand(3, 120, 90, generator=g)
    vgg = VGG19Features.random_init(seed=3)
    sd = {k: v.cpu() for k, v in vgg.state_dict().items()}
    # ---- the product: one iteration as train_st.py writes it
    pkg = render(cam, pc, pipe, bg, sampling_offsets=offs.cuda())
    image = pkg["render"]; image.retain_grad()
    image_feats = get_features(image, vgg)
    with torch.no_grad():
        gt_feats = get_features(gt_image.cuda(), vgg); style_feats = get_features(style_image.cuda(), vgg)
    l_cont = ops.content_loss([gt_feats[i] for i in (2, 3)], [image_feats[i] for i in (2, 3)]) * 1e1
    l_tv = ops.tv_loss(image) * 1e3
    l_style = ops.style_loss([style_feats[i] for i in (0, 1)], [image_feats[i] for i in (0, 1)]) * 1e-3
    loss = l_cont + l_tv + l_style
    loss.backward()
    g_image_gpu = image.grad.detach().cpu()
    pc.optimizer.step(); pc.optimizer.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    # ---- the checker: oracle rasteriser + torch-CPU VGG / autograd + oracle Adam
    xyz, fdc, frest, opr, scr, rotr = raw0
    sc = np.exp(scr); nq = np.maximum(np.linalg.norm(rotr, axis=1, keepdims=True), 1e-12); q = rotr / nq; op = 1 / (1 + np.exp(-opr))
    shs = np.concatenate([fdc, frest], 1)
    f = lambda t: t.detach().cpu().numpy().astype(np.float32)
    r = RR.Raster(f(bg), xyz, op, f(cam.world_view_transform), f(cam.full_proj_transform), f(cam.camera_center), math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                  H, W, offs.numpy(), shs=shs, sh_degree=3, scales=sc, rotations=q)
    color, depth, _ = r.forward()
    assert np.abs(image.detach().cpu().numpy() - color).max() < 2e-5
    x = torch.tensor(color).requires_grad_(True)
    fr = torch_reference_features(x, sd)
    with torch.no_grad():
        fg = torch_reference_features(gt_image, sd); fs = torch_reference_features(style_image, sd)
    gram = lambda a: (lambda A: A @ A.t())(a.reshape(a.shape[1], -1))
    c_ref = sum(torch.mean((fr[i] - fg[i]) ** 2) for i in (2, 3)) * 1e1
    tv_ref = 0.5 * ((x[:, 1:, :] - x[:, :-1, :]).abs().mean() + (x[:, :, 1:] - x[:, :, :-1]).abs().mean()) * 1e3
    s_ref = sum(torch.mean((gram(fr[i]) - gram(fs[i])) ** 2) for i in (0, 1)) * 1e-3
    (c_ref + tv_ref + s_ref).backward()
    tv_o, _ = O.tv_loss(color)
    assert abs(float(tv_ref) - 1e3 * float(tv_o)) <= 1e-5 * float(tv_ref)                       # the torch expression is the oracle's tv_loss
    for name, got, ref in (("content", l_cont, c_ref), ("tv", l_tv, tv_ref), ("style", l_style, s_ref)):
        assert abs(float(got) - float(ref)) <= 5e-4 * abs(float(ref)), (name, float(got), float(ref))
    gi = x.grad
    rel = (g_image_gpu - gi).abs() / float(gi.abs().max())
    assert float(rel.max()) < 5e-2 and float(rel.mean()) < 2e-3, (float(rel.max()), float(rel.mean()))   # (ReLU units at rounding distance from 0: see the test above)
    lrs = [pc.xyz_scheduler_args(0) if False else opt.position_lr_init * pc.spatial_lr_scale, opt.feature_lr, opt.feature_lr / 20.0, opt.opacity_lr, opt.scaling_lr, opt.rotation_lr]
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]

    def oracle_step(g_color):
        gr = r.backward(g_color, np.zeros((H, W), np.float32))
        g_rot = (gr["dL_drots"] - q * np.sum(q * gr["dL_drots"], axis=1, keepdims=True)) / nq
        grads = [gr["dL_dmeans3D"], gr["dL_dsh"][:, :1], gr["dL_dsh"][:, 1:], gr["dL_dopacity"] * op * (1 - op), None, g_rot]
        out = []
        for p0, gg, lr in zip(raw0, grads, lrs):
            out.append(p0 if gg is None else O.adam_step(p0, gg.reshape(p0.shape), np.zeros_like(p0), np.zeros_like(p0), 1, float(lr), eps=1e-15)[0])
        return out
    # (a) the independent chain end to end; (b) the same with the product's dL/d(image) handed to the oracle rasteriser: takes the VGG's ReLU-flip noise out of the
    # comparison of everything downstream of the image (blend backward, projection chain, activations, Adam)
    for tag, gcol, need in (("independent", gi.numpy(), 0.97), ("shared dL/dimage", g_image_gpu.numpy(), 0.995)):
        ref = oracle_step(gcol.astype(np.float32))
        for name, p0, rf, lr, p_gpu in zip(names, raw0, ref, lrs, params):
            got = p_gpu.data.cpu().numpy()
            if name == "scaling":
                assert np.array_equal(got, p0)                                                # frozen (train_st.py:229): untouched by the step
                continue
            touched = np.abs(rf - p0) > 0                                                    # Gaussians no pixel saw have zero gradient: Adam leaves them where they are
            assert np.array_equal(np.abs(got - p0) > 0, touched) or (np.abs(got - p0) > 0).mean() > 0.5 * touched.mean()
            same = np.abs(got - rf) <= 0.02 * lr + 1e-12                                     # first Adam step = lr * sign(g): compare the moves
            assert same.mean() > need, (tag, name, same.mean())
