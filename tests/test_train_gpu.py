"""GPU parity of the training-side kernels (a11 / config 5) against PyTorch CPU autograd.  PARITY UNPINNED (own spec)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _head_reference(params, hx0, dres):
    """The head of oracle/seflow_oracle.py::head written on [h | x] rows, with autograd."""
    T = lambda k: torch.from_numpy(params[k].copy()).requires_grad_(True)
    P = {k: T(k) for k in ("head.gru.z.weight", "head.gru.z.bias", "head.gru.r.weight", "head.gru.r.bias", "head.gru.q.weight",
                           "head.gru.q.bias", "head.dec1.weight", "head.dec1.bias", "head.dec2.weight", "head.dec2.bias")}
    hx = torch.from_numpy(hx0.copy()).requires_grad_(True)
    h, x = hx[:, :128], hx[:, 128:]
    for _ in range(4):
        cat = torch.cat([h, x], 1)
        z = torch.sigmoid(cat @ P["head.gru.z.weight"] + P["head.gru.z.bias"])
        r = torch.sigmoid(cat @ P["head.gru.r.weight"] + P["head.gru.r.bias"])
        q = torch.tanh(torch.cat([r * h, x], 1) @ P["head.gru.q.weight"] + P["head.gru.q.bias"])
        h = (1 - z) * h + z * q
    y = F.gelu(torch.cat([h, x], 1) @ P["head.dec1.weight"] + P["head.dec1.bias"])
    res = y @ P["head.dec2.weight"] + P["head.dec2.bias"]
    (res * torch.from_numpy(dres[:, :3])).sum().backward()
    return res.detach().numpy(), hx.grad.numpy(), {k: v.grad.numpy() for k, v in P.items()}


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "mixed"])
def test_head_forward_backward_matches_autograd(gpu, precision):
    from himo_amd.seflow import spec
    from himo_amd.seflow.train import HeadTrainer
    params = spec.init_params(2)
    rng = np.random.default_rng(0)
    n = 3001
    hx0 = rng.normal(0, 0.7, (n, 192)).astype(np.float32)
    dres = np.zeros((n, 4), np.float32)
    dres[:, :3] = rng.normal(0, 1.0 / n, (n, 3)).astype(np.float32)
    ht = HeadTrainer(params, device=gpu, precision=precision)        # row GEMMs: float32 MFMA | split-bf16 | fp16-split forward
    res = ht.forward(torch.from_numpy(hx0).to(gpu))
    dhx = ht.backward(torch.from_numpy(dres).to(gpu))
    torch.cuda.synchronize()
    ref_res, ref_dhx, ref_g = _head_reference(params, hx0, dres)
    assert np.abs(res.cpu().numpy()[:, :3] - ref_res).max() <= 2e-5
    assert np.abs(dhx.cpu().numpy() - ref_dhx).max() <= (1e-4 if precision == "f32" else 3e-4) * max(np.abs(ref_dhx).max(), 1e-12) + 1e-9
    got = {k: v.cpu().numpy() for k, v in ht.g.items()}
    pairs = [("zr.weight", np.concatenate([ref_g["head.gru.z.weight"], ref_g["head.gru.r.weight"]], 1)),
             ("zr.bias", np.concatenate([ref_g["head.gru.z.bias"], ref_g["head.gru.r.bias"]])),
             ("q.weight", ref_g["head.gru.q.weight"]), ("q.bias", ref_g["head.gru.q.bias"]),
             ("dec1.weight", ref_g["head.dec1.weight"]), ("dec1.bias", ref_g["head.dec1.bias"]),
             ("dec2.weight", ref_g["head.dec2.weight"]), ("dec2.bias", ref_g["head.dec2.bias"])]
    for name, ref in pairs:
        g = got[name]
        if name.startswith("dec2"):
            g = g[..., :3]
        assert np.abs(g - ref).max() <= 3e-4 * max(np.abs(ref).max(), 1e-12), name


def test_tiled_wgrad_with_accumulate(gpu):
    from himo_amd import _lib
    import himo_amd.seflow.train  # noqa: F401  (registers signatures)
    lib = _lib.load()
    rng = np.random.default_rng(1)
    n, cin, cout = 4097, 192, 256
    x = torch.from_numpy(rng.normal(size=(n, cin)).astype(np.float32)).to(gpu)
    dz = torch.from_numpy(rng.normal(size=(n, cout)).astype(np.float32)).to(gpu)
    dw, db = torch.zeros((cin, cout), device=gpu), torch.zeros(cout, device=gpu)
    ws = torch.empty(int(lib.himo_wgrad_workspace_bytes_ex(n, cin, cout)), dtype=torch.uint8, device=gpu)
    for flag in (0, 1):
        _lib.check(lib.himo_linear_wgrad_ex(n, x.data_ptr(), cin, cin, dz.data_ptr(), cout, cout, dw.data_ptr(), db.data_ptr(), flag,
                                            ws.data_ptr(), ws.numel(), _lib.stream_handle()))
    ref = (x.double().T @ dz.double()).float() * 2
    assert (dw - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert (db - 2 * dz.double().sum(0).float()).abs().max().item() <= 2e-2


@pytest.mark.parametrize("stride,h,w,cin,cout,n", [(1, 24, 40, 64, 64, 2), (2, 32, 48, 32, 64, 3), (1, 16, 16, 192, 128, 1), (1, 16, 64, 64, 128, 2), (1, 8, 32, 128, 64, 3), (1, 64, 96, 64, 64, 1), (2, 16, 64, 32, 64, 3), (2, 8, 128, 64, 128, 2), (2, 32, 64, 128, 256, 1),
                                                   (2, 16, 24, 128, 256, 1)])
def test_conv3x3_backward_matches_autograd(gpu, stride, h, w, cin, cout, n):
    from himo_amd.seflow.train import conv3x3_backward_nhwc
    rng = np.random.default_rng(stride * 100 + cin)
    x = rng.normal(size=(n, h, w, cin)).astype(np.float32)
    wt = (rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    ho, wo = (h // 2, w // 2) if stride == 2 else (h, w)
    dy = rng.normal(size=(n, ho, wo, cout)).astype(np.float32)
    dx, dw, db = conv3x3_backward_nhwc(*(torch.from_numpy(a).to(gpu) for a in (x, wt)), torch.from_numpy(dy).to(gpu), stride)
    tx = torch.from_numpy(x).permute(0, 3, 1, 2).double().requires_grad_(True)
    tw = torch.from_numpy(wt).permute(3, 2, 0, 1).double().requires_grad_(True)
    y = F.conv2d(tx, tw, None, stride=stride, padding=1)
    (y * torch.from_numpy(dy).permute(0, 3, 1, 2).double()).sum().backward()
    ref_dx = tx.grad.permute(0, 2, 3, 1).numpy()
    ref_dw = tw.grad.permute(2, 3, 1, 0).numpy()
    assert np.abs(dx.cpu().numpy() - ref_dx).max() <= 1e-4 * np.abs(ref_dx).max()
    assert np.abs(dw.cpu().numpy() - ref_dw).max() <= 1e-4 * np.abs(ref_dw).max()
    assert np.abs(db.cpu().numpy() - dy.sum((0, 1, 2))).max() <= 1e-3


@pytest.mark.parametrize("h,w,cin,cout,n,stride", [(24, 64, 64, 64, 2, 1), (16, 32, 192, 128, 1, 1), (64, 96, 64, 128, 3, 1), (128, 128, 128, 64, 2, 1),
                                                   (32, 64, 32, 64, 3, 2), (16, 128, 64, 128, 2, 2), (64, 192, 128, 256, 1, 2), (2, 64, 32, 64, 1, 2)])
def test_split_bf16_weight_gradient_matches_autograd(gpu, h, w, cin, cout, n, stride):
    """himo_conv3x3_wgrad_batch with flag 2 (the training default, precision "mixed"): operands x = h + m in bf16 (16 significant
    bits) on the 16-bit matrix instructions, float32 accumulation -- against float64 autograd, and against the float32-MFMA
    kernel of the same entry point.  Image borders, several images and ragged tile runs per block are all in the shapes."""
    from himo_amd.seflow.train import conv3x3_backward_nhwc
    rng = np.random.default_rng(h + cin)
    x = (rng.normal(size=(n, h, w, cin)) * rng.uniform(0.1, 3.0, size=(1, 1, 1, cin))).astype(np.float32)
    wt = (rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    dy = (rng.normal(size=(n, h // stride, w // stride, cout)) * 10.0 ** rng.uniform(-6, 0, size=(1, 1, 1, cout))).astype(np.float32)   # gradients span decades
    args = [torch.from_numpy(a).to(gpu) for a in (x, wt, dy)]
    _, dw_split, _ = conv3x3_backward_nhwc(*args, stride, wgrad_flags=2)       # stride 2: the de-interleaved halo (conv_wgrad_split2_kernel)
    _, dw_f32, _ = conv3x3_backward_nhwc(*args, stride, wgrad_flags=0)
    tx = torch.from_numpy(x).permute(0, 3, 1, 2).double()
    tw = torch.from_numpy(wt).permute(3, 2, 0, 1).double().requires_grad_(True)
    (F.conv2d(tx, tw, None, stride=stride, padding=1) * torch.from_numpy(dy).permute(0, 3, 1, 2).double()).sum().backward()
    ref = tw.grad.permute(2, 3, 1, 0).numpy()
    # per output channel (the columns' scales differ by six decades): error against that channel's largest gradient
    scale = np.abs(ref).max(axis=(0, 1, 2))
    err_split = (np.abs(dw_split.cpu().numpy() - ref) / scale).max()
    err_f32 = (np.abs(dw_f32.cpu().numpy() - ref) / scale).max()
    assert err_f32 <= 1e-4 and err_split <= 1e-4, (err_f32, err_split)
    # flags bit 2: one block per CU (the launch runs beside another stream's kernels): the same products over another split of the pixel chunks
    _, dw_beside, _ = conv3x3_backward_nhwc(*args, stride, wgrad_flags=2 | 4)
    assert (np.abs(dw_beside.cpu().numpy() - ref) / scale).max() <= 1e-4
    assert (np.abs(dw_beside.cpu().numpy() - dw_split.cpu().numpy()) / scale).max() <= 2e-6
    if stride == 1:                      # the bias gradient from the same pass over dY (himo_conv3x3_wgrad_batch_bias)
        from himo_amd import _lib
        lib = _lib.load()
        X, _, DY = args
        dw2, db2 = torch.empty_like(dw_split), torch.empty(cout, device=gpu)
        ws = torch.empty(int(lib.himo_conv_wgrad_batch_workspace_bytes(n, h, w, cin, cout, 1)), dtype=torch.uint8, device=gpu)
        _lib.check(lib.himo_conv3x3_wgrad_batch_bias(n, X.data_ptr(), h * w * cin, cin, h, w, cin, DY.data_ptr(), h * w * cout, cout, cout, 1,
                                                     dw2.data_ptr(), db2.data_ptr(), 2, ws.data_ptr(), ws.numel(), _lib.stream_handle()), "wgrad_bias")
        assert torch.equal(dw2, dw_split)
        ref_db = dy.astype(np.float64).sum((0, 1, 2))
        assert (np.abs(db2.cpu().numpy() - ref_db) / np.abs(dy).sum((0, 1, 2))).max() <= 1e-6


@pytest.mark.parametrize("stride,h,w,cin,cout,n", [(1, 24, 40, 64, 64, 2), (1, 16, 64, 128, 256, 1), (1, 64, 96, 256, 64, 1), (2, 32, 64, 64, 128, 2),
                                                   (2, 16, 128, 128, 32, 1)])
def test_two_term_bf16_convolution_for_the_data_gradients(gpu, stride, h, w, cin, cout, n):
    """HIMO_PACK_BF16X2 (x = h + m in bf16, three matrix products per block): the arithmetic of the mixed-precision training
    step's 3x3 data-gradient convolutions.  Inputs span eight decades (gradients do): relative to each output's own scale the
    error stays at the 2^-16 of the operands, where the fp16 split would have flushed most of the input to zero."""
    from himo_amd.seflow.model import conv2d_nhwc
    rng = np.random.default_rng(h * 7 + cin)
    x = (rng.normal(size=(n, h, w, cin)) * 10.0 ** rng.uniform(-8, 0, size=(n, h, w, 1))).astype(np.float32)
    wt = (rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = np.zeros(cout, np.float32)
    y = conv2d_nhwc(*(torch.from_numpy(a).to(gpu) for a in (x, wt, b)), stride=stride, precision="bf16x2").cpu().numpy()
    tx, tw = torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(wt).permute(3, 2, 0, 1).double()
    ref = F.conv2d(tx, tw, None, stride=stride, padding=1).permute(0, 2, 3, 1).numpy()
    # scale of an output = sum of |x| |w| over its window (what the rounding errors are relative to)
    mag = F.conv2d(tx.abs(), tw.abs(), None, stride=stride, padding=1).permute(0, 2, 3, 1).numpy()
    assert (np.abs(y - ref) / mag).max() <= 2.0 ** -15
    if stride == 1:                      # HIMO_ACT_ACCUMULATE: the same product added into an existing map, in the epilogue
        from himo_amd.seflow.model import ACT_ACCUMULATE
        base = rng.normal(size=y.shape).astype(np.float32)
        out = torch.from_numpy(base.copy()).to(gpu)
        conv2d_nhwc(*(torch.from_numpy(a).to(gpu) for a in (x, wt, b)), stride=1, precision="bf16x2", act_layout=ACT_ACCUMULATE, out=out)
        assert np.array_equal(out.cpu().numpy(), base + y)


@pytest.mark.parametrize("h,w,cin,cout,n", [(16, 32, 64, 32, 2), (24, 64, 128, 64, 3), (8, 96, 256, 128, 1), (64, 64, 64, 32, 1)])
def test_stride2_data_gradient_reads_the_compact_map_as_its_zero_stuffed_image(gpu, h, w, cin, cout, n):
    """HIMO_ACT_STUFFED_2X: the two-term bf16 3x3 kernel convolving a compact [h][w] gradient map as its zero-stuffed [2h][2w] image (the
    data gradient of a stride-2 layer, himo_amd/seflow/train.py) returns the bits of the same kernel on the materialised image
    (himo_zero_stuff2x) -- every tile variant, alone and with HIMO_ACT_ACCUMULATE -- and equals the float64 transposed convolution."""
    from himo_amd import _lib
    from himo_amd.seflow.model import ACT_ACCUMULATE, ACT_STUFFED_2X, conv2d_nhwc
    import himo_amd.seflow.train  # noqa: F401  (registers himo_zero_stuff2x)
    lib = _lib.load()
    rng = np.random.default_rng(h + 3 * cin)
    g = (rng.normal(size=(n, h, w, cin)) * 10.0 ** rng.uniform(-6, 0, size=(n, h, w, 1))).astype(np.float32)
    wt = (rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = np.zeros(cout, np.float32)
    tg, tw, tb = (torch.from_numpy(a).to(gpu) for a in (g, wt, b))
    z = torch.empty((n, 2 * h, 2 * w, cin), dtype=torch.float32, device=gpu)
    _lib.check(lib.himo_zero_stuff2x(n, h, w, cin, tg.data_ptr(), h * w * cin, cin, z.data_ptr(), 4 * h * w * cin, cin, _lib.stream_handle()), "stuff")
    from himo_amd.seflow.train import SeFlowTrainer
    for hint in SeFlowTrainer.TILE_HINTS:
        try:
            want = conv2d_nhwc(z, tw, tb, precision="bf16x2", tile_hint=hint)
        except Exception:
            continue                                          # a variant this shape does not admit
        got = conv2d_nhwc(tg, tw, tb, precision="bf16x2", tile_hint=hint, act_layout=ACT_STUFFED_2X)
        assert got.shape == want.shape and torch.equal(got, want), hint
    base = torch.from_numpy(rng.normal(size=tuple(want.shape)).astype(np.float32)).to(gpu)
    out = base.clone()
    conv2d_nhwc(tg, tw, tb, precision="bf16x2", act_layout=ACT_STUFFED_2X | ACT_ACCUMULATE, out=out)
    assert torch.equal(out, base + want)
    # = the adjoint of the stride-2 convolution with the flipped kernel: conv_transpose2d(g, flip(w)) cropped to [2h][2w]
    ref = F.conv2d(z.permute(0, 3, 1, 2).double().cpu(), torch.from_numpy(wt).permute(3, 2, 0, 1).double(), None, padding=1).permute(0, 2, 3, 1).numpy()
    mag = F.conv2d(z.permute(0, 3, 1, 2).double().cpu().abs(), torch.from_numpy(wt).permute(3, 2, 0, 1).double().abs(), None, padding=1).permute(0, 2, 3, 1).numpy()
    assert (np.abs(want.cpu().numpy() - ref) / np.maximum(mag, 1e-30)).max() <= 2.0 ** -15
    # odd stuffed sizes / other formats are refused
    with pytest.raises(Exception):
        conv2d_nhwc(tg, tw, tb, precision="f16x2", act_layout=ACT_STUFFED_2X)


@pytest.mark.parametrize("n,cin,cout", [(5000, 128, 128), (70_001, 192, 256), (33_333, 384, 64), (4096, 3, 64)])
def test_split_bf16_linear_weight_gradient(gpu, n, cin, cout):
    """himo_linear_wgrad_ex with flag 2 (split-bf16 operands, the mixed training default for the 1x1 layers and the head):
    dW = X^T dZ and the column sums against float64, with the columns of dZ spanning six decades and an accumulate pass."""
    from himo_amd import _lib
    import himo_amd.seflow.train  # noqa: F401  (registers the entry points)
    lib = _lib.load()
    rng = np.random.default_rng(n)
    x = torch.from_numpy((rng.normal(size=(n, cin)) * rng.uniform(0.1, 3.0, size=(1, cin))).astype(np.float32)).to(gpu)
    dz = torch.from_numpy((rng.normal(size=(n, cout)) * 10.0 ** rng.uniform(-6, 0, size=(1, cout))).astype(np.float32)).to(gpu)
    dw = torch.zeros((cin, cout), dtype=torch.float32, device=gpu)
    db = torch.zeros(cout, dtype=torch.float32, device=gpu)
    ws = torch.empty(int(lib.himo_wgrad_workspace_bytes_ex(n, cin, cout)), dtype=torch.uint8, device=gpu)
    for flag in (2, 3):                                                      # overwrite, then accumulate on top
        _lib.check(lib.himo_linear_wgrad_ex(n, x.data_ptr(), cin, cin, dz.data_ptr(), cout, cout, dw.data_ptr(), db.data_ptr(), flag,
                                            ws.data_ptr(), ws.numel(), _lib.stream_handle()))
    ref = (x.double().T @ dz.double()).cpu().numpy() * 2
    scale = np.abs(ref).max(axis=0)
    assert (np.abs(dw.cpu().numpy() - ref) / scale).max() <= 1e-4
    ref_b = dz.double().sum(0).cpu().numpy() * 2
    assert np.abs(db.cpu().numpy() - ref_b).max() <= 1e-4 * np.abs(ref_b).max() + 1e-3 * np.abs(dz.cpu().numpy()).max() * n ** 0.5 * 1e-3


def test_two_term_bf16_row_gemm(gpu):
    """HIMO_PACK_BF16X2 for 1x1 layers / row GEMMs (dX = dZ W^T of the head and the 1x1 convolutions in the mixed training step)."""
    from himo_amd.seflow.model import conv2d_nhwc
    rng = np.random.default_rng(17)
    for n, cin, cout in ((70_001, 256, 192), (4097, 32, 192), (9000, 128, 384)):
        x = (rng.normal(size=(1, 1, n, cin)) * 10.0 ** rng.uniform(-8, 0, size=(1, 1, n, 1))).astype(np.float32)
        wt = (rng.normal(size=(1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
        y = conv2d_nhwc(*(torch.from_numpy(a).to(gpu) for a in (x, wt, np.zeros(cout, np.float32))), precision="bf16x2").cpu().numpy()[0, 0]
        ref = x[0, 0].astype(np.float64) @ wt[0, 0].astype(np.float64)
        mag = np.abs(x[0, 0]).astype(np.float64) @ np.abs(wt[0, 0]).astype(np.float64)
        assert (np.abs(y - ref) / mag).max() <= 2.0 ** -15, (n, cin, cout)


def test_upsample2x_backward_is_the_adjoint(gpu):
    from himo_amd.seflow.train import upsample2x_backward_nhwc
    rng = np.random.default_rng(5)
    for h, w, c in [(8, 12, 8), (1, 5, 4), (32, 32, 64)]:
        dy = rng.normal(size=(2 * h, 2 * w, c)).astype(np.float32)
        dx = upsample2x_backward_nhwc(torch.from_numpy(dy).to(gpu)).cpu().numpy()
        tx = torch.zeros((1, c, h, w), dtype=torch.float64, requires_grad=True)
        y = F.interpolate(tx, scale_factor=2, mode="bilinear", align_corners=True)
        (y * torch.from_numpy(dy).permute(2, 0, 1)[None].double()).sum().backward()
        ref = tx.grad[0].permute(1, 2, 0).numpy()
        assert np.abs(dx - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1.0), (h, w, c)


# per-tensor bar of the full-network gradient test: max |g - autograd| / max |autograd|.  2e-3 for every arithmetic until round 6;
# now what the recorded margins show (profiles/r06_grad_margins.json: worst tensor 1.2e-5 .. 3.1e-5 in f32 / bf16x3, 3.0e-5 .. 4.3e-5
# in `mixed`, at 6k and at 120k points) with a factor ~2.3 of head-room: the 16-significant-bit products of `mixed` cost nothing here
GRAD_BAR = {"f32": 1e-4, "bf16x3": 1e-4, "mixed": 1e-4}


def _record_margins(case: str, worst: dict):
    """gpurun_out/r06_grad_margins.json: {case: {tensor: relative error}} (+ the worst tensor per case), merged across the test's cases"""
    import json
    from pathlib import Path
    path = Path(__file__).resolve().parents[1] / "gpurun_out" / "r06_grad_margins.json"
    try:
        path.parent.mkdir(exist_ok=True)
        data = json.loads(path.read_text()) if path.exists() else {}
        k_worst = max(worst, key=worst.get)
        data[case] = {"worst_tensor": k_worst, "worst": float(worst[k_worst]), "per_tensor": {k: float(v) for k, v in sorted(worst.items())}}
        path.write_text(json.dumps(data, indent=1, sort_keys=True))
    except OSError:
        pass


def _sample(n, seed=0):
    from himo_amd.synthetic import make_frame
    f = make_frame(seed, n_points=n, n_instances=6)
    rng = np.random.default_rng(seed)
    pose0, pose1 = np.asarray(f["pose0"], np.float64), np.asarray(f["pose1"], np.float64)
    pose_h = pose0.copy(); pose_h[:3, 3] -= pose1[:3, 3] - pose0[:3, 3]
    pc0 = np.asarray(f["pc0"], np.float32)[:, :3]
    pc1 = (pc0 + rng.normal(0, 0.05, pc0.shape)).astype(np.float32)[rng.permutation(len(pc0))[: n - 37]]
    pch = (pc0 + rng.normal(0, 0.05, pc0.shape)).astype(np.float32)[rng.permutation(len(pc0))[: n - 101]]
    return pch, pc0, pc1, pose_h, pose0, pose1


@pytest.mark.parametrize("precision,n_points,batchnorm", [
    ("bf16x3", 6000, "frozen"), ("mixed", 6000, "frozen"), ("f32", 6000, "frozen"), ("mixed", 120_000, "frozen"),
    ("f32", 6000, "batch"), ("mixed", 6000, "batch"), ("bf16x3", 6000, "batch"), ("mixed", 120_000, "batch")])
def test_full_network_gradients_match_autograd(gpu, precision, n_points, batchnorm):
    """Every trainable tensor's gradient (pillar net, 16 encoder convs, decoder, head) against CPU autograd through the
    oracle network, for the linear functional L = sum(res * G) -- at test size in all three arithmetics and once at
    BASELINE size (3 x 120k points) in the training default.  ``batchnorm="batch"`` (BASELINE config 5: the reference's job
    trains from scratch): BatchNorm in training mode on both sides -- batch statistics, gamma / beta gradients, and the
    running statistics after the forward pass equal to torch's."""
    import oracle.seflow_oracle as so
    from himo_amd.seflow import spec
    from himo_amd.seflow.train import SeFlowTrainer
    params = spec.init_params(4)
    pch, pc0, pc1, pose_h, pose0, pose1 = _sample(n_points, seed=3)
    tr = SeFlowTrainer(params, device=gpu, max_points=n_points + 2000, precision=precision, batchnorm=batchnorm)
    res = tr.forward(pch, pc0, pc1, pose_h, pose0, pose1)
    rng = np.random.default_rng(9)
    G = np.zeros((len(pc0), 4), np.float32)
    G[:, :3] = rng.normal(0, 1.0, (len(pc0), 3)).astype(np.float32)
    tr.backward(torch.from_numpy(G).to(gpu))
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in tr.g.items()}
    res_gpu = res.cpu().numpy()
    batch = batchnorm == "batch"
    assert any(k.endswith(".bn.gamma") for k in got) == batch

    torch.set_num_threads(max(1, torch.get_num_threads()))
    P = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    for k in P:
        if k.endswith(".weight") or k.endswith(".bias") or (batch and (k.endswith(".gamma") or k.endswith(".beta"))):
            P[k].requires_grad_(True)
    ref_res, valid, _ = so.forward_train(P, pch, pc0, pc1, pose_h, pose0, pose1, training=batch)
    v = valid.numpy()
    assert np.abs(res_gpu[v][:, :3] - ref_res.detach().numpy()).max() <= 1e-4
    assert np.all(res_gpu[~v] == 0)
    (ref_res * torch.from_numpy(G[v][:, :3])).sum().backward()
    _compare_with_autograd(tr, params, P, got, batch, precision, f"{precision}/{n_points}/{batchnorm}")


def _compare_with_autograd(tr, params, P, got, batch, precision, case):
    """every trainable tensor's gradient in ``got`` against the autograd leaves of ``P`` (+ the running statistics in batch mode)"""
    from himo_amd.seflow import spec
    ref = {k: P[k].grad.numpy() for k in P if P[k].grad is not None}
    pairs = {k: ref[k] for k in got if k in ref}
    pairs["head.zr.weight"] = np.concatenate([ref["head.gru.z.weight"], ref["head.gru.r.weight"]], 1)
    pairs["head.zr.bias"] = np.concatenate([ref["head.gru.z.bias"], ref["head.gru.r.bias"]])
    pairs["head.q.weight"], pairs["head.q.bias"] = ref["head.gru.q.weight"], ref["head.gru.q.bias"]
    assert set(pairs) | {"head.dec2.weight", "head.dec2.bias"} >= set(got), set(got) - set(pairs)
    worst = {}
    for k, r in pairs.items():
        g = got[k]
        if k.startswith("head.dec2"):
            g = g[..., :3]
        if batch and k.startswith("enc") and k.endswith(".bias") and ".bn." not in k:
            # a bias in front of a training-mode BatchNorm has NO gradient (the batch mean absorbs it): both sides hold
            # rounding noise around zero, far below the layer's weight gradient
            floor = 1e-4 * np.abs(pairs[k[:-4] + "weight"]).max()
            assert np.abs(g).max() <= floor and np.abs(r).max() <= floor, (k, np.abs(g).max(), np.abs(r).max(), floor)
            continue
        scale = max(np.abs(r).max(), 1e-12)
        worst[k] = np.abs(g - r).max() / scale
    _record_margins(case, worst)
    bad = {k: e for k, e in worst.items() if not e <= GRAD_BAR[precision]}
    assert not bad, bad
    if batch:
        # running statistics after ONE training-mode forward (momentum 0.1, unbiased variance; the pillar net's moved three
        # times, once per frame slot in call order) against torch's in-place updates of the oracle's tensors
        for prefix in ["pfn.bn"] + [f"{n}.bn" for n, *_ in spec.ENCODER]:
            for stat in ("mean", "var"):
                mine, theirs = tr.net.p[f"{prefix}.{stat}"].cpu().numpy(), P[f"{prefix}.{stat}"].numpy()
                assert not np.array_equal(theirs, params[f"{prefix}.{stat}"]), prefix          # it did move
                assert np.abs(mine - theirs).max() <= 2e-5 * max(np.abs(theirs).max(), 1.0), (prefix, stat, np.abs(mine - theirs).max())


@pytest.mark.parametrize("precision,n_points,batchnorm,B", [
    ("mixed", 6000, "batch", 2), ("bf16x3", 6000, "batch", 2), ("f32", 6000, "batch", 2), ("mixed", 6000, "frozen", 3),
    ("mixed", 120_000, "batch", 8)])
def test_batched_pass_gradients_match_autograd(gpu, precision, n_points, batchnorm, B):
    """a per-process BATCH as ONE pass (the launcher's ``batch_size=8``, assets/slurm/ssl-train-av2.sh:32-34) --
    the encoder's launches over B x F images, the decoder's over B, BatchNorm statistics over the whole batch (pillar net: all the
    batch's sweeps of a frame slot; encoder: all B x F images), as torch.nn.BatchNorm does -- against CPU autograd through the oracle's
    batched network (``forward_train_batch``), for L = sum over the samples of sum(res_b * G_b): every trainable tensor's gradient, the
    per-sample outputs and the running statistics.  B = 2 in the three arithmetics, B = 3 with frozen statistics, and B = 8 at BASELINE
    size (8 x 3 x 120k points) in the training default; the samples differ in size."""
    import oracle.seflow_oracle as so
    from himo_amd.seflow import spec
    from himo_amd.seflow.train import SeFlowTrainer
    params = spec.init_params(4)
    samples = [_sample(n_points - 431 * b, seed=3 + b) for b in range(B)]
    tr = SeFlowTrainer(params, device=gpu, max_points=n_points + 2000, precision=precision, batchnorm=batchnorm, batch=B)
    res = tr.forward_batch(samples)
    rng = np.random.default_rng(9)
    Gs = []
    for smp in samples:
        G = np.zeros((len(smp[1]), 4), np.float32)
        G[:, :3] = rng.normal(0, 1.0, (len(smp[1]), 3)).astype(np.float32)
        Gs.append(G)
    tr.backward_batch([torch.from_numpy(G).to(gpu) for G in Gs])
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in tr.g.items()}
    res_gpu = [r.cpu().numpy() for r in res]
    batch = batchnorm == "batch"
    P = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    for k in P:
        if k.endswith(".weight") or k.endswith(".bias") or (batch and (k.endswith(".gamma") or k.endswith(".beta"))):
            P[k].requires_grad_(True)
    outs = so.forward_train_batch(P, samples, training=batch)
    total = 0.0
    for (ref_res, valid, _), G, rg in zip(outs, Gs, res_gpu):
        v = valid.numpy()
        assert np.abs(rg[v][:, :3] - ref_res.detach().numpy()).max() <= 1e-4
        assert np.all(rg[~v] == 0)
        total = total + (ref_res * torch.from_numpy(G[v][:, :3])).sum()
    total.backward()
    _compare_with_autograd(tr, params, P, got, batch, precision, f"{precision}/{n_points}/{batchnorm}/batch{B}")


@pytest.mark.parametrize("batchnorm", ["batch", "frozen"])
def test_a_batch_of_one_in_a_larger_trainer_is_the_single_sample_pass(gpu, batchnorm):
    """the batch layout (all images as channel groups of pixel-major maps, pitch C * F * batch) changes addresses, not arithmetic: one
    sample through a trainer built for four gives the loss BITS of the trainer built for one and its gradients to float32 round-off
    (the weight-gradient kernels split their reductions by the size of the workspace they are handed, which grows with the batch
    capacity: another summation order, not another sum); and a different batch through the same buffers first (2 samples, then 1)
    leaves nothing behind"""
    from himo_amd.seflow import spec
    from himo_amd.seflow.train import SeFlowTrainer
    params = spec.init_params(6)
    (args, lab0, lab1), (args2, lab0b, lab1b) = _labelled_sample(5000, 21), _labelled_sample(4600, 22)
    mk = lambda a, l0, l1: (*a, torch.from_numpy(l0).to(gpu), torch.from_numpy(l1).to(gpu), int(l0.max()) + 1)
    smp, smp2 = mk(args, lab0, lab1), mk(args2, lab0b, lab1b)
    outs = []
    for B in (1, 4):
        tr = SeFlowTrainer(params, device=gpu, max_points=6000, batchnorm=batchnorm, batch=B)
        if B == 4:
            tr.loss_and_grad_batch([smp2, smp])                  # something else first: the buffers are not fresh
        terms, total = tr.loss_and_grad(*smp)
        torch.cuda.synchronize()
        outs.append((float(total.item()), {k: v.clone() for k, v in tr.g.items()}, tr.net.p["enc2.1.bn.mean"].clone()))
        del tr
        torch.cuda.empty_cache()
    assert outs[0][0] == outs[1][0]
    for k, a in outs[0][1].items():
        if batchnorm == "batch" and k.startswith("enc") and k.endswith(".bias") and ".bn." not in k:
            continue                                              # (exactly zero by construction on both sides)
        assert (a - outs[1][1][k]).abs().max().item() <= 3e-6 * max(a.abs().max().item(), 1e-12), k
    if batchnorm == "frozen":
        assert torch.equal(outs[0][2], outs[1][2])


def _labelled_sample(n, seed):
    from himo_amd.synthetic import make_frame
    f = make_frame(seed, n_points=n, n_instances=8)
    rng = np.random.default_rng(seed)
    pose0, pose1 = np.asarray(f["pose0"], np.float64), np.asarray(f["pose1"], np.float64)
    ego = np.linalg.inv(pose1) @ pose0
    pc0 = np.asarray(f["pc0"], np.float32)[:, :3]
    lab0 = np.asarray(f["flow_instance_id"], np.int32)
    moved = (pc0.astype(np.float64) + f["flow"].astype(np.float64)).astype(np.float32)      # pc0 at t1, in pc1's frame
    perm = rng.permutation(n)
    pc1, lab1 = moved[perm], lab0[perm]
    pose_h = pose0 @ np.linalg.inv(ego)                                                    # one step back in time
    pch = (pc0.astype(np.float64) - (f["flow"].astype(np.float64) - (pc0 @ ego[:3, :3].T + ego[:3, 3] - pc0))).astype(np.float32)
    return (pch, pc0, pc1, pose_h, pose0, pose1), lab0, lab1


@pytest.mark.parametrize("batchnorm,lr,steps", [("frozen", 1e-3, 6), ("batch", 1e-4, 8)])
def test_train_steps_reduce_the_loss(gpu, batchnorm, lr, steps):
    """(learning rates from scripts/exp_bn_lr.py: with BatchNorm in training mode Adam at 1e-3 overshoots on the first steps --
    the loss spikes 5.8 -> 96 before it recovers -- while 1e-4, next to the launcher's 6e-5, falls monotonically)"""
    from himo_amd.seflow import spec
    from himo_amd.seflow.train import SeFlowTrainer
    (pch, pc0, pc1, pose_h, pose0, pose1), lab0, lab1 = _labelled_sample(8000, 11)
    tr = SeFlowTrainer(spec.init_params(5), device=gpu, max_points=8000, batchnorm=batchnorm)
    l0, l1 = torch.from_numpy(lab0).to(gpu), torch.from_numpy(lab1).to(gpu)
    totals = []
    for _ in range(steps):
        _, total = tr.train_step(pch, pc0, pc1, pose_h, pose0, pose1, l0, l1, n_labels=int(lab0.max()) + 1, lr=lr)
        totals.append(float(total.item()))
    assert all(np.isfinite(totals)), totals
    assert totals[-1] < 0.9 * totals[0], totals
    # the exported parameters drive the inference network to the trainer's own forward result
    from himo_amd.seflow.model import SeFlowNet
    res = tr.forward(pch, pc0, pc1, pose_h, pose0, pose1, training=False)[:, :3].cpu().numpy()      # eval mode: running statistics
    net = SeFlowNet(tr.export_params(), device=gpu, max_points=8000, precision="f32")
    flow = net.forward(pch, pc0, pc1, pose_h, pose0, pose1).cpu().numpy()
    ego = np.linalg.inv(pose1) @ pose0
    valid = net.pid[1][: len(pc0)].cpu().numpy() >= 0
    pose_flow = net.xyz_t[1][: len(pc0)].cpu().numpy() - pc0
    assert np.abs((flow - pose_flow)[valid] - res[valid]).max() <= 2e-4


def test_mixed_and_fp32_class_runs_follow_the_same_trajectory(gpu):
    """the training default (`mixed`: fp16-split forward, two-term bf16 = 16-significant-bit products in every
    data and weight gradient) against the float32-class arithmetics (`bf16x3` split, `f32` MFMA): 200 optimiser steps from ONE
    initialisation on 5 rotating samples of one synthetic drive (BatchNorm in training mode, Adam), then a held-out sample.

    What holds, and is asserted: the three loss curves agree to < 1 % over the first 50 steps; afterwards they drift apart -- ALL of
    them, f32 from bf16x3 as much as mixed from bf16x3 (Adam turns last-bit gradient differences into full-size steps) -- and
    `mixed` stays inside that spread: its mean distance to the bf16x3 curve, its held-out loss and its held-out flow differ from
    bf16x3's by no more than twice what f32's do (plus a floor).  Measured at BASELINE size, 3 x 120k points
    (profiles/r06_train_trajectory.txt; also 60k and 30k, where ALL the curves spread further): first-50 0.05 % (mixed) / 0.06 %
    (f32); mean 0.57 % / 0.65 %; held-out loss 1.2 % / 0.4 %; held-out flow mean end-point difference 3.1 cm / 2.7 cm.  "Held-out
    flow within 1e-3 m" holds for NO pair of arithmetics, float32-class pairs included."""
    from himo_amd.dataset import ListDataset
    from himo_amd.seflow import spec
    from himo_amd.seflow.fit import make_sample, triplets
    from himo_amd.seflow.train import SeFlowTrainer
    from himo_amd.synthetic import make_scene
    P, steps = 120_000, 200
    ds, held = ListDataset(make_scene(7, 7, n_points=P)), ListDataset(make_scene(8, 3, n_points=P))
    samples = [make_sample(ds, t, gpu, "flow_instance_id") for t in triplets(ds)[:5]]
    held_s = make_sample(held, (0, 1, 2), gpu, "flow_instance_id")
    runs = {}
    for prec in ("mixed", "bf16x3", "f32"):
        tr = SeFlowTrainer(spec.init_params(3, fresh_bn=True), device=gpu, max_points=P, precision=prec, batchnorm="batch")
        losses = [tr.train_batch([samples[k % len(samples)]], lr=1e-4) for k in range(steps)]
        curve = torch.stack(losses).cpu().numpy().astype(np.float64)
        flow = tr.forward(*held_s[:6], training=False)[:, :3].cpu().numpy()
        runs[prec] = (curve, flow, float(tr.loss_only(*held_s).item()))
        del tr
        torch.cuda.empty_cache()
    ref_curve, ref_flow, ref_val = runs["bf16x3"]
    assert np.isfinite(ref_curve).all() and ref_curve[-20:].mean() < 0.5 * ref_curve[0]              # it trains
    d = {}
    for prec in ("mixed", "f32"):
        curve, flow, val = runs[prec]
        rel = np.abs(curve - ref_curve) / np.abs(ref_curve)
        d[prec] = {"first50": rel[:50].max(), "mean": rel.mean(), "val": abs(val - ref_val) / ref_val,
                   "epe": np.linalg.norm(flow - ref_flow, axis=1).mean(), "tail": abs(curve[-20:].mean() - ref_curve[-20:].mean()) / ref_curve[-20:].mean()}
        assert np.isfinite(curve).all()
        assert d[prec]["first50"] <= 1e-2, (prec, d[prec])
        assert d[prec]["tail"] <= 0.08 and d[prec]["val"] <= 0.08, (prec, d[prec])
    assert d["mixed"]["mean"] <= 2 * d["f32"]["mean"] + 0.01, d
    assert d["mixed"]["val"] <= 2 * d["f32"]["val"] + 0.03, d
    assert d["mixed"]["epe"] <= 2 * d["f32"]["epe"] + 0.02, d


@pytest.mark.parametrize("batchnorm", ["frozen", "batch"])
def test_train_batch_averages_the_per_sample_gradients(gpu, batchnorm):
    """(batch mode: the statistics are per forward call, i.e. per sample, so the batch gradient is still the plain average)"""
    from himo_amd.seflow import spec
    from himo_amd.seflow.train import SeFlowTrainer
    tr = SeFlowTrainer(spec.init_params(6), device=gpu, max_points=6000, batchnorm=batchnorm)
    batch, grads = [], []
    for seed in (21, 22):
        args, lab0, lab1 = _labelled_sample(5000, seed)
        smp = (*args, torch.from_numpy(lab0).to(gpu), torch.from_numpy(lab1).to(gpu), int(lab0.max()) + 1)
        batch.append(smp)
        tr.loss_and_grad(*smp)
        grads.append(tr.flat_g.clone())
    p0 = tr.flat_p.clone()
    loss = tr.train_batch(batch, lr=0.0)                     # lr 0: parameters stay, flat_g holds the batch gradient
    assert np.isfinite(float(loss.item()))
    assert torch.equal(tr.flat_p, p0)
    want = (grads[0] + grads[1]) * 0.5
    assert (tr.flat_g - want).abs().max().item() <= 1e-6 * max(want.abs().max().item(), 1e-12)
    assert SeFlowTrainer.step_lr(0) == 6e-5 and SeFlowTrainer.step_lr(3) == 3e-5 and SeFlowTrainer.step_lr(7) == 1.5e-5


@pytest.mark.parametrize("n_img,rows,ch,pad", [(3, 4097, 64, 0), (1, 300, 32, 32), (2, 70_001, 128, 0), (3, 1024, 256, 256), (2, 5000, 48, 16),
                                               (1, 9, 8, 0)])
def test_batchnorm_training_kernels_match_autograd(gpu, n_img, rows, ch, pad):
    """himo_bn_train_fwd / himo_bn_train_bwd (csrc/batchnorm.hip) against float64 autograd of torch's batch_norm + GELU: statistics,
    running estimates, xhat, y, dx, dgamma, dbeta -- over narrow (every lane busy), wide and pitched maps and ragged row counts."""
    from himo_amd import _lib
    from himo_amd.seflow import train  # noqa: F401  (registers the signatures)
    lib = _lib.load()
    g = torch.Generator().manual_seed(7 + ch + rows)
    pitch = ch + pad
    x = torch.randn((n_img, rows, pitch), generator=g) * 1.7 + 0.4
    dy = torch.randn((n_img, rows, pitch), generator=g)
    gamma, beta = torch.rand(ch, generator=g) + 0.5, torch.randn(ch, generator=g) * 0.2
    rm, rv = torch.randn(ch, generator=g) * 0.1, torch.rand(ch, generator=g) + 0.5
    eps, mom = 1e-3, 0.1
    # float64 reference
    xr = x[:, :, :ch].double().reshape(-1, ch).clone().requires_grad_(True)
    gr, br = gamma.double().clone().requires_grad_(True), beta.double().clone().requires_grad_(True)
    rm_r, rv_r = rm.double().clone(), rv.double().clone()
    pre = F.batch_norm(xr, rm_r, rv_r, gr, br, training=True, momentum=mom, eps=eps)
    y_ref = F.gelu(pre)
    (y_ref * dy[:, :, :ch].double().reshape(-1, ch)).sum().backward()
    mean = xr.detach().mean(0); var = xr.detach().var(0, unbiased=False)
    xhat_ref = (xr.detach() - mean) / torch.sqrt(var + eps)
    # device
    dev = gpu
    X, DY = x.to(dev), dy.to(dev)
    XH, Y, DX = torch.zeros_like(X), torch.zeros_like(X), torch.zeros_like(X)
    G, B, RM, RV = gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev)
    M, IS, DG, DB = (torch.zeros(ch, device=dev) for _ in range(4))
    ws = torch.empty(int(lib.himo_bn_workspace_bytes(n_img * rows, ch)), dtype=torch.uint8, device=dev)
    s = _lib.stream_handle()
    _lib.check(lib.himo_bn_train_fwd(n_img, rows, ch, X.data_ptr(), rows * pitch, pitch, G.data_ptr(), B.data_ptr(), eps, mom, RM.data_ptr(),
                                     RV.data_ptr(), M.data_ptr(), IS.data_ptr(), XH.data_ptr(), rows * pitch, pitch, Y.data_ptr(), rows * pitch,
                                     pitch, ws.data_ptr(), ws.numel(), s), "bn_train_fwd")
    _lib.check(lib.himo_bn_train_bwd(n_img, rows, ch, DY.data_ptr(), rows * pitch, pitch, XH.data_ptr(), rows * pitch, pitch, G.data_ptr(),
                                     B.data_ptr(), IS.data_ptr(), DX.data_ptr(), rows * pitch, pitch, DG.data_ptr(), DB.data_ptr(), 0,
                                     ws.data_ptr(), ws.numel(), s), "bn_train_bwd")
    torch.cuda.synchronize()
    flat = lambda t: t.cpu()[:, :, :ch].reshape(-1, ch).double()
    np.testing.assert_allclose(M.cpu().double(), mean, rtol=0, atol=2e-6)
    np.testing.assert_allclose(IS.cpu().double(), 1.0 / torch.sqrt(var + eps), rtol=2e-6, atol=0)
    np.testing.assert_allclose(RM.cpu().double(), rm_r, rtol=0, atol=2e-6)
    np.testing.assert_allclose(RV.cpu().double(), rv_r, rtol=3e-6, atol=0)
    np.testing.assert_allclose(flat(XH), xhat_ref, rtol=0, atol=1e-5)
    np.testing.assert_allclose(flat(Y), y_ref.detach(), rtol=0, atol=1e-5)
    scale = float(xr.grad.abs().max())
    np.testing.assert_allclose(flat(DX), xr.grad, rtol=0, atol=2e-5 * max(scale, 1.0))
    np.testing.assert_allclose(DG.cpu().double(), gr.grad, rtol=2e-5, atol=2e-5 * float(gr.grad.abs().max()))
    np.testing.assert_allclose(DB.cpu().double(), br.grad, rtol=2e-5, atol=2e-5 * float(br.grad.abs().max()))
    if pad:                                              # the padding columns of the pitched maps stay untouched
        assert float(Y.cpu()[:, :, ch:].abs().max()) == 0.0 and float(DX.cpu()[:, :, ch:].abs().max()) == 0.0
    # the trainer's form: the forward pass writes no xhat, the backward pass re-forms it from x and the saved mean / invstd -- same bits
    Y2, DX2 = torch.zeros_like(X), torch.zeros_like(X)
    M2, IS2, DG2, DB2 = (torch.zeros(ch, device=dev) for _ in range(4))
    _lib.check(lib.himo_bn_train_fwd(n_img, rows, ch, X.data_ptr(), rows * pitch, pitch, G.data_ptr(), B.data_ptr(), eps, mom, None,
                                     None, M2.data_ptr(), IS2.data_ptr(), None, 0, 0, Y2.data_ptr(), rows * pitch,
                                     pitch, ws.data_ptr(), ws.numel(), s), "bn_train_fwd (no xhat)")
    _lib.check(lib.himo_bn_train_bwd_x(n_img, rows, ch, DY.data_ptr(), rows * pitch, pitch, X.data_ptr(), rows * pitch, pitch, G.data_ptr(),
                                       B.data_ptr(), M2.data_ptr(), IS2.data_ptr(), DX2.data_ptr(), rows * pitch, pitch, DG2.data_ptr(),
                                       DB2.data_ptr(), 0, ws.data_ptr(), ws.numel(), s), "bn_train_bwd_x")
    assert torch.equal(Y2, Y) and torch.equal(M2, M) and torch.equal(IS2, IS)
    assert torch.equal(DX2, DX) and torch.equal(DG2, DG) and torch.equal(DB2, DB)


@pytest.mark.parametrize("precision,n", [("mixed", 5000), ("bf16x3", 777), ("mixed", 64)])
def test_fused_training_head_matches_the_unfused_path(gpu, precision, n):
    """himo_gru_head_train (one launch) against himo_head_gather + HeadTrainer.forward + the row mask: the residual flow and every
    saved state of every iteration; then himo_gru_head_backward (one launch + one weight-gradient product per matrix) against the
    unfused backward pass: the same d loss / d [h0 | x] and parameter gradients."""
    from himo_amd import _lib
    from himo_amd.seflow import spec
    from himo_amd.seflow.train import HeadTrainer
    lib = _lib.load()
    params = spec.init_params(3)
    rng = np.random.default_rng(11)
    H = W = 32
    F_ = 3
    dev = gpu
    B0 = torch.from_numpy(rng.standard_normal((H * W, 32 * F_)).astype(np.float32)).to(dev)
    DEC = torch.from_numpy(rng.standard_normal((H * W, 64)).astype(np.float32)).to(dev)
    pid = rng.integers(0, H * W, n).astype(np.int32)
    pid[rng.random(n) < 0.1] = -1                                   # dropped points
    off = (rng.standard_normal((n, 3)) * 0.1).astype(np.float32)
    off[pid < 0] = 0
    PID, OFF = torch.from_numpy(pid).to(dev), torch.from_numpy(off).to(dev)
    w_off = torch.from_numpy(params["head.offset.weight"]).to(dev)
    b_off = torch.from_numpy(params["head.offset.bias"]).to(dev)
    s = _lib.stream_handle

    def run(fused):
        ht = HeadTrainer(params, device=dev, precision=precision)
        ht.fused_backward = fused               # csrc/gruheadbwd.hip against the three element-wise kernels + two row products per iteration
        if fused:
            res = ht.forward_fused(n, PID.data_ptr(), OFF.data_ptr(), B0.data_ptr() + 4 * 32, B0.data_ptr() + 4 * 64, 32 * F_,
                                   DEC.data_ptr(), 64, w_off.data_ptr(), b_off.data_ptr())
        else:
            hx0 = torch.empty((n, 192), dtype=torch.float32, device=dev)
            rhx = torch.empty((n, 192), dtype=torch.float32, device=dev)
            _lib.check(lib.himo_head_gather(n, PID.data_ptr(), OFF.data_ptr(), B0.data_ptr() + 4 * 32, B0.data_ptr() + 4 * 64, 32 * F_,
                                            DEC.data_ptr(), 64, w_off.data_ptr(), b_off.data_ptr(), hx0.data_ptr(), rhx.data_ptr(), 192, s()),
                       "head_gather")
            res = ht.forward(hx0)
            _lib.check(lib.himo_mask_rows(n, 4, PID.data_ptr(), res.data_ptr(), 4, s()), "mask_rows")
        torch.cuda.synchronize()
        saved = {"res": res.cpu().numpy().copy(), "pre1": ht.PRE1.cpu().numpy().copy(), "y1": ht.Y1.cpu().numpy().copy()}
        for t in range(spec.GRU_ITERS):
            for k, v in (("hx", ht.HX), ("rhx", ht.RHX), ("z", ht.Z), ("r", ht.R), ("q", ht.Q)):
                saved[f"{k}{t}"] = v[t].cpu().numpy().copy()
        saved["hxT"] = ht.HX[-1].cpu().numpy().copy()
        dres = torch.from_numpy(np.random.default_rng(5).standard_normal((n, 4)).astype(np.float32)).to(dev)
        dres[:, 3] = 0
        dres[PID < 0] = 0
        dhx0 = ht.backward(dres)
        torch.cuda.synchronize()
        return saved, dhx0.cpu().numpy(), {k: v.cpu().numpy().copy() for k, v in ht.g.items()}

    sa, da, ga = run(False)
    sb, db, gb = run(True)
    tol = 3e-5 if precision == "mixed" else 1e-5
    inr = pid >= 0
    for k in sa:
        a, b = sa[k], sb[k]
        if k in ("pre1", "y1"):                 # the unfused path leaves the dropped rows' decoder values unmasked; res is masked in both
            a, b = a[inr], b[inr]
        np.testing.assert_allclose(b, a, rtol=0, atol=tol, err_msg=k)
    assert np.all(sb["res"][~inr] == 0) and np.all(sb["res"][:, 3] == 0)
    np.testing.assert_allclose(db, da, rtol=0, atol=2e-4 * max(1.0, float(np.abs(da).max())))
    for k in ga:
        np.testing.assert_allclose(gb[k], ga[k], rtol=0, atol=2e-4 * max(1.0, float(np.abs(ga[k]).max())), err_msg=k)


def test_head_trainer_reuses_its_buffers_across_sweep_sizes(gpu):
    """One HeadTrainer fed sweeps of 5000, 3000, 700 and 5200 points (capacity only grows; the stacked weight-gradient products read the
    padding rows): every call gives what a fresh trainer gives for that sweep -- nothing of a larger earlier sweep leaks in."""
    from himo_amd import _lib
    from himo_amd.seflow import spec
    from himo_amd.seflow.train import HeadTrainer
    params = spec.init_params(4)
    rng = np.random.default_rng(2)
    H = W = 32
    B0 = torch.from_numpy(rng.standard_normal((H * W, 96)).astype(np.float32)).to(gpu)
    DEC = torch.from_numpy(rng.standard_normal((H * W, 64)).astype(np.float32)).to(gpu)
    w_off = torch.from_numpy(params["head.offset.weight"]).to(gpu)
    b_off = torch.from_numpy(params["head.offset.bias"]).to(gpu)

    def run(ht, n, seed):
        r = np.random.default_rng(seed)
        pid = torch.from_numpy(r.integers(0, H * W, n).astype(np.int32)).to(gpu)
        off = torch.from_numpy((r.standard_normal((n, 3)) * 0.1).astype(np.float32)).to(gpu)
        res = ht.forward_fused(n, pid.data_ptr(), off.data_ptr(), B0.data_ptr() + 128, B0.data_ptr() + 256, 96, DEC.data_ptr(), 64,
                               w_off.data_ptr(), b_off.data_ptr()).clone()
        dres = torch.from_numpy(r.standard_normal((n, 4)).astype(np.float32)).to(gpu)
        dres[:, 3] = 0
        d0 = ht.backward(dres).clone()
        torch.cuda.synchronize()
        return res.cpu().numpy(), d0.cpu().numpy(), {k: v.cpu().numpy().copy() for k, v in ht.g.items()}

    shared = HeadTrainer(params, device=gpu, precision="mixed")
    for seed, n in enumerate((5000, 3000, 700, 5200)):
        got = run(shared, n, seed)
        want = run(HeadTrainer(params, device=gpu, precision="mixed"), n, seed)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), n
        for k in want[2]:
            np.testing.assert_allclose(got[2][k], want[2][k], rtol=0, atol=1e-6 * max(1.0, float(np.abs(want[2][k]).max())), err_msg=f"{k} n={n}")


@pytest.mark.parametrize("prec,stride", [("f16x2", 1), ("f16x2", 2), ("bf16x2", 1)])
def test_tile_variants_of_the_training_convolutions_return_identical_bits(gpu, prec, stride):
    """SeFlowTrainer picks the fastest tile variant per 3x3 layer shape at the shape's first launch (train.py _tune_tile): the
    choice may differ between runs, so the variants must agree bit for bit -- float32 maps in, both split formats, stride 1 and 2."""
    from himo_amd import _lib
    from himo_amd.seflow.model import conv2d_nhwc
    from himo_amd.seflow.train import SeFlowTrainer
    torch.manual_seed(5)
    for n, h, w, cin, cout in ((3, 64, 64, 64, 64), (1, 128, 128, 128, 64), (3, 32, 64, 256, 256), (2, 96, 64, 32, 64)):
        if prec == "bf16x2" and stride == 2:
            continue
        x = torch.randn(n, h, w, cin, device=gpu) * (1e-3 if prec == "bf16x2" else 1.0)
        wt = torch.randn(3, 3, cin, cout, device=gpu) * 0.05
        b = torch.randn(cout, device=gpu) * 0.1
        ref = conv2d_nhwc(x, wt, b, stride=stride, precision=prec, tile_hint=0)
        for hint in SeFlowTrainer.TILE_HINTS[1:]:
            try:
                got = conv2d_nhwc(x, wt, b, stride=stride, precision=prec, tile_hint=hint)
            except (_lib.HimoError, ValueError):
                continue                                   # a variant the shape does not admit
            assert torch.equal(got, ref), (prec, stride, (n, h, w, cin, cout), hex(hint))

