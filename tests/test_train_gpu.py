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


def test_head_forward_backward_matches_autograd(gpu):
    from himo_amd.seflow import spec
    from himo_amd.seflow.train import HeadTrainer
    params = spec.init_params(2)
    rng = np.random.default_rng(0)
    n = 3001
    hx0 = rng.normal(0, 0.7, (n, 192)).astype(np.float32)
    dres = np.zeros((n, 4), np.float32)
    dres[:, :3] = rng.normal(0, 1.0 / n, (n, 3)).astype(np.float32)
    ht = HeadTrainer(params, device=gpu)
    res = ht.forward(torch.from_numpy(hx0).to(gpu))
    dhx = ht.backward(torch.from_numpy(dres).to(gpu))
    torch.cuda.synchronize()
    ref_res, ref_dhx, ref_g = _head_reference(params, hx0, dres)
    assert np.abs(res.cpu().numpy()[:, :3] - ref_res).max() <= 2e-5
    assert np.abs(dhx.cpu().numpy() - ref_dhx).max() <= 1e-4 * max(np.abs(ref_dhx).max(), 1e-12) + 1e-9
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
