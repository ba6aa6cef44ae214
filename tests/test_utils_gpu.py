"""GPU versions of the reference's utils/__init__.py functions against the golden vectors."""
import numpy as np
import pytest
import torch

from conftest import RES, golden_frames

pytestmark = pytest.mark.gpu


def test_flow2compdis_dtypes_and_values(gpu, gold, oracle):
    from himo_amd import utils
    f = golden_frames(gold, "av2")[0]
    est64 = oracle.remove_ego_motion(f["pc0"], f["pose0"], f["pose1"], f[RES])
    dt0 = oracle.dt0_from_lidar_dt(f["lidar_dt"])
    out64 = utils.flow2compDis(est64, dt0, sensor_dt=0.1)
    assert isinstance(out64, np.ndarray) and out64.dtype == np.float64
    assert np.array_equal(out64, gold["av2/0/ref_comp_dis_f64"])               # f64 divide+multiply: bit exact
    out32 = utils.flow2compDis(est64.astype(np.float32), dt0, sensor_dt=0.1)
    assert out32.dtype == np.float32
    assert np.array_equal(out32, gold["av2/0/ref_comp_dis_f32chain"])
    assert utils.flow2compDis(est64.astype(np.float32), dt0.astype(np.float64), 0.1).dtype == np.float64
    # default sensor_dt = 10 like the reference signature
    assert np.array_equal(utils.flow2compDis(est64, dt0), oracle.flow2compDis(est64, dt0))
    # tensors in -> tensors out, on the device
    t = utils.flow2compDis(torch.from_numpy(est64).cuda(), torch.from_numpy(dt0).cuda(), 0.1)
    assert isinstance(t, torch.Tensor) and t.is_cuda and np.array_equal(t.cpu().numpy(), out64)
    with pytest.raises(ValueError):
        utils.flow2compDis(est64, dt0[:-1], 0.1)


def test_refine_pts(gpu, gold):
    from himo_amd import utils
    f = golden_frames(gold, "scania")[0]
    r64 = utils.refine_pts(f["pc0"], gold["scania/0/ref_comp_dis_f64"])
    assert r64.dtype == np.float64 and np.array_equal(r64, gold["scania/0/ref_refined_f64"])
    r32 = utils.refine_pts(f["pc0"], gold["scania/0/ref_comp_dis_f32chain"])
    assert r32.dtype == np.float32 and np.array_equal(r32, gold["scania/0/ref_refined_f32chain"])
    assert utils.refine_pts(f["pc0"][:0], np.zeros((0, 3), np.float32)).shape == (0, 3)


@pytest.mark.parametrize("data_name", ["av2", "scania"])
def test_ego_pts_mask(gpu, gold, data_name):
    from himo_amd import utils
    for i, f in enumerate(golden_frames(gold, data_name)):
        m = utils.ego_pts_mask(f["pc0"])
        assert m.dtype == bool and np.array_equal(m, gold[f"{data_name}/{i}/ref_ego_mask_default"])
        m2 = utils.ego_pts_mask(f["pc0"], min_bound=[-1.5, -1.5, -2.0], max_bound=[1.5, 1.5, 2.0])
        assert np.array_equal(m2, gold[f"{data_name}/{i}/ref_ego_mask_av2"])
        assert (~m).sum() > 0 or (~m2).sum() > 0
    # points exactly on the box faces are OUTSIDE (strict compares)
    edge = np.array([[-1.5, 0, 0], [1.5, 0, 0], [0, 0, 2.0], [0, 0, 0]], np.float32)
    assert utils.ego_pts_mask(edge, [-1.5, -1.5, -2.0], [1.5, 1.5, 2.0]).tolist() == [True, True, True, False]


def test_dt0(gpu, gold, oracle):
    from himo_amd import utils
    for n in (1, 5, 4096, 4097, 3000):
        dt = np.random.default_rng(n).uniform(0, 0.1, n).astype(np.float32)
        assert np.array_equal(utils.dt0_from_lidar_dt(dt), oracle.dt0_from_lidar_dt(dt))
    neg = np.array([-0.3, -0.1, -0.2], np.float32)                       # negative offsets: max is -0.1
    assert np.array_equal(utils.dt0_from_lidar_dt(neg), oracle.dt0_from_lidar_dt(neg))
    with pytest.raises(ValueError, match="empty sequence"):
        utils.dt0_from_lidar_dt(np.zeros(0, np.float32))
