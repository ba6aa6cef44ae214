"""Loss trajectories of the training step with BatchNorm in training mode, for a few learning rates (experiment behind the
thresholds of tests/test_train_gpu.py::test_train_steps_reduce_the_loss)."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from himo_amd.seflow import spec
from himo_amd.seflow.train import SeFlowTrainer
from test_train_gpu import _labelled_sample

gpu = torch.device("cuda", 0)
(pch, pc0, pc1, pose_h, pose0, pose1), lab0, lab1 = _labelled_sample(8000, 11)
l0, l1 = torch.from_numpy(lab0).to(gpu), torch.from_numpy(lab1).to(gpu)
for bn, fresh in (("frozen", False), ("batch", False), ("batch", True)):
    for lr in (1e-3, 3e-4, 1e-4):
        tr = SeFlowTrainer(spec.init_params(5, fresh_bn=fresh), device=gpu, max_points=8000, batchnorm=bn)
        tot = []
        for _ in range(12):
            _, total = tr.train_step(pch, pc0, pc1, pose_h, pose0, pose1, l0, l1, n_labels=int(lab0.max()) + 1, lr=lr)
            tot.append(round(float(total.item()), 3))
        print(bn, "fresh" if fresh else "random-bn", lr, tot, flush=True)
        del tr
