"""CPU tests of the host logic above the kernels, with the C oracle standing in for the HIP
library (oracle/cpu_backend.py -- allowed in tests only): the full clip model steps end to end
(BASELINE config c1 plumbing), and the data-parallel path gives the same gradients as a single
process on the same global batch (world_size 2, gloo)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_clip_step_runs_on_cpu_oracle_backend():
    from multimodal_gar_amd import workload as W
    from oracle.cpu_backend import use_cpu_oracle
    dev = torch.device("cpu")
    batch = W.make_batch(0, 1, 2, 3, 256, 32, 48, dev)
    with use_cpu_oracle():
        step = W.TrainStep(3, 256, dev)
        out = step.module(batch)
        assert len(out) == 16 and out[0].shape == (2, 4, 4)
        l0 = float(step.run(batch)); l1 = float(step.run(batch))
    assert np.isfinite(l0) and np.isfinite(l1)
    # every trainable parameter took part in the step, except the cardinality head: it is outside the reference's L_total
    # objective (train_func.py:240-247) and therefore gets no gradient on any rank (the flat all-reduce pins that layout)
    missing = [n for n, p in step.module.named_parameters() if p.requires_grad and p.grad is None]
    assert missing and all("card_net" in n for n in missing), missing
    syn = W.TrainStep(3, 256, dev, loss="synthetic")
    with use_cpu_oracle():
        syn.run(batch)
    assert all(p.grad is not None for p in syn.module.parameters() if p.requires_grad)


def test_product_path_is_restored_after_the_context():
    from multimodal_gar_amd import _lib
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as pb
    from oracle.cpu_backend import use_cpu_oracle
    with use_cpu_oracle():
        idx = pb.ball_query(1.0, 4, torch.zeros(1, 8, 3), torch.zeros(1, 2, 3))
        assert idx.shape == (1, 2, 4)
    with pytest.raises(_lib.MgarError):
        pb.ball_query(1.0, 4, torch.zeros(1, 8, 3), torch.zeros(1, 2, 3))


def _no_dropout(module):
    """Dropout masks come from per-process RNG streams; switch them off to compare gradients."""
    for m in module.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
            m.dropout = 0.0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _ddp_worker(rank, world, port, out_path, manual=False, loss="synthetic"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from multimodal_gar_amd import workload as W
    from oracle.cpu_backend import use_cpu_oracle
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    with use_cpu_oracle():
        step = W.TrainStep(3, 256, dev, ddp=True, seed=5, manual_allreduce=manual, loss=loss)
        _no_dropout(step.module)
        full = W.make_batch(11, world, 1, 3, 256, 32, 48, dev)            # the GLOBAL batch: `world` clips
        mine = {k: (v[rank:rank + 1] if torch.is_tensor(v) else v) for k, v in full.items()}
        mine["n_clips"] = 1
        if manual:       # the graph-mode exchange: one all-reduce of the flattened gradients after backward
            step._forward_backward(mine)
            step._exchange_gradients()
        else:
            step._forward_backward(mine)
    if rank == 0:
        g = {n: p.grad.clone() for n, p in step.module.named_parameters() if p.grad is not None}
        torch.save(g, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("manual,loss", [(False, "synthetic"), (True, "synthetic"), (True, "reference"), (False, "reference")])
def test_ddp_gloo_world2_matches_single_process(tmp_path, manual, loss):
    """DistributedDataParallel (manual=False) and the flattened manual all-reduce of the HIP-graph mode (manual=True)
    against a single process on the same global batch.  loss="reference": the objective is a SUM over scenes, so two ranks
    must produce the one-rank gradient of the global batch (sum of the ranks' gradients, not their average: ADVICE r2)."""
    import torch.multiprocessing as mp
    from multimodal_gar_amd import workload as W
    from oracle.cpu_backend import use_cpu_oracle
    out_path = str(tmp_path / "ddp_grads.pt")
    mp.spawn(_ddp_worker, args=(2, _free_port(), out_path, manual, loss), nprocs=2, join=True)
    ddp_grads = torch.load(out_path)
    dev = torch.device("cpu")
    # same thread count as the workers: the tiny I3D feature maps make BatchNorm statistics
    # sensitive to the summation order of the convolution kernels
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    with use_cpu_oracle():
        step = W.TrainStep(3, 256, dev, ddp=False, seed=5, loss=loss)
        _no_dropout(step.module)
        full = W.make_batch(11, 2, 1, 3, 256, 32, 48, dev)
        # DDP averages the per-rank losses' gradients: reference = mean over the two clips, each run
        # as its own forward (BatchNorm statistics are per replica, as under the reference's DataParallel)
        step.opt.zero_grad(set_to_none=True)
        for r in range(2):
            mine = {k: (v[r:r + 1] if torch.is_tensor(v) else v) for k, v in full.items()}
            mine["n_clips"] = 1
            if loss == "synthetic":
                (W.synthetic_loss(step.model(mine)) / 2).backward()
            else:
                W.reference_loss(step.model(mine), mine).backward()
    torch.set_num_threads(threads)
    checked = 0
    for n, p in step.module.named_parameters():
        if p.grad is None:
            continue
        a, b = ddp_grads[n], p.grad
        scale = b.abs().max().item() + 1e-12
        assert (a - b).abs().max().item() <= 1e-5 + 1e-4 * scale, n
        checked += 1
    assert checked > 100


def test_trunk_geometry_ahead_of_time_equals_inline_geometry():
    """PointNet2MSG.geometry() (FPS centres, ball queries of the folded scales, 3-NN weights: what workload.ClipModel issues on
    a side stream ahead of the feature path) must hand the SA / FP modules exactly what they compute themselves."""
    from multimodal_gar_amd import workload as W
    from oracle.cpu_backend import use_cpu_oracle
    torch.manual_seed(0)
    model = W.ClipModel(4, 1024).train()
    batch = W.make_batch(1, 1, 2, 4, 1024, 64, 96, torch.device("cpu"))
    with use_cpu_oracle():
        trunk = model.net.LiDAR_backbone.model.backbone_3d
        geo = trunk.geometry(batch["points"])
        assert len(geo.centres) == len(trunk.SA_modules) and set(geo.nn) == {-1, -2, -3, -4}
        f, p, _ = batch["points"].shape
        bidx = torch.arange(f, dtype=batch["points"].dtype).view(f, 1, 1).expand(f, p, 1)
        plain = {"batch_size": f, "points": torch.cat([bidx, batch["points"]], -1).view(f * p, 5)}
        ahead = dict(plain, trunk_geometry=geo)
        a = trunk(plain)
        b = trunk(ahead)
    assert torch.equal(a["point_features_cm"], b["point_features_cm"]) and torch.equal(a["point_coords"], b["point_coords"])
