"""Data parallelism over torch.distributed (gloo on CPU, world size 2; RCCL on the GPUs uses the same code).

Each rank runs the fused trainer (C ABI served by the numpy emulator -- host logic only) on its shard of a
global batch with the GradReducer; the averaged flat gradients must equal the single-process gradients on the
whole batch (InstanceNorm is per sample, losses are means: model/networks.py:30, model/pix2pix.py:195-257)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _setup_path():
    for p in (os.path.join(ROOT, "nir-gan_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def _build(z):
    from emu_backend import EmuBackend
    from model import networks
    from nirgan_hip import lib as L
    L.set_backend(EmuBackend())
    netG = networks.define_G(3, 1, 8, "resnet_6blocks", "instance", False, "normal", 0.02)
    netD = networks.define_D(4, 8, "basic", 3, "instance", "normal", 0.02)
    netG.load_state_dict({k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("G0/")})
    netD.load_state_dict({k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("D0/")})
    return netG, netD


def _batch():
    g = torch.Generator().manual_seed(77)
    return 0.02 + 0.58 * torch.rand(4, 3, 32, 32, generator=g), 0.05 + 0.75 * torch.rand(4, 1, 32, 32, generator=g)


def _worker(rank, world, port, out_dir, micro, perturb=False):
    _setup_path()
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nirgan_hip.parallel import GradReducer, shard_batch
    from nirgan_hip.trainer import Pix2PixTrainer
    z = np.load(os.path.join(ROOT, "tests", "golden", "f1_g6_d.npz"))
    netG, netD = _build(z)
    if perturb and rank != 0:          # ranks that did not seed alike: DDP semantics = everyone starts from rank 0's weights
        torch.manual_seed(100 + rank)
        with torch.no_grad():
            for p in list(netG.parameters()) + list(netD.parameters()):
                p.add_(0.05 * torch.randn_like(p))
    red = GradReducer()
    one_bucket = micro == 0            # the opt-out: one blocking all-reduce per network after its backward (OPT.dp_buckets = False)
    if one_bucket:
        from nirgan_hip.options import OPT
        OPT.dp_buckets, micro = False, 1
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, reducer=red, micro_batches=micro)
    rgb, nir = _batch()
    out = tr.step(shard_batch(rgb, rank, world), shard_batch(nir, rank, world)).as_dict()
    if one_bucket:
        assert not tr._state.bucketed and not any(n == "__hook__" for n, _ in tr.G.bwd.ops)
    elif micro == 1:                   # two buckets per network: the tail went out from inside the backward plans
        assert tr._state.bucketed and any(n == "__hook__" for n, _ in tr.G.bwd.ops) and any(n == "__hook__" for n, _ in tr.D2.bwd.ops)
        # ... and the middle from in front of the first layer's backward: what is left for after the plan is the first layer's gradient only
        assert sum(1 for n, _ in tr.G.bwd.ops if n == "__hook__") == 2 and sum(1 for n, _ in tr.D2.bwd.ops if n == "__hook__") == 2
        assert sum(p_.numel() for p_ in tr._state.headG) == sum(tr.flatG.slices[k][1] for k in ("model.1.weight", "model.1.bias"))
        assert sum(p_.numel() for p_ in tr._state.headD) == sum(tr.flatD.slices[k][1] for k in ("model.0.weight", "model.0.bias")) and not red._pending
    torch.save({"gD": tr.flatD.grad.clone(), "gG": tr.flatG.grad.clone(), "pD": tr.flatD.flat.clone(),
                "pG": tr.flatG.flat.clone(), "loss": out}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("micro,perturb", [(1, False), (2, False), (1, True), (0, False)])
def test_two_rank_gradients_equal_single_process(tmp_path, micro, perturb):
    """micro = 0: one part, OPT.dp_buckets = False (the single blocking all-reduce).  micro = 2: every rank additionally cuts its shard into two parts whose gradients are summed before the all-reduce.
    perturb: rank 1 starts from different weights; the trainer broadcasts rank 0's (what DDP does when it wraps the module)."""
    _setup_path()
    port = 29500 + (os.getpid() % 2000) + 7 * micro + 3 * perturb
    mp.spawn(_worker, args=(2, port, str(tmp_path), micro, perturb), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in (0, 1))
    # both ranks hold identical reduced gradients and identical updated parameters
    for k in ("gD", "gG", "pD", "pG"):
        assert torch.equal(r0[k], r1[k]), k
    from nirgan_hip import lib as L
    from nirgan_hip.trainer import Pix2PixTrainer
    z = np.load(os.path.join(ROOT, "tests", "golden", "f1_g6_d.npz"))
    netG, netD = _build(z)
    try:
        tr = Pix2PixTrainer(netG, netD, n_blocks=6)
        rgb, nir = _batch()
        out = tr.step(rgb, nir).as_dict()
    finally:
        L.set_backend(None)
    for k, ref in (("gD", tr.flatD.grad), ("gG", tr.flatG.grad)):
        err = (r0[k] - ref).norm().item() / ref.norm().item()
        assert err < 1e-4, (k, err)
    # mean of the per-rank losses = loss of the global batch
    assert abs(0.5 * (r0["loss"]["loss_D"] + r1["loss"]["loss_D"]) - out["loss_D"]) < 1e-5 * abs(out["loss_D"])


def test_shard_batch_and_single_rank_noop():
    _setup_path()
    from nirgan_hip.parallel import shard_batch
    t = torch.arange(24.).view(8, 3)
    assert torch.equal(torch.cat([shard_batch(t, r, 4) for r in range(4)]), t)
    with pytest.raises(AssertionError):
        shard_batch(t, 0, 3)
