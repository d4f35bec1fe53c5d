"""RCCL on the device under the driver's `pytest -m gpu` (the reference's only distribution mode is Lightning's
strategy "ddp", train.py:118-120, configs/config_px2px.yaml:60-63).

One rank is all a 1-GPU box has, but it is the REAL path: a `nccl` (= RCCL) process group, `GradReducer.begin` started from inside
the backward plans (`Plan.insert_hook`) as async all-reduces with `ReduceOp.AVG` on slices of the flat gradient, `finish()` ordering
Adam behind them on the launch stream, the initial broadcast.  With one rank the average is the identity, so the bucketed step must
reproduce the reducer-less step bit for bit (tolerance 1e-6).  Multi-rank arithmetic is covered on the CPU with gloo
(tests/test_parallel_gloo.py, tests/test_bench_launch.py)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture()
def nccl_world1():
    import torch.distributed as dist
    assert not dist.is_initialized()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def _nets(nb=6, ngf=64):
    from model import networks
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, ngf, f"resnet_{nb}blocks", "instance", False, "normal", 0.02).to(DEV)
    netD = networks.define_D(4, ngf, "basic", 3, "instance", "normal", 0.02).to(DEV)
    return netG, netD


def test_bucketed_fused_step_over_rccl_matches_the_plain_step(nccl_world1):
    from nirgan_hip.parallel import GradReducer
    from nirgan_hip.trainer import Pix2PixTrainer
    assert nccl_world1.get_backend() == "nccl"
    g = torch.Generator().manual_seed(21)
    rgb = (0.02 + 0.58 * torch.rand(4, 3, 128, 128, generator=g)).to(DEV)
    nir = (0.05 + 0.75 * torch.rand(4, 1, 128, 128, generator=g)).to(DEV)
    # reference: no reducer
    netG, netD = _nets()
    plain = Pix2PixTrainer(netG, netD, n_blocks=6)
    o_plain = plain.step(rgb, nir).as_dict()
    want = [t.clone() for t in (plain.flatD.grad, plain.flatG.grad, plain.flatD.flat, plain.flatG.flat)]
    del plain
    # the same step with the gradients going through RCCL in two buckets per network
    netG, netD = _nets()
    red = GradReducer()
    assert red.world == 1 and red._avg, "nccl backend: the division rides in the collective (ReduceOp.AVG)"
    begun = []
    orig_begin = red.begin
    red.begin = lambda part: (begun.append((part.data_ptr(), part.numel())), orig_begin(part))[1]
    red.exposed_events = []
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, reducer=red)
    o = tr.step(rgb, nir).as_dict()
    torch.cuda.synchronize()
    st = tr._state
    assert st.bucketed and st.headG is not None and st.headD is not None
    # the tail and the middle buckets were started from INSIDE the backward plans (hooks), the heads -- the first layers' gradients
    # only -- after them: 3 begins per network
    nD, nG = tr.flatD.total, tr.flatG.total
    headD, headG = sum(p.numel() for p in st.headD), sum(p.numel() for p in st.headG)
    assert headG == sum(tr.flatG.slices[k][1] for k in ("model.1.weight", "model.1.bias")) and headD == sum(tr.flatD.slices[k][1] for k in ("model.0.weight", "model.0.bias"))
    assert len(begun) == 6 and sum(n for _, n in begun) == nD + nG, (begun, nD, nG)
    tailG = tr.flatG.total - tr.flatG.slices["model.10.conv_block.1.weight"][0]
    tailD = tr.flatD.total - tr.flatD.slices["model.8.weight"][0]
    sizes = [n for _, n in begun]
    assert sizes[:3] == [tailD, nD - tailD - headD, headD] and sizes[3:] == [tailG, nG - tailG - headG, headG], (sizes, tailD, tailG)
    assert len(red.exposed_events) == 2 and all(e0.elapsed_time(e1) >= 0.0 for e0, e1 in red.exposed_events)
    assert not red._pending
    for k in o_plain:
        assert abs(o[k] - o_plain[k]) <= 1e-6 * max(abs(o_plain[k]), 1e-12), (k, o[k], o_plain[k])
    got = (tr.flatD.grad, tr.flatG.grad, tr.flatD.flat, tr.flatG.flat)
    for a, b, what in zip(got, want, ("grad D", "grad G", "params D", "params G")):
        err = (a - b).abs().max().item()
        assert err <= 1e-6 * max(b.abs().max().item(), 1e-20), f"{what} over RCCL: {err:.3e}"
    # a second step keeps working (collectives re-armed, hooks fire again)
    n0 = len(begun)
    tr.step(rgb, nir)
    torch.cuda.synchronize()
    assert len(begun) == 2 * n0


def test_rccl_broadcast_and_single_bucket_path(nccl_world1):
    """The un-bucketed path (micro-batches: one all-reduce per network after the join) and the initial-weight broadcast."""
    from nirgan_hip.parallel import GradReducer
    from nirgan_hip.trainer import Pix2PixTrainer
    g = torch.Generator().manual_seed(22)
    rgb = (0.02 + 0.58 * torch.rand(4, 3, 64, 64, generator=g)).to(DEV)
    nir = (0.05 + 0.75 * torch.rand(4, 1, 64, 64, generator=g)).to(DEV)
    netG, netD = _nets(6, 16)
    plain = Pix2PixTrainer(netG, netD, n_blocks=6, micro_batches=2)
    plain.step(rgb, nir)
    want = plain.flatG.flat.clone()
    netG, netD = _nets(6, 16)
    red = GradReducer()
    tr = Pix2PixTrainer(netG, netD, n_blocks=6, reducer=red, micro_batches=2)
    tr.step(rgb, nir)
    assert not tr._state.bucketed and tr._synced_ptrs is not None
    err = (tr.flatG.flat - want).abs().max().item()
    assert err <= 1e-6 * want.abs().max().item(), err
    t = torch.arange(8, dtype=torch.float32, device=DEV)
    red.all_reduce_mean(t)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
