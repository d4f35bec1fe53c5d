#!/usr/bin/env python3
"""Throughput of the NIR-GAN Pix2Pix train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one batch of the reference's loop (model/pix2pix.py:165-257): generator forward,
PatchGAN forward on fake+real, both backward passes, both Adam steps -- on synthetic random
tiles already resident in HBM.  Workload = BASELINE.json configs[1]: 6-block
ResnetGenerator + 3-layer PatchGAN, bs=16 256x256 per GPU, GAN + L1 loss, fp32 (exact-fp32
MFMA).  N > 1: one process per GPU, tile batches sharded data-parallel, flat gradients averaged
with RCCL in two buckets per network; per-GPU work is fixed ("weak").  Started without a
launcher (`python bench.py --gpus 8`) the parent spawns the N ranks itself (a child
`python -m torch.distributed.run`; the parent never touches the GPU and exits with the
children's status); started under torchrun it is one rank.  WORLD_SIZE must equal --gpus.
`--verify-dp` checks, before timing, that the N-rank averaged gradients equal the single-process
gradients on the concatenated batch.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel by share of step time
(round 2: wino6_pair16p_kernel -- the data-gradient plane GEMMs of a Winograd F(6x6,3x3)
residual-block layer as persistent workgroups + its 64 transform-domain weight-gradient planes in one
grid, 12 launches per step; `roofline_other` lists the other MFMA kernels): EXECUTED FLOPs
of its launches (2*M*N*K from the descriptors) / their duration, bracketed by HIP events on
the launch stream inside the timed steps.  `cpu_baseline` times the CPU oracle (a port, on a
bounded sample) on rank 0 at N = 1.

Other workloads (not the headline): --blocks 9 --lambda-rs 1 --bs 32 (configs[2]); --inject
--size 512 --padding 10 --bs 8 (configs[3]); --mixed --precision bf16 --blocks 9 --lambda-rs 1
(configs[4]: resolution buckets, bf16 MFMA); --precision bf16x3 (split-fp32 on the bf16 pipe).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nir-gan_amd"))

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 (v_mfma_f32_32x32x16_bf16); never the 2:1-sparsity headline
# peak the ALGORITHMIC flops are priced against, per operand precision (bf16x3 issues three bf16 products per product)
PEAKS = {"fp32": PEAK_FP32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3.0}
# the three-term split tiles of the exact-fp32 mode (csrc/igemm_x3.h, descriptor precision 3): six bf16 products per fp32 product on the
# bf16 pipe -- their fp32-equivalent FLOPs are priced against a sixth of the dense bf16 peak
PEAK_X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def _is_x3(label):
    """launches of the three-term split tiles (csrc/igemm_x3.h, igemm_x3r.h)"""
    return "_x3_kernel" in label or "_x3r_kernel" in label


def kernel_peak(label: str, precision: str) -> float:
    return PEAK_X3_TFLOPS if _is_x3(label) else PEAKS[precision]


def synth(B, H, W, seed, device):
    g = torch.Generator().manual_seed(seed)
    rgb = 0.02 + 0.58 * torch.rand(B, 3, H, W, generator=g)
    nir = 0.05 + 0.75 * torch.rand(B, 1, H, W, generator=g)
    return rgb.to(device), nir.to(device)



def w6_planes(r):
    """planes of a wino6 descriptor's variant code: 0 / 3 = F(4x4,3x3): 36, 4 = F(4x4,4x4): 49, 6 = F(6x6,3x3): 64"""
    return 64 if r == 6 else (max(r, 3) + 3) ** 2


def w6_tiles(d):
    mo = 6 if d.r == 6 else 4
    return d.B * (-(-d.H // mo)) * (-(-d.W // mo))


def op_direct_flops(name, args):
    """DIRECT-convolution FLOPs of a Winograd launch (SURVEY 8d's count for the layer: 2 * pixels * C * K * taps), None for launches whose
    executed count is the direct count already.  `algorithmic_over_peak` of a roofline entry is built from it: what the launch is worth in
    the reference's arithmetic, next to the executed `frac`."""
    if name == "nirgan_conv_igemm_group":
        # paired sub-pixel phases (nirgan_conv_desc.out_span = 2, a 3 x 3 stride-2 kernel): 9 of the 12 tap blocks of the two problems'
        # weight matrices are non-zero -- the launch EXECUTES 4/3 of the layer's multiplies
        ds = [args[0][j].contents for j in range(args[1])]
        if any(d.out_span == 2 for d in ds):
            return sum(2.0 * d.B * d.OH * d.OW * d.N * d.ntaps * d.run * (0.75 if d.out_span == 2 else 1.0) for d in ds)
        return None
    if name not in ("nirgan_wino6_gemm", "nirgan_wino6_gemm_wgrad_pair"):
        return None
    d = args[0]._obj
    taps = 16 if d.r == 4 else 9
    fl = 2.0 * d.B * d.H * d.W * d.C * d.K * taps                      # the plane GEMMs' layer over the extent the descriptor covers
    if name == "nirgan_wino6_gemm_wgrad_pair":                         # + the layer's weight gradient over the forward's output extent
        shrink = 3 if d.r == 4 else 2
        fl += 2.0 * d.B * (d.H - shrink) * (d.W - shrink) * d.C * d.K * taps
    return fl


def op_mfma_work(name, args):
    """(kernel label, EXECUTED matrix-pipe FLOPs, algorithmic HBM bytes or None) of one plan op, from its descriptor(s); None for
    ops that do not run on the matrix pipe.  Labels of the Winograd launches come from the library itself
    (nirgan_wino6_*_kernel_name mirror the launchers' dispatch), the others from the descriptor's tile width."""
    from nirgan_hip import lib as L
    be = L.backend()
    def conv_bytes(d):           # input buffer read once, packed weights once, output written once (fp32 or the producers' bf16 twins)
        return d.B * d.in_hp * d.in_wp * d.in_cs * (2.0 if d.in_bf16 else 4.0) + d.N * d.ntaps * d.run * (2.0 if d.w_bf16 else 4.0) + d.B * d.OH * d.OW * d.N * 4.0

    def wgrad_bytes(w):          # both operand images read once, slabs written once
        e = 2.0 if w.pq_bf16 else 4.0
        npl = max(w.nplanes, 1)
        return npl * (w.B * w.p_hp * w.p_wp * w.p_cs * e + w.B * w.q_hp * w.q_wp * w.q_cs * e) + npl * w.nsplit * w.N * w.ntaps * w.run * 4.0
    if name == "nirgan_conv_igemm":
        d = args[0]._obj
        k = be.nirgan_conv_kernel_name(args[0]) if hasattr(be, "nirgan_conv_kernel_name") else None
        return (k.decode() if k else f"conv_igemm_kernel<{128 if d.N > 64 else 64}>"), 2.0 * d.B * d.OH * d.OW * d.N * d.ntaps * d.run, conv_bytes(d)
    if name == "nirgan_conv_igemm_group":
        ds = [args[0][j].contents for j in range(args[1])]
        by = conv_bytes(ds[0]) + sum(d.N * d.ntaps * d.run * (2.0 if d.w_bf16 else 4.0) + d.B * d.OH * d.OW * d.N * 4.0 for d in ds[1:])      # the phases share the input
        label = f"conv_group_kernel<{128 if ds[0].N > 64 else 64}>"
        if all(d.precision == 3 for d in ds) and hasattr(be, "nirgan_conv_kernel_name"):      # the three-term split tile: one persistent launch of conv_x3_kernel over the phases
            k = be.nirgan_conv_kernel_name(C.byref(ds[0]))
            if k and k.decode().startswith("conv_x3"):
                # (the register-fed four-wave tile takes a group only if it takes every phase: igemm_conv.hip::ng_launch_conv_x3)
                ks = [be.nirgan_conv_kernel_name(C.byref(d)) for d in ds]
                r4 = all(kk and kk.decode().startswith("conv_x3r") for kk in ks)
                label = f"conv_x3r_kernel<128>" if r4 else f"conv_x3_kernel<{128 if ds[0].N % 128 == 0 else 64}>"
        return label, sum(2.0 * d.B * d.OH * d.OW * d.N * d.ntaps * d.run for d in ds), by
    if name == "nirgan_wgrad_igemm":
        w = args[0]._obj
        k = be.nirgan_wgrad_kernel_name(args[0]) if hasattr(be, "nirgan_wgrad_kernel_name") else None
        label = k.decode() if k and (k.decode().endswith("256_kernel") or k.decode().startswith("wgrad_x3")) else f"wgrad_igemm_kernel<{128 if w.N > 64 else 64}>" + ("(bf16 twins)" if w.pq_bf16 else "")
        return label, 2.0 * max(w.nplanes, 1) * w.B * w.OH * w.OW * w.N * w.ntaps * w.run, wgrad_bytes(w)
    if name == "nirgan_conv_wgrad_pair":
        c, w = args[0]._obj, args[1]._obj
        # (the weight gradient's p operand is the data gradient's input: counted once)
        by = conv_bytes(c) + wgrad_bytes(w) - max(w.nplanes, 1) * w.B * w.p_hp * w.p_wp * w.p_cs * (2.0 if w.pq_bf16 else 4.0)
        k = be.nirgan_conv_wgrad_pair_kernel_name(args[0], args[1]) if hasattr(be, "nirgan_conv_wgrad_pair_kernel_name") else None
        return (k.decode() if k and k.decode().startswith("conv_wgrad") else "conv_wgrad_pair_kernel"), 2.0 * c.B * c.OH * c.OW * c.N * c.ntaps * c.run + 2.0 * w.B * w.OH * w.OW * w.N * w.ntaps * w.run, by
    if name == "nirgan_wino6_gemm":
        d = args[0]._obj
        T = w6_tiles(d)
        k = be.nirgan_wino6_gemm_kernel_name(args[0]).decode() if hasattr(be, "nirgan_wino6_gemm_kernel_name") else "wino6_gemm"
        k = k.replace(" (planes)", "")         # (the plane batches of the split tile are launches of conv_x3_kernel: one row of a kernel trace)
        # EXECUTED flops: the plane GEMMs [T x C] x [C x K] (64/324 of the direct layer's multiplies for F(6x6,3x3)); V read once, U once, M written once
        return k, 2.0 * w6_planes(d.r) * T * d.C * d.K, 4.0 * w6_planes(d.r) * (T * d.C + d.K * d.C + T * d.K)
    if name == "nirgan_wino6_gemm_wgrad_pair":
        d, w = args[0]._obj, args[1]._obj
        T = w6_tiles(d)
        k = be.nirgan_wino6_pair_kernel_name(args[0], args[1]).decode() if hasattr(be, "nirgan_wino6_pair_kernel_name") else "wino6_pair"
        fl = 2.0 * w6_planes(d.r) * T * d.C * d.K + 2.0 * max(w.nplanes, 1) * w.B * w.OH * w.OW * w.N * w.ntaps * w.run
        # V of dY read once, U once, M written once; Yt and the forward's V read once, slabs written once
        by = 4.0 * w6_planes(d.r) * (T * d.C + d.K * d.C + T * d.K) + 4.0 * w.nplanes * (w.OW * (w.N + w.run) + w.nsplit * w.N * w.run)
        return (k or "wino6_gemm+wgrad"), fl, by
    if name in ("nirgan_endconv_dgrad", "nirgan_endconv_wgrad"):
        d = args[0]._obj
        px = d.B * d.x_hp * d.x_wp if name.endswith("dgrad") else d.B * d.OH * d.OW
        return name.replace("nirgan_", "") + "_kernel", 2.0 * px * 49 * 64, None
    return None


PEAK_HBM_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s (about 6.3 achievable)


def mfma_probes(trainer, want=("conv_igemm_kernel<128>", "conv_igemm256_kernel", "conv_group_kernel<128>", "wgrad_igemm_kernel<128>", "wgrad_igemm256_kernel", "conv_wgrad_pair", "wino6_",
                                "conv_x3_kernel", "conv_x3r_kernel", "wgrad_x3_kernel")):
    """EXECUTED FLOPs of one step's launches of the big MFMA kernels, and the op indices to bracket with HIP events."""
    plans = [trainer.G.fwd, trainer.G.bwd, trainer.D2.fwd, trainer.D2.bwd, trainer.D1.fwd, trainer.D1.bwd_pred]
    kinds, algo_bytes, direct = {}, {}, {}
    for pl in plans:
        pl.probe_idx, pl.probe_events = {}, []
        for i, (name, args) in enumerate(pl.ops):
            if not isinstance(name, str):
                continue
            w = op_mfma_work(name, args)
            if w is None or not w[0].startswith(tuple(want)):
                continue
            k, fl, by = w
            kinds.setdefault(k, [0.0, 0])
            kinds[k][0] += fl
            kinds[k][1] += 1
            dfl = op_direct_flops(name, args)
            direct[k] = direct.get(k, 0.0) + (fl if dfl is None else dfl)
            if by is not None:
                algo_bytes[k] = algo_bytes.get(k, 0.0) + by
            pl.probe_idx[i] = k
    mfma_probes.algo_bytes = algo_bytes
    mfma_probes.direct = direct
    return kinds, plans


def plan_executed_flops(plan) -> float:
    """Executed matrix-pipe FLOPs of every launch of a plan (all tile widths; the Winograd layers at their executed count; the three-term
    split tiles at their fp32-EQUIVALENT count, a sixth of the bf16 products they issue)."""
    return sum(w[1] for w in (op_mfma_work(n, a) for n, a in plan.ops if isinstance(n, str)) if w is not None)


def plan_pipe_ms_at_peak(plan, precision) -> float:
    """Time the matrix pipe needs for a plan's launches at each kernel's own peak (fp32 MFMA, or a sixth of the bf16 peak for the split
    tiles), in ms: divided by the elapsed time it is the share of the elapsed time the pipe would be busy at peak -- the utilisation of a
    step that mixes the two pipes."""
    return sum(w[1] / (kernel_peak(w[0], precision) * 1e12) for w in (op_mfma_work(n, a) for n, a in plan.ops if isinstance(n, str)) if w is not None) * 1e3


def predict_scaling(tr, ms1: float, tiles_per_rank: int):
    """What the first N > 1 measurement should be compared with (no multi-GPU node is available to the builder): weak scaling of the
    data-parallel step from quantities measured on ONE GPU plus an explicit model of the two exposed all-reduces.  Every input is in the
    block; nothing here is a measurement of N > 1."""
    def buckets(eng, flat):
        """bytes of the tail / middle / head bucket (trainer._hook_tail: the head is the first layer's gradient only)"""
        def span(first, last):
            lo = flat.slices[first][0]
            o, k, _ = flat.slices[last]
            return lo, min(flat.total, o + -(-k // 4) * 4)
        lo, hi = span(*eng.bwd_tail[1:])
        head = 0
        if getattr(eng, "bwd_mid", None) is not None:
            hlo, hhi = span(*eng.bwd_mid[1:])
            head = hhi - hlo
        else:
            head = flat.total - (hi - lo)
        return 4 * (hi - lo), 4 * (flat.total - (hi - lo) - head), 4 * head
    tailG, midG, headG = buckets(tr.G, tr.flatG)
    tailD, midD, headD = buckets(tr.D2, tr.flatD)
    # what covers the middle buckets: the first layer's backward (its weight gradient and slab sum), measured per op on one GPU
    # (profiles/r05_per_op_times.txt: generator 7x7 weight gradient 0.21 ms + reduce; PatchGAN first layer's weight gradient 0.1 ms)
    cover_mid_G_ms, cover_mid_D_ms = 0.25, 0.12
    # one rank over RCCL against the plain run, same box (scripts/refresh_profiles.sh: *_bench_rccl_one_rank / *_bench_final): the fixed
    # cost of going through torch.distributed (stream waits, two async launches per network)
    fixed_ms, src = 0.18, "profiles/r04_bench_rccl_one_rank.json.log vs r04_bench_final.json.log (19.29 vs 19.11 ms)"
    try:
        tags = sorted({f.split("_")[0] for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench_rccl_one_rank.json.log")}, reverse=True)
        for tag in tags:
            one = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", f"{tag}_bench_rccl_one_rank.json.log")) if l.startswith("{")]
            fin = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", f"{tag}_bench_final.json.log")) if l.startswith("{")]
            if one and fin:
                fixed_ms = max(0.0, one[-1]["ms_per_step"] - fin[-1]["ms_per_step"])
                src = f"profiles/{tag}_bench_rccl_one_rank.json.log vs {tag}_bench_final.json.log ({one[-1]['ms_per_step']} vs {fin[-1]['ms_per_step']} ms)"
                break
    except Exception:
        pass
    link_gbps, links = 153.0, 7            # MI355X_MICROARCH.md / task statement: 7 xGMI links x ~153 GB/s per GPU, point to point
    eff, lat_us = 0.6, 25.0                # ASSUMED: fraction of the link rate a ring step sustains on 1-3 MB messages; latency per ring step
    res = []
    for n in (2, 4, 8):
        def allreduce_ms(nbytes):
            # ring all-reduce: 2 (n - 1) steps of nbytes / n each; over the fully connected mesh min(n - 1, links) rings run on disjoint links
            rings = min(n - 1, links)
            return 2 * (n - 1) * (lat_us * 1e-3 + (nbytes / n / rings) / (link_gbps * eff * 1e9) * 1e3)
        # the head buckets (the first layers' gradients) start after their backward plans: nothing covers them; the middle buckets
        # start in front of the first layer's backward: exposed by what that backward does not cover
        exposed = (allreduce_ms(headG) + allreduce_ms(headD) + max(0.0, allreduce_ms(midG) - cover_mid_G_ms) + max(0.0, allreduce_ms(midD) - cover_mid_D_ms))
        hidden = allreduce_ms(tailG) + allreduce_ms(tailD)           # the tails run under the rest of the backward plans
        ms = ms1 + fixed_ms + exposed
        res.append({"n_gpus": n, "ms_per_step": round(ms, 3), "tiles_per_s": round(n * tiles_per_rank / ms * 1e3, 1),
                    "weak_scaling_efficiency": round(ms1 / ms, 4), "exposed_allreduce_ms": round(exposed, 4), "tail_allreduce_ms_under_backward": round(hidden, 4)})
    return {"what": "PREDICTED from one-GPU measurements + a stated all-reduce model; not a measurement of N > 1 (with more than one rank the "
                    "line carries `measured_vs_predicted`: this run's own numbers beside the entry for its N)",
            "by_n_gpus": res,
            "inputs": {"ms_per_step_1gpu": round(ms1, 3), "rccl_fixed_ms_per_step": round(fixed_ms, 3), "rccl_fixed_source": src,
                       "bucket_bytes": {"G_tail": tailG, "G_mid": midG, "G_head": headG, "D_tail": tailD, "D_mid": midD, "D_head": headD},
                       "mid_bucket_cover_ms": {"G": cover_mid_G_ms, "D": cover_mid_D_ms},
                       "xgmi": {"links_per_gpu": links, "gbps_per_link": link_gbps, "assumed_ring_efficiency": eff, "assumed_latency_us_per_ring_step": lat_us}},
            "model": "ms(N) = ms(1) + fixed RCCL cost + sum over G, D of [all-reduce(head bucket) + max(0, all-reduce(middle bucket) - the first "
                     "layer's backward)]; all-reduce(b) = 2 (N - 1) x (latency + b / N / min(N - 1, 7) rings / (153 GB/s x efficiency)); the tail "
                     "buckets (started inside the backward plans) are assumed hidden: their modelled time is listed next to the backward they run under"}


def host_cores() -> int:
    """CPU share of this process (affinity mask / cgroup quota), not the machine's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(n_blocks, size, bs=16, steps=3):
    """The CPU oracle (port of the reference's step) on the metric's own configuration: bs = 16 (BASELINE.md section 3:
    1 warm-up + >= 3 timed steps), all host cores of this process's share."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nirgan_oracle as O
    from model import networks
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    netG = networks.define_G(3, 1, 64, f"resnet_{n_blocks}blocks", "instance", False, "normal", 0.02)
    netD = networks.define_D(4, 64, "basic", 3, "instance", "normal", 0.02)
    tr = O.OracleTrainer(netG.state_dict(), netD.state_dict(), n_blocks)
    rgb, nir = synth(bs, size, size, 1234, "cpu")
    tr.step(rgb, nir)
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        tr.step(rgb, nir)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": round(bs / dt, 3), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port", "cpu": cpu_model(),
            "s_per_step": round(dt, 3),
            "sample": f"oracle (plain PyTorch CPU fp32) train step, {n_blocks}-block, bs={bs}, {size}x{size}, 1 warm-up + {steps} timed steps (median)"}


def kernel_source_sha16() -> str:
    """Hash of the HIP sources + headers the library is built from: profiles recorded for another version are stale."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "nir-gan_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def free_port() -> int:
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """No launcher around us: start the N ranks as a CHILD torchrun (never exec / never touch the GPU in this process)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def verify_dp(a, dev, rank, world, reducer, netG, netD, make_trainer, rgb, nir, embeds):
    """SURVEY 8e "Verification": the gradients the N ranks hold after the all-reduce equal the single-process gradients on
    the concatenated global batch (InstanceNorm is per sample, losses are means; lr 0 keeps the parameters in place)."""
    import torch.distributed as dist
    tr = make_trainer(reducer, 0.0)
    tr.step(rgb, nir, embeds)
    gD, gG = tr.flatD.grad.clone(), tr.flatG.grad.clone()
    del tr
    def gather(t):
        if t is None:
            return None
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous())
        return torch.cat(parts, 0)
    one = make_trainer(None, 0.0)
    one.step(gather(rgb), gather(nir), gather(embeds))
    res = {}
    for k, dp, ref in (("D", gD, one.flatD.grad), ("G", gG, one.flatG.grad)):
        res["rel_l2_" + k] = float(((dp - ref).norm() / ref.norm().clamp_min(1e-30)).item())
    del one
    for f in (netG._flat(), netD._flat()):           # the two lr-0 steps must not count as training
        f.step_count = 0
        if f.m is not None:
            f.m.zero_()
            f.v.zero_()
    torch.cuda.empty_cache() if dev.type == "cuda" else None
    worst = torch.tensor([max(res.values())], dtype=torch.float64, device=dev)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    res["max_over_ranks"] = float(worst.item())
    res["tol"] = a.verify_tol
    res["global_batch"] = int(rgb.shape[0] * world)
    res["ok"] = res["max_over_ranks"] <= a.verify_tol
    return res


class Tick:
    """A HIP event on the launch stream, or a host timestamp on the CPU test seam (same elapsed_time interface, ms)."""

    def __init__(self, dev):
        self.ev = torch.cuda.Event(enable_timing=True) if dev.type == "cuda" else None
        self.t = 0.0

    def record(self):
        if self.ev is not None:
            self.ev.record()
        else:
            self.t = time.perf_counter()

    def elapsed_time(self, other) -> float:
        return self.ev.elapsed_time(other.ev) if self.ev is not None else (other.t - self.t) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bs", type=int, default=16, help="tiles per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--padding", type=int, default=0, help="Data.padding_amount (YAML default 10); 0 = BASELINE.md headline case")
    ap.add_argument("--lambda-rs", type=float, default=0.0)
    ap.add_argument("--inject", action="store_true", help="SatCLIP-inject generator (configs[3]): 9 blocks, fc 256->128x128, multiply")
    ap.add_argument("--precision", choices=["fp32", "bf16", "bf16x3"], default="fp32",
                    help="operand precision of the MFMA contractions; fp32 is the BASELINE.json configs[1] headline, "
                         "bf16 is configs[4]'s 'bf16 MFMA' (fp32 accumulate, fp32 master weights, everything else fp32)")
    ap.add_argument("--mixed", action="store_true",
                    help="configs[4]: every rank-step draws one resolution bucket (4*bs @128, bs @256, bs/4 @512: equal tile "
                         "area); value is reported in 256x256-equivalent tiles/s")
    ap.add_argument("--micro", type=int, default=1,
                    help="cut each batch into this many parts that run concurrently on separate HIP streams (the HBM-bound "
                         "kernels of one part under the matrix-pipe kernels of the other); per-kernel event times then "
                         "include the sharing, so the roofline entry is not a clean single-kernel figure")
    ap.add_argument("--api", choices=["fused", "lightning"], default="fused",
                    help="fused: Px2Px_PL.train_batch / Pix2PixTrainer.step (one call per batch).  lightning: the drop-in surface exactly as "
                         "Lightning 1.9's two-optimizer loop drives the reference (train.py:136): toggle_optimizer, training_step(batch, i, 0), "
                         "zero_grad, backward, HipAdam.step, then the same for optimizer 1 -- through the autograd bridges")
    ap.add_argument("--sustain", type=float, default=10.0,
                    help="seconds of back-to-back steps AFTER the timed ones (the reference trains 200 000 steps, train.py:124; a 0.4 s burst "
                         "says nothing about clocks under sustained load); 0 = off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--verify-dp", action="store_true",
                    help="before timing: N-rank averaged gradients == single-process gradients on the concatenated batch")
    ap.add_argument("--no-verify-dp", action="store_true",
                    help="with more than one rank the check of --verify-dp runs by default (one extra step before the timing; a failure is "
                         "REPORTED in the JSON line, `dp_verify.ok`, and only fatal with an explicit --verify-dp): this switches it off")
    ap.add_argument("--verify-tol", type=float, default=1e-4, help="relative L2 bound of --verify-dp (measured values are in the JSON)")
    ap.add_argument("--ngf", type=int, default=64, help="network width (64 = the reference's; smaller only for the CPU launch test)")
    ap.add_argument("--emulate-cpu", action="store_true",
                    help="TEST SEAM (tests/test_bench_launch.py): gloo + the numpy emulator of the C ABI instead of RCCL + the HIP "
                         "library, to exercise launching / sharding / verification without a GPU; the numbers mean nothing")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: the launcher and the request disagree")
    if a.emulate_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu_backend import EmuBackend
        from nirgan_hip import lib as _L
        _L.set_backend(EmuBackend())
        torch.set_num_threads(2)
        dev = torch.device("cpu")
        a.no_probe = a.no_cpu_baseline = True
        a.sustain = 0.0
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    reducer = None
    if world > 1 or os.environ.get("NIRGAN_FORCE_DIST") == "1":     # the env var exercises RCCL on a single GPU (tests)
        import torch.distributed as dist
        if a.emulate_cpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        from nirgan_hip.parallel import GradReducer
        reducer = GradReducer()

    from model import networks
    from nirgan_hip.trainer import Pix2PixTrainer
    torch.manual_seed(0)
    inject, embeds = None, None
    if a.inject:
        import types
        from model.generator_inject import define_G_inject
        ns = types.SimpleNamespace
        a.blocks = 9
        cfg = ns(base_configs=ns(input_nc=3, output_nc=1, ngf=a.ngf, netG="resnet_9blocks", norm="instance", no_dropout=True,
                                 init_type="normal", init_gain=0.02),
                 satclip=ns(satclip_inject_style="multiply", post_correction=False, post_correction_init=1.0,
                            scaling_param=True, scaling_param_init=0.01))
        netG = define_G_inject(cfg).to(dev)
        inject = {"style": "multiply", "use_scale": True}
        embeds = torch.randn(a.bs, 256, generator=torch.Generator().manual_seed(99 + rank)).to(dev)
    else:
        netG = networks.define_G(3, 1, a.ngf, f"resnet_{a.blocks}blocks", "instance", False, "normal", 0.02).to(dev)
    netD = networks.define_D(4, a.ngf, "basic", 3, "instance", "normal", 0.02).to(dev)
    rs_w = {"lambda_ndvi": 0.3333, "lambda_ndwi": 0.3333, "lambda_evi": 0.3333}

    def make_trainer(red, lr=2e-4):
        return Pix2PixTrainer(netG, netD, n_blocks=a.blocks, lr=lr, padding=a.padding, lambda_rs=a.lambda_rs, rs_weights=rs_w,
                              inject=inject, reducer=red, precision=a.precision, micro_batches=a.micro)
    rgb, nir = synth(a.bs, a.size, a.size, 1234 + rank, dev)
    dp_check = None
    # the first real multi-GPU run checks itself: with more than one rank the gradient check runs unless switched off (mixed-resolution
    # ranks hold different shapes: no concatenated batch to compare with)
    auto_verify = reducer is not None and world > 1 and not a.no_verify_dp and not a.mixed
    if a.verify_dp or auto_verify:
        if reducer is None:
            sys.exit("bench.py: --verify-dp needs more than one rank (or NIRGAN_FORCE_DIST=1)")
        try:
            dp_check = verify_dp(a, dev, rank, world, reducer, netG, netD, make_trainer, rgb, nir, embeds)
        except Exception as exc:            # (the default check must never cost the measurement)
            if a.verify_dp:
                raise
            dp_check = {"ok": False, "error": repr(exc)}
        dp_check["requested"] = "--verify-dp" if a.verify_dp else "default with more than one rank"
        if not dp_check["ok"] and a.verify_dp:
            print(json.dumps({"dp_verify": dp_check}), flush=True)
            sys.exit(f"bench.py: data-parallel gradients differ from the single-process gradients: {dp_check}")
    tr = make_trainer(reducer)          # with a reducer: broadcasts rank 0's weights first (what DDP does at wrap time)
    _step = tr.step
    tr.step = lambda r, n: _step(r, n, embeds)
    pl_model = None
    if a.api == "lightning":
        assert not a.mixed and a.micro == 1 and reducer is None and not a.inject and a.precision == "fp32", "--api lightning: single GPU, plain fp32 step"
        from model.pix2pix import Px2Px_PL
        from utils.config import to_attr
        cfg = to_attr({
            "base_configs": {"isTrain": True, "input_nc": 3, "output_nc": 1, "ngf": a.ngf, "ndf": a.ngf, "netD": "basic",
                             "netG": f"resnet_{a.blocks}blocks", "norm": "instance", "no_dropout": True, "init_type": "normal", "init_gain": 0.02,
                             "n_layers_D": 3, "gan_mode": "lsgan", "lr": 0.0002, "beta1": 0.5, "direction": "AtoB", "lambda_GAN": 1.0,
                             "lambda_L1": 100.0, "lambda_ssim": 0.0, "lambda_hist": 0.0, "lambda_rs_losses": a.lambda_rs,
                             "rs_losses_criterium": "l1", "internal_rs_loss_weights": dict(rs_w)},
            "satclip": {"use_satclip": False, "satclip_style": "inject", "satclip_inject_style": "multiply", "scaling_param": True,
                        "scaling_param_init": 0.01, "post_correction": False, "post_correction_init": 1.0},
            "Schedulers": {"metric": "val/L1", "patience_g": 25, "patience_d": 25},
            "custom_configs": {"Logging": {"num_val_images": 0, "log_input_stats": False}},
            "Data": {"padding": a.padding > 0, "padding_amount": a.padding}})
        torch.manual_seed(0)
        pl_model = Px2Px_PL(cfg).to(dev).train()
        (opt_d, opt_g), _ = pl_model.configure_optimizers()
        pG, pD = list(pl_model.netG.parameters()), list(pl_model.netD.parameters())
        batch_idx = [0]

        def toggle(on, off):             # LightningModule.toggle_optimizer: only the current optimizer's parameters take gradients
            for q in off:
                q.requires_grad_(False)
            for q in on:
                q.requires_grad_(True)

        def pl_step(r, n):
            batch = pl_step.batch if (r is rgb and n is nir) else {"rgb": r, "nir": n}
            i = batch_idx[0]
            batch_idx[0] += 1
            toggle(pD, pG)
            loss_d = pl_model.training_step(batch, i, 0)
            opt_d.zero_grad()
            loss_d.backward()
            opt_d.step()
            toggle(pG, pD)
            loss_g = pl_model.training_step(batch, i, 1)
            opt_g.zero_grad()
            loss_g.backward()
            opt_g.step()
            toggle(pD + pG, [])
            pl_step.last = (loss_d, loss_g)
        pl_step.batch = {"rgb": rgb, "nir": nir}
        # every batch a Lightning loop sees is a NEW pair of tensors: re-using one object would let the forward record of the
        # previous batch's second pass match (functional._ForwardRecord is keyed on tensor identity + version); clone per step
        _pl = pl_step

        def pl_fresh(r, n):
            pl_step.batch = {"rgb": r.clone(), "nir": n.clone()}
            return _pl(r, n)
        tr.step = pl_fresh
    buckets, bucket_ms = None, None
    if a.mixed:
        assert not a.inject and a.bs % 4 == 0, "--mixed: plain generator, bs % 4 == 0"
        buckets = [(4 * a.bs, a.size // 2), (a.bs, a.size), (a.bs // 4, 2 * a.size)]      # --size 256 (default): 128 / 256 / 512
        data = [synth(b, sz, sz, 1234 + 17 * i + rank, dev) for i, (b, sz) in enumerate(buckets)]
        bucket_ms = [[] for _ in buckets]
        counter = [rank]                   # ranks start on different buckets
        timed_order = []                   # bucket index of every timed step of this rank, in order

        def mixed_step(_r, _n, timed=False):
            i = counter[0] % len(buckets)
            counter[0] += 1
            if timed:
                timed_order.append(i)
                e0, e1 = Tick(dev), Tick(dev)
                e0.record()
            out = _step(data[i][0], data[i][1], None)
            if timed:
                e1.record()
                bucket_ms[i].append((e0, e1))
            return out
        tr.step = mixed_step

    for _ in range(max(a.warmup, 3 if a.mixed else 1)):
        tr.step(rgb, nir)
    if a.mixed:
        tr._prepare(a.bs, a.size, a.size)        # the probes bracket the middle (256x256) bucket's launches
    if pl_model is not None:
        a.no_probe = True               # the autograd bridge leases its own engines: per-kernel figures come from the fused run
        kinds, plans = {}, []
    else:
        kinds, plans = mfma_probes(tr)
    # generator-forward MFMA utilisation (second half of BASELINE.json's metric): HIP events around the generator's
    # forward plan inside the timed steps (the launch stream is torch's current stream)
    gen_fwd_events, gen_bwd_events = [], []
    if not a.no_probe:
        for eng in ([m.G for st in tr._states.values() for m in st.micros] if (a.mixed or a.micro > 1) else [tr.G]):
            def timed_forward(*args, _orig=eng.forward, _hw=(eng.Hg, eng.Wg), _b=eng.B, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = _orig(*args, **kw)
                e1.record()
                gen_fwd_events.append((e0, e1, _b * _hw[0] * _hw[1]))
                return out
            eng.forward = timed_forward

            def timed_backward(*args, _orig=eng.backward, _hw=(eng.Hg, eng.Wg), _b=eng.B, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = _orig(*args, **kw)
                e1.record()
                gen_bwd_events.append((e0, e1, _b * _hw[0] * _hw[1]))
                return out
            eng.backward = timed_backward
    if a.no_probe:
        for pl in plans:
            pl.probe_idx = None
    # Which kernel is the dominant one is found in 3 UNTIMED steps with every matrix-pipe launch kind bracketed; the timed steps then
    # bracket the launches of that kernel alone (12-13 event pairs per step instead of ~100: with all of them the burst read 2 % low in
    # fp32 and 6 % low in the bf16 mode against the probe-free sustained leg).  The other kernels' entries (`roofline_other`) come from
    # the untimed steps and say so.
    untimed_acc, dominant_kind = None, None
    if not a.no_probe and plans and dev.type == "cuda":
        for _ in range(3):
            _step(data[1][0], data[1][1], None) if a.mixed else tr.step(rgb, nir)
        torch.cuda.synchronize()
        untimed_acc = {k: [0.0, 0] for k in kinds}
        for pl in plans:
            for kind, st_ev, en_ev in pl.probe_events:
                untimed_acc[kind][0] += st_ev.elapsed_time(en_ev)
                untimed_acc[kind][1] += 1
            pl.probe_events = []
        dominant_kind = max(untimed_acc, key=lambda k: untimed_acc[k][0]) if untimed_acc else None
        for pl in plans:
            pl.probe_idx = {i: k for i, k in pl.probe_idx.items() if k == dominant_kind}

    def barrier():
        if reducer is not None:
            torch.distributed.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    if reducer is not None and dev.type == "cuda":
        reducer.exposed_events = []        # HIP events around every wait for the gradient collectives
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if a.mixed:
            tr.step(rgb, nir, True)
        else:
            tr.step(rgb, nir)
    barrier()
    dt = time.perf_counter() - t0
    rank_ms, comm_ms = [dt / a.steps * 1e3], None
    if reducer is not None:
        exposed = 0.0
        if reducer.exposed_events:
            exposed = sum(e0.elapsed_time(e1) for e0, e1 in reducer.exposed_events) / a.steps
        reducer.exposed_events = None
        t = torch.tensor([dt / a.steps * 1e3, exposed], device=dev, dtype=torch.float64)
        allt = [torch.empty_like(t) for _ in range(world)]
        torch.distributed.all_gather(allt, t)
        rank_ms = [round(float(x[0]), 3) for x in allt]
        comm_ms = [round(float(x[1]), 4) for x in allt]
        dt = max(float(x[0]) for x in allt) * a.steps / 1e3          # MAX over ranks
    # ---- the same steps WITHOUT the collectives, in this process on this rank's own shard (a second trainer with no reducer): what the
    # step costs when nothing is exchanged -- measured here instead of taken from another run, so that `measured - no_comm` is this
    # node's communication cost and the `predicted` line has its own one-GPU input
    nocomm_ms = None
    if reducer is not None and world > 1 and pl_model is None and not a.mixed and a.micro == 1:
        try:
            solo = make_trainer(None)
            for _ in range(2):
                solo.step(rgb, nir, embeds)
            barrier()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                solo.step(rgb, nir, embeds)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            mine = torch.tensor([(time.perf_counter() - t1) / a.steps * 1e3], device=dev, dtype=torch.float64)
            alln = [torch.empty_like(mine) for _ in range(world)]
            torch.distributed.all_gather(alln, mine)
            nocomm_ms = [round(float(x[0]), 3) for x in alln]
            del solo
            # (the solo steps moved each rank's weights apart: put rank 0's back before anything else runs on them)
            reducer.broadcast_params(netG._flat())
            reducer.broadcast_params(netD._flat())
        except Exception as exc:
            nocomm_ms = {"error": repr(exc)}
    sched_by_rank = None
    if a.mixed:
        # configs[4]'s load balance: which bucket every rank's timed steps drew (first one = where the rank started)
        mine = torch.tensor((timed_order + [-1] * a.steps)[:a.steps], device=dev, dtype=torch.int64)
        if reducer is not None:
            alls = [torch.empty_like(mine) for _ in range(world)]
            torch.distributed.all_gather(alls, mine)
            sched_by_rank = [[int(v) for v in t.tolist()] for t in alls]
        else:
            sched_by_rank = [[int(v) for v in mine.tolist()]]
    # ---- share of the step spent outside the matrix pipe: 3 untimed steps with EVERY matrix-pipe launch bracketed by HIP events
    mfma_share = None
    if not a.no_probe and pl_model is None and a.micro == 1 and dev.type == "cuda":
        saved = [(pl.probe_idx, pl.probe_events) for pl in plans]
        for pl in plans:
            pl.probe_idx = {i: "mfma" for i, (n, ar) in enumerate(pl.ops) if isinstance(n, str) and op_mfma_work(n, ar) is not None}
            pl.probe_events = []
        e0, e1 = Tick(dev), Tick(dev)
        barrier()
        e0.record()
        for _ in range(3):        # (mixed: the middle bucket, whose plans the probes bracket)
            _step(data[1][0], data[1][1], None) if a.mixed else tr.step(rgb, nir)
        e1.record()
        barrier()
        t_all = e0.elapsed_time(e1)
        t_mfma = sum(s_.elapsed_time(e_) for pl in plans for _, s_, e_ in pl.probe_events)
        mfma_share = t_mfma / t_all
        for pl, (pi, pe) in zip(plans, saved):
            pl.probe_idx, pl.probe_events = pi, pe
    # ---- sustained leg: back-to-back steps for >= --sustain seconds (no probes, chunks of 10 steps between two HIP events)
    sustained = None
    if a.sustain > 0 and dev.type == "cuda":
        for pl in plans:
            pl.probe_idx = None
        chunk, ticks = 10, []
        barrier()
        s0 = time.perf_counter()
        est = max(dt / a.steps, 1e-4)
        n_chunks = max(3, int(a.sustain / (est * chunk)) + 1)
        for _ in range(n_chunks):
            e0, e1 = Tick(dev), Tick(dev)
            e0.record()
            for _ in range(chunk):
                tr.step(rgb, nir, True) if a.mixed else tr.step(rgb, nir)
            e1.record()
            ticks.append((e0, e1))
        barrier()
        wall = time.perf_counter() - s0
        per = sorted(e0.elapsed_time(e1) / chunk for e0, e1 in ticks)
        third = max(1, len(per) // 3)
        seq = [e0.elapsed_time(e1) / chunk for e0, e1 in ticks]
        sustained = {"seconds": round(wall, 2), "steps": n_chunks * chunk, "tiles_per_s": round(a.bs * world * n_chunks * chunk / wall, 2),
                     "ms_per_step_p50": round(per[len(per) // 2], 3), "ms_per_step_p95": round(per[min(len(per) - 1, int(0.95 * len(per)))], 3),
                     "ms_per_step_first_third": round(sum(seq[:third]) / third, 3), "ms_per_step_last_third": round(sum(seq[-third:]) / third, 3),
                     "vs_burst": round((a.bs * world * n_chunks * chunk / wall) / (a.bs * world * a.steps / dt), 4),
                     "note": "steps issued back to back after the timed ones; chunks of 10 steps between HIP events on the launch stream "
                             "(rank 0's own clock; the headline `value` stays the barrier-bracketed K steps)"}
    out_l = tr.step(rgb, nir)
    if pl_model is not None:
        losses = {"loss_D": float(pl_step.last[0]), "loss_G": float(pl_step.last[1])}
    else:
        losses = out_l.as_dict()
    assert all(v == v and abs(v) < 1e30 for v in losses.values()), f"non-finite losses {losses}"

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = a.bs * world * a.steps / dt
        roof, roof_other = None, None
        if not a.no_probe:
            acc = {k: [0.0, 0] for k in kinds}
            for pl in plans:
                for kind, st_ev, en_ev in pl.probe_events:
                    acc[kind][0] += st_ev.elapsed_time(en_ev)
                    acc[kind][1] += 1
            roofs = []
            for k, (flops, nlaunch) in kinds.items():
                ev_ms, n_ev = acc[k]
                live = n_ev > 0                                   # bracketed inside the timed steps (the dominant kernel)
                if not live and untimed_acc is not None:
                    ev_ms, n_ev = untimed_acc[k]
                if not nlaunch or not n_ev:
                    continue
                per_launch_flop = flops / nlaunch
                avg_ms = ev_ms / n_ev
                ach = per_launch_flop / (avg_ms * 1e-3) / 1e12
                by = (mfma_probes.algo_bytes[k] / nlaunch) if k in getattr(mfma_probes, "algo_bytes", {}) else None
                gbps = None if by is None else by / (avg_ms * 1e-3) / 1e9
                pk = kernel_peak(k, a.precision)
                roofs.append({"bound": "mfma", "achieved": round(ach, 2), "peak": round(pk, 1), "unit": "TFLOP/s",
                              "frac": round(ach / pk, 4),
                              # the same launches priced by the DIRECT convolution's FLOPs (the Winograd layers execute 64/324 and 49/256 of them)
                              "algorithmic_over_peak": round(getattr(mfma_probes, "direct", {}).get(k, flops) / nlaunch / (avg_ms * 1e-3) / 1e12 / pk, 4),
                              "traffic": None, "kernel": k,
                              "launches_per_step": nlaunch, "avg_launch_ms": round(avg_ms, 5),
                              "algorithmic_gflop_per_launch": round(per_launch_flop / 1e9, 3),
                              "algorithmic_bytes_per_launch": None if by is None else int(by),
                              # the OTHER bound of the same launch: its algorithmic bytes over the same time against the HBM peak
                              "hbm_bound": None if by is None else {"achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4)},
                              "binding": None if by is None else ("mfma" if ach / pk >= gbps / PEAK_HBM_GBPS else "hbm"),
                              "share_of_step_time": round(avg_ms * nlaunch / ms, 3),
                              "measured": "HIP events on the launch stream inside the timed steps" if live else "HIP events on the launch stream, 3 untimed steps before the timed ones"})
            # HBM traffic per launch: NOT measured in this run (PMC needs rocprofv3 passes around the process).  It is replayed from
            # the PMC summary recorded under profiles/ by scripts/refresh_profiles.sh over this same command (rocprofv3 --pmc
            # FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x2-corrected as MI355X_MICROARCH.md prescribes; mean over the
            # kernel's launches in a step) -- only for the workload it was measured on and only while the kernel sources are the
            # ones it was measured with (kernel_src_sha16); otherwise traffic stays null.
            pmc, pmc_file = {}, None
            for cand in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_bench_summary.json")), reverse=True):
                try:
                    pmc, pmc_file = json.load(open(os.path.join(ROOT, "profiles", cand))), "profiles/" + cand
                    break
                except Exception:
                    continue
            meta = pmc.get("_meta", {}) if isinstance(pmc.get("_meta", {}), dict) else {}
            src_now = kernel_source_sha16()
            fresh = meta.get("kernel_src_sha16") == src_now
            headline = (a.bs == 16 and a.size == 256 and a.blocks == 6 and a.padding == 0 and a.precision == "fp32"
                        and not a.mixed and a.micro == 1 and not a.inject and a.lambda_rs == 0.0 and a.ngf == 64)
            for r in roofs:
                ent = next((v for k, v in pmc.items() if k != "_meta" and k.split("<")[0] == r["kernel"].split("<")[0]
                            and ("<" not in r["kernel"] or k == r["kernel"] or k.startswith(r["kernel"][:-1] + ","))), None)
                r["traffic_source"] = None
                if ent and headline and fresh and "hbm_read_bytes_per_launch_corrected" in ent:
                    r["traffic"] = int(ent["hbm_read_bytes_per_launch_corrected"] + ent.get("hbm_write_bytes_per_launch", 0.0))
                    r["traffic_unit"] = "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, mean over the kernel's launches)"
                    r["traffic_source"] = f"replayed from {pmc_file} (recorded by scripts/refresh_profiles.sh; kernel_src_sha16 {src_now}), not measured in this run"
                    if "mfma_busy_fraction_of_active_cycles" in ent:
                        r["pmc_mfma_busy"] = round(ent["mfma_busy_fraction_of_active_cycles"], 4)
                        r["pmc_source"] = r["traffic_source"]
                elif ent and headline and not fresh:
                    r["traffic_source"] = (f"{pmc_file} was recorded for kernel sources {meta.get('kernel_src_sha16')}, this build is {src_now}: "
                                           "stale, not reported (rerun scripts/refresh_profiles.sh)")
            for r in roofs:
                if _is_x3(r["kernel"]):
                    r["arithmetic"] = ("fp32-equivalent on the bf16 matrix pipe: every fp32 operand as three bf16 terms, six v_mfma_f32_16x16x32_bf16 products per "
                                       "fp32 product, fp32 accumulate (csrc/igemm_x3.h); `achieved` counts fp32-EQUIVALENT FLOPs, `peak` = 2.5 PFLOP/s dense bf16 / 6")
                    r["executed_bf16_tflops"] = round(6.0 * r["achieved"], 1)
                    r["flops_counted"] = ("fp32-equivalent flops of every launch of this kernel in a step: direct convolutions / sub-pixel groups / weight gradients "
                                          "at 2 M N K, the Winograd layers' plane GEMMs and transform-domain weight gradients at their EXECUTED count (64/324 of "
                                          "the direct layer's multiplies for F(6x6,3x3), 49/256 for F(4x4,4x4))")
                    continue
                if r["kernel"].startswith("wino6"):
                    r["flops_counted"] = ("EXECUTED matrix-pipe flops: the Winograd plane GEMMs (and, in the pair launch, the transform-domain weight-gradient "
                                          "planes) perform 64/324 of the direct layer's multiplies for the F(6x6,3x3) residual-block layers (36/144 with "
                                          "NIRGAN_OPTIONS=winograd=f4) and 49/256 for the PatchGAN's F(4x4,4x4) layer; rows of a plane's last, partly filled M tile "
                                          "(T = 1936 = 15.1 tiles of 128) are executed but not counted")
                elif r["kernel"].startswith("wino"):
                    r["flops_counted"] = ("EXECUTED matrix-pipe flops: the Winograd F(2x2,3x3) part performs 16/36 of the direct layer's "
                                          "multiplies")
                elif r["kernel"].startswith("wgrad_igemm"):
                    r["flops_counted"] = ("EXECUTED matrix-pipe flops (includes transform-domain weight-gradient planes of Winograd layers that are "
                                          "launched on their own)")
            roofs.sort(key=lambda r: (not r["measured"].endswith("inside the timed steps"), -r["share_of_step_time"]))      # the live-measured (dominant) kernel first
            roof = roofs[0] if roofs else None
            roof_other = roofs[1:] or None
        gflop_tile = {(6, 0): 257.0, (6, 10): 290.5, (9, 0): 344.0, (9, 10): 391.6}.get((a.blocks, a.padding))
        from nirgan_hip.options import OPT as _OPT
        x3_on = a.precision == "fp32" and _OPT.split3
        dtype = {"fp32": "f32", "bf16": "bf16 operands, f32 accumulate", "bf16x3": "f32 as 2 bf16 terms (3 products), f32 accumulate"}[a.precision]
        mfma = {"fp32": "fp32 MFMA", "bf16": "bf16 MFMA (fp32 accumulate/master)", "bf16x3": "bf16x3 split-fp32 MFMA"}[a.precision]
        if x3_on:
            dtype = ("f32 (3xbf16 split, 6 products, f32 accumulate) for the contractions with 32-channel runs" + (" incl. the Winograd plane GEMMs" if _OPT.split3_wino else "")
                     + "; exact f32 MFMA for the first / last layers; f32 everywhere else")
            mfma = "fp32-equivalent MFMA (three bf16 terms, six products)"
        if a.mixed:
            workload = (f"configs[4]: mixed-resolution buckets {[f'{b}@{sz}' for b, sz in buckets]} per GPU (one bucket per rank-step, "
                        f"equal tile area), {a.blocks}-block ResnetGenerator + PatchGAN, GAN+L1"
                        + (f"+RS(l={a.lambda_rs})" if a.lambda_rs else "") + f", padding={a.padding}, {mfma}; value in 256x256-equivalent tiles/s")
        else:
            if a.inject:
                tag = "configs[3]" if (a.size == 512 and a.bs == 8) else "configs[3]-like"
            elif a.blocks == 9 and a.lambda_rs > 0:
                tag = "configs[2]" if (a.size == 256 and a.bs == 32 and a.precision == "fp32") else "configs[2]-like"
            elif a.blocks == 6 and a.lambda_rs == 0 and a.size == 256 and a.bs == 16 and a.precision == "fp32" and a.ngf == 64:
                tag = "configs[1]" if a.padding == 0 else "configs[1] with the YAML's padding"
            else:
                tag = "custom"
            workload = (f"{tag}: {a.blocks}-block ResnetGenerator" + (f" (ngf {a.ngf})" if a.ngf != 64 else "") + f" + 3-layer PatchGAN, bs={a.bs}/GPU, "
                        f"{a.size}x{a.size}, GAN+L1" + (f"+RS(l={a.lambda_rs})" if a.lambda_rs else "")
                        + (", SatCLIP inject" if a.inject else "") + f", padding={a.padding}, {mfma}")
        out = {"metric": "256x256 RGB tiles/sec (G+D fwd+bwd+step)", "value": round(value, 3), "unit": "tiles/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
               "config": {"workload": workload, "global_batch": a.bs * world, "parallelism": f"dp{world}"},
               "roofline": roof, "kernel_src_sha16": kernel_source_sha16()}
        if reducer is not None:
            out["rccl_ranks"] = world if not a.emulate_cpu else 0
            out["collective_backend"] = torch.distributed.get_backend()
            out["ms_per_step_by_rank"] = rank_ms
            out["comm_exposed_ms_per_step_by_rank"] = comm_ms
        else:
            # one process, no process group: the same fields as the N > 1 line (a scaling table reads them column by column)
            out["rccl_ranks"] = 0 if a.emulate_cpu else 1
            out["collective_backend"] = None
            out["ms_per_step_by_rank"] = [round(ms, 3)]
            out["comm_exposed_ms_per_step_by_rank"] = [0.0]
        if isinstance(nocomm_ms, list):
            out["ms_per_step_no_comm_by_rank"] = nocomm_ms
        elif nocomm_ms is not None:
            out["ms_per_step_no_comm_by_rank"] = nocomm_ms
        if pl_model is None and not a.mixed and a.micro == 1 and not a.emulate_cpu:
            try:
                # (N > 1: the model's one-GPU input is THIS run's no-communication leg, not the N-rank time)
                ms1 = ms if world == 1 else (min(nocomm_ms) if isinstance(nocomm_ms, list) else min(rank_ms))
                out["predicted"] = predict_scaling(tr, ms1, a.bs)
                if world > 1:
                    mine = [r for r in out["predicted"]["by_n_gpus"] if r["n_gpus"] == world]
                    # measured beside predicted for THIS N: the record is evidence of the model's error as well as of the speed
                    out["measured_vs_predicted"] = {
                        "n_gpus": world, "measured_ms_per_step": round(ms, 3), "measured_no_comm_ms_per_step": ms1,
                        "measured_comm_cost_ms": round(ms - ms1, 3), "measured_exposed_wait_ms_max_rank": max(comm_ms) if comm_ms else None,
                        "measured_weak_scaling_efficiency_vs_no_comm": round(ms1 / ms, 4),
                        "predicted": mine[0] if mine else None,
                        "note": "efficiency against this run's own no-communication leg (same process, same GPUs); the driver computes the "
                                "efficiency against the N = 1 run itself"}
            except Exception as exc:              # (never lose the measured line to the model)
                out["predicted"] = {"error": repr(exc)}
        if reducer is not None:
            out["comm"] = ("three gradient buckets per network (tail and middle started inside the backward plan, the first layer's gradient -- "
                           "16 / 38 KB -- after it); exposed = launch-stream time spent waiting for the collectives before each Adam step (HIP events)")
        if dp_check is not None:
            out["dp_verify"] = dp_check
        if a.micro > 1:
            out["config"]["micro_batches"] = a.micro
            if roof:
                roof["note"] = "kernels of the micro-batches share the chip: per-launch time includes the sharing"
        if a.mixed and sched_by_rank is not None:
            per_bucket = [[sum(1 for v in sch if v == i) for i in range(len(buckets))] for sch in sched_by_rank]
            out["bucket_schedule"] = {"first_bucket_by_rank": [sch[0] for sch in sched_by_rank], "timed_steps_per_bucket_by_rank": per_bucket,
                                      "timed_steps_per_bucket_all_ranks": [sum(pb[i] for pb in per_bucket) for i in range(len(buckets))],
                                      "note": "rank r starts on bucket r mod 3 and walks the buckets round-robin: every step all-reduces gradients that come from "
                                              "different tile sizes; with steps a multiple of 3 every rank spends the same time per bucket"}
        if a.mixed:
            out["buckets_rank0"] = [{"tiles": b, "size": sz, "steps": len(ev),
                                     "raw_tiles_per_s": round(b * len(ev) / (sum(x.elapsed_time(y) for x, y in ev) * 1e-3), 2) if ev else None}
                                    for (b, sz), ev in zip(buckets, bucket_ms)]
        if roof_other:
            out["roofline_other"] = roof_other
        if gen_fwd_events:
            # generator forward = 69.29 (6-block) / 98.28 (9-block) GFLOP per 256x256 generator input
            g256 = {6: 69.29, 9: 98.28}[a.blocks]          # per 65536 generator-input pixels (padded size counted below)
            t_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in gen_fwd_events)
            px = sum(n for _, _, n in gen_fwd_events)
            tf = g256 * (px / 65536.0) / t_ms          # GFLOP / ms = TFLOP/s
            ex = plan_executed_flops(tr.G.fwd) / 1e9 if not (a.mixed or a.micro > 1) else None      # GFLOP per call, executed on the matrix pipe
            ms_call = t_ms / len(gen_fwd_events)
            out["gen_fwd"] = {"ms_per_call": round(ms_call, 3), "tflops_algorithmic": round(tf, 2),
                              "algorithmic_over_peak": round(tf / PEAKS[a.precision], 4), "peak": round(PEAKS[a.precision], 1),
                              "executed_gflop_per_call": None if ex is None else round(ex, 2),
                              "executed_mfma_util": None if ex is None else round(plan_pipe_ms_at_peak(tr.G.fwd, a.precision) / ms_call, 4),
                              "note": "generator forward incl. its instance-norm / transform / layout kernels.  algorithmic_over_peak = direct-"
                                      "convolution FLOPs (SURVEY 8d) / elapsed / dense MFMA peak: NOT a utilisation -- the Winograd layers execute 64/324 "
                                      "of those multiplies, so it can exceed 1.  executed_mfma_util = FLOPs the matrix pipe actually executes (from "
                                      "the launch descriptors) / elapsed / peak, each launch at its own pipe's peak (fp32 MFMA 157.3, or 2500 / 6 for the "
                                      "three-term split tiles' fp32-equivalent FLOPs): BASELINE.json's 'gen-fwd MFMA util'"}
        if gen_fwd_events and gen_bwd_events:
            # backward = data + weight gradients of every conv except the first layer's data gradient: 2g - g1,
            # g1 = 7x7x3x64 MACs per pixel (SURVEY 8d formula)
            g1 = 2.0 * 49 * 3 * 64 * 65536 / 1e9
            gfb = 3.0 * {6: 69.29, 9: 98.28}[a.blocks] - g1
            t_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in gen_fwd_events + gen_bwd_events)
            px = sum(n for _, _, n in gen_fwd_events)
            tf = gfb * (px / 65536.0) / t_ms
            ex = (plan_executed_flops(tr.G.fwd) + plan_executed_flops(tr.G.bwd)) / 1e9 if not (a.mixed or a.micro > 1) else None
            ms_fb = t_ms / len(gen_fwd_events)
            out["gen_fwd_bwd"] = {"ms_per_step": round(ms_fb, 3), "tflops_algorithmic": round(tf, 2),
                                  "algorithmic_over_peak": round(tf / PEAKS[a.precision], 4),
                                  "executed_gflop_per_step": None if ex is None else round(ex, 2),
                                  "executed_mfma_util": None if ex is None else round((plan_pipe_ms_at_peak(tr.G.fwd, a.precision) + plan_pipe_ms_at_peak(tr.G.bwd, a.precision)) / ms_fb, 4)}
        if gflop_tile and not a.mixed and a.size == 256:
            out["step_tflops_algorithmic"] = round(gflop_tile * value / 1e3, 2)
        if pl_model is None and not a.no_probe and not (a.mixed or a.micro > 1):
            exs = sum(plan_executed_flops(pl) for pl in plans) / 1e9
            out["whole_step"] = {"executed_mfma_gflop": round(exs, 1), "executed_mfma_util": round(sum(plan_pipe_ms_at_peak(pl, a.precision) for pl in plans) / ms, 4),
                                 "note": "FLOPs the matrix pipe executes per step (launch descriptors; fp32-equivalent for the three-term split tiles); util = each "
                                         "launch's FLOPs over its own pipe's dense peak (fp32 MFMA 157.3, split tiles 2500 / 6), summed, over ms_per_step"}
            if mfma_share is not None:
                out["whole_step"]["matrix_pipe_launch_share"] = round(mfma_share, 4)
                out["whole_step"]["hbm_bound_share"] = round(1.0 - mfma_share, 4)
                out["whole_step"]["share_note"] = ("3 untimed steps with every matrix-pipe launch between HIP events: share of the step inside them; the "
                                                   "rest (transforms, instance norm, layout, losses, Adam, launch gaps) is HBM / latency bound")
            wr = pmc.get("_whole_run") if (not a.no_probe and isinstance(pmc.get("_whole_run"), dict)) else None
            if wr and headline and fresh:
                gb = wr["hbm_read_gb_per_step"] + wr["hbm_write_gb_per_step"]
                out["whole_step"].update({"pmc_mfma_busy": round(wr["mfma_busy_fraction_of_active_cycles"], 4),
                                          "pmc_executed_mfma_gflop": round(wr["executed_mfma_gflop_per_step_fp32"], 1),
                                          "pmc_hbm_gb": round(gb, 2), "hbm_tbps": round(gb / ms, 3),
                                          "pmc_source": f"replayed from {pmc_file} (rocprofv3 --pmc passes over this command, scripts/refresh_profiles.sh), not measured in this run"})
        if a.mixed and mfma_share is not None:
            out["middle_bucket"] = {"matrix_pipe_launch_share": round(mfma_share, 4), "hbm_bound_share": round(1.0 - mfma_share, 4),
                                    "note": f"{a.bs} tiles @{a.size}: 3 untimed steps with every matrix-pipe launch between HIP events"}
        if sustained is not None:
            out["sustained"] = sustained
        if pl_model is not None:
            out["config"]["api"] = ("Px2Px_PL.training_step(batch, i, 0/1) + zero_grad + backward + HipAdam.step under toggle_optimizer, as "
                                    "Lightning 1.9 drives the reference (train.py:136); generator forwards re-used: "
                                    f"{pl_model.netG.__dict__.get('_fwd_reused', 0)}")
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.blocks, a.size, bs=16 if a.size <= 256 else 4)
        print(json.dumps(out), flush=True)
    if reducer is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
