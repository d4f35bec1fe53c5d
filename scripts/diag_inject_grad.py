import sys, types, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/nir-gan_amd"); sys.path.insert(0, "/root/repo/oracle")
import torch
import nirgan_oracle as O
from test_gpu_nets import synth, leaf64, DEV
from model.generator_inject import define_G_inject
ns = types.SimpleNamespace
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = ns(base_configs=ns(input_nc=3, output_nc=1, ngf=64, netG="resnet_9blocks", norm="instance", no_dropout=True, init_type="normal", init_gain=0.02),
         satclip=ns(satclip_inject_style="multiply", post_correction=False, post_correction_init=1.0, scaling_param=True, scaling_param_init=0.5))
torch.manual_seed(0)
net = define_G_inject(cfg)
sd = {k: v.clone() for k, v in net.state_dict().items()}
rgb, _ = synth(1, size, size, 33)
emb = torch.randn(1, 256, generator=torch.Generator().manual_seed(34))
dout = torch.randn(1, 1, size, size, generator=torch.Generator().manual_seed(35))
net = net.to(DEV); net.data_pad = 10
pred = net(rgb.to(DEV), emb.to(DEV)); pred.backward(dout.to(DEV))
p64 = leaf64(sd)
ref = O.px_forward(p64, rgb.double(), 9, 10, emb.double(), {"style": "multiply", "use_scale": True})
ref.backward(dout.double())
print("pred err", (pred.detach().cpu().double() - ref.detach()).abs().max().item())
for k, p in net.named_parameters():
    if k in p64 and p64[k].grad is not None:
        a, b = p.grad.double().cpu(), p64[k].grad
        print(f"{k:28s} relL2 {((a-b).norm()/b.norm().clamp_min(1e-30)).item():.3e}  |b| {b.norm().item():.3e}")
