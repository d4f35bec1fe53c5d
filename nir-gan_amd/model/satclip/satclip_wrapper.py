"""Counterpart of the reference's ``model/satclip/satclip_wrapper.py``: ``SatClIP_wrapper(satclip_path, device).predict(x)``
returns the detached float32 location embeddings of ``x`` = [B, 2] lon/lat (satclip_wrapper.py:8-35)."""
import torch

from model.satclip.load_lightweight import get_satclip_loc_encoder


class SatClIP_wrapper(torch.nn.Module):
    def __init__(self, satclip_path=None, device="cuda"):
        super().__init__()
        if satclip_path is None:
            satclip_path = "model/satclip/satclip-resnet50-l10.ckpt"
        self.encoder_model = get_satclip_loc_encoder(satclip_path, device)

    def predict(self, x):
        with torch.no_grad():
            return self.encoder_model(x.double()).float().detach()

    def forward(self, x):
        print("Don't use fwd, use 'predict' step instead")
        return self.encoder_model(x.double())
