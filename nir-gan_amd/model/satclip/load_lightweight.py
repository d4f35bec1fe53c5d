"""Counterpart of the reference's ``model/satclip/load_lightweight.py``: build the location encoder from a SatCLIP
checkpoint's hyper-parameters and load only its ``nnet`` tensors (load_lightweight.py:5-35)."""
import torch

from model.satclip.location_encoder import LocationEncoder, get_neural_network, get_positional_encoding


def loc_encoder_from_checkpoint(ckpt: dict, device) -> LocationEncoder:
    hp = ckpt['hyper_parameters']
    posenc = get_positional_encoding(hp['le_type'], hp['legendre_polys'], hp['harmonics_calculation'],
                                     hp['min_radius'], hp['max_radius'], hp['frequency_num'])
    nnet = get_neural_network(hp['pe_type'], posenc.embedding_dim, hp['embed_dim'], hp['capacity'], hp['num_hidden_layers'])
    state_dict = ckpt['state_dict']
    state_dict = {k[k.index('nnet'):]: state_dict[k] for k in state_dict.keys() if 'nnet' in k}
    loc_encoder = LocationEncoder(posenc, nnet).double()
    loc_encoder.load_state_dict(state_dict)
    loc_encoder.eval()
    return loc_encoder.to(device)


def get_satclip_loc_encoder(ckpt_path, device):
    return loc_encoder_from_checkpoint(torch.load(ckpt_path, map_location=device), device)
