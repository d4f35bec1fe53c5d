"""MI355X-native counterpart of the reference's ``utils/remote_sensing_indices.py``.

``RemoteSensingIndices(mode, criterion).get_and_weight_losses(rgb, nir, nir_pred, loss_config, mode)``
keeps the reference's semantics (utils/remote_sensing_indices.py:6-71): in 'loss' mode the
indices with weight > 0 are accumulated in the dict order ndvi, ndwi, gndvi, savi, msavi, evi;
'logging_dict' returns the six unweighted errors.  All of it is one fused HIP pass
(nirgan_pix_loss) that is differentiable wrt ``nir_pred``.  The per-index ``*_calculation``
methods return the same scalar in 'loss' mode; 'index' mode (plots only, outside the hot
path) evaluates the formulas with tensor expressions.
"""
import torch

from nirgan_hip import functional as HF

_ORDER = ["ndvi", "ndwi", "gndvi", "savi", "msavi", "evi"]
_LOG_NAMES = {k: f"indices_loss/{k}_error" for k in _ORDER}


class RemoteSensingIndices():

    def __init__(self, mode="loss", criterion="l1"):
        assert mode in ["loss", "index"], f"Mode '{mode}' not implemented. 'loss', 'index' are supported."
        self.mode = mode
        if criterion == "l1":
            self._crit = 0
        elif criterion == "l2":
            self._crit = 1
        else:
            raise NotImplementedError(f"Criterion '{criterion}' not implemented. 'l1' or 'l2' are supported.")

    def prepare_tensor_for_loss(self, rgb, nir, nir_pred):
        if len(rgb.shape) == 3:
            rgb = rgb.unsqueeze(0)
        if len(nir.shape) == 3:
            nir = nir.unsqueeze(0)
        if len(nir_pred.shape) == 3:
            nir_pred = nir_pred.unsqueeze(0)
        return (rgb, nir, nir_pred)

    def get_and_weight_losses(self, rgb, nir, nir_pred, loss_config=None, mode="loss"):
        if loss_config is None:
            loss_config = {"lambda_ndvi": 0.333, "lambda_ndwi": 0.333, "lambda_evi": 0.333,
                           "lambda_savi": 0.0, "lambda_msavi": 0.0, "lambda_gndvi": 0.0}
        rgb, nir, nir_pred = self.prepare_tensor_for_loss(rgb, nir, nir_pred)
        if mode == "loss":
            w = [0.0] + [float(loss_config.get("lambda_" + k, 0.0)) for k in _ORDER]
            w = [x if x > 0.0 else 0.0 for x in w]
            if not any(w):
                return 0.0
            return HF.PixLossFn.apply(rgb, nir, nir_pred, tuple(w), self._crit)
        elif mode == "logging_dict":
            s = HF.index_sums(rgb, nir, nir_pred, self._crit)
            return {_LOG_NAMES[k]: s[1 + i] for i, k in enumerate(_ORDER)}
        else:
            raise NotImplementedError(f"Mode '{mode}' not implemented. 'loss' or 'logging_dict' are supported.")

    # ------------------------------------------------------------------ single indices
    def _single(self, which, rgb, nir, nir_pred):
        rgb, nir, nir_pred = self.prepare_tensor_for_loss(rgb, nir, nir_pred)
        if self.mode == "loss":
            w = [0.0] * 7
            w[1 + _ORDER.index(which)] = 1.0
            return HF.PixLossFn.apply(rgb, nir, nir_pred, tuple(w), self._crit)
        elif self.mode == "index":
            return _index_images(which, rgb, nir, nir_pred)
        raise NotImplementedError(f"Mode '{self.mode}' not implemented. 'loss' or 'index' are supported.")

    def ndvi_calculation(self, rgb, nir, nir_pred):
        return self._single("ndvi", rgb, nir, nir_pred)

    def ndwi_calculation(self, rgb, nir, nir_pred):
        return self._single("ndwi", rgb, nir, nir_pred)

    def gndvi_calculation(self, rgb, nir, nir_pred):
        return self._single("gndvi", rgb, nir, nir_pred)

    def savi_calculation(self, rgb, nir, nir_pred):
        return self._single("savi", rgb, nir, nir_pred)

    def msavi_calculation(self, rgb, nir, nir_pred):
        return self._single("msavi", rgb, nir, nir_pred)

    def evi_calculation(self, rgb, nir, nir_pred):
        return self._single("evi", rgb, nir, nir_pred)


def _index_images(which, rgb, nir, pred):
    """'index' mode: index images for plotting (no epsilon), remote_sensing_indices.py:116-117,156-157,..."""
    red, green, blue = rgb[:, 0:1], rgb[:, 1:2], rgb[:, 2:3]

    def f(v):
        if which == "ndvi":
            return (v - red) / (v + red)
        if which == "ndwi":
            return (v - green) / (v + green)
        if which == "gndvi":
            return (v - green) / ((v - red) / (v + red) + green)
        if which == "savi":
            return 1.5 * (v - red) / (v + red + 0.5)
        if which == "msavi":
            return (2 * v + 1 - torch.sqrt((2 * v + 1) ** 2 - 8 * (v - red))) / 2
        return 2.5 * ((v - red) / ((v + 6) * (red - 7.5) * (blue + 1)))
    return (f(nir), f(pred))
