"""Drop-in ``model`` package: same module names as the reference (model/networks.py,
model/generator_inject.py, model/pix2pix.py, model/pix2pix_model.py), HIP engines inside."""
