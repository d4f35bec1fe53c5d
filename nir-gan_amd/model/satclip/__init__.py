"""MI355X-native counterpart of the part of ``model/satclip`` the NIR-GAN hot path touches: the frozen location
encoder (spherical harmonics + SirenNet) that turns lon/lat into the 256-d embedding the inject generator consumes.
The CLIP training code, image encoders and data modules of SatCLIP are not on the path."""
