"""Engine options that are NOT part of the reference's API: A/B switches for measurements and for the kernel tests.

One process-wide object, read when an engine is BUILT (descriptors are static afterwards; the C library reads no environment
variables and keeps no switches of its own -- kernel choices travel in the descriptors' ``algo`` fields).  Tests and scripts set
attributes (``monkeypatch.setattr(OPT, "w6_pair", False)``); the defaults are the measured-fastest configuration.
"""
from __future__ import annotations


class Options:
    # plane GEMMs of the Winograd layers: nirgan_wino6_desc.algo (0 = persistent workgroups on 32-k stages where they apply,
    # lib.W6_ONE_TILE / W6_PERSIST16 / W6_DIRECT_TILE)
    w6_gemm_algo: int = 0
    # weight-gradient launches of plane-matrix form: nirgan_wgrad_desc.algo (0 = persistent walk, lib.WGRAD_ONE_UNIT)
    wgrad_algo: int = 0
    # data-gradient plane GEMMs and the transform-domain weight gradient of a layer in ONE grid (nirgan_wino6_gemm_wgrad_pair)
    w6_pair: bool = True

    def reset(self):
        for k, v in vars(Options).items():
            if not k.startswith("_") and not callable(v):
                setattr(self, k, v)


OPT = Options()
