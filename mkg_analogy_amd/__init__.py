"""mkg_analogy_amd -- MI355X-native (gfx950) implementation of the MarT / MKGformer analogy hot path.

Host-side mirror of the reference's operator API (``models.MKGformerKGC``) and trainer surface
(``lit_models.TransformerLitModel``) over hand-written HIP kernels reached through a C ABI
(``include/mart_hip.h`` -> ``lib/libmart_hip.so``).  There is no CPU / eager fallback.
"""
__version__ = "0.1.0"
