"""MI355X-native implementation of the synthesis hot path of decode_tonal_langauge.

Sub-packages mirror the reference's module paths for the path they replace
(``models.synthesis_models``, ``models.synthesis_trainer``, ``preprocess.signal.frequency_filter`` ...).
All arithmetic runs in hand-written HIP kernels (``csrc/``) reached through the C ABI of
``libtonal_hip.so``; there is no CPU fallback.
"""
__version__ = "0.1.0"
